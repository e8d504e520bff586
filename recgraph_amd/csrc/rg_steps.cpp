// Step tables of the pathwise sweeps (k_sweep / k_sweep16): one 16-byte record per (row, edge group) in sweep order, the
// split form of the same records (TAILs moved behind their register runs) and the PATH RETIREMENT tables that go with
// either.  Host-only code (no HIP): rg_path_driver.hip uploads what this builds, tests/test_host_cpu.py and the sanitizer
// build (tests/c/host_asan.cpp) read it back through rg_graph_dump codes 31-34.
// Reference: the order of the rows and of a row's predecessor groups is the one of pathwise_alignment.rs:185-299 /
// pathwise_alignment_recombination.rs:436-745 (forward) and :129-435 (reverse), as flattened by rg_graph.cpp.
#include "rg_host.hpp"

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include <cstdlib>

namespace rg {

std::atomic<int>& retire_shift_option() {
    static std::atomic<int> v{[] {
        const char* e = getenv("RG_RETIRE_SHIFT");
        const int k = e ? atoi(e) : RG_SWEEP16_RETIRE_SHIFT;
        return k < 2 ? 2 : (k > 12 ? 12 : k);
    }()};
    return v;
}

void build_step_tables(const HostGraph& h, bool forward, bool want_split, StepTables& T) {
    const int L = h.L;
    const std::vector<int32_t>& goff_ = forward ? h.fgoff : h.rgoff;
    const std::vector<GroupDesc>& groups_ = forward ? h.fgroups : h.rgroups;
struct StepMeta { int row, pred; unsigned long long mask; bool low, cont; };   // per record: what split_tails needs
std::vector<StepMeta> meta;
auto steps = [&](const std::vector<int32_t>& goff, const std::vector<GroupDesc>& groups, bool fwd, std::vector<StepRec>& out) {
    out.clear();
    meta.clear();
    for (int step = 1; step + 1 < L; ++step) {
        const int i = fwd ? step : L - 1 - step;
        const bool inner = fwd ? (h.node_id[i] == h.node_id[i - 1] && i > 1) : (h.node_id[i] == h.node_id[i + 1]);
        const int li = (int)std::string("ACGTN").find(h.lnz[i]);
        for (int gi = goff[i]; gi < goff[i + 1]; ++gi) {
            // x: row (20) | base code (3) | flags (3: first / last entry of the row, inner row of a one-entry
            //    segment run) | group alpha bit (6) — for an INNER row the alpha is the lowest member (k_sweep* take
            //    it from the mask) and the field holds the number of inner rows left in the run, this one
            //    included (capped at 63): k_sweep16 handles long runs of wide groups in one piece (gather runs)
            // y: slot (20) | knm + 1 (9) | page (2) | continuation entry (1)        z, w: members of the page
            int flags = 0;
            if (gi == goff[i]) flags |= 1;
            if (gi + 1 == goff[i + 1]) flags |= 2;
            const bool cont = groups[gi].ga == GroupDesc::GA_CONT;
            if (goff[i + 1] - goff[i] == 1 && !cont && groups[gi].mask &&
                groups[gi].ga == (uint32_t)__builtin_ctzll(groups[gi].mask))
                // one group led by its lowest member: an inner row of a segment run (7), or — the first row of a
                // segment, e.g. of an allele between two shared segments — the HEAD of one (4 alone): the kernels
                // start a register / gather run on either, and continue one only into a 7
                flags = inner ? 7 : 4;
            StepRec r;
            r.x = (int)((unsigned)i | ((unsigned)li << 20) | ((unsigned)flags << 23) | ((cont ? 0u : groups[gi].ga) << 26));
            r.y = (int)((unsigned)groups[gi].slot | ((unsigned)(h.knm[i] + 1) << 20) | ((unsigned)groups[gi].page << 29) |
                        (cont ? 0x80000000u : 0u));
            r.z = (int)(unsigned)(groups[gi].mask & 0xffffffffull);
            r.w = (int)(unsigned)(groups[gi].mask >> 32);
            out.push_back(r);
            meta.push_back({i, groups[gi].pred, groups[gi].mask,
                            !cont && groups[gi].mask && groups[gi].ga == (uint32_t)__builtin_ctzll(groups[gi].mask), cont || groups[gi].page != 0});
        }
    }
    // inner / head records: alpha field := rows left in the run, this one included, capped at 63 (consecutive inner
    // records = rows of one segment): k_sweep16 decides from it whether a gather run pays and knows where the run
    // ends (the rows of what follows are touched early).
    int left = 0;
    for (size_t t = out.size(); t-- > 0;) {
        const bool inner = ((unsigned)out[t].x >> 23) & 4u;
        left = inner ? std::min(left + 1, 63) : 0;
        if (inner) {
            out[t].x = (int)(((unsigned)out[t].x & 0x03ffffffu) | ((unsigned)left << 26));
            if ((((unsigned)out[t].x >> 23) & 7u) == 4u) left = 0;      // a HEAD: the record before it belongs to another segment
        }
    }
};
// SPLIT TABLES (k_sweep16, record variants).  A row with several groups (the first row of a segment that several
// segments lead into: one group per predecessor) is processed group by group, each loading and storing its members'
// rolling rows — right after the register runs of those predecessors stored the very same rows.  Here a group whose
// paths are exactly the paths of a register run (<= 4 paths, led by the lowest) on its predecessor row moves
// directly behind that run as a TAIL (flag bit 4 with a zero run field): the run continues into it with the rows
// in registers, and the row's keys fold into bkey across its groups as before (its first / last record in the NEW
// order carry the first / last bits).  Only runs that the kernel handles as REGISTER runs may lie between the groups
// of one row (they leave the row's keys alone, and unlike gather runs they do not use the LDS words the keys wait in):
// the block of such runs directly before the row.
auto split_tails = [&](const std::vector<StepRec>& in, std::vector<StepRec>& out) {
    out.clear();
    std::vector<StepMeta> om;
    auto fl = [](const StepRec& r) { return ((unsigned)r.x >> 23) & 7u; };
    auto field = [](const StepRec& r) { return ((unsigned)r.x >> 26) & 63u; };
    auto is_run = [&](const StepRec& r) { return (fl(r) & 4u) && field(r) != 0; };
    size_t q = 0;
    while (q < in.size()) {
        size_t e = q;
        while (e < in.size() && meta[e].row == meta[q].row) ++e;
        bool plain = e - q < 2;
        for (size_t j = q; j < e; ++j) plain = plain || meta[j].cont;
        if (plain) { for (size_t j = q; j < e; ++j) { out.push_back(in[j]); om.push_back(meta[j]); } q = e; continue; }
        // runs of the safe block that ends at out.size(): [start, end) index pairs, in order
        std::vector<std::pair<size_t, size_t>> runs;
        size_t pos = out.size();
        while (pos > 0 && is_run(out[pos - 1])) {
            size_t b = pos - 1;     // a run: back to its HEAD, or to the inner row that follows a general first row
            while (fl(out[b]) == 7u && b > 0 && is_run(out[b - 1]) && om[b - 1].mask == om[b].mask) --b;
            const int nm = __builtin_popcountll(om[b].mask);
            const int len = (int)(pos - b);
            // register runs only (<= 4 paths: rg_codes.hpp RG_SWEEP16_KRUN).  Round 4 also admitted gather runs here; since
            // round 5 the keys of a row in progress wait in the LDS words of the gather table (k_sweep16: keys_ld / keys_st),
            // so a run that uses that table must not lie between the groups of one row.
            (void)len;
            const bool safe = nm <= 4;
            if (!safe) break;
            runs.push_back({b, pos});
            pos = b;
        }
        std::reverse(runs.begin(), runs.end());
        std::vector<int> tail_of(runs.size(), -1);
        std::vector<bool> taken(e - q, false);
        for (size_t j = q; j < e; ++j) {
            if (!meta[j].low || __builtin_popcountll(meta[j].mask) > 4) continue;
            for (size_t u = 0; u < runs.size(); ++u) {
                const StepMeta& last = om[runs[u].second - 1];
                if (tail_of[u] < 0 && last.row == meta[j].pred && last.mask == meta[j].mask) { tail_of[u] = (int)j; taken[j - q] = true; break; }
            }
        }
        // rebuild the block with the tails behind their runs, then the other groups of the row
        std::vector<StepRec> blk;
        std::vector<StepMeta> bm;
        std::vector<std::pair<size_t, bool>> rowrecs;      // (index in blk, tail?) of this row's records, in the new order
        for (size_t u = 0; u < runs.size(); ++u) {
            for (size_t x = runs[u].first; x < runs[u].second; ++x) { blk.push_back(out[x]); bm.push_back(om[x]); }
            if (tail_of[u] >= 0) { rowrecs.push_back({blk.size(), true}); blk.push_back(in[tail_of[u]]); bm.push_back(meta[tail_of[u]]); }
        }
        for (size_t j = q; j < e; ++j)
            if (!taken[j - q]) { rowrecs.push_back({blk.size(), false}); blk.push_back(in[j]); bm.push_back(meta[j]); }
        for (size_t z = 0; z < rowrecs.size(); ++z) {
            StepRec& r = blk[rowrecs[z].first];
            const bool tail = rowrecs[z].second;
            const unsigned f = (z == 0 ? 1u : 0u) | (z + 1 == rowrecs.size() ? 2u : 0u) | (tail ? 4u : 0u);
            unsigned x = (unsigned)r.x & ~(7u << 23);
            if (tail) x &= 0x03ffffffu;                     // zero run field: the mark of a tail (its alpha is its lowest member)
            r.x = (int)(x | (f << 23));
        }
        out.resize(pos);
        om.resize(pos);
        out.insert(out.end(), blk.begin(), blk.end());
        om.insert(om.end(), bm.begin(), bm.end());
        q = e;
    }
};
// WIDE RUNS (more than 64 paths, k_sweep16 record variants; round 6).  In the plain table only a row with ONE ENTRY starts a register
// run, and in a graph with more than 64 paths the group of nearly every row spans pages (an alpha entry + continuation entries): no
// runs at all, every member row loaded and stored in every row.  This form of the table — it takes the place of the split table,
// which exists only up to 64 paths — flags the alpha entry of a row with ONE GROUP led by its lowest member as the HEAD / inner row
// of a run like the narrow tables do (run field = rows left in the segment, this one included, capped at 63), continuation entries
// or not: the kernel gathers the members that are still needed from the alpha entry and the continuation entries behind it,
// and when at most KRUN are left it runs the segment's rows with those in registers, stepping over the continuation entries.
auto wide_runs = [&](const std::vector<StepRec>& in, std::vector<StepRec>& out) {
    out = in;
    std::vector<size_t> alpha_of_row;          // index of the flagged alpha entry per flagged row, in table order
    size_t q = 0;
    while (q < in.size()) {
        size_t e = q;
        while (e < in.size() && meta[e].row == meta[q].row) ++e;
        bool one_group = in[q].y >= 0 && meta[q].low;
        for (size_t j = q + 1; j < e; ++j) one_group = one_group && in[j].y < 0;
        // (the alpha must be the lowest path of the WHOLE group: the entries of a group come in ascending page order, the
        // alpha entry first, so its page must not lie above a continuation entry's)
        const unsigned pg0 = ((unsigned)in[q].y >> 29) & 3u;
        for (size_t j = q + 1; j < e; ++j) one_group = one_group && ((((unsigned)in[j].y >> 29) & 3u) > pg0);
        if (one_group) {
            const int i = meta[q].row;
            const bool inner = forward ? (h.node_id[i] == h.node_id[i - 1] && i > 1) : (h.node_id[i] == h.node_id[i + 1]);
            const unsigned fl = inner ? 7u : 4u;
            out[q].x = (int)(((unsigned)out[q].x & ~(7u << 23)) | (fl << 23));
        }
        q = e;
    }
    // run fields (the narrow rule, with the continuation entries stepped over)
    int left = 0;
    for (size_t t = out.size(); t-- > 0;) {
        if (out[t].y < 0) continue;                                   // a continuation entry: part of the row of its alpha entry
        const bool inner = ((unsigned)out[t].x >> 23) & 4u;
        left = inner ? std::min(left + 1, 63) : 0;
        if (inner) {
            out[t].x = (int)(((unsigned)out[t].x & 0x03ffffffu) | ((unsigned)left << 26));
            if ((((unsigned)out[t].x >> 23) & 7u) == 4u) left = 0;
        }
    }
};
// PATH RETIREMENT (k_sweep16): per evaluation point e (record e << retire_shift) and path k, the union of the member masks of the
// groups k LEADS in the records from there on (groups with other members only): a path that is hopeless for a read may
// stop being computed once no path that is still needed appears in that union
auto lead_table = [&](const std::vector<StepRec>& recs) {
    // Layout: [evaluation point][path, padded to whole 64-path pages][word of the member set]: out[(e * PP + k) * NW + w], NW =
    // 64-bit words per path set (1 up to 64 paths: the round-4 layout), PP = 64 * NW.  A group that spans pages is an alpha
    // entry followed by continuation entries (y < 0) with the members of the other pages: walking backwards the continuation
    // entries come first and are held until their alpha entry closes the group.
    const size_t EV = (size_t)1 << T.retire_shift;                    // records per evaluation point
    const size_t E = recs.size() / EV + 2;
    const size_t NW = (size_t)((h.P + 63) / 64), PP = 64 * NW;
    std::vector<unsigned long long> out(E * PP * NW, 0ull);
    std::vector<unsigned long long> cur(PP * NW, 0ull);
    unsigned long long grp[RG_PW] = {};
    for (size_t t = recs.size(); t-- > 0;) {
        const unsigned x = (unsigned)recs[t].x;
        const unsigned long long mask = ((unsigned long long)(unsigned)recs[t].w << 32) | (unsigned)recs[t].z;
        const size_t page = ((unsigned)recs[t].y >> 29) & 3u;
        grp[page] |= mask;
        if (recs[t].y >= 0) {                                         // the group's alpha entry: the group is complete
            int members = 0;
            for (size_t w = 0; w < NW; ++w) members += __builtin_popcountll(grp[w]);
            if (mask && members > 1) {
                const size_t alpha = page * 64 + (size_t)(((x >> 23) & 4u) ? __builtin_ctzll(mask) : (int)((x >> 26) & 63u));
                for (size_t w = 0; w < NW; ++w) cur[alpha * NW + w] |= grp[w];
            }
            for (size_t w = 0; w < RG_PW; ++w) grp[w] = 0ull;
        }
        if (t % EV == 0) std::copy(cur.begin(), cur.end(), out.begin() + (long long)((t / EV) * PP * NW));
    }
    return out;
};

    steps(goff_, groups_, forward, T.plain);
    T.retire_shift = retire_shift_option().load();
    T.members = 0;
    for (const StepRec& r : T.plain) T.members += (unsigned long long)(__builtin_popcount((unsigned)r.z) + __builtin_popcount((unsigned)r.w));
    T.lead_plain.clear();
    T.lead_split.clear();
    T.split.clear();
    T.lead_plain = lead_table(T.plain);
    if (want_split) {
        if (h.P <= 64) split_tails(T.plain, T.split);
        else wide_runs(T.plain, T.split);
        T.lead_split = lead_table(T.split);
    }
}

}  // namespace rg
