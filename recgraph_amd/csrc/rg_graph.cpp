// Graph ingestion for librecgraph_hip: GFA text (or already flattened arrays) -> HostGraph.
//
// Behavioural contract (what the arrays must equal), by reference location:
//   linearisation, nwp, pred lists ........ src/graph.rs:31-123
//   r-values ............................... src/utils.rs:103-126
//   row -> segment id ...................... src/utils.rs:144-165, src/pathwise_graph.rs:151-165
//   path masks, alphas, PredHash ........... src/pathwise_graph.rs:135-248
//   reverse PredHash ....................... src/pathwise_graph.rs:250-282
//   distance heuristics for displacement ... src/pathwise_graph.rs:306-354
// Where the reference iterates a HashMap (sinks of F, PredHash edges) this code uses ascending
// row order; single-sink graphs give identical results for every order.
#include <algorithm>
#include <cstring>
#include <map>
#include <unordered_map>

#include "rg_host.hpp"

namespace rg {

thread_local std::string g_last_error;
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

namespace {

struct Segment {
    uint64_t id;
    std::string seq;
    int32_t first = 0, last = 0;  // rows
};

bool parse_u64(const char* b, const char* e, uint64_t& v) {
    if (b == e) return false;
    v = 0;
    for (const char* p = b; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (uint64_t)(*p - '0');
    }
    return true;
}

void finish_linear(HostGraph& g, const std::vector<Segment>& segs) {
    g.lnz = "$";
    g.node_id.assign(1, 0);
    g.seg_off.assign(1, 0);
    for (const Segment& s : segs) {
        int32_t k = 1;
        for (char c : s.seq) {
            g.lnz.push_back(c);
            g.node_id.push_back(s.id);
            g.seg_off.push_back(k++);
        }
    }
    g.lnz.push_back('F');
    g.node_id.push_back(0);
    g.seg_off.push_back(0);
    g.L = (int32_t)g.lnz.size();
}

// utils.rs:103-126 over the CSR arrays
void compute_r_values(HostGraph& g) {
    const int32_t L = g.L;
    std::vector<int64_t> r(L, -1);
    r[L - 1] = 0;
    for (int32_t e = g.pred_off[L - 1]; e < g.pred_off[L]; ++e) r[g.pred_rows[e]] = 0;
    for (int32_t i = L - 2; i >= 1; --i) {
        if (r[i] == -1 || r[i] > r[i + 1] + 1) r[i] = r[i + 1] + 1;
        for (int32_t e = g.pred_off[i]; e < g.pred_off[i + 1]; ++e) {
            int32_t p = g.pred_rows[e];
            if (r[p] == -1 || r[p] > r[i] + 1) r[p] = r[i] + 1;
        }
    }
    g.r_values.resize(L);
    // `*x as usize` of a still -1 entry is 2^64-1 in the reference (utils.rs:125); band arithmetic
    // then casts it `as i32` = -1 (utils.rs:56) and compares seq_len > r (false).  Keep -1: the
    // kernels reproduce both uses from the signed value.
    for (int32_t i = 0; i < L; ++i) g.r_values[i] = (int32_t)r[i];
    g.min_pred.assign(L, 0);
    for (int32_t i = 1; i < L; ++i) {
        if (g.pred_off[i + 1] == g.pred_off[i]) g.min_pred[i] = i - 1;
        else {
            int32_t m = g.pred_rows[g.pred_off[i]];
            for (int32_t e = g.pred_off[i]; e < g.pred_off[i + 1]; ++e) m = std::min(m, g.pred_rows[e]);
            g.min_pred[i] = m;
        }
    }
}

int finish_path_view(HostGraph& g) {
    const int32_t L = g.L, P = g.P;
    // alphas: lowest path id through the row (pathwise_graph.rs:200-205); rows 0 and L-1 -> 0
    g.alphas.assign(L, P + 1);
    for (int32_t i = 0; i < L; ++i)
        if (g.row_mask[i].any()) g.alphas[i] = g.row_mask[i].lowest();
    g.alphas[0] = 0;
    g.alphas[L - 1] = 0;
    for (int32_t i = 1; i + 1 < L; ++i)
        if (!g.row_mask[i].any())
            return fail(RG_ERR_GRAPH, "segment of row " + std::to_string(i) +
                                          " is on no path (the reference indexes out of bounds: pathwise_graph.rs:182)");
    g.pnwp.assign(L, 0);
    for (int32_t i = 0; i < L; ++i) g.pnwp[i] = g.eoff[i + 1] > g.eoff[i];
    // reverse PredHash: for every forward edge (node <- pred, mask): rev[pred] gets (node, mask)
    std::vector<std::map<int32_t, PMask>> rev(L);
    for (int32_t i = 0; i < L; ++i)
        for (int32_t e = g.eoff[i]; e < g.eoff[i + 1]; ++e) rev[g.epred[e]][i] |= g.emask[e];
    g.roff.assign(L + 1, 0);
    g.rsucc.clear();
    g.rmask.clear();
    g.rnwp.assign(L, 0);
    for (int32_t i = 0; i < L; ++i) {
        for (auto& kv : rev[i]) { g.rsucc.push_back(kv.first); g.rmask.push_back(kv.second); }
        g.roff[i + 1] = (int32_t)g.rsucc.size();
        g.rnwp[i] = !rev[i].empty();
    }
    // distance from end on the forward graph (pathwise_graph.rs:330-354)
    {
        std::vector<int64_t> r(L, -1);
        r[L - 1] = 0;
        for (int32_t e = g.eoff[L - 1]; e < g.eoff[L]; ++e) r[g.epred[e]] = 1;
        for (int32_t i = L - 2; i >= 1; --i) {
            if (r[i] == -1 || r[i] > r[i + 1] + 1) r[i] = r[i + 1] + 1;
            if (g.pnwp[i])
                for (int32_t e = g.eoff[i]; e < g.eoff[i + 1]; ++e) {
                    int32_t p = g.epred[e];
                    if (r[p] == -1 || r[p] > r[i] + 1) r[p] = r[i] + 1;
                }
        }
        g.dfe.assign(r.begin(), r.end());
    }
    // distance from start on the reverse graph (pathwise_graph.rs:306-329)
    {
        std::vector<int64_t> r(L, -1);
        r[0] = 0;
        for (int32_t e = g.roff[0]; e < g.roff[1]; ++e) r[g.rsucc[e]] = 1;
        for (int32_t i = 1; i + 1 < L; ++i) {
            if (r[i] == -1 || r[i] > r[i - 1] + 1) r[i] = r[i - 1] + 1;
            if (g.rnwp[i])
                for (int32_t e = g.roff[i]; e < g.roff[i + 1]; ++e) {
                    int32_t p = g.rsucc[e];
                    if (r[p] == -1 || r[p] > r[i] + 1) r[p] = r[i] + 1;
                }
        }
        g.dfs.assign(r.begin(), r.end());
    }
    g.knm.assign(L, -1);
    const PMask all = PMask::first(P);
    for (int32_t i = 0; i < L; ++i) g.knm[i] = all.andnot(g.row_mask[i]).highest();
    // DP programs: group alpha = alphas[pred] if member, else alphas[row] if member,
    // else lowest member
    auto build = [&](bool fwd, std::vector<int32_t>& goff, std::vector<GroupDesc>& groups, int32_t& nslots) {
        goff.assign(L + 1, 0);
        groups.clear();
        nslots = 0;
        for (int32_t i = 0; i < L; ++i) {
            if (i >= 1 && i + 1 < L) {
                auto add = [&](int32_t p, PMask m) {
                    m = m & g.row_mask[i];
                    if (!m.any()) return;
                    const int ap = g.alphas[p], ai = g.alphas[i];
                    const int ga = m.test(ap) ? ap : (m.test(ai) ? ai : m.lowest());
                    GroupDesc d;
                    d.pred = p;
                    d.slot = nslots++;
                    d.page = ga >> 6;
                    d.ga = (uint32_t)(ga & 63);
                    d.mask = m.w[d.page];
                    groups.push_back(d);                       // the alpha's page runs the alpha
                    for (int pg = 0; pg < RG_PW; ++pg)         // members of the other pages follow its directions
                        if (pg != (ga >> 6) && m.w[pg]) {
                            GroupDesc c = d;
                            c.page = pg;
                            c.ga = GroupDesc::GA_CONT;
                            c.mask = m.w[pg];
                            groups.push_back(c);
                        }
                };
                if (fwd) {
                    if (g.pnwp[i]) for (int32_t e = g.eoff[i]; e < g.eoff[i + 1]; ++e) add(g.epred[e], g.emask[e]);
                    else add(i - 1, g.row_mask[i - 1]);
                } else {
                    if (g.rnwp[i]) for (int32_t e = g.roff[i]; e < g.roff[i + 1]; ++e) add(g.rsucc[e], g.rmask[e]);
                    else add(i + 1, g.row_mask[i + 1]);
                }
            }
            goff[i + 1] = (int32_t)groups.size();
        }
    };
    build(true, g.fgoff, g.fgroups, g.fslots);
    build(false, g.rgoff, g.rgroups, g.rslots);
    std::vector<int32_t> cnt(P, 0);
    for (int32_t i = 1; i + 1 < L; ++i)
        for (int32_t k = 0; k < P; ++k) cnt[k] += g.row_mask[i].test(k) ? 1 : 0;
    g.max_path_rows = 0;
    for (int32_t k = 0; k < P; ++k) g.max_path_rows = std::max(g.max_path_rows, cnt[k]);
    g.has_path = true;
    build_path_lists(g);
    return RG_OK;
}

}  // namespace

int build_from_gfa(const char* text, int64_t len, HostGraph& g) {
    std::vector<Segment> segs;
    std::vector<std::pair<uint64_t, uint64_t>> links;
    std::vector<std::vector<uint64_t>> paths;
    std::string path_error;
    const char* p = text;
    const char* end = text + len;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        const char* ln_end = le;
        if (ln_end > p && ln_end[-1] == '\r') --ln_end;
        std::vector<std::pair<const char*, const char*>> f;
        const char* q = p;
        while (true) {
            const char* t = (const char*)memchr(q, '\t', (size_t)(ln_end - q));
            if (!t) { f.emplace_back(q, ln_end); break; }
            f.emplace_back(q, t);
            q = t + 1;
        }
        if (ln_end > p) {
            char kind = *p;
            if (kind == 'S' && f.size() >= 3) {
                Segment s;
                if (!parse_u64(f[1].first, f[1].second, s.id)) return fail(RG_ERR_GFA, "segment names must be unsigned integers (graph.rs:13)");
                s.seq.assign(f[2].first, f[2].second);
                segs.push_back(std::move(s));
            } else if (kind == 'L' && f.size() >= 5) {
                uint64_t a, b;
                if (!parse_u64(f[1].first, f[1].second, a) || !parse_u64(f[3].first, f[3].second, b)) return fail(RG_ERR_GFA, "bad L line");
                if (*f[2].first != '+' || *f[4].first != '+') return fail(RG_ERR_GFA, "only '+' orientations are supported");
                links.emplace_back(a, b);
            } else if (kind == 'P' && f.size() >= 3) {
                // A P line the PathGraph view cannot take does not fail the graph: modes 0-3 only need graph::read_graph
                // (main.rs:29); the reason is kept and reported when a pathwise mode is requested.
                std::vector<uint64_t> steps;
                const char* a = f[2].first;
                while (a < f[2].second) {
                    const char* c = (const char*)memchr(a, ',', (size_t)(f[2].second - a));
                    const char* se = c ? c : f[2].second;
                    if (se > a) {
                        uint64_t v = 0;
                        if (se[-1] != '+') { if (path_error.empty()) path_error = "only '+' path steps are supported"; }
                        else if (!parse_u64(a, se - 1, v)) { if (path_error.empty()) path_error = "bad P line"; }
                        steps.push_back(v);
                    }
                    a = se + 1;
                }
                paths.push_back(std::move(steps));
            }
        }
        p = nl ? nl + 1 : end;
    }
    if (segs.empty()) return fail(RG_ERR_GFA, "no segments");
    std::stable_sort(segs.begin(), segs.end(), [](const Segment& a, const Segment& b) { return a.id < b.id; });
    for (size_t k = 1; k < segs.size(); ++k)
        if (segs[k].id == segs[k - 1].id) return fail(RG_ERR_GFA, "duplicate segment id " + std::to_string(segs[k].id));
    std::unordered_map<uint64_t, int32_t> idx;
    int32_t row = 1;
    for (size_t s = 0; s < segs.size(); ++s) {
        if (segs[s].seq.empty()) return fail(RG_ERR_GFA, "empty segment");
        segs[s].first = row;
        row += (int32_t)segs[s].seq.size();
        segs[s].last = row - 1;
        idx[segs[s].id] = (int32_t)s;
    }
    finish_linear(g, segs);
    const int32_t L = g.L;
    // ---- LnzGraph view ----
    std::vector<std::vector<int32_t>> left(segs.size());
    std::vector<uint8_t> has_out(segs.size(), 0);
    for (auto& l : links) {
        auto a = idx.find(l.first), b = idx.find(l.second);
        if (a == idx.end() || b == idx.end()) return fail(RG_ERR_GFA, "link to unknown segment");
        // every DP row reads rows above it only: a link into the same or an earlier segment (ids are the topological
        // order, graph.rs:32-33) would make the kernels read rows not yet written
        if (segs[a->second].last >= segs[b->second].first)
            return fail(RG_ERR_GRAPH, "link " + std::to_string(l.first) + " -> " + std::to_string(l.second) +
                                          " does not follow the id order (graph not topological)");
        left[b->second].push_back(segs[a->second].last);  // L-line order (graph.rs:75)
        has_out[a->second] = 1;
    }
    g.pred_off.assign(L + 1, 0);
    g.pred_rows.clear();
    {
        std::vector<std::vector<int32_t>> preds(L);
        for (size_t s = 0; s < segs.size(); ++s) {
            if (left[s].empty()) preds[segs[s].first].push_back(0);   // graph.rs:64-74
            else preds[segs[s].first] = left[s];
        }
        for (size_t s = 0; s < segs.size(); ++s)
            if (!has_out[s]) preds[L - 1].push_back(segs[s].last);   // graph.rs:112-123, id order
        for (int32_t i = 0; i < L; ++i) {
            for (int32_t v : preds[i]) g.pred_rows.push_back(v);
            g.pred_off[i + 1] = (int32_t)g.pred_rows.size();
        }
    }
    compute_r_values(g);
    g.has_lnz = true;
    build_rev_ids(g);
    // ---- PathGraph view ----
    if (!paths.empty()) {
        auto no_path_view = [&](const std::string& why) {
            g.has_path = false;
            g.P = 0;
            g.path_error = why;
            return RG_OK;       // LnzGraph view stays usable (modes 0-3); rg_batch_create reports `why` for modes 4+
        };
        if (!path_error.empty()) return no_path_view(path_error);
        if (paths.size() > (size_t)RG_MAXP) return no_path_view("more than 256 paths are not supported by the pathwise kernels");
        g.P = (int32_t)paths.size();
        g.row_mask.assign(L, PMask());
        const PMask all = PMask::first(g.P);
        g.row_mask[0] = all;
        g.row_mask[L - 1] = all;
        std::vector<std::map<int32_t, PMask>> ed(L);
        for (size_t k = 0; k < paths.size(); ++k) {
            const auto& st = paths[k];
            for (size_t s = 0; s < st.size(); ++s) {
                auto it = idx.find(st[s]);
                if (it == idx.end()) return no_path_view("path step on unknown segment");
                const Segment& sg = segs[it->second];
                for (int32_t r = sg.first; r <= sg.last; ++r) g.row_mask[r].set((int)k);
                if (s == 0) ed[sg.first][0].set((int)k);
                else {
                    const Segment& pv = segs[idx[st[s - 1]]];
                    if (pv.last >= sg.first) return no_path_view("path steps must follow the topological id order");
                    ed[sg.first][pv.last].set((int)k);
                    if (s + 1 == st.size()) ed[L - 1][sg.last].set((int)k);   // pathwise_graph.rs:225-232
                }
            }
        }
        g.eoff.assign(L + 1, 0);
        g.epred.clear();
        g.emask.clear();
        for (int32_t i = 0; i < L; ++i) {
            for (auto& kv : ed[i]) { g.epred.push_back(kv.first); g.emask.push_back(kv.second); }
            g.eoff[i + 1] = (int32_t)g.epred.size();
        }
        int rc = finish_path_view(g);
        if (rc != RG_OK) return no_path_view(g_last_error);
    }
    return RG_OK;
}

static int fill_rows_from(const char* lnz, int64_t L, const uint64_t* node_id, HostGraph& g) {
    if (!lnz || L < 3) return fail(RG_ERR_ARG, "lnz too short");
    g.L = (int32_t)L;
    g.lnz.assign(lnz, lnz + L);
    g.node_id.assign(L, 0);
    g.seg_off.assign(L, 0);
    if (node_id) {
        for (int64_t i = 0; i < L; ++i) g.node_id[i] = node_id[i];
        for (int64_t i = 1; i + 1 < L; ++i) g.seg_off[i] = (g.node_id[i] == g.node_id[i - 1] && i > 1) ? g.seg_off[i - 1] + 1 : 1;
    }
    return RG_OK;
}

int build_from_lnz(const char* lnz, int64_t L, const int64_t* pred_off, const int64_t* pred_rows,
                   const uint64_t* node_id, HostGraph& g) {
    if (!pred_off || !pred_rows) return fail(RG_ERR_ARG, "null pred arrays");
    int rc = fill_rows_from(lnz, L, node_id, g);
    if (rc) return rc;
    if (pred_off[0] != 0) return fail(RG_ERR_ARG, "pred_off[0] must be 0");
    for (int64_t i = 0; i < L; ++i)
        if (pred_off[i + 1] < pred_off[i]) return fail(RG_ERR_ARG, "pred_off must be non-decreasing");
    g.pred_off.resize(L + 1);
    for (int64_t i = 0; i <= L; ++i) g.pred_off[i] = (int32_t)pred_off[i];
    g.pred_rows.resize(pred_off[L]);
    if (pred_off[1] != 0) return fail(RG_ERR_GRAPH, "row 0 has no predecessors");
    for (int64_t i = 1; i < L; ++i)
        for (int64_t e = pred_off[i]; e < pred_off[i + 1]; ++e) {
            // a DP row reads rows above it only (a later, equal or F predecessor row is never written when it is read)
            if (pred_rows[e] < 0 || pred_rows[e] >= std::min(i, L - 1))
                return fail(RG_ERR_GRAPH, "predecessor " + std::to_string(pred_rows[e]) + " of row " + std::to_string(i) +
                                              " is not an earlier row (graph not topological)");
            g.pred_rows[e] = (int32_t)pred_rows[e];
        }
    if (g.pred_off[L] == g.pred_off[L - 1]) return fail(RG_ERR_GRAPH, "row F has no predecessor");
    if (!node_id) {
        // no ids given: number segments by their start rows, as utils.rs:144-165 would count them
        uint64_t cur = 0;
        for (int64_t i = 1; i + 1 < L; ++i) {
            if (g.pred_off[i + 1] > g.pred_off[i]) { cur += 1; g.seg_off[i] = 1; }
            else g.seg_off[i] = g.seg_off[i - 1] + 1;
            g.node_id[i] = cur;
        }
    }
    compute_r_values(g);
    g.has_lnz = true;
    build_rev_ids(g);
    return RG_OK;
}

int build_from_path(const char* lnz, int64_t L, int32_t P, const uint64_t* row_mask, const int64_t* edge_off,
                    const int64_t* edge_pred, const uint64_t* edge_mask, const uint64_t* node_id, HostGraph& g) {
    if (!row_mask || !edge_off || !edge_pred || !edge_mask || !node_id) return fail(RG_ERR_ARG, "null path arrays");
    if (P < 1 || P > RG_MAXP) return fail(RG_ERR_GRAPH, "paths_number must be in 1..256");
    int rc = fill_rows_from(lnz, L, node_id, g);
    if (rc) return rc;
    g.P = P;
    const int nw = (P + 63) / 64;                 // mask words per row / edge in the caller's arrays
    auto load = [&](const uint64_t* src) { PMask m; for (int w = 0; w < nw; ++w) m.w[w] = src[w]; return m & PMask::first(P); };
    g.row_mask.resize(L);
    for (int64_t i = 0; i < L; ++i) g.row_mask[i] = load(row_mask + i * nw);
    g.eoff.resize(L + 1);
    std::vector<std::map<int32_t, PMask>> ed(L);
    for (int64_t i = 0; i < L; ++i)
        for (int64_t e = edge_off[i]; e < edge_off[i + 1]; ++e) {
            if (edge_pred[e] < 0 || edge_pred[e] >= L || (i < L - 1 && edge_pred[e] >= i))
                return fail(RG_ERR_GRAPH, "edge predecessor must be an earlier row");
            ed[i][(int32_t)edge_pred[e]] |= load(edge_mask + e * nw);
        }
    g.epred.clear();
    g.emask.clear();
    for (int64_t i = 0; i < L; ++i) {
        g.eoff[i] = (int32_t)g.epred.size();
        for (auto& kv : ed[i]) { g.epred.push_back(kv.first); g.emask.push_back(kv.second); }
    }
    g.eoff[L] = (int32_t)g.epred.size();
    return finish_path_view(g);
}

// text dumps in the same format as the test oracle's orc_graph_dump
std::string dump_graph(const HostGraph& g, int which) {
    std::string s;
    auto bits = [&](const PMask& m) { std::string o; for (int k = 0; k < g.P; ++k) o += m.test(k) ? '1' : '0'; return o; };
    auto csv = [&](const std::vector<int32_t>& v) { std::string o; for (size_t i = 0; i < v.size(); ++i) { if (i) o += ","; o += std::to_string(v[i]); } return o; };
    auto ph = [&](const std::vector<int32_t>& off, const std::vector<int32_t>& pr, const std::vector<PMask>& mk) {
        std::string o;
        for (int32_t i = 0; i < g.L; ++i) {
            if (off[i + 1] == off[i]) continue;
            o += std::to_string(i) + ":";
            for (int32_t e = off[i]; e < off[i + 1]; ++e) { if (e > off[i]) o += ","; o += std::to_string(pr[e]) + "=" + bits(mk[e]); }
            o += ";";
        }
        return o;
    };
    switch (which) {
        case 0: case 10: s = g.lnz; break;
        case 1: for (int32_t i = 0; i < g.L; ++i) s += (g.pred_off[i + 1] > g.pred_off[i]) ? '1' : '0'; break;
        case 2:
            for (int32_t i = 0; i < g.L; ++i) {
                if (g.pred_off[i + 1] == g.pred_off[i]) continue;
                s += std::to_string(i) + ":";
                for (int32_t e = g.pred_off[i]; e < g.pred_off[i + 1]; ++e) { if (e > g.pred_off[i]) s += ","; s += std::to_string(g.pred_rows[e]); }
                s += ";";
            }
            break;
        case 3: for (int32_t i = 0; i + 1 < g.L; ++i) { if (i) s += ","; s += i == 0 ? std::string("-1") : std::to_string(g.node_id[i]); } break;
        case 4: for (int32_t i = 0; i < g.L; ++i) { if (i) s += ","; s += std::to_string((long long)(g.r_values[i] < 0 ? -1LL : (long long)g.r_values[i])); } break;
        case 11: for (int32_t i = 0; i < g.L; ++i) s += g.pnwp[i] ? '1' : '0'; break;
        case 30:    // product only: the formatter's path row lists and whether every step of them equals the PredHash step
            s = g.pl_ok ? "ok;" : "fallback;";
            for (int32_t k = 0; k < g.P && !g.pl_off.empty(); ++k) {
                for (int32_t t = g.pl_off[k]; t < g.pl_off[k + 1]; ++t) { if (t > g.pl_off[k]) s += ","; s += std::to_string(g.pl_row[t]); }
                s += ";";
            }
            break;
        case 31: case 32: case 33: case 34: {
            // product only: the step tables of the pathwise sweeps (31 forward plain, 32 forward split, 33 / 34 reverse):
            // "members=..;points=..;" then one "x:y:z:w" record (hex) per step, ";", then the path-retirement table as
            // "point:path:mask" entries (non-zero masks only)
            if (g.P == 0) break;
            StepTables T;
            build_step_tables(g, which <= 32, true, T);
            const bool split = which == 32 || which == 34;
            const std::vector<StepRec>& recs = split ? T.split : T.plain;
            const std::vector<unsigned long long>& lead = split ? T.lead_split : T.lead_plain;
            char tmp[96];
            // (lead tables: [point][path, padded to whole 64-path pages][word of the member set], rg_steps.cpp)
            const size_t NW = (size_t)((g.P + 63) / 64), PP = 64 * NW;
            snprintf(tmp, sizeof tmp, "members=%llu;points=%zu;", T.members, lead.size() / (PP * NW));
            s = tmp;
            for (size_t t = 0; t < recs.size(); ++t) {
                snprintf(tmp, sizeof tmp, "%s%x:%x:%x:%x", t ? "," : "", (unsigned)recs[t].x, (unsigned)recs[t].y, (unsigned)recs[t].z, (unsigned)recs[t].w);
                s += tmp;
            }
            s += ";";
            bool first = true;
            for (size_t e = 0; e + NW <= lead.size(); e += NW) {
                bool any = false;
                for (size_t w = 0; w < NW; ++w) any = any || lead[e + w];
                if (!any) continue;
                snprintf(tmp, sizeof tmp, "%s%zu:%zu:", first ? "" : ",", (e / NW) / PP, (e / NW) % PP);
                s += tmp;
                bool lead0 = true;      // the member set as ONE hexadecimal number, most significant word first
                for (size_t w = NW; w-- > 0;) {
                    if (lead0 && !lead[e + w] && w > 0) continue;
                    snprintf(tmp, sizeof tmp, lead0 ? "%llx" : "%016llx", lead[e + w]);
                    s += tmp;
                    lead0 = false;
                }
                first = false;
            }
            s += ";";
            break;
        }
        case 12: s = ph(g.eoff, g.epred, g.emask); break;
        case 13: for (int32_t i = 0; i < g.L; ++i) { s += bits(g.row_mask[i]); s += ";"; } break;
        case 14: s = csv(g.alphas); break;
        case 15: for (int32_t i = 0; i < g.L; ++i) { if (i) s += ","; s += std::to_string(g.node_id[i]); } break;
        case 16: for (int32_t i = 0; i < g.L; ++i) s += g.rnwp[i] ? '1' : '0'; break;
        case 17: s = ph(g.roff, g.rsucc, g.rmask); break;
        case 18: s = csv(g.dfs); break;
        case 19: s = csv(g.dfe); break;
        default: break;
    }
    return s;
}

}  // namespace rg
