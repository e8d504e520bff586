// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp header).
// Pathwise modes, SECOND restatement: absolute-score formulation.
// Structurally different from orc_pathwise.cpp: no L x (n+1) x P matrix, one rolling row per
// path, 2-bit direction words per (edge group, column), per-(row,column) best member for the
// recombination search, layers of the chosen paths rebuilt from the direction words for the
// traceback.  Agreement between the two restatements on randomised graphs is one of the pins
// listed in the header of orc_common.hpp.  It is also the "pruned" CPU baseline timed by
// bench.py (kind "port").
#include <algorithm>
#include <climits>

#include "orc_common.hpp"

namespace orc {
namespace {

struct Group {
    size_t pred;
    std::vector<size_t> members;
    size_t ga;   // group alpha: alphas[pred] if member, else alphas[row] if member, else lowest member
    size_t slot; // index of this (row,group) in the direction-word store
};

struct Program {  // per direction
    std::vector<std::vector<Group>> rows;  // indexed by row
    size_t nslots = 0;
};

Program build_program(const PathGraph& g, bool forward) {
    const size_t L = g.lnz.size(), P = g.paths_number;
    Program pr;
    pr.rows.resize(L);
    auto mk = [&](size_t i, size_t p, const std::vector<uint8_t>& paths) {
        Group gr;
        gr.pred = p;
        for (size_t k = 0; k < P; ++k)
            if (paths[k] && g.paths_nodes[i][k]) gr.members.push_back(k);
        if (gr.members.empty()) return;
        auto has = [&](size_t k) { return std::find(gr.members.begin(), gr.members.end(), k) != gr.members.end(); };
        if (has(g.alphas[p])) gr.ga = g.alphas[p];
        else if (has(g.alphas[i])) gr.ga = g.alphas[i];
        else gr.ga = gr.members[0];
        gr.slot = pr.nslots++;
        pr.rows[i].push_back(gr);
    };
    for (size_t i = 1; i + 1 < L; ++i) {
        if (g.nwp[i]) {
            auto it = g.pred_hash.find(i);
            if (it != g.pred_hash.end())
                for (auto& pk : it->second) mk(i, pk.first, pk.second);
        } else {
            size_t p = forward ? i - 1 : i + 1;
            mk(i, p, g.paths_nodes[p]);
        }
    }
    return pr;
}

struct Pass {
    // outputs of one DP sweep
    std::vector<std::vector<int>> roll;   // final rolling rows per path
    std::vector<int> best_val;            // [L*W] best member value (INT_MIN if row has no valid winner)
    std::vector<uint8_t> best_path;       // [L*W]
    std::vector<uint32_t> dirs;           // [nslots * words] 2 bits per column
    size_t words = 0;
    std::vector<int> sink_val;            // forward only: A[pred of F][n][k] (INT_MIN if k not registered)
    std::vector<size_t> sink_row;
    std::vector<int> endval;              // semiglobal: A[i][n][k] of every member (INT_MIN elsewhere), [L*P]
};

enum : uint32_t { DIR_D = 1, DIR_U = 2, DIR_L = 3 };

void sweep(const PathGraph& g, const Program& pr, const std::string& seq, const Scores& sc, bool forward,
           bool want_best, Pass& out, bool semi = false) {
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size(), P = g.paths_number;
    out.words = (W + 15) / 16;
    out.dirs.assign(pr.nslots * out.words, 0);
    if (want_best) { out.best_val.assign(L * W, INT_MIN); out.best_path.assign(L * W, 0); }
    out.roll.assign(P, std::vector<int>(W, 0));
    if (semi && forward) out.endval.assign(L * P, INT_MIN);
    // row 0 (forward) / row L-1 (reverse): gap-only row, identical for every path
    {
        std::vector<int> r0(W, 0);
        if (forward) for (size_t j = 1; j < W; ++j) r0[j] = r0[j - 1] + sc.get(seq[j], '-');
        else for (size_t j = W - 1; j-- > 1;) r0[j] = r0[j + 1] + sc.get(seq[j], '-');
        for (size_t k = 0; k < P; ++k) out.roll[k] = r0;
    }
    std::vector<int> knm(L, -1);  // highest NON-member path id of each row (its dpm entry stays 0)
    for (size_t i = 0; i < L; ++i)
        for (size_t k = 0; k < P; ++k)
            if (!g.paths_nodes[i][k]) knm[i] = (int)k;

    std::vector<int> na(W), tmp(W);
    std::vector<uint8_t> dir(W);
    auto do_row = [&](size_t i) {
        const int g_i = sc.get(lnz[i], '-');
        for (const Group& gr : pr.rows[i]) {
            std::vector<int>& ra = out.roll[gr.ga];
            // group alpha row + directions
            if (forward) {
                na[0] = semi ? 0 : ra[0] + g_i; dir[0] = DIR_U;
                for (size_t j = 1; j < W; ++j) {
                    int d = ra[j - 1] + sc.get(lnz[i], seq[j]);
                    int u = ra[j] + g_i;
                    int l = na[j - 1] + sc.get(seq[j], '-');
                    int b = std::max(std::max(d, u), l);
                    na[j] = b;
                    dir[j] = b == d ? DIR_D : b == u ? DIR_U : DIR_L;
                }
            } else {
                na[W - 1] = semi ? 0 : ra[W - 1] + g_i; dir[W - 1] = DIR_U;
                for (size_t j = W - 1; j-- > 1;) {
                    int d = ra[j + 1] + sc.get(lnz[i], seq[j]);
                    int u = ra[j] + g_i;
                    int l = na[j + 1] + sc.get(seq[j], '-');
                    int b = std::max(std::max(d, u), l);
                    na[j] = b;
                    dir[j] = b == d ? DIR_D : b == u ? DIR_U : DIR_L;
                }
            }
            // every other member follows the alpha's directions with its own values
            for (size_t k : gr.members) {
                if (k == gr.ga) continue;
                std::vector<int>& rk = out.roll[k];
                if (forward) {
                    tmp[0] = semi ? 0 : rk[0] + g_i;
                    for (size_t j = 1; j < W; ++j)
                        tmp[j] = dir[j] == DIR_D ? rk[j - 1] + sc.get(lnz[i], seq[j])
                               : dir[j] == DIR_U ? rk[j] + g_i
                                                 : tmp[j - 1] + sc.get(seq[j], '-');
                } else {
                    tmp[W - 1] = semi ? 0 : rk[W - 1] + g_i;
                    for (size_t j = W - 1; j-- > 1;)
                        tmp[j] = dir[j] == DIR_D ? rk[j + 1] + sc.get(lnz[i], seq[j])
                               : dir[j] == DIR_U ? rk[j] + g_i
                                                 : tmp[j + 1] + sc.get(seq[j], '-');
                    tmp[0] = 0;
                }
                rk.swap(tmp);
            }
            if (!forward) na[0] = 0;
            ra.swap(na);
            uint32_t* dw = &out.dirs[gr.slot * out.words];
            for (size_t j = 0; j < W; ++j) dw[j >> 4] |= (uint32_t)dir[j] << ((j & 15) * 2);
        }
        if (semi && forward) {
            for (const Group& gr : pr.rows[i])
                for (size_t k : gr.members) out.endval[i * P + k] = out.roll[k][W - 1];
        }
        if (want_best) {
            // argmax over ALL P entries of (value, path id) where non-members hold 0
            // (pathwise_alignment_recombination.rs:809-830); row is usable only if the winner is a member
            for (size_t j = 0; j < W; ++j) {
                int bv = INT_MIN; int bk = -1;
                for (const Group& gr : pr.rows[i])
                    for (size_t k : gr.members) {
                        int v = out.roll[k][j];
                        if (v > bv || (v == bv && (int)k > bk)) { bv = v; bk = (int)k; }
                    }
                if (bk < 0) continue;
                bool valid = knm[i] < 0 || bv > 0 || (bv == 0 && bk > knm[i]);
                if (valid) { out.best_val[i * W + j] = bv; out.best_path[i * W + j] = (uint8_t)bk; }
            }
        }
    };
    if (forward) for (size_t i = 1; i + 1 < L; ++i) do_row(i);
    else for (size_t i = L - 2; i >= 1; --i) do_row(i);
    if (forward) {
        out.sink_val.assign(P, INT_MIN);
        out.sink_row.assign(P, 0);
    }
}

// rebuild the absolute layer of one path from the stored direction words: rows on the path only
struct Layer {
    std::vector<long> row_index;            // row -> index into rows (or -1)
    std::vector<std::vector<int>> rows;
    const std::vector<int>& at(size_t i) const { return rows[(size_t)row_index[i]]; }
};

Layer rebuild_layer(const PathGraph& g, const Program& pr, const Pass& ps, const std::string& seq,
                    const Scores& sc, bool forward, size_t path, bool semi = false) {
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size();
    Layer ly;
    ly.row_index.assign(L, -1);
    std::vector<int> cur(W, 0), nxt(W);
    if (forward) for (size_t j = 1; j < W; ++j) cur[j] = cur[j - 1] + sc.get(seq[j], '-');
    else for (size_t j = W - 1; j-- > 1;) cur[j] = cur[j + 1] + sc.get(seq[j], '-');
    auto do_row = [&](size_t i) {
        for (const Group& gr : pr.rows[i]) {
            if (std::find(gr.members.begin(), gr.members.end(), path) == gr.members.end()) continue;
            const uint32_t* dw = &ps.dirs[gr.slot * ps.words];
            const int g_i = sc.get(lnz[i], '-');
            auto dirat = [&](size_t j) { return (dw[j >> 4] >> ((j & 15) * 2)) & 3u; };
            if (forward) {
                nxt[0] = semi ? 0 : cur[0] + g_i;
                for (size_t j = 1; j < W; ++j) {
                    uint32_t d = dirat(j);
                    nxt[j] = d == DIR_D ? cur[j - 1] + sc.get(lnz[i], seq[j]) : d == DIR_U ? cur[j] + g_i
                                                                                           : nxt[j - 1] + sc.get(seq[j], '-');
                }
            } else {
                nxt[W - 1] = semi ? 0 : cur[W - 1] + g_i;
                for (size_t j = W - 1; j-- > 1;) {
                    uint32_t d = dirat(j);
                    nxt[j] = d == DIR_D ? cur[j + 1] + sc.get(lnz[i], seq[j]) : d == DIR_U ? cur[j] + g_i
                                                                                           : nxt[j + 1] + sc.get(seq[j], '-');
                }
                nxt[0] = 0;
            }
            cur = nxt;
            ly.row_index[i] = (long)ly.rows.size();
            ly.rows.push_back(cur);
        }
    };
    if (forward) {
        ly.row_index[0] = 0;
        ly.rows.push_back(cur);
        for (size_t i = 1; i + 1 < L; ++i) do_row(i);
    } else {
        for (size_t i = L - 2; i >= 1; --i) do_row(i);
    }
    return ly;
}

std::vector<uint64_t> dedup(const std::vector<uint64_t>& v) {
    std::vector<uint64_t> o;
    for (uint64_t x : v) if (o.empty() || o.back() != x) o.push_back(x);
    return o;
}

bool pred_on_path(const PathGraph& g, size_t i, size_t path, size_t& out) {
    bool found = false;
    auto it = g.pred_hash.find(i);
    if (it == g.pred_hash.end()) return false;
    for (auto& pk : it->second)
        if (pk.second[path]) { out = pk.first; found = true; }
    return found;
}

void path_len_start_end(const std::vector<uint64_t>& ids, size_t start, size_t end, size_t pl,
                        size_t& path_len, size_t& path_start, size_t& path_end) {
    path_start = 0;
    if (start > 0) {
        uint64_t f = ids[start]; size_t c = start - 1;
        while (c > 0 && ids[c] == f) { c -= 1; path_start += 1; }
    }
    path_end = pl > 0 ? path_start + pl - 1 : 0;
    size_t eo = 0;
    if (end > 0) {
        uint64_t l = ids[end]; size_t c = end + 1;
        while (c < ids.size() - 1 && ids[c] == l) { c += 1; eo += 1; }
    }
    path_len = path_end + eo + 1;
}

// forward-layer traceback from (i, j) back to the source; appends in walk order
void trace_forward(const PathGraph& g, const Layer& ly, const std::string& seq, const Scores& sc, size_t path,
                   size_t& i, size_t& j, std::vector<char>& cigar, std::vector<uint64_t>& hia,
                   std::vector<char>& pseq, size_t& plen, bool pad_to_source = true) {
    const std::string& lnz = g.lnz;
    while (i > 0 && j > 0) {
        size_t p = i - 1;
        if (g.nwp[i]) { size_t q; if (pred_on_path(g, i, path, q)) p = q; }
        int d = ly.at(p)[j - 1] + sc.get(lnz[i], seq[j]);
        int u = ly.at(p)[j] + sc.get(lnz[i], '-');
        int l = ly.at(i)[j - 1] + sc.get('-', seq[j]);
        int mx = std::max(std::max(d, u), l);
        if (mx == d) {
            cigar.push_back(lnz[i] != seq[j] ? 'd' : 'D');
            hia.push_back(g.nodes_id_pos[i]); pseq.push_back(lnz[i]); i = p; j -= 1; plen += 1;
        } else if (mx == u) {
            cigar.push_back('U'); hia.push_back(g.nodes_id_pos[i]); pseq.push_back(lnz[i]); i = p; plen += 1;
        } else { cigar.push_back('L'); j -= 1; }
    }
    while (j > 0) { cigar.push_back('L'); j -= 1; }
    while (pad_to_source && i > 0) {
        cigar.push_back('U'); hia.push_back(g.nodes_id_pos[i]); pseq.push_back(lnz[i]);
        size_t p = i - 1;
        if (g.nwp[i]) { size_t q; if (pred_on_path(g, i, path, q)) p = q; }
        i = p; plen += 1;
    }
}

}  // namespace

// =================================================================================
Result m4_abs(const std::string& seq, const std::string& name, const PathGraph& g, const Scores& sc) {
    Result res;
    sc.panicked = false;
    const size_t L = g.lnz.size(), W = seq.size(), P = g.paths_number;
    Program pr = build_program(g, true);
    Pass ps;
    sweep(g, pr, seq, sc, true, false, ps);
    // pathwise_alignment.rs:305-325: results default 0, ending_nodes default 0
    std::vector<int> results(P, 0);
    std::vector<size_t> ending(P, 0);
    for (auto& pk : g.pred_hash.at(L - 1))
        for (size_t k = 0; k < P; ++k)
            if (pk.second[k]) { results[k] = ps.roll[k][W - 1]; ending[k] = pk.first; }
    size_t bp = 0;
    for (size_t k = 0; k < P; ++k)
        if (std::make_pair(results[k], k) >= std::make_pair(results[bp], bp)) bp = k;
    size_t ending_node = ending[bp];
    Layer ly = rebuild_layer(g, pr, ps, seq, sc, true, bp);
    size_t i = ending_node, j = W - 1, plen = 0;
    int score = ly.at(i)[j];
    res.score = score;
    std::vector<char> cigar, pseq; std::vector<uint64_t> hia;
    trace_forward(g, ly, seq, sc, bp, i, j, cigar, hia, pseq, plen);
    std::reverse(cigar.begin(), cigar.end());
    std::reverse(pseq.begin(), pseq.end());
    GAF gaf;
    gaf.query_name = name; gaf.query_length = W - 1; gaf.query_start = 0; gaf.query_end = W - 2; gaf.strand = '+';
    auto dd = dedup(hia); std::reverse(dd.begin(), dd.end()); gaf.path = dd;
    path_len_start_end(g.nodes_id_pos, 0, ending_node, plen, gaf.path_length, gaf.path_start, gaf.path_end);
    gaf.alignment_block_length = "*"; gaf.mapping_quality = "*";
    gaf.comments = build_cigar(cigar) + ", best path: " + std::to_string(bp) + ", score: " + std::to_string(score) +
                   "\t" + std::string(pseq.begin(), pseq.end());
    res.out = gaf.to_string() + "\n";
    res.would_panic = sc.panicked;
    return res;
}

// =================================================================================
Result m8_abs(const std::string& seq, const std::string& name, const PathGraph& g, const PathGraph& rg,
              const std::vector<int64_t>& dfs, const std::vector<int64_t>& dfe, const Scores& sc, int brc,
              float mrc, float rbw) {
    Result res;
    sc.panicked = false;
    if (brc < 0 || mrc < 0) { res.would_panic = true; return res; }
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size(), P = g.paths_number;
    const auto& ids = g.nodes_id_pos;
    Program pf = build_program(g, true), prv = build_program(rg, false);
    std::string r_seq = seq.substr(1) + "F";
    Pass F, R;
    sweep(g, pf, seq, sc, true, true, F);
    sweep(rg, prv, r_seq, sc, false, true, R);

    // seed (pathwise_alignment_recombination.rs:778-788): strict '<' over F's preds then paths ascending
    bool have = false; int mx = 0; size_t bp = 0;
    for (auto& pk : g.pred_hash.at(L - 1))
        for (size_t k = 0; k < P; ++k)
            if (pk.second[k]) {
                int v = F.roll[k][W - 1];
                if (!have || mx < v) { mx = v; bp = k; have = true; }
            }
    if (!have) { res.would_panic = true; return res; }
    float curr = (float)mx;
    size_t fbp = bp, rbp = bp, fen = 0, rsn = 0, rec_col = 0;
    bool onedge = false; int rec_penalty = 0;
    int oob = std::max((int)((float)W * (1.0f - rbw) / 2.0f), 1);
    auto dms = [&](size_t a, size_t b) -> int {
        if (a == b) return 0;
        return (int)(std::llabs(dfs[a] - dfs[b]) + std::llabs(dfe[a] - dfe[b]));
    };
    std::vector<size_t> fi, ri;
    for (size_t j = (size_t)oob; j + (size_t)oob < W; ++j) {
        long mfmax = LONG_MIN, wrmax = LONG_MIN;
        for (size_t i = 1; i + 1 < L; ++i) {
            if (F.best_val[i * W + j] != INT_MIN) mfmax = std::max<long>(mfmax, F.best_val[i * W + j]);
            if (R.best_val[i * W + j] != INT_MIN) wrmax = std::max<long>(wrmax, R.best_val[i * W + j]);
        }
        if (mfmax == LONG_MIN || wrmax == LONG_MIN) continue;
        fi.clear(); ri.clear();
        for (size_t i = 1; i + 1 < L; ++i) {
            int a = F.best_val[i * W + j], b = R.best_val[i * W + j];
            if (a != INT_MIN && (long)a + wrmax - brc >= (long)mx) fi.push_back(i);
            if (b != INT_MIN && (long)b + mfmax - brc >= (long)mx) ri.push_back(i);
        }
        for (size_t i : fi) {
            size_t fpk = F.best_path[i * W + j];
            for (size_t r : ri) {
                if (ids[i] == ids[r]) continue;
                size_t rpk = R.best_path[r * W + j];
                if (fpk == rpk) continue;
                float penalty = (float)brc + (mrc * (float)dms(i, r));
                float ns = (float)(F.best_val[i * W + j] + R.best_val[r * W + j]) - penalty;
                bool cond = (i + 1 == L || ids[i] != ids[i + 1]) && ids[r] != ids[r - 1];
                if (ns > curr || (ns == curr && !onedge && cond)) {
                    onedge = cond; curr = ns; fen = i; rsn = r; fbp = fpk; rbp = rpk; rec_col = j;
                    rec_penalty = dms(i, r);
                }
            }
        }
    }

    GAF gaf;
    gaf.query_name = name; gaf.query_length = W - 1; gaf.query_start = 0; gaf.query_end = W - 2; gaf.strand = '+';
    gaf.alignment_block_length = "*"; gaf.mapping_quality = "*";
    std::vector<char> cigar, pseq; std::vector<uint64_t> hia;
    if (fbp == rbp) {
        Layer ly = rebuild_layer(g, pf, F, seq, sc, true, fbp);
        size_t i = 0;
        for (auto& pk : g.pred_hash.at(L - 1)) if (pk.second[fbp]) i = pk.first;
        size_t ending_node = i, j = W - 1, plen = 0;
        int score = ly.at(i)[j];
        res.score = score;
        trace_forward(g, ly, seq, sc, fbp, i, j, cigar, hia, pseq, plen);
        std::reverse(cigar.begin(), cigar.end());
        std::reverse(pseq.begin(), pseq.end());
        auto dd = dedup(hia); std::reverse(dd.begin(), dd.end()); gaf.path = dd;
        path_len_start_end(ids, 0, ending_node, plen, gaf.path_length, gaf.path_start, gaf.path_end);
        gaf.comments = build_cigar(cigar) + ", best path: " + std::to_string(fbp) + ", score: " +
                       std::to_string(score) + "\t" + std::string(pseq.begin(), pseq.end());
    } else {
        Layer lf = rebuild_layer(g, pf, F, seq, sc, true, fbp);
        Layer lr = rebuild_layer(rg, prv, R, r_seq, sc, false, rbp);
        // row L-1 of w is left delta-encoded by absolute_scores (:748): path 0 absolute, others 0
        std::vector<int> wF(W, 0);
        if (rbp == 0) for (size_t j = W - 1; j-- > 1;) wF[j] = wF[j + 1] + sc.get(r_seq[j], '-');
        auto wrow = [&](size_t i) -> const std::vector<int>& { return i == L - 1 ? wF : lr.at(i); };
        size_t rlen = 0, i = rsn, j = rec_col, rev_ending = i;
        while (i > 0 && i < L - 1 && j < W - 1) {
            size_t p = i + 1;
            if (rg.nwp[i]) { size_t q; if (pred_on_path(rg, i, rbp, q)) p = q; }
            int d = wrow(p)[j + 1] + sc.get(lnz[i], r_seq[j]);
            int u = wrow(p)[j] + sc.get(lnz[i], '-');
            int l = wrow(i)[j + 1] + sc.get('-', r_seq[j]);
            int mxv = std::max(std::max(d, u), l);
            rev_ending = i;
            if (mxv == d) {
                cigar.push_back(lnz[i] != r_seq[j] ? 'd' : 'D');
                hia.push_back(ids[i]); pseq.push_back(lnz[i]); i = p; j += 1; rlen += 1;
            } else if (mxv == u) {
                cigar.push_back('U'); hia.push_back(ids[i]); pseq.push_back(lnz[i]); i = p; rlen += 1;
            } else { cigar.push_back('L'); j += 1; }
        }
        while (j < W - 1) { cigar.push_back('L'); j += 1; }
        while (i < L - 1) {
            cigar.push_back('U'); hia.push_back(ids[i]); pseq.push_back(lnz[i]);
            size_t p = i + 1;
            if (rg.nwp[i]) { size_t q; if (pred_on_path(rg, i, rbp, q)) p = q; }
            i = p; rlen += 1;
        }
        std::vector<char> tc, tp; std::vector<uint64_t> th;
        size_t plen = 0; i = fen; j = rec_col;
        trace_forward(g, lf, seq, sc, fbp, i, j, tc, th, tp, plen);
        size_t rec_edge = tp.size() - 1;
        std::reverse(tc.begin(), tc.end()); tc.insert(tc.end(), cigar.begin(), cigar.end());
        std::reverse(th.begin(), th.end()); th.insert(th.end(), hia.begin(), hia.end());
        std::reverse(tp.begin(), tp.end()); tp.insert(tp.end(), pseq.begin(), pseq.end());
        gaf.path = dedup(th);
        // utils.rs:256-323
        {
            size_t path_start = 0;  // start == 0
            size_t fpe = plen > 0 ? path_start + plen - 1 : 0, feo = 0;
            if (fen > 0) { uint64_t l = ids[fen]; size_t c = fen + 1; while (c < ids.size() - 1 && ids[c] == l) { c++; feo++; } }
            size_t fpl = fpe + feo + 1, rps = 0;
            if (rsn > 0) { uint64_t f = ids[rsn]; size_t c = rsn - 1; while (c > 0 && ids[c] == f) { c--; rps++; } }
            size_t rpe = rlen > 0 ? rps + rlen - 1 : 0;
            size_t path_end = fpl + rpe, eo = 0;
            if (rev_ending > 0) { uint64_t l = ids[rev_ending]; size_t c = rev_ending + 1; while (c < ids.size() - 1 && ids[c] == l) { c++; eo++; } }
            gaf.path_length = fpl + (rpe + eo + 1); gaf.path_start = path_start; gaf.path_end = path_end;
        }
        auto noff = [&](size_t node) { uint64_t h = ids[node]; if (!h) return 0; size_t c = node; int o = 0; while (ids[c - 1] == h) { c--; o++; } return o; };
        gaf.comments = build_cigar(tc) + ", recombination path " + std::to_string(fbp) + " " + std::to_string(rbp) +
                       ", nodes " + std::to_string(ids[fen]) + "[" + std::to_string(noff(fen)) + "] " +
                       std::to_string(ids[rsn]) + "[" + std::to_string(noff(rsn)) + "], score: " + f32_display(curr) +
                       ", displacement: " + std::to_string(rec_penalty) + "\t" + std::string(tp.begin(), tp.end()) +
                       "\t" + std::to_string(rec_edge);
        res.score = (int)curr;
    }
    res.out = gaf.to_string() + "\n";
    res.would_panic = sc.panicked;
    return res;
}

// =================================================================================
// semiglobal modes, absolute-score form
Result m5_abs(const std::string& seq, const std::string& name, const PathGraph& g, const Scores& sc) {
    Result res;
    sc.panicked = false;
    const size_t L = g.lnz.size(), W = seq.size(), P = g.paths_number;
    Program pr = build_program(g, true);
    Pass ps;
    sweep(g, pr, seq, sc, true, false, ps, true);
    // best_ending_node (pathwise_alignment_semiglobal.rs:244-277): per row lowest path id among the row maxima,
    // across rows the first row with the strictly largest value
    bool have = false; int mx = 0; size_t ending_node = 0, bp = 0;
    for (size_t i = 1; i + 1 < L; ++i) {
        bool hb = false; int bs = 0; size_t bk = 0;
        for (size_t k = 0; k < P; ++k)
            if (g.paths_nodes[i][k]) { int v = ps.endval[i * P + k]; if (!hb || bs < v) { bs = v; bk = k; hb = true; } }
        if (!hb) { res.would_panic = true; return res; }
        if (!have || bs > mx) { mx = bs; ending_node = i; bp = bk; have = true; }
    }
    Layer ly = rebuild_layer(g, pr, ps, seq, sc, true, bp, true);
    size_t i = ending_node, j = W - 1, plen = 0;
    int score = ly.at(i)[j];
    res.score = score;
    std::vector<char> cigar, pseq; std::vector<uint64_t> hia;
    trace_forward(g, ly, seq, sc, bp, i, j, cigar, hia, pseq, plen, false);
    std::reverse(cigar.begin(), cigar.end());
    std::reverse(pseq.begin(), pseq.end());
    GAF gaf;
    gaf.query_name = name; gaf.query_length = W - 1; gaf.query_start = 0; gaf.query_end = W - 2; gaf.strand = '+';
    auto dd = dedup(hia); std::reverse(dd.begin(), dd.end()); gaf.path = dd;
    path_len_start_end(g.nodes_id_pos, i == 0 ? i : i + 1, ending_node, plen, gaf.path_length, gaf.path_start, gaf.path_end);
    gaf.alignment_block_length = "*"; gaf.mapping_quality = "*";
    gaf.comments = build_cigar(cigar) + ", best path: " + std::to_string(bp) + ", score: " + std::to_string(score) + "\t" +
                   std::string(pseq.begin(), pseq.end());
    res.out = gaf.to_string() + "\n";
    res.would_panic = sc.panicked;
    return res;
}

Result m9_abs(const std::string& seq, const std::string& name, const PathGraph& g, const PathGraph& rg,
              const std::vector<int64_t>& dfs, const std::vector<int64_t>& dfe, const Scores& sc, int brc, float mrc,
              float rbw) {
    Result res;
    sc.panicked = false;
    if (brc < 0 || mrc < 0) { res.would_panic = true; return res; }
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size(), P = g.paths_number;
    const auto& ids = g.nodes_id_pos;
    Program pf = build_program(g, true), prv = build_program(rg, false);
    std::string r_seq = seq.substr(1) + "F";
    Pass F, R;
    sweep(g, pf, seq, sc, true, true, F, true);
    sweep(rg, prv, r_seq, sc, false, true, R, true);
    // seed (pathwise_alignment_recombination.rs:789-800): rows 0..L-2 ascending, member paths ascending, strict '<';
    // row 0 holds the all-gap row for every path
    int gapsum = 0;
    for (size_t j = 1; j < W; ++j) gapsum += sc.get(seq[j], '-');
    int mx = gapsum; size_t bp = 0;
    for (size_t i = 1; i + 1 < L; ++i)
        for (size_t k = 0; k < P; ++k)
            if (g.paths_nodes[i][k]) { int v = F.endval[i * P + k]; if (mx < v) { mx = v; bp = k; } }
    float curr = (float)mx;
    size_t fbp = bp, rbp = bp, fen = 0, rsn = 0, rec_col = 0;
    bool onedge = false; int rec_penalty = 0;
    int oob = std::max((int)((float)W * (1.0f - rbw) / 2.0f), 1);
    auto dms = [&](size_t a, size_t b) -> int { return a == b ? 0 : (int)(std::llabs(dfs[a] - dfs[b]) + std::llabs(dfe[a] - dfe[b])); };
    std::vector<size_t> fi, ri;
    for (size_t j = (size_t)oob; j + (size_t)oob < W; ++j) {
        long mfmax = LONG_MIN, wrmax = LONG_MIN;
        for (size_t i = 1; i + 1 < L; ++i) {
            if (F.best_val[i * W + j] != INT_MIN) mfmax = std::max<long>(mfmax, F.best_val[i * W + j]);
            if (R.best_val[i * W + j] != INT_MIN) wrmax = std::max<long>(wrmax, R.best_val[i * W + j]);
        }
        if (mfmax == LONG_MIN || wrmax == LONG_MIN) continue;
        fi.clear(); ri.clear();
        for (size_t i = 1; i + 1 < L; ++i) {
            int a = F.best_val[i * W + j], b = R.best_val[i * W + j];
            if (a != INT_MIN && (long)a + wrmax - brc >= (long)mx) fi.push_back(i);
            if (b != INT_MIN && (long)b + mfmax - brc >= (long)mx) ri.push_back(i);
        }
        for (size_t i : fi) {
            size_t fpk = F.best_path[i * W + j];
            for (size_t r : ri) {
                if (ids[i] == ids[r]) continue;
                size_t rpk = R.best_path[r * W + j];
                if (fpk == rpk) continue;
                float penalty = (float)brc + (mrc * (float)dms(i, r));
                float ns = (float)(F.best_val[i * W + j] + R.best_val[r * W + j]) - penalty;
                bool cond = (i + 1 == L || ids[i] != ids[i + 1]) && ids[r] != ids[r - 1];
                if (ns > curr || (ns == curr && !onedge && cond)) {
                    onedge = cond; curr = ns; fen = i; rsn = r; fbp = fpk; rbp = rpk; rec_col = j; rec_penalty = dms(i, r);
                }
            }
        }
    }
    GAF gaf;
    gaf.query_name = name; gaf.query_length = W - 1; gaf.query_start = 0; gaf.query_end = W - 2; gaf.strand = '+';
    gaf.alignment_block_length = "*"; gaf.mapping_quality = "*";
    std::vector<char> cigar, pseq; std::vector<uint64_t> hia;
    if (fbp == rbp) {
        // ending_node (:885-897): first row of the path with its largest last-column value
        bool hb = false; int bs = 0; size_t en = 0;
        for (size_t i = 1; i + 1 < L; ++i)
            if (g.paths_nodes[i][fbp]) { int v = F.endval[i * P + fbp]; if (!hb || v > bs) { bs = v; en = i; hb = true; } }
        Layer ly = rebuild_layer(g, pf, F, seq, sc, true, fbp, true);
        size_t i = en, j = W - 1, plen = 0;
        int score = ly.at(i)[j];
        res.score = score;
        trace_forward(g, ly, seq, sc, fbp, i, j, cigar, hia, pseq, plen, false);
        std::reverse(cigar.begin(), cigar.end());
        std::reverse(pseq.begin(), pseq.end());
        auto dd = dedup(hia); std::reverse(dd.begin(), dd.end()); gaf.path = dd;
        path_len_start_end(ids, i == 0 ? i : i + 1, en, plen, gaf.path_length, gaf.path_start, gaf.path_end);
        gaf.comments = build_cigar(cigar) + ", best path: " + std::to_string(fbp) + ", score: " + std::to_string(score) + "\t" +
                       std::string(pseq.begin(), pseq.end());
    } else {
        Layer lf = rebuild_layer(g, pf, F, seq, sc, true, fbp, true);
        Layer lr = rebuild_layer(rg, prv, R, r_seq, sc, false, rbp, true);
        std::vector<int> wF(W, 0);
        if (rbp == 0) for (size_t j = W - 1; j-- > 1;) wF[j] = wF[j + 1] + sc.get(r_seq[j], '-');
        auto wrow = [&](size_t i) -> const std::vector<int>& { return i == L - 1 ? wF : lr.at(i); };
        size_t rlen = 0, i = rsn, j = rec_col, rev_ending = i;
        while (i > 0 && i < L - 1 && j < W - 1) {
            size_t p = i + 1;
            if (rg.nwp[i]) { size_t q; if (pred_on_path(rg, i, rbp, q)) p = q; }
            int d = wrow(p)[j + 1] + sc.get(lnz[i], r_seq[j]);
            int u = wrow(p)[j] + sc.get(lnz[i], '-');
            int l = wrow(i)[j + 1] + sc.get('-', r_seq[j]);
            int mxv = std::max(std::max(d, u), l);
            rev_ending = i;
            if (mxv == d) { cigar.push_back(lnz[i] != r_seq[j] ? 'd' : 'D'); hia.push_back(ids[i]); pseq.push_back(lnz[i]); i = p; j += 1; rlen += 1; }
            else if (mxv == u) { cigar.push_back('U'); hia.push_back(ids[i]); pseq.push_back(lnz[i]); i = p; rlen += 1; }
            else { cigar.push_back('L'); j += 1; }
        }
        while (j < W - 1) { cigar.push_back('L'); j += 1; }
        std::vector<char> tc, tp; std::vector<uint64_t> th;
        size_t plen = 0; i = fen; j = rec_col;
        trace_forward(g, lf, seq, sc, fbp, i, j, tc, th, tp, plen, false);
        if (tp.empty()) { res.would_panic = true; return res; }
        size_t rec_edge = tp.size() - 1;
        std::reverse(tc.begin(), tc.end()); tc.insert(tc.end(), cigar.begin(), cigar.end());
        std::reverse(th.begin(), th.end()); th.insert(th.end(), hia.begin(), hia.end());
        std::reverse(tp.begin(), tp.end()); tp.insert(tp.end(), pseq.begin(), pseq.end());
        gaf.path = dedup(th);
        {   // utils.rs:256-323
            const size_t start = i == 0 ? i : i + 1;
            size_t path_start = 0;
            if (start > 0) { uint64_t f = ids[start]; size_t c = start - 1; while (c > 0 && ids[c] == f) { c--; path_start++; } }
            size_t fpe = plen > 0 ? path_start + plen - 1 : 0, feo = 0;
            if (fen > 0) { uint64_t l = ids[fen]; size_t c = fen + 1; while (c < ids.size() - 1 && ids[c] == l) { c++; feo++; } }
            size_t fpl = fpe + feo + 1, rps = 0;
            if (rsn > 0) { uint64_t f = ids[rsn]; size_t c = rsn - 1; while (c > 0 && ids[c] == f) { c--; rps++; } }
            size_t rpe = rlen > 0 ? rps + rlen - 1 : 0;
            size_t path_end = fpl + rpe, eo = 0;
            if (rev_ending > 0) { uint64_t l = ids[rev_ending]; size_t c = rev_ending + 1; while (c < ids.size() - 1 && ids[c] == l) { c++; eo++; } }
            gaf.path_length = fpl + (rpe + eo + 1); gaf.path_start = path_start; gaf.path_end = path_end;
        }
        auto noff = [&](size_t node) { uint64_t h = ids[node]; if (!h) return 0; size_t c = node; int o = 0; while (ids[c - 1] == h) { c--; o++; } return o; };
        gaf.comments = build_cigar(tc) + ", recombination path " + std::to_string(fbp) + " " + std::to_string(rbp) + ", nodes " +
                       std::to_string(ids[fen]) + "[" + std::to_string(noff(fen)) + "] " + std::to_string(ids[rsn]) + "[" +
                       std::to_string(noff(rsn)) + "], score: " + f32_display(curr) + ", displacement: " + std::to_string(rec_penalty) +
                       "\t" + std::string(tp.begin(), tp.end()) + "\t" + std::to_string(rec_edge);
        res.score = (int)curr;
    }
    res.out = gaf.to_string() + "\n";
    res.would_panic = sc.panicked;
    return res;
}

}  // namespace orc
