// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp header).
// Pathwise modes, LITERAL restatement: the dpm[L][n+1][P] matrix keeps the reference's
// alpha/delta encoding (absolute score for alphas[row], deltas for the other member paths)
// and every cell type is transcribed branch by branch.
//   -m 4: src/pathwise_alignment.rs:5-340  + src/pathwise_alignment_output.rs:7-184
//   -m 8: src/pathwise_alignment_recombination.rs:23-883 + src/recombination_output.rs:363-782
#include <algorithm>

#include <chrono>

#include "orc_common.hpp"

namespace orc {

thread_local FaithfulProbe* g_faithful_probe = nullptr;

namespace {

struct Dpm {
    size_t L, W, P;
    std::vector<int> v;
    Dpm(size_t L_, size_t W_, size_t P_) : L(L_), W(W_), P(P_), v(L_ * W_ * P_, 0) {}
    int& at(size_t i, size_t j, size_t k) { return v[(i * W + j) * P + k]; }
    int at(size_t i, size_t j, size_t k) const { return v[(i * W + j) * P + k]; }
};

// One DP fill, forward (di=dj=-1: pathwise_alignment_recombination.rs:436-745 ==
// pathwise_alignment.rs:16-304) or reverse (di=dj=+1: :129-435).  `seq` is the read with '$'
// for forward, get_rev_sequence(read) (:875-883) for reverse.
void fill(Dpm& D, const std::string& seq, const PathGraph& g, const Scores& sc, bool forward, bool semi = false) {
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size(), P = g.paths_number;
    const auto& alphas = g.alphas;
    const auto& pn = g.paths_nodes;
    const long di = forward ? -1 : 1, dj = forward ? -1 : 1;
    const size_t jborder = forward ? 0 : W - 1;

    auto fixup = [&](size_t i, size_t j, const std::map<size_t, std::vector<size_t>>& ad) {
        for (auto& kv : ad) {  // "remove multiple alpha"
            size_t a = kv.first;
            if (a != alphas[i]) {
                D.at(i, j, a) -= D.at(i, j, alphas[i]);
                for (size_t path : kv.second)
                    if (path != a) D.at(i, j, path) += D.at(i, j, a);
            }
        }
    };

    auto border_cell = [&](size_t i, size_t j) {
        int g_i = sc.get(lnz[i], '-');
        if (!g.nwp[i]) {
            size_t ip = (size_t)((long)i + di);
            std::vector<uint8_t> common(P);
            for (size_t k = 0; k < P; ++k) common[k] = pn[i][k] & pn[ip][k];
            if (common[alphas[ip]]) {
                for (size_t path = 0; path < P; ++path)
                    if (common[path]) {
                        if (path == alphas[i]) D.at(i, j, path) = D.at(ip, j, path) + g_i;
                        else D.at(i, j, path) = D.at(ip, j, path);
                    }
            } else {
                D.at(i, j, alphas[i]) = D.at(ip, j, alphas[i]) + D.at(ip, j, alphas[ip]) + g_i;
                for (size_t path = 0; path < P; ++path)
                    if (common[path] && path != alphas[i])
                        D.at(i, j, path) = D.at(ip, j, path) - D.at(ip, j, alphas[i]);
            }
        } else {
            std::map<size_t, std::vector<size_t>> ad;
            for (auto& pk : g.pred_hash.at(i)) {
                size_t p = pk.first;
                std::vector<uint8_t> common(P);
                std::vector<size_t> plist;
                for (size_t k = 0; k < P; ++k) { common[k] = pn[i][k] & pk.second[k]; if (common[k]) plist.push_back(k); }
                if (common[alphas[p]]) {
                    ad[alphas[p]] = plist;
                    D.at(i, j, alphas[p]) = D.at(p, j, alphas[p]) + g_i;
                    for (size_t path : plist)
                        if (path != alphas[p]) D.at(i, j, path) = D.at(p, j, path);
                } else {
                    size_t ta = common[alphas[i]] ? alphas[i] : plist.at(0);
                    ad[ta] = plist;
                    D.at(i, j, ta) = D.at(p, j, alphas[p]) + D.at(p, j, ta) + g_i;
                    for (size_t path : plist)
                        if (path != ta) D.at(i, j, path) = D.at(p, j, path) - D.at(p, j, ta);
                }
            }
            fixup(i, j, ad);
        }
    };

    auto inner_cell = [&](size_t i, size_t j) {
        const size_t jp = (size_t)((long)j + dj);
        const int g_i = sc.get(lnz[i], '-');
        const int s_ij = sc.get(lnz[i], seq[j]);
        const int g_j = sc.get(seq[j], '-');
        if (!g.nwp[i]) {
            size_t ip = (size_t)((long)i + di);
            std::vector<uint8_t> common(P);
            for (size_t k = 0; k < P; ++k) common[k] = pn[i][k] & pn[ip][k];
            const size_t ai = alphas[i], ap = alphas[ip];
            if (common[ap]) {
                int u = D.at(ip, j, ap) + g_i;
                int d = D.at(ip, jp, ap) + s_ij;
                int l = D.at(i, jp, ai) + g_j;
                int best = std::max(std::max(d, u), l);
                D.at(i, j, ai) = best;
                for (size_t path = 0; path < P; ++path)
                    if (common[path] && path != ai) {
                        if (best == d) D.at(i, j, path) = D.at(ip, jp, path);
                        else if (best == u) D.at(i, j, path) = D.at(ip, j, path);
                        else D.at(i, j, path) = D.at(i, jp, path);
                    }
            } else {
                int u = D.at(ip, j, ap) + D.at(ip, j, ai) + g_i;
                int d = D.at(ip, jp, ap) + D.at(ip, jp, ai) + s_ij;
                int l = D.at(i, jp, ai) + g_j;
                int best = std::max(std::max(d, u), l);
                D.at(i, j, ai) = best;
                for (size_t path = 0; path < P; ++path)
                    if (common[path] && path != ai) {
                        if (best == d) D.at(i, j, path) = D.at(ip, jp, path) - D.at(ip, jp, ai);
                        else if (best == u) D.at(i, j, path) = D.at(ip, j, path) - D.at(ip, j, ai);
                        else D.at(i, j, path) = D.at(i, jp, path);
                    }
            }
        } else {
            std::map<size_t, std::vector<size_t>> ad;
            const size_t ai = alphas[i];
            for (auto& pk : g.pred_hash.at(i)) {
                size_t p = pk.first;
                std::vector<uint8_t> common(P);
                std::vector<size_t> plist;
                for (size_t k = 0; k < P; ++k) { common[k] = pn[i][k] & pk.second[k]; if (common[k]) plist.push_back(k); }
                const size_t ap = alphas[p];
                if (common[ap]) {
                    ad[ap] = plist;
                    int u = D.at(p, j, ap) + g_i;
                    int d = D.at(p, jp, ap) + s_ij;
                    int l = (ai == ap) ? D.at(i, jp, ap) + g_j : D.at(i, jp, ap) + D.at(i, jp, ai) + g_j;
                    int best = std::max(std::max(d, u), l);
                    D.at(i, j, ap) = best;
                    for (size_t path : plist)
                        if (path != ap) {
                            if (best == d) D.at(i, j, path) = D.at(p, jp, path);
                            else if (best == u) D.at(i, j, path) = D.at(p, j, path);
                            else if (ap == ai) D.at(i, j, path) = D.at(i, jp, path);
                            else D.at(i, j, path) = D.at(i, jp, path) - D.at(i, jp, ap);
                        }
                } else {
                    size_t ta = common[ai] ? ai : plist.at(0);
                    ad[ta] = plist;
                    int u = D.at(p, j, ap) + D.at(p, j, ta) + g_i;
                    int d = D.at(p, jp, ap) + D.at(p, jp, ta) + s_ij;
                    int l = (ai == ta) ? D.at(i, jp, ta) + g_j : D.at(i, jp, ta) + D.at(i, jp, ai) + g_j;
                    int best = std::max(std::max(d, u), l);
                    D.at(i, j, ta) = best;
                    for (size_t path : plist)
                        if (path != ta) {
                            if (best == d) D.at(i, j, path) = D.at(p, jp, path) - D.at(p, jp, ta);
                            else if (best == u) D.at(i, j, path) = D.at(p, j, path) - D.at(p, j, ta);
                            else if (ta == ai) D.at(i, j, path) = D.at(i, jp, path);
                            else D.at(i, j, path) = D.at(i, jp, path) - D.at(i, jp, ta);
                        }
                }
            }
            fixup(i, j, ad);
        }
    };

    if (forward) {
        for (size_t i = 0; i + 1 < L; ++i)
            for (size_t j = 0; j < W; ++j) {
                if (i == 0 && j == 0) { /* zeros */ }
                else if (j == 0) { if (!semi) border_cell(i, j); /* semiglobal: dpm[i][0] = 0 */ }
                else if (i == 0) {
                    D.at(0, j, alphas[0]) = D.at(0, j - 1, alphas[0]) + sc.get(seq[j], '-');
                    for (size_t k = alphas[0] + 1; k < P; ++k) D.at(0, j, k) = D.at(0, j - 1, k);
                } else inner_cell(i, j);
            }
    } else {
        const size_t last_node = L - 1, last_char = W - 1;
        for (size_t i = last_node; i >= 1; --i)
            for (size_t j = last_char; j >= 1; --j) {
                if (i == last_node && j == last_char) { /* zeros */ }
                else if (i == last_node) {
                    D.at(i, j, alphas[i]) = D.at(i, j + 1, alphas[i]) + sc.get(seq[j], '-');
                    for (size_t k = alphas[i] + 1; k < P; ++k) D.at(i, j, k) = D.at(i, j + 1, k);
                } else if (j == jborder) { if (!semi) border_cell(i, j); /* aln_mode 9: zeros (:157-159) */ }
                else inner_cell(i, j);
            }
    }
}

// pathwise_alignment_recombination.rs:747-757
void absolute_scores(Dpm& D, const PathGraph& g) {
    for (size_t i = 0; i + 1 < D.L; ++i)
        for (size_t j = 0; j < D.W; ++j)
            for (size_t path = 0; path < D.P; ++path)
                if (path != g.alphas[i] && g.paths_nodes[i][path]) D.at(i, j, path) += D.at(i, j, g.alphas[i]);
}

// utils.rs:221-254
void get_path_len_start_end(const std::vector<uint64_t>& ids, size_t start, size_t end,
                            size_t path_len_in, size_t& path_len, size_t& path_start, size_t& path_end) {
    path_start = 0;
    if (start > 0) {
        uint64_t first = ids[start];
        size_t counter = start - 1;
        while (counter > 0 && ids[counter] == first) { counter -= 1; path_start += 1; }
    }
    path_end = path_len_in > 0 ? path_start + path_len_in - 1 : 0;
    size_t end_offset = 0;
    if (end > 0) {
        uint64_t last = ids[end];
        size_t counter = end + 1;
        while (counter < ids.size() - 1 && ids[counter] == last) { counter += 1; end_offset += 1; }
    }
    path_len = path_end + end_offset + 1;
}

// utils.rs:256-323
void get_rec_path_len_start_end(const std::vector<uint64_t>& ids, size_t fen, size_t rsn, size_t start,
                                size_t end, size_t fwd_len, size_t rev_len, size_t& path_len,
                                size_t& path_start, size_t& path_end) {
    path_start = 0;
    if (start > 0) {
        uint64_t first = ids[start];
        size_t counter = start - 1;
        while (counter > 0 && ids[counter] == first) { counter -= 1; path_start += 1; }
    }
    size_t forw_path_end = fwd_len > 0 ? path_start + fwd_len - 1 : 0;
    size_t forw_end_offset = 0;
    if (fen > 0) {
        uint64_t last = ids[fen];
        size_t counter = fen + 1;
        while (counter < ids.size() - 1 && ids[counter] == last) { counter += 1; forw_end_offset += 1; }
    }
    size_t forw_path_len = forw_path_end + forw_end_offset + 1;
    size_t rev_path_start = 0;
    if (rsn > 0) {
        uint64_t first = ids[rsn];
        size_t counter = rsn - 1;
        while (counter > 0 && ids[counter] == first) { counter -= 1; rev_path_start += 1; }
    }
    size_t rev_path_end = rev_len > 0 ? rev_path_start + rev_len - 1 : 0;
    path_end = forw_path_len + rev_path_end;
    size_t end_offset = 0;
    if (end > 0) {
        uint64_t last = ids[end];
        size_t counter = end + 1;
        while (counter < ids.size() - 1 && ids[counter] == last) { counter += 1; end_offset += 1; }
    }
    size_t rev_path_len = rev_path_end + end_offset + 1;
    path_len = forw_path_len + rev_path_len;
}

// pathwise_alignment_recombination.rs:9-22
int get_node_offset(const std::vector<uint64_t>& ids, size_t node) {
    uint64_t h = ids[node];
    if (h == 0) return 0;
    size_t counter = node; int offset = 0;
    while (ids[counter - 1] == h) { counter -= 1; offset += 1; }
    return offset;
}

std::vector<uint64_t> dedup(const std::vector<uint64_t>& v) {
    std::vector<uint64_t> o;
    for (uint64_t x : v) if (o.empty() || o.back() != x) o.push_back(x);
    return o;
}

// predecessor of row i on `path` in a PredHash (last match in iteration order wins, as in
// the reference's `for (pred, paths) in preds.iter() { if paths[best_path] {..} }`)
bool pred_on_path(const PathGraph& g, size_t i, size_t path, size_t& out) {
    bool found = false;
    auto it = g.pred_hash.find(i);
    if (it == g.pred_hash.end()) return false;
    for (auto& pk : it->second)
        if (pk.second[path]) { out = pk.first; found = true; }
    return found;
}

}  // namespace

// =================================================================================
// -m 4
// =================================================================================
Result m4_literal(const std::string& seq, const std::string& name, const PathGraph& g, const Scores& sc) {
    Result res;
    sc.panicked = false;
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size(), P = g.paths_number;
    const auto& alphas = g.alphas;
    Dpm D(L, W, P);
    fill(D, seq, g, sc, true);
    // pathwise_alignment.rs:305-325
    std::vector<size_t> ending_nodes(P, 0);
    std::vector<int> results(P, 0);
    for (auto& pk : g.pred_hash.at(L - 1)) {
        size_t pred = pk.first;
        for (size_t path = 0; path < P; ++path)
            if (pk.second[path]) {
                if (path == alphas[pred]) results[path] = D.at(pred, W - 1, path);
                else results[path] = D.at(pred, W - 1, path) + D.at(pred, W - 1, alphas[pred]);
                ending_nodes[path] = pred;
            }
    }
    size_t best_path = 0;
    for (size_t k = 0; k < P; ++k)
        if (std::make_pair(results[k], k) >= std::make_pair(results[best_path], best_path)) best_path = k;
    size_t ending_node = ending_nodes[best_path];

    // ---- pathwise_alignment_output.rs:7-184 build_alignment (global) ----
    auto ABS = [&](size_t i, size_t j) {  // best_path's absolute score at (i,j)
        return alphas[i] == best_path ? D.at(i, j, best_path) : D.at(i, j, best_path) + D.at(i, j, alphas[i]);
    };
    std::vector<char> cigar, pseq;
    std::vector<uint64_t> hia;
    size_t path_length = 0, i = ending_node, j = W - 1;
    int score = ABS(i, j);
    res.score = score;
    while (i > 0 && j > 0) {
        bool has_pred = false; size_t predecessor = 0;
        int d = 0, u = 0, l = 0;
        if (!g.nwp[i]) {
            d = ABS(i - 1, j - 1) + sc.get(lnz[i], seq[j]);
            u = ABS(i - 1, j) + sc.get(lnz[i], '-');
            l = ABS(i, j - 1) + sc.get('-', seq[j]);
        } else {
            for (auto& pk : g.pred_hash.at(i))
                if (pk.second[best_path]) {
                    size_t pred = pk.first;
                    predecessor = pred; has_pred = true;
                    d = ABS(pred, j - 1) + sc.get(lnz[i], seq[j]);
                    u = ABS(pred, j) + sc.get(lnz[i], '-');
                    l = ABS(i, j - 1) + sc.get('-', seq[j]);
                }
        }
        int mx = std::max(std::max(d, u), l);
        if (mx == d) {
            cigar.push_back(lnz[i] != seq[j] ? 'd' : 'D');
            hia.push_back(g.nodes_id_pos[i]); pseq.push_back(lnz[i]);
            i = has_pred ? predecessor : i - 1; j -= 1; path_length += 1;
        } else if (mx == u) {
            cigar.push_back('U');
            hia.push_back(g.nodes_id_pos[i]); pseq.push_back(lnz[i]);
            i = has_pred ? predecessor : i - 1; path_length += 1;
        } else { cigar.push_back('L'); j -= 1; }
    }
    while (j > 0) { cigar.push_back('L'); j -= 1; }
    while (i > 0) {
        cigar.push_back('U'); hia.push_back(g.nodes_id_pos[i]); pseq.push_back(lnz[i]); path_length += 1;
        size_t p = 0;
        if (!g.nwp[i]) p = i - 1; else pred_on_path(g, i, best_path, p);
        i = p;
    }
    std::reverse(cigar.begin(), cigar.end());
    std::reverse(pseq.begin(), pseq.end());
    GAF gaf;
    gaf.query_name = name;  // main.rs:259
    gaf.query_length = W - 1; gaf.query_start = 0; gaf.query_end = W - 2; gaf.strand = '+';
    auto dd = dedup(hia); std::reverse(dd.begin(), dd.end());
    gaf.path = dd;
    get_path_len_start_end(g.nodes_id_pos, i == 0 ? i : i + 1, ending_node, path_length,
                           gaf.path_length, gaf.path_start, gaf.path_end);
    gaf.residue_matches_number = 0; gaf.alignment_block_length = "*"; gaf.mapping_quality = "*";
    gaf.comments = build_cigar(cigar) + ", best path: " + std::to_string(best_path) +
                   ", score: " + std::to_string(score) + "\t" + std::string(pseq.begin(), pseq.end());
    res.out = gaf.to_string() + "\n";
    res.would_panic = sc.panicked;
    return res;
}

// =================================================================================
// -m 8
// =================================================================================
Result m8_literal(const std::string& seq, const std::string& name, const PathGraph& g,
                  const PathGraph& rg, const std::vector<int64_t>& dfs,
                  const std::vector<int64_t>& dfe, const Scores& sc, int brc, float mrc, float rbw,
                  bool pruned) {
    Result res;
    sc.panicked = false;
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size(), P = g.paths_number;
    const auto& ids = g.nodes_id_pos;
    const auto& np = g.paths_nodes;
    FaithfulProbe* probe = g_faithful_probe;
    const auto t_start = std::chrono::steady_clock::now();
    Dpm m(L, W, P), w(L, W, P);
    fill(m, seq, g, sc, true);
    absolute_scores(m, g);
    std::string r_seq = seq.substr(1) + "F";  // :875-883
    fill(w, r_seq, rg, sc, false);
    absolute_scores(w, rg);
    auto dms = [&](size_t a, size_t b) -> int {  // pathwise_graph.rs:292-301
        if (a == b) return 0;
        return (int)(std::llabs(dfs[a] - dfs[b]) + std::llabs(dfe[a] - dfe[b]));
    };
    const auto t_dp = std::chrono::steady_clock::now();
    if (probe) probe->dp_secs += std::chrono::duration<double>(t_dp - t_start).count();

    // ---- best_alignment :759-873 (aln_mode 8) ----
    size_t fen = 0, rsn = 0, rec_col = 0;
    bool have = false; int mx = 0; size_t bp = 0;
    for (auto& pk : g.pred_hash.at(L - 1))
        for (size_t path = 0; path < P; ++path)
            if (pk.second[path]) {
                int v = m.at(pk.first, W - 1, path);
                if (!have || mx < v) { mx = v; bp = path; have = true; }
            }
    if (!have) { res.would_panic = true; return res; }
    float curr = (float)mx;
    size_t fbp = bp, rbp = bp;
    bool onedge = false;
    int oob = std::max((int)((float)W * (1.0f - rbw) / 2.0f), 1);
    int rec_penalty = 0;
    std::vector<size_t> fp(L), rp(L);
    for (size_t j = (size_t)oob; j + (size_t)oob < W; ++j) {
        if (probe) {   // timing probe only (see FaithfulProbe): sample the columns of the scan
            probe->cols_total += 1;
            if ((j - (size_t)oob) % (size_t)probe->col_stride != 0) continue;
            probe->cols_visited += 1;
        }
        for (size_t i = 0; i < L; ++i) {
            size_t bf = 0, br = 0;
            for (size_t k = 0; k < P; ++k) {
                if (std::make_pair(m.at(i, j, k), k) >= std::make_pair(m.at(i, j, bf), bf)) bf = k;
                if (std::make_pair(w.at(i, j, k), k) >= std::make_pair(w.at(i, j, br), br)) br = k;
            }
            fp[i] = bf; rp[i] = br;
        }
        std::vector<size_t> fi, ri;
        if (pruned) {
            // exact pruning: a pair can only change the state if
            // (m+w) - R >= seed; keep rows that can reach that with the column's best partner.
            if (brc < 0 || mrc < 0) { res.would_panic = true; return res; }
            long mfmax = INT32_MIN, wrmax = INT32_MIN;
            for (size_t i = 1; i + 1 < L; ++i) {
                if (np[i][fp[i]]) mfmax = std::max<long>(mfmax, m.at(i, j, fp[i]));
                if (np[i][rp[i]]) wrmax = std::max<long>(wrmax, w.at(i, j, rp[i]));
            }
            for (size_t i = 1; i + 1 < L; ++i) {
                if (np[i][fp[i]] && (long)m.at(i, j, fp[i]) + wrmax - brc >= (long)mx) fi.push_back(i);
                if (np[i][rp[i]] && (long)w.at(i, j, rp[i]) + mfmax - brc >= (long)mx) ri.push_back(i);
            }
        } else {
            for (size_t i = 1; i + 1 < L; ++i) { fi.push_back(i); ri.push_back(i); }
        }
        for (size_t i : fi) {
            size_t forw_path = fp[i];
            if (!np[i][forw_path]) continue;
            for (size_t rev_i : ri) {
                if (ids[i] != ids[rev_i]) {
                    size_t rev_path = rp[rev_i];
                    if (forw_path != rev_path && np[rev_i][rev_path]) {
                        float penalty = (float)brc + (mrc * (float)dms(i, rev_i));
                        float new_score = (float)(m.at(i, j, forw_path) + w.at(rev_i, j, rev_path)) - penalty;
                        bool cond = (i + 1 == L || ids[i] != ids[i + 1]) && ids[rev_i] != ids[rev_i - 1];
                        if (new_score > curr || (new_score == curr && !onedge && cond)) {
                            onedge = cond;
                            curr = new_score;
                            fen = i; rsn = rev_i; fbp = forw_path; rbp = rev_path; rec_col = j;
                            rec_penalty = dms(i, rev_i);
                        }
                    }
                }
            }
        }
    }

    if (probe) probe->scan_secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dp).count();

    std::vector<char> cigar, pseq;
    std::vector<uint64_t> hia;
    GAF gaf;
    gaf.query_name = name;  // main.rs:309
    gaf.query_length = W - 1; gaf.query_start = 0; gaf.query_end = W - 2; gaf.strand = '+';
    gaf.residue_matches_number = 0; gaf.alignment_block_length = "*"; gaf.mapping_quality = "*";

    if (fbp == rbp) {
        // ---- recombination_output.rs:633-782 gaf_output_global_no_rec ----
        size_t best_path = fbp, i = 0;
        for (auto& pk : g.pred_hash.at(L - 1)) if (pk.second[best_path]) i = pk.first;
        size_t ending_node = i, j = W - 1, path_length = 0;
        int score = m.at(i, j, best_path);
        res.score = score;
        while (i > 0 && j > 0) {
            bool has_pred = false; size_t predecessor = 0; int d = 0, u = 0, l = 0;
            if (!g.nwp[i]) {
                d = m.at(i - 1, j - 1, best_path) + sc.get(lnz[i], seq[j]);
                u = m.at(i - 1, j, best_path) + sc.get(lnz[i], '-');
                l = m.at(i, j - 1, best_path) + sc.get('-', seq[j]);
            } else {
                for (auto& pk : g.pred_hash.at(i))
                    if (pk.second[best_path]) {
                        predecessor = pk.first; has_pred = true;
                        d = m.at(pk.first, j - 1, best_path) + sc.get(lnz[i], seq[j]);
                        u = m.at(pk.first, j, best_path) + sc.get(lnz[i], '-');
                        l = m.at(i, j - 1, best_path) + sc.get('-', seq[j]);
                    }
            }
            int mxv = std::max(std::max(d, u), l);
            if (mxv == d) {
                cigar.push_back(lnz[i] == seq[j] ? 'D' : 'd');
                hia.push_back(ids[i]); pseq.push_back(lnz[i]);
                i = has_pred ? predecessor : i - 1; j -= 1; path_length += 1;
            } else if (mxv == u) {
                cigar.push_back('U'); hia.push_back(ids[i]); pseq.push_back(lnz[i]);
                i = has_pred ? predecessor : i - 1; path_length += 1;
            } else { cigar.push_back('L'); j -= 1; }
        }
        while (j > 0) { cigar.push_back('L'); j -= 1; }
        while (i > 0) {
            cigar.push_back('U'); hia.push_back(ids[i]); pseq.push_back(lnz[i]);
            size_t p; bool hp = g.nwp[i] && pred_on_path(g, i, best_path, p);
            i = hp ? p : i - 1; path_length += 1;
        }
        std::reverse(cigar.begin(), cigar.end());
        std::reverse(pseq.begin(), pseq.end());
        auto dd = dedup(hia); std::reverse(dd.begin(), dd.end());
        gaf.path = dd;
        get_path_len_start_end(ids, i == 0 ? i : i + 1, ending_node, path_length, gaf.path_length,
                               gaf.path_start, gaf.path_end);
        gaf.comments = build_cigar(cigar) + ", best path: " + std::to_string(best_path) +
                       ", score: " + std::to_string(score) + "\t" + std::string(pseq.begin(), pseq.end());
    } else {
        // ---- recombination_output.rs:363-631 gaf_output_global_rec ----
        size_t rev_path_length = 0, i = rsn, j = rec_col, rev_ending_node = i;
        while (i > 0 && i < L - 1 && j < W - 1) {  // reverse half
            bool has_pred = false; size_t predecessor = 0; int d = 0, u = 0, l = 0;
            if (!rg.nwp[i]) {
                d = w.at(i + 1, j + 1, rbp) + sc.get(lnz[i], r_seq[j]);
                u = w.at(i + 1, j, rbp) + sc.get(lnz[i], '-');
                l = w.at(i, j + 1, rbp) + sc.get('-', r_seq[j]);
            } else {
                for (auto& pk : rg.pred_hash.at(i))
                    if (pk.second[rbp]) {
                        predecessor = pk.first; has_pred = true;
                        d = w.at(pk.first, j + 1, rbp) + sc.get(lnz[i], r_seq[j]);
                        u = w.at(pk.first, j, rbp) + sc.get(lnz[i], '-');
                        l = w.at(i, j + 1, rbp) + sc.get('-', r_seq[j]);
                    }
            }
            int mxv = std::max(std::max(d, u), l);
            rev_ending_node = i;
            if (mxv == d) {
                cigar.push_back(lnz[i] != r_seq[j] ? 'd' : 'D');
                hia.push_back(ids[i]); pseq.push_back(lnz[i]);
                i = has_pred ? predecessor : i + 1; j += 1; rev_path_length += 1;
            } else if (mxv == u) {
                cigar.push_back('U'); hia.push_back(ids[i]); pseq.push_back(lnz[i]);
                i = has_pred ? predecessor : i + 1; rev_path_length += 1;
            } else { cigar.push_back('L'); j += 1; }
        }
        while (j < W - 1) { cigar.push_back('L'); j += 1; }
        while (i < L - 1) {
            cigar.push_back('U'); hia.push_back(ids[i]); pseq.push_back(lnz[i]);
            size_t p; bool hp = rg.nwp[i] && pred_on_path(rg, i, rbp, p);
            i = hp ? p : i + 1; rev_path_length += 1;
        }
        size_t path_length = 0;
        std::vector<char> tcigar, tpseq; std::vector<uint64_t> thia;
        i = fen; j = rec_col;
        while (i > 0 && j > 0) {
            bool has_pred = false; size_t predecessor = 0; int d = 0, u = 0, l = 0;
            if (!g.nwp[i]) {
                d = m.at(i - 1, j - 1, fbp) + sc.get(lnz[i], seq[j]);
                u = m.at(i - 1, j, fbp) + sc.get(lnz[i], '-');
                l = m.at(i, j - 1, fbp) + sc.get('-', seq[j]);
            } else {
                for (auto& pk : g.pred_hash.at(i))
                    if (pk.second[fbp]) {
                        predecessor = pk.first; has_pred = true;
                        d = m.at(pk.first, j - 1, fbp) + sc.get(lnz[i], seq[j]);
                        u = m.at(pk.first, j, fbp) + sc.get(lnz[i], '-');
                        l = m.at(i, j - 1, fbp) + sc.get('-', seq[j]);
                    }
            }
            int mxv = std::max(std::max(d, u), l);
            if (mxv == d) {
                tcigar.push_back(lnz[i] != seq[j] ? 'd' : 'D');
                thia.push_back(ids[i]); tpseq.push_back(lnz[i]);
                i = has_pred ? predecessor : i - 1; j -= 1; path_length += 1;
            } else if (mxv == u) {
                tcigar.push_back('U'); thia.push_back(ids[i]); tpseq.push_back(lnz[i]);
                i = has_pred ? predecessor : i - 1; path_length += 1;
            } else { tcigar.push_back('L'); j -= 1; }
        }
        while (j > 0) { tcigar.push_back('L'); j -= 1; }
        while (i > 0) {
            tcigar.push_back('U'); thia.push_back(ids[i]); tpseq.push_back(lnz[i]);
            size_t p; bool hp = g.nwp[i] && pred_on_path(g, i, fbp, p);
            i = hp ? p : i - 1; path_length += 1;
        }
        if (tpseq.empty()) { res.would_panic = true; return res; }
        size_t rec_edge = tpseq.size() - 1;
        std::reverse(tcigar.begin(), tcigar.end());
        tcigar.insert(tcigar.end(), cigar.begin(), cigar.end());
        std::reverse(thia.begin(), thia.end());
        thia.insert(thia.end(), hia.begin(), hia.end());
        std::reverse(tpseq.begin(), tpseq.end());
        tpseq.insert(tpseq.end(), pseq.begin(), pseq.end());
        gaf.path = dedup(thia);
        size_t start = i == 0 ? i : i + 1;
        get_rec_path_len_start_end(ids, fen, rsn, start, rev_ending_node, path_length, rev_path_length,
                                   gaf.path_length, gaf.path_start, gaf.path_end);
        std::string pss(tpseq.begin(), tpseq.end());
        std::string rec = "recombination path " + std::to_string(fbp) + " " + std::to_string(rbp) +
                          ", nodes " + std::to_string(ids[fen]) + "[" + std::to_string(get_node_offset(ids, fen)) +
                          "] " + std::to_string(ids[rsn]) + "[" + std::to_string(get_node_offset(ids, rsn)) +
                          "], score: " + f32_display(curr) + ", displacement: " + std::to_string(rec_penalty) +
                          "\t" + pss + "\t" + std::to_string(rec_edge);
        gaf.comments = build_cigar(tcigar) + ", " + rec;
        res.score = (int)curr;
    }
    res.out = gaf.to_string() + "\n";
    res.would_panic = sc.panicked;
    return res;
}

// =================================================================================
// -m 5  (src/pathwise_alignment_semiglobal.rs:6-277 + pathwise_alignment_output.rs:7-184, global_align=false)
// =================================================================================
Result m5_literal(const std::string& seq, const std::string& name, const PathGraph& g, const Scores& sc) {
    Result res;
    sc.panicked = false;
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size(), P = g.paths_number;
    const auto& alphas = g.alphas;
    Dpm D(L, W, P);
    fill(D, seq, g, sc, true, true);
    auto ABSv = [&](size_t i, size_t j, size_t k) { return alphas[i] == k ? D.at(i, j, k) : D.at(i, j, k) + D.at(i, j, alphas[i]); };
    // best_ending_node :244-277
    bool have = false; int mx = 0; size_t ending_node = 0, best_path = 0;
    for (size_t i = 1; i + 1 < L; ++i) {
        bool hb = false; int bs = 0; size_t bp = 0;
        for (size_t k = 0; k < P; ++k)
            if (g.paths_nodes[i][k]) {
                int v = ABSv(i, W - 1, k);
                if (!hb || bs < v) { bs = v; bp = k; hb = true; }
            }
        if (!hb) { res.would_panic = true; return res; }   // best_path.unwrap() on None
        if (!have || bs > mx) { mx = bs; ending_node = i; best_path = bp; have = true; }
    }
    // build_alignment(global_align = false)
    auto ABS = [&](size_t i, size_t j) { return ABSv(i, j, best_path); };
    std::vector<char> cigar, pseq; std::vector<uint64_t> hia;
    size_t path_length = 0, i = ending_node, j = W - 1;
    int score = ABS(i, j);
    res.score = score;
    while (i > 0 && j > 0) {
        bool has_pred = false; size_t predecessor = 0; int d = 0, u = 0, l = 0;
        if (!g.nwp[i]) {
            d = ABS(i - 1, j - 1) + sc.get(lnz[i], seq[j]); u = ABS(i - 1, j) + sc.get(lnz[i], '-'); l = ABS(i, j - 1) + sc.get('-', seq[j]);
        } else {
            for (auto& pk : g.pred_hash.at(i))
                if (pk.second[best_path]) {
                    predecessor = pk.first; has_pred = true;
                    d = ABS(pk.first, j - 1) + sc.get(lnz[i], seq[j]); u = ABS(pk.first, j) + sc.get(lnz[i], '-'); l = ABS(i, j - 1) + sc.get('-', seq[j]);
                }
        }
        int m = std::max(std::max(d, u), l);
        if (m == d) { cigar.push_back(lnz[i] != seq[j] ? 'd' : 'D'); hia.push_back(g.nodes_id_pos[i]); pseq.push_back(lnz[i]); i = has_pred ? predecessor : i - 1; j -= 1; path_length += 1; }
        else if (m == u) { cigar.push_back('U'); hia.push_back(g.nodes_id_pos[i]); pseq.push_back(lnz[i]); i = has_pred ? predecessor : i - 1; path_length += 1; }
        else { cigar.push_back('L'); j -= 1; }
    }
    while (j > 0) { cigar.push_back('L'); j -= 1; }
    std::reverse(cigar.begin(), cigar.end());
    std::reverse(pseq.begin(), pseq.end());
    GAF gaf;
    gaf.query_name = name; gaf.query_length = W - 1; gaf.query_start = 0; gaf.query_end = W - 2; gaf.strand = '+';
    auto dd = dedup(hia); std::reverse(dd.begin(), dd.end()); gaf.path = dd;
    get_path_len_start_end(g.nodes_id_pos, i == 0 ? i : i + 1, ending_node, path_length, gaf.path_length, gaf.path_start, gaf.path_end);
    gaf.alignment_block_length = "*"; gaf.mapping_quality = "*";
    gaf.comments = build_cigar(cigar) + ", best path: " + std::to_string(best_path) + ", score: " + std::to_string(score) + "\t" + std::string(pseq.begin(), pseq.end());
    res.out = gaf.to_string() + "\n";
    res.would_panic = sc.panicked;
    return res;
}

// =================================================================================
// -m 9  (pathwise_alignment_recombination.rs aln_mode 9 + recombination_output.rs:12-361)
// =================================================================================
Result m9_literal(const std::string& seq, const std::string& name, const PathGraph& g, const PathGraph& rg,
                  const std::vector<int64_t>& dfs, const std::vector<int64_t>& dfe, const Scores& sc, int brc, float mrc,
                  float rbw, bool pruned) {
    Result res;
    sc.panicked = false;
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size(), P = g.paths_number;
    const auto& ids = g.nodes_id_pos;
    const auto& np = g.paths_nodes;
    Dpm m(L, W, P), w(L, W, P);
    fill(m, seq, g, sc, true, true);
    absolute_scores(m, g);
    std::string r_seq = seq.substr(1) + "F";
    fill(w, r_seq, rg, sc, false, true);
    absolute_scores(w, rg);
    auto dms = [&](size_t a, size_t b) -> int { return a == b ? 0 : (int)(std::llabs(dfs[a] - dfs[b]) + std::llabs(dfe[a] - dfe[b])); };
    // seed :789-800: every row 0..L-2, member paths ascending, strict '<'
    bool have = false; int mx = 0; size_t bp = 0;
    for (size_t i = 0; i + 1 < L; ++i)
        for (size_t k = 0; k < P; ++k)
            if (np[i][k]) { int v = m.at(i, W - 1, k); if (!have || mx < v) { mx = v; bp = k; have = true; } }
    float curr = (float)mx;
    size_t fbp = bp, rbp = bp, fen = 0, rsn = 0, rec_col = 0;
    bool onedge = false; int rec_penalty = 0;
    int oob = std::max((int)((float)W * (1.0f - rbw) / 2.0f), 1);
    std::vector<size_t> fp(L), rp(L);
    for (size_t j = (size_t)oob; j + (size_t)oob < W; ++j) {
        for (size_t i = 0; i < L; ++i) {
            size_t bf = 0, br = 0;
            for (size_t k = 0; k < P; ++k) {
                if (std::make_pair(m.at(i, j, k), k) >= std::make_pair(m.at(i, j, bf), bf)) bf = k;
                if (std::make_pair(w.at(i, j, k), k) >= std::make_pair(w.at(i, j, br), br)) br = k;
            }
            fp[i] = bf; rp[i] = br;
        }
        std::vector<size_t> fi, ri;
        if (pruned) {
            if (brc < 0 || mrc < 0) { res.would_panic = true; return res; }
            long mfmax = INT32_MIN, wrmax = INT32_MIN;
            for (size_t i = 1; i + 1 < L; ++i) {
                if (np[i][fp[i]]) mfmax = std::max<long>(mfmax, m.at(i, j, fp[i]));
                if (np[i][rp[i]]) wrmax = std::max<long>(wrmax, w.at(i, j, rp[i]));
            }
            for (size_t i = 1; i + 1 < L; ++i) {
                if (np[i][fp[i]] && (long)m.at(i, j, fp[i]) + wrmax - brc >= (long)mx) fi.push_back(i);
                if (np[i][rp[i]] && (long)w.at(i, j, rp[i]) + mfmax - brc >= (long)mx) ri.push_back(i);
            }
        } else for (size_t i = 1; i + 1 < L; ++i) { fi.push_back(i); ri.push_back(i); }
        for (size_t i : fi) {
            size_t forw_path = fp[i];
            if (!np[i][forw_path]) continue;
            for (size_t rev_i : ri) {
                if (ids[i] == ids[rev_i]) continue;
                size_t rev_path = rp[rev_i];
                if (forw_path == rev_path || !np[rev_i][rev_path]) continue;
                float penalty = (float)brc + (mrc * (float)dms(i, rev_i));
                float ns = (float)(m.at(i, j, forw_path) + w.at(rev_i, j, rev_path)) - penalty;
                bool cond = (i + 1 == L || ids[i] != ids[i + 1]) && ids[rev_i] != ids[rev_i - 1];
                if (ns > curr || (ns == curr && !onedge && cond)) {
                    onedge = cond; curr = ns; fen = i; rsn = rev_i; fbp = forw_path; rbp = rev_path; rec_col = j; rec_penalty = dms(i, rev_i);
                }
            }
        }
    }
    GAF gaf;
    gaf.query_name = name; gaf.query_length = W - 1; gaf.query_start = 0; gaf.query_end = W - 2; gaf.strand = '+';
    gaf.alignment_block_length = "*"; gaf.mapping_quality = "*";
    auto fwd_walk = [&](size_t& i, size_t& j, size_t path, std::vector<char>& cg, std::vector<uint64_t>& hi, std::vector<char>& ps, size_t& plen) {
        while (i > 0 && j > 0) {
            bool has_pred = false; size_t predecessor = 0; int d = 0, u = 0, l = 0;
            if (!g.nwp[i]) {
                d = m.at(i - 1, j - 1, path) + sc.get(lnz[i], seq[j]); u = m.at(i - 1, j, path) + sc.get(lnz[i], '-'); l = m.at(i, j - 1, path) + sc.get('-', seq[j]);
            } else {
                for (auto& pk : g.pred_hash.at(i))
                    if (pk.second[path]) {
                        predecessor = pk.first; has_pred = true;
                        d = m.at(pk.first, j - 1, path) + sc.get(lnz[i], seq[j]); u = m.at(pk.first, j, path) + sc.get(lnz[i], '-'); l = m.at(i, j - 1, path) + sc.get('-', seq[j]);
                    }
            }
            int mv = std::max(std::max(d, u), l);
            if (mv == d) { cg.push_back(lnz[i] != seq[j] ? 'd' : 'D'); hi.push_back(ids[i]); ps.push_back(lnz[i]); i = has_pred ? predecessor : i - 1; j -= 1; plen += 1; }
            else if (mv == u) { cg.push_back('U'); hi.push_back(ids[i]); ps.push_back(lnz[i]); i = has_pred ? predecessor : i - 1; plen += 1; }
            else { cg.push_back('L'); j -= 1; }
        }
        while (j > 0) { cg.push_back('L'); j -= 1; }
    };
    if (fbp == rbp) {
        // ending_node :885-897, then gaf_output_semiglobal_no_rec (recombination_output.rs:239-361)
        bool hb = false; int bs = 0; size_t en = 0;
        for (size_t i = 1; i + 1 < L; ++i)
            if (np[i][fbp]) { int v = m.at(i, W - 1, fbp); if (!hb || v > bs) { bs = v; en = i; hb = true; } }
        size_t i = en, j = W - 1, plen = 0;
        int score = m.at(i, j, fbp);
        res.score = score;
        std::vector<char> cigar, pseq; std::vector<uint64_t> hia;
        fwd_walk(i, j, fbp, cigar, hia, pseq, plen);
        std::reverse(cigar.begin(), cigar.end());
        std::reverse(pseq.begin(), pseq.end());
        auto dd = dedup(hia); std::reverse(dd.begin(), dd.end()); gaf.path = dd;
        get_path_len_start_end(ids, i == 0 ? i : i + 1, en, plen, gaf.path_length, gaf.path_start, gaf.path_end);
        gaf.comments = build_cigar(cigar) + ", best path: " + std::to_string(fbp) + ", score: " + std::to_string(score) + "\t" + std::string(pseq.begin(), pseq.end());
    } else {
        // gaf_output_semiglobal_rec (recombination_output.rs:12-237)
        std::vector<char> cigar, pseq; std::vector<uint64_t> hia;
        size_t rlen = 0, i = rsn, j = rec_col, rev_ending = i;
        while (i > 0 && i < L - 1 && j < W - 1) {
            bool has_pred = false; size_t predecessor = 0; int d = 0, u = 0, l = 0;
            if (!rg.nwp[i]) {
                d = w.at(i + 1, j + 1, rbp) + sc.get(lnz[i], r_seq[j]); u = w.at(i + 1, j, rbp) + sc.get(lnz[i], '-'); l = w.at(i, j + 1, rbp) + sc.get('-', r_seq[j]);
            } else {
                for (auto& pk : rg.pred_hash.at(i))
                    if (pk.second[rbp]) {
                        predecessor = pk.first; has_pred = true;
                        d = w.at(pk.first, j + 1, rbp) + sc.get(lnz[i], r_seq[j]); u = w.at(pk.first, j, rbp) + sc.get(lnz[i], '-'); l = w.at(i, j + 1, rbp) + sc.get('-', r_seq[j]);
                    }
            }
            int mv = std::max(std::max(d, u), l);
            rev_ending = i;
            if (mv == d) { cigar.push_back(lnz[i] != r_seq[j] ? 'd' : 'D'); hia.push_back(ids[i]); pseq.push_back(lnz[i]); i = has_pred ? predecessor : i + 1; j += 1; rlen += 1; }
            else if (mv == u) { cigar.push_back('U'); hia.push_back(ids[i]); pseq.push_back(lnz[i]); i = has_pred ? predecessor : i + 1; rlen += 1; }
            else { cigar.push_back('L'); j += 1; }
        }
        while (j < W - 1) { cigar.push_back('L'); j += 1; }
        std::vector<char> tc, tp; std::vector<uint64_t> th;
        size_t plen = 0; i = fen; j = rec_col;
        fwd_walk(i, j, fbp, tc, th, tp, plen);
        if (tp.empty()) { res.would_panic = true; return res; }   // usize underflow of rec_edge
        size_t rec_edge = tp.size() - 1;
        std::reverse(tc.begin(), tc.end()); tc.insert(tc.end(), cigar.begin(), cigar.end());
        std::reverse(th.begin(), th.end()); th.insert(th.end(), hia.begin(), hia.end());
        std::reverse(tp.begin(), tp.end()); tp.insert(tp.end(), pseq.begin(), pseq.end());
        gaf.path = dedup(th);
        get_rec_path_len_start_end(ids, fen, rsn, i == 0 ? i : i + 1, rev_ending, plen, rlen, gaf.path_length, gaf.path_start, gaf.path_end);
        gaf.comments = build_cigar(tc) + ", recombination path " + std::to_string(fbp) + " " + std::to_string(rbp) + ", nodes " +
                       std::to_string(ids[fen]) + "[" + std::to_string(get_node_offset(ids, fen)) + "] " + std::to_string(ids[rsn]) + "[" +
                       std::to_string(get_node_offset(ids, rsn)) + "], score: " + f32_display(curr) + ", displacement: " +
                       std::to_string(rec_penalty) + "\t" + std::string(tp.begin(), tp.end()) + "\t" + std::to_string(rec_edge);
        res.score = (int)curr;
    }
    res.out = gaf.to_string() + "\n";
    res.would_panic = sc.panicked;
    return res;
}

}  // namespace orc
