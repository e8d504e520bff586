// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp header).
// Input preparation shared by every mode: score tables, GFA reading, LnzGraph / PathGraph
// flattening, r-values, adaptive band, GAF record text.
#include "orc_common.hpp"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>

namespace orc {

// src/score_matrix.rs:35-51
Scores make_scores_match_mis(int m, int x) {
    Scores s;
    const char* al = "ACGTN-";
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            if (i == j) s.t[i][j] = m;
            else if (al[i] == '-' || al[j] == '-') s.t[i][j] = x * 2;
            else s.t[i][j] = x;
        }
    s.t[4][4] = x;        // ('N','N') = x  (:48)
    s.t[5][5] = MISSING;  // ('-','-') removed (:49)
    return s;
}

// src/score_matrix.rs:52-66
Scores make_scores_match_mis_f32(int m, int x) {
    Scores s;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) s.t[i][j] = (i == j) ? m : x;
    s.t[4][4] = x;
    s.t[5][5] = MISSING;
    return s;
}

// src/score_matrix.rs:67-105: first line = column letters, following lines = row letter +
// values; every (base,'-') and ('-',base) = -200; ('-','-') removed.
Scores make_scores_from_mtx(const std::string& text) {
    Scores s;
    for (auto& r : s.t) for (int& v : r) v = MISSING;
    std::vector<std::vector<std::string>> matrix;
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        std::vector<std::string> toks;
        std::istringstream ls(line);
        std::string tok;
        while (ls >> tok) toks.push_back(tok);
        matrix.push_back(toks);
    }
    if (matrix.empty()) return s;
    matrix[0].insert(matrix[0].begin(), "X");
    for (size_t i = 1; i < matrix.size(); ++i) {
        if (matrix[i].empty()) continue;
        for (size_t j = 1; j < matrix[0].size() && j < matrix[i].size(); ++j) {
            int a = base_idx(matrix[i][0][0]), b = base_idx(matrix[0][j][0]);
            if (a >= 0 && b >= 0) s.t[a][b] = std::stoi(matrix[i][j]);
        }
    }
    for (int c = 0; c < 5; ++c) { s.t[c][5] = -200; s.t[5][c] = -200; }
    s.t[5][5] = MISSING;
    return s;
}

// ---------------------------------------------------------------------------------
bool parse_gfa_text(const std::string& text, Gfa& out, std::string& err) {
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> f;
        size_t p = 0;
        while (true) {
            size_t q = line.find('\t', p);
            if (q == std::string::npos) { f.push_back(line.substr(p)); break; }
            f.push_back(line.substr(p, q - p));
            p = q + 1;
        }
        if (f[0] == "S") {
            if (f.size() < 3) { err = "bad S line"; return false; }
            uint64_t id = std::stoull(f[1]);  // GFA<usize,()>: numeric names (graph.rs:13)
            out.seg_ids.push_back(id);
            out.seg_seq[id] = f[2];
        } else if (f[0] == "L") {
            if (f.size() < 5) { err = "bad L line"; return false; }
            if (f[2] != "+" || f[4] != "+") { err = "only '+' orientations are supported"; return false; }
            out.links.emplace_back(std::stoull(f[1]), std::stoull(f[3]));
        } else if (f[0] == "P") {
            if (f.size() < 3) { err = "bad P line"; return false; }
            out.path_names.push_back(f[1]);
            std::vector<uint64_t> steps;
            std::string s = f[2];
            size_t a = 0;
            while (a < s.size()) {
                size_t b = s.find(',', a);
                if (b == std::string::npos) b = s.size();
                std::string st = s.substr(a, b - a);
                if (!st.empty()) {
                    if (st.back() != '+') { err = "only '+' path steps are supported"; return false; }
                    steps.push_back(std::stoull(st.substr(0, st.size() - 1)));
                }
                a = b + 1;
            }
            out.paths.push_back(steps);
        }
    }
    return true;
}

// src/graph.rs:31-123 with amb_mode=false; hofp from src/utils.rs:144-165.
LnzGraph create_graph_struct(const Gfa& g) {
    LnzGraph out;
    std::vector<uint64_t> sorted = g.seg_ids;  // graph.rs:32-33: handles sorted by id
    std::sort(sorted.begin(), sorted.end());
    // left edges of each node in L-line order (handle_edges_iter(h, Left), graph.rs:75)
    std::unordered_map<uint64_t, std::vector<uint64_t>> left_edges;
    for (auto& l : g.links) left_edges[l.second].push_back(l.first);

    std::string lin = "$";
    std::unordered_map<uint64_t, long> visited;    // id -> last row (graph.rs:55)
    std::map<uint64_t, long> last_nodes;          // graph.rs:56 (HashMap there; id order here)
    for (uint64_t id : sorted) {
        lin += g.seg_seq.at(id);
        visited[id] = (long)lin.size() - 1;
        last_nodes[id] = (long)lin.size() - 1;
    }
    std::vector<uint8_t> nwp(lin.size() + 1, 0);  // graph.rs:58
    for (uint64_t id : sorted) {
        long h_last = visited[id];
        size_t start = (size_t)h_last - g.seg_seq.at(id).size() + 1;
        auto it = left_edges.find(id);
        if (it == left_edges.end() || it->second.empty()) {  // graph.rs:64-74
            nwp[start] = 1;
            out.pred_hash[start].push_back(0);
        } else {
            for (uint64_t pid : it->second) {  // graph.rs:75-86
                long pred_last = visited.at(pid);
                last_nodes.erase(pid);
                nwp[start] = 1;
                out.pred_hash[start].push_back((size_t)pred_last);
            }
        }
    }
    // graph.rs:112-123 set_last_node
    lin += 'F';
    nwp[lin.size() - 1] = 1;
    for (auto& kv : last_nodes) out.pred_hash[lin.size() - 1].push_back((size_t)kv.second);
    out.lnz = lin;
    nwp.resize(lin.size());
    out.nwp = nwp;
    // utils.rs:144-165 (rows 1..L-2 -> id string, row 0 -> "-1")
    out.hofp.assign(lin.size() - 1, "");
    long curr = 0;
    for (size_t i = 1; i + 1 < lin.size(); ++i) {
        if (out.nwp[i]) curr += 1;
        out.hofp[i] = std::to_string(sorted[(size_t)(curr - 1)]);
    }
    out.hofp[0] = "-1";
    // create_handle_pos_in_lnz(.., amb_mode = true): same nwp walk over the handles reversed (and flipped: same id)
    std::vector<uint64_t> rsorted(sorted.rbegin(), sorted.rend());
    out.hofp_rev.assign(lin.size() - 1, "");
    curr = 0;
    for (size_t i = 1; i + 1 < lin.size(); ++i) {
        if (out.nwp[i]) curr += 1;
        out.hofp_rev[i] = std::to_string(rsorted[(size_t)(curr - 1)]);
    }
    out.hofp_rev[0] = "-1";
    return out;
}

// src/pathwise_graph.rs:135-248 (is_reversed=false)
PathGraph create_path_graph(const Gfa& g) {
    PathGraph out;
    std::vector<uint64_t> sorted = g.seg_ids;
    std::sort(sorted.begin(), sorted.end());
    std::string lin = "$";
    std::unordered_map<uint64_t, std::pair<size_t, size_t>> pos;
    out.nodes_id_pos.push_back(0);
    for (uint64_t id : sorted) {
        size_t start = lin.size();
        for (char c : g.seg_seq.at(id)) { lin += c; out.nodes_id_pos.push_back(id); }
        pos[id] = {start, lin.size() - 1};
    }
    lin += 'F';
    out.nodes_id_pos.push_back(0);
    size_t L = lin.size();
    size_t P = g.paths.size();
    out.nwp.assign(L, 0);
    out.alphas.assign(L, P + 1);
    out.paths_nodes.assign(L, std::vector<uint8_t>(P, 0));
    out.paths_nodes[0].assign(P, 1);
    out.alphas[0] = 0;
    out.alphas[L - 1] = 0;
    auto set_pp = [&](size_t node, size_t pred, size_t path) {  // :95-124
        auto& m = out.pred_hash[node];
        auto it = m.find(pred);
        if (it == m.end()) it = m.emplace(pred, std::vector<uint8_t>(P, 0)).first;
        it->second[path] = 1;
    };
    for (size_t pid = 0; pid < P; ++pid) {
        const auto& steps = g.paths[pid];
        for (size_t k = 0; k < steps.size(); ++k) {
            auto [hs, he] = pos.at(steps[k]);
            for (size_t idx = hs; idx <= he; ++idx) {
                out.paths_nodes[idx][pid] = 1;
                if (out.alphas[idx] == P + 1) out.alphas[idx] = pid;
            }
            out.nwp[hs] = 1;
            if (k == 0) {
                set_pp(hs, 0, pid);
            } else {
                size_t pred_end = pos.at(steps[k - 1]).second;
                set_pp(hs, pred_end, pid);
                if (k == steps.size() - 1) set_pp(L - 1, he, pid);  // :225-232
            }
        }
    }
    out.nwp[L - 1] = 1;
    out.paths_nodes[L - 1].assign(P, 1);
    out.lnz = lin;
    out.paths_number = P;
    return out;
}

// src/pathwise_graph.rs:250-282
PathGraph create_reverse_path_graph(const PathGraph& f) {
    PathGraph r;
    r.lnz = f.lnz;
    r.nwp.assign(f.lnz.size(), 0);
    r.paths_nodes = f.paths_nodes;
    r.alphas = f.alphas;
    r.paths_number = f.paths_number;
    r.nodes_id_pos = f.nodes_id_pos;
    for (auto& nk : f.pred_hash)
        for (auto& pk : nk.second) {
            r.nwp[pk.first] = 1;
            for (size_t path = 0; path < pk.second.size(); ++path)
                if (pk.second[path]) {
                    auto& m = r.pred_hash[pk.first];
                    auto it = m.find(nk.first);
                    if (it == m.end())
                        it = m.emplace(nk.first, std::vector<uint8_t>(f.paths_number, 0)).first;
                    it->second[path] = 1;
                }
        }
    return r;
}

// src/pathwise_graph.rs:306-329 (called with the REVERSE graph, :289)
std::vector<int64_t> get_distance_from_start(const PathGraph& g) {
    size_t L = g.lnz.size();
    std::vector<int64_t> r(L, -1);
    r[0] = 0;
    auto it0 = g.pred_hash.find(0);
    if (it0 != g.pred_hash.end())
        for (auto& kv : it0->second) r[kv.first] = 1;
    for (size_t i = 1; i + 1 < L; ++i) {
        if (r[i] == -1 || r[i] > r[i - 1] + 1) r[i] = r[i - 1] + 1;
        if (g.nwp[i]) {
            auto it = g.pred_hash.find(i);
            if (it != g.pred_hash.end())
                for (auto& kv : it->second)
                    if (r[kv.first] == -1 || r[kv.first] > r[i] + 1) r[kv.first] = r[i] + 1;
        }
    }
    return r;
}

// src/pathwise_graph.rs:330-354 (called with the FORWARD graph, :287)
std::vector<int64_t> get_distance_from_end(const PathGraph& g) {
    size_t L = g.lnz.size();
    std::vector<int64_t> r(L, -1);
    r[L - 1] = 0;
    auto itl = g.pred_hash.find(L - 1);
    if (itl != g.pred_hash.end())
        for (auto& kv : itl->second) r[kv.first] = 1;
    for (size_t i = L - 2; i >= 1; --i) {
        if (r[i] == -1 || r[i] > r[i + 1] + 1) r[i] = r[i + 1] + 1;
        if (g.nwp[i]) {
            auto it = g.pred_hash.find(i);
            if (it != g.pred_hash.end())
                for (auto& kv : it->second)
                    if (r[kv.first] == -1 || r[kv.first] > r[i] + 1) r[kv.first] = r[i] + 1;
        }
    }
    return r;
}

// src/utils.rs:103-126
std::vector<size_t> set_r_values(const std::vector<uint8_t>& nwp,
                                 const std::map<size_t, std::vector<size_t>>& pred_hash,
                                 size_t L) {
    std::vector<int64_t> r(L, -1);
    r[L - 1] = 0;
    for (size_t p : pred_hash.at(L - 1)) r[p] = 0;
    for (size_t i = L - 2; i >= 1; --i) {
        if (r[i] == -1 || r[i] > r[i + 1] + 1) r[i] = r[i + 1] + 1;
        if (nwp[i])
            for (size_t p : pred_hash.at(i))
                if (r[p] == -1 || r[p] > r[i] + 1) r[p] = r[i] + 1;
    }
    std::vector<size_t> out(L);
    for (size_t i = 0; i < L; ++i) out[i] = (size_t)r[i];  // `*x as usize` (:125)
    return out;
}

// src/utils.rs:74-98 (usize arithmetic of a release build: wraps)
static std::pair<size_t, size_t> set_left_right_x64(size_t left, size_t right, size_t seq_len) {
    size_t nr = right, nl = left;
    while ((nr - nl) % 8 != 0) {
        if ((nr - nl) % 2 == 0 && nr < seq_len) nr += 1;
        else if (nl > 0) nl -= 1;
        else break;
    }
    if (nl == 0)
        while ((nr - 1) % 8 != 0 && nr < seq_len) nr += 1;
    if (nr == seq_len)
        while ((nr - nl) % 8 != 0 && nl > 1) nl -= 1;
    return {nl, nr};
}

// src/utils.rs:17-72
std::pair<size_t, size_t> set_ampl_for_row(size_t i, const std::vector<size_t>& p_arr,
                                           size_t r_val, const std::vector<size_t>& bsp,
                                           size_t seq_len, size_t bta, bool simd_version) {
    size_t ms, me;
    if (i == 0) { ms = 0; me = 0; }
    else if (p_arr.empty()) { size_t pl = bsp[i - 1]; ms = pl + 1; me = pl + 1; }
    else {
        size_t pl = 0, pr = 0; bool first = true;
        for (size_t p : p_arr) {
            size_t cb = bsp[p];
            if (first) { pl = cb; pr = cb; first = false; }
            if (cb < pl) pl = cb;
            if (cb > pr) pr = cb;
        }
        ms = pl + 1; me = pr + 1;
    }
    int32_t tmp_bs = std::min((int32_t)ms, ((int32_t)seq_len - (int32_t)r_val) - (int32_t)bta);
    size_t band_start = tmp_bs < 0 ? 0 : (size_t)tmp_bs;
    size_t band_end = seq_len > r_val ? std::min(seq_len, std::max(me, seq_len - r_val) + bta)
                                      : std::min(seq_len, me + bta);
    if (simd_version) return set_left_right_x64(band_start, band_end, seq_len);
    return {band_start, band_end};
}

// Rust `{}` of an f32 (library/core/src/fmt/float.rs -> flt2dec::to_shortest_str): the shortest decimal digit string that
// reads back as the same f32, in positional notation, zero-padded.  Written differently from the product's formatter on
// purpose (test infrastructure should not share a reading with what it checks): the digit count is found by trying
// 1..9 significant digits with printf and reading each back.
std::string f32_display(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    if (v == 0.0f) return std::signbit(v) ? "-0" : "0";
    char b[64];
    int prec = 0;
    for (; prec < 9; ++prec) {
        snprintf(b, sizeof b, "%.*e", prec, (double)v);
        if (strtof(b, nullptr) == v) break;
    }
    // b = [-]d.ddd...e[+-]XX with prec digits behind the point
    std::string t(b), digits, out;
    size_t i = 0;
    if (t[0] == '-') { out = "-"; i = 1; }
    const size_t epos = t.find('e');
    for (; i < epos; ++i) if (t[i] != '.') digits += t[i];
    const int e10 = atoi(t.c_str() + epos + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const int before = e10 + 1;
    if (before <= 0) return out + "0." + std::string((size_t)(-before), '0') + digits;
    if ((size_t)before >= digits.size()) return out + digits + std::string((size_t)before - digits.size(), '0');
    return out + digits.substr(0, (size_t)before) + "." + digits.substr((size_t)before);
}

// src/gaf_output.rs:70-94
std::string GAF::to_string() const {
    std::string pm;
    for (size_t i = 0; i < path.size(); ++i) {
        if (i) pm += ">";
        pm += std::to_string(path[i]);
    }
    std::string s = query_name;
    s += "\t" + std::to_string(query_length) + "\t" + std::to_string(query_start) + "\t" +
         std::to_string(query_end) + "\t";
    s += strand;
    s += "\t>" + pm + "\t" + std::to_string(path_length) + "\t" + std::to_string(path_start) +
         "\t" + std::to_string(path_end) + "\t" + std::to_string(residue_matches_number) + "\t" +
         alignment_block_length + "\t" + mapping_quality + "\t" + comments;
    return s;
}

// src/pathwise_alignment_output.rs:471-556
std::string build_cigar(const std::vector<char>& cigar) {
    std::string out;
    size_t d = 0, u = 0, l = 0, mm = 0;
    auto flush = [&](size_t& c, char op) {
        if (c != 0) { out += std::to_string(c); out += op; c = 0; }
    };
    for (char ch : cigar) {
        switch (ch) {
            case 'D': flush(u, 'I'); flush(l, 'D'); flush(mm, 'X'); d += 1; break;
            case 'U': flush(d, 'M'); flush(l, 'D'); flush(mm, 'X'); u += 1; break;
            case 'd': flush(d, 'M'); flush(l, 'D'); flush(u, 'I'); mm += 1; break;
            default:  flush(d, 'M'); flush(u, 'I'); flush(mm, 'X'); l += 1; break;
        }
    }
    flush(d, 'M'); flush(u, 'I'); flush(l, 'D'); flush(mm, 'X');
    return out;
}

}  // namespace orc
