// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp header).
// Banded POA modes: -m 0 (AVX2 f32 path and scalar i32 path) and -m 2 (affine gaps),
// with their traceback / GAF builders.  Literal restatements: same matrices, same
// loop order, same tie rules as the reference.
#include <algorithm>
#include <cmath>

#include "orc_common.hpp"

namespace orc {

static const std::vector<size_t> kEmpty;

static inline const std::vector<size_t>& preds_of(const LnzGraph& g, size_t i) {
    return g.pred_hash.at(i);
}

// gaf_output.rs:867-874
static size_t node_start(const std::vector<std::string>& hofp, size_t row) {
    const std::string& id = hofp[row];
    size_t i = row;
    while (hofp[i] == id && i > 0) i -= 1;
    return row - i;
}

// gaf_output.rs:876-892
static bool set_cigar_substring(int cm, int ci, int cd, std::string& cs) {
    if (cm * ci + ci * cd + cm * cd != 0) return false;  // panic!("wrong format in cigar string")
    if (cm > 0) cs = std::to_string(cm) + "M" + cs;
    else if (ci > 0) cs = std::to_string(ci) + "I" + cs;
    else if (cd > 0) cs = std::to_string(cd) + "D" + cs;
    return true;
}

static const char* kEmptyGaf = "\t0\t0\t0\t \t>0\t0\t0\t0\t0\t\t\t";  // GAFStruct::new().to_string()

// =================================================================================
// -m 0, AVX2 path.  src/global_abpoa.rs:10-257.  8 f32 lanes are restated as a loop
// over 8 scalars; every comparison is the ordered, strict `_CMP_GT_OS`.
// =================================================================================
Result m0_simd(const std::string& read, const std::string& name, size_t idx, const LnzGraph& g,
               const Scores& sc, size_t bta, const std::vector<size_t>& r_values,
               uint64_t* cells) {
    Result res;
    sc.panicked = false;
    const size_t L = g.lnz.size(), W = read.size();
    const std::string& lnz = g.lnz;
    auto S = [&](char a, char b) { return (float)sc.get(a, b); };

    const float min_score = 2.0f * (float)W * S(read[1], '-');  // :20
    std::vector<std::vector<float>> m(L, std::vector<float>(W, min_score));
    std::vector<std::vector<float>> path(L, std::vector<float>(W, -1.0f));
    std::vector<size_t> bsp(L, 0);
    uint64_t ncells = 0;

    m[0][0] = 0.0f;
    path[0][0] = 0.0f;
    for (size_t i = 1; i + 1 < L; ++i) {  // :36-46
        if (!g.nwp[i]) {
            m[i][0] = m[i - 1][0] + S(lnz[i], '-');
            path[i][0] = (float)(i - 1) + 0.2f;
        } else {
            const auto& pr = preds_of(g, i);
            size_t best_p = *std::min_element(pr.begin(), pr.end());
            m[i][0] = m[best_p][0] + S(lnz[i], '-');
            path[i][0] = (float)best_p + 0.2f;
        }
    }
    {
        auto lr = set_ampl_for_row(0, kEmpty, r_values[0], bsp, W, bta, true);  // :48-56
        for (size_t j = 1; j < lr.second; ++j) {
            m[0][j] = m[0][j - 1] + S(read[j], '-');
            path[0][j] = 0.3f;
        }
    }
    for (size_t i = 1; i + 1 < L; ++i) {  // :63-226
        const std::vector<size_t>& p_arr = g.nwp[i] ? preds_of(g, i) : kEmpty;
        auto lr = set_ampl_for_row(i, p_arr, r_values[i], bsp, W, bta, true);
        size_t left = lr.first, right = lr.second;
        size_t best_col = left;
        size_t start = left == 0 ? 1 : left;
        size_t end = right == W ? ((right - start) / 8) * 8 + start : right;
        const float us_update = S(lnz[i], '-');
        for (size_t j = start; j < end; j += 8) {  // :89-166
            float ds_update[8];
            for (int k = 0; k < 8; ++k) {
                if (j + k >= W) { res.would_panic = true; return res; }  // read[j+k] OOB
                ds_update[k] = S(lnz[i], read[j + k]);
            }
            if (!g.nwp[i]) {
                for (int k = 0; k < 8; ++k) {
                    float us = m[i - 1][j + k] + us_update;
                    float ds = m[i - 1][j + k - 1] + ds_update[k];
                    bool bc = ds > us;
                    m[i][j + k] = bc ? ds : us;
                    path[i][j + k] = (float)(i - 1) + (bc ? 0.1f : 0.2f);
                }
            } else {
                const auto& preds = p_arr;
                for (int k = 0; k < 8; ++k) {
                    float best_us = m[preds[0]][j + k], best_ds = m[preds[0]][j + k - 1];
                    float pus = (float)preds[0], pds = (float)preds[0];
                    for (size_t q = 1; q < preds.size(); ++q) {
                        float us = m[preds[q]][j + k], ds = m[preds[q]][j + k - 1];
                        if (us > best_us) { best_us = us; pus = (float)preds[q]; }
                        if (ds > best_ds) { best_ds = ds; pds = (float)preds[q]; }
                    }
                    best_us = best_us + us_update;
                    best_ds = best_ds + ds_update[k];
                    bool bc = best_ds > best_us;
                    m[i][j + k] = bc ? best_ds : best_us;
                    pds = pds + 0.1f;
                    pus = pus + 0.2f;
                    path[i][j + k] = bc ? pds : pus;
                }
            }
            for (size_t x = j; x < j + 8; ++x) {  // :156-165 (gap key uses read[j], the chunk head)
                float l = m[i][x - 1] + S(read[j], '-');
                if (l > m[i][x]) { m[i][x] = l; path[i][x] = (float)i + 0.3f; }
                if (m[i][x] >= m[i][best_col]) best_col = x;
            }
            ncells += 8;
        }
        if (end < right) {  // :168-224 scalar tail
            for (size_t j = end; j < right; ++j) {
                if (!g.nwp[i]) {
                    float l = m[i][j - 1] + S(read[j], '-');
                    float u = m[i - 1][j] + S(lnz[i], '-');
                    float d = m[i - 1][j - 1] + S(lnz[i], read[j]);
                    m[i][j] = std::max(std::max(l, u), d);
                    if (m[i][j] == d) path[i][j] = (float)(i - 1) + 0.1f;
                    else if (m[i][j] == u) path[i][j] = (float)(i - 1) + 0.2f;
                    else path[i][j] = (float)i + 0.3f;
                } else {
                    float u = 0, d = 0; size_t u_pred = 0, d_pred = 0; bool first = true;
                    for (size_t p : p_arr) {
                        if (first) { u = m[p][j]; d = m[p][j - 1]; u_pred = p; d_pred = p; first = false; }
                        if (m[p][j] > u) { u = m[p][j]; u_pred = p; }
                        if (m[p][j - 1] > d) { d = m[p][j - 1]; d_pred = p; }
                    }
                    u += S(lnz[i], '-');
                    d += S(read[j], lnz[i]);  // :206 swapped key
                    float l = m[i][j - 1] + S(read[j], '-');
                    m[i][j] = std::max(std::max(l, u), d);
                    if (m[i][j] == d) path[i][j] = (float)d_pred + 0.1f;
                    else if (m[i][j] == u) path[i][j] = (float)u_pred + 0.2f;
                    else path[i][j] = (float)i + 0.3f;
                }
                if (m[i][j] >= m[i][best_col]) best_col = j;
                ncells += 1;
            }
        }
        bsp[i] = best_col;
    }
    float best_result = 0; bool first = true; size_t last_row = 0;  // :227-240
    for (size_t p : preds_of(g, L - 1)) {
        if (first) { best_result = m[p][W - 1]; last_row = p; first = false; }
        if (m[p][W - 1] > best_result) { best_result = m[p][W - 1]; last_row = p; }
    }
    res.score = (int)best_result;
    if (cells) *cells = ncells;
    if (sc.panicked) { res.would_panic = true; return res; }
    if (idx == 0) return res;

    // ---- gaf_output.rs:753-865 gaf_of_global_abpoa_simd ----
    size_t col = W - 1, row = last_row;
    const size_t last_col = W - 1;
    std::vector<const std::string*> hia;
    std::vector<char> cigar, pseq;
    size_t path_length = 0, residue = 0;
    bool out_ok = true;
    while (path[row][col] != 0.0f) {
        float val = path[row][col];
        if (val == -1.0f) { out_ok = false; break; }
        std::string vs = f32_display(val);  // :783-786
        size_t dot = vs.find('.');
        if (dot == std::string::npos || vs.find('.', dot + 1) != std::string::npos || vs[0] == '-') {
            res.would_panic = true; return res;
        }
        size_t pred = std::stoull(vs.substr(0, dot));
        std::string frac = vs.substr(dot + 1);
        if (frac.size() > 9) { res.would_panic = true; return res; }
        long dir = std::stol(frac);
        if (dir == 1) {
            hia.push_back(&g.hofp[row]);
            pseq.push_back(lnz[row]);
            row = pred;
            if (col == 0) { res.would_panic = true; return res; }
            col -= 1;
            cigar.push_back(lnz[row] == read[col] ? 'D' : 'd');  // tested on the destination cell
            path_length += 1;
            residue += 1;
        } else if (dir == 3) {
            if (col == 0) { res.would_panic = true; return res; }
            col -= 1;
            cigar.push_back('L');
        } else if (dir == 2) {
            hia.push_back(&g.hofp[row]);
            pseq.push_back(lnz[row]);
            row = pred;
            cigar.push_back('U');
            path_length += 1;
        } else { res.would_panic = true; return res; }
        if (row >= L - 1) { res.would_panic = true; return res; }
    }
    if (!out_ok) {
        res.out = std::string("band not enough for correct output\n") + kEmptyGaf + "\n";
        return res;
    }
    std::reverse(cigar.begin(), cigar.end());
    std::string cigar_out = build_cigar(cigar);
    std::reverse(pseq.begin(), pseq.end());
    // dedup consecutive, then reverse (:825-826)
    std::vector<const std::string*> dd;
    for (auto* s : hia) if (dd.empty() || *dd.back() != *s) dd.push_back(s);
    std::reverse(dd.begin(), dd.end());
    GAF gaf;
    gaf.query_name = name;
    gaf.query_length = W - 1;
    gaf.query_start = col;
    gaf.query_end = last_col;
    gaf.strand = g.strand;
    gaf.path.clear();
    for (auto* s : dd) gaf.path.push_back(std::stoull(*s));
    gaf.path_length = path_length;
    gaf.path_start = node_start(g.hofp, row);
    gaf.path_end = node_start(g.hofp, last_row);
    gaf.residue_matches_number = residue;
    gaf.alignment_block_length = "*";
    gaf.mapping_quality = "*";
    gaf.comments = cigar_out + ", score: " + f32_display(best_result) + "\t" +
                   std::string(pseq.begin(), pseq.end());
    res.out = gaf.to_string() + "\n";
    return res;
}

// =================================================================================
// band-relative helpers shared by the scalar -m 0 and -m 2 (global_abpoa.rs:477-566,
// gap_global_abpoa.rs:254-368).  A 32-bit path cell is (pred as u16, dir) (bitfield_path.rs).
// =================================================================================
struct PCell { uint16_t pred = 0; char dir = 'O'; };
static inline PCell pc(size_t pred, char dir) { return PCell{(uint16_t)pred, dir}; }  // `pred as u16`

using Ampl = std::vector<std::pair<size_t, size_t>>;

// rust `if left_p < left_i { j + (left_i-left_p) } else { j - (left_p-left_i) }`; wrap -> OOB
static inline bool jpos(size_t j, size_t left_i, size_t left_p, size_t& out) {
    if (left_p < left_i) { out = j + (left_i - left_p); return true; }
    if (j < left_p - left_i) return false;
    out = j - (left_p - left_i);
    return true;
}

static bool best_d(const std::vector<size_t>& p_arr, const std::vector<std::vector<int>>& m,
                   const Ampl& a, size_t i, size_t j, int& d, size_t& d_idx) {
    bool first = true; size_t left_i = a[i].first;
    for (size_t p : p_arr) {
        size_t left_p = a[p].first;
        if (j + left_i > a[p].first && j + left_i <= a[p].second) {
            size_t jp; jpos(j, left_i, left_p, jp);
            int cur = m[p][jp - 1];
            if (first) { d = cur; d_idx = p; first = false; }
            if (cur > d) { d = cur; d_idx = p; }
        }
    }
    return !first;
}

static bool best_u0(const std::vector<size_t>& p_arr, const std::vector<std::vector<int>>& m,
                    const Ampl& a, size_t i, size_t j, int& u, size_t& u_idx) {
    bool first = true; size_t left_i = a[i].first;
    for (size_t p : p_arr) {
        size_t left_p = a[p].first;
        if (j + left_i >= a[p].first && j + left_i < a[p].second) {
            size_t jp; jpos(j, left_i, left_p, jp);
            int cur = m[p][jp];
            if (first) { first = false; u = cur; u_idx = p; }
            if (cur > u) { u = cur; u_idx = p; }
        }
    }
    return !first;
}

// =================================================================================
// -m 0 scalar.  src/global_abpoa.rs:260-476; GAF gaf_output.rs:254-381
// =================================================================================
Result m0_scalar(const std::string& seq, const std::string& name, size_t idx, const LnzGraph& g,
                 const Scores& sc, size_t bta, uint64_t* cells) {
    Result res;
    sc.panicked = false;
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size();
    auto r_values = set_r_values(g.nwp, g.pred_hash, L);  // :274 (per read)
    std::vector<size_t> bsp(L, 0);
    std::vector<std::vector<int>> m(L);
    std::vector<std::vector<PCell>> path(L);
    Ampl ampl(L, {0, 0});
    uint64_t ncells = 0;
    auto min_pred = [&](size_t i) -> size_t {
        if (!g.nwp[i]) return i - 1;
        const auto& pr = preds_of(g, i);
        return *std::min_element(pr.begin(), pr.end());
    };
    for (size_t i = 0; i + 1 < L; ++i) {
        const std::vector<size_t>& pa0 = g.nwp[i] ? preds_of(g, i) : kEmpty;
        auto lr = set_ampl_for_row(i, pa0, r_values[i], bsp, W, bta, false);
        size_t left = lr.first, right = lr.second;
        ampl[i] = lr;
        size_t best_val_pos = 0;
        if (right < left) { res.would_panic = true; return res; }
        m[i].assign(right - left, 0);
        path[i].assign(right - left, PCell{0, 'O'});
        if (right == left) { res.would_panic = true; return res; }  // m[i][best_val_pos] OOB later
        std::vector<size_t> single{i ? i - 1 : 0};
        for (size_t j = 0; j < right - left; ++j) {
            if (i == 0 && j == 0) {
                m[i][j] = 0; path[i][j] = pc(0, 'O');
            } else if (i == 0) {
                m[i][j] = m[i][j - 1] + sc.get('-', seq[j + left]);  // :307
                path[i][j] = pc(i, 'L');
            } else if (j == 0 && left == 0) {
                size_t bp = min_pred(i);
                if (m[bp].empty()) { res.would_panic = true; return res; }
                m[i][j] = m[bp][j] + sc.get('-', lnz[i]);  // :316
                path[i][j] = pc(bp, 'U');
            } else {
                const std::vector<size_t>& p_arr = g.nwp[i] ? preds_of(g, i) : single;
                int l; size_t l_pred;
                if (j > 0) { l = m[i][j - 1] + sc.get(seq[j + left], '-'); l_pred = i; }
                else { l = sc.get(seq[j + left], '-') * (int)(i + left + j); l_pred = min_pred(i); }
                int u = 0; size_t u_pred = 0;
                if (best_u0(p_arr, m, ampl, i, j, u, u_pred)) u += sc.get(lnz[i], '-');
                else { u = sc.get(lnz[i], '-') * (int)(i + left + j); u_pred = min_pred(i); }
                int d = 0; size_t d_pred = 0;
                if (best_d(p_arr, m, ampl, i, j, d, d_pred)) d += sc.get(lnz[i], seq[j + left]);
                else { d = sc.get(lnz[i], '-') * (int)(i + left); d_pred = min_pred(i); }
                // utils.rs:129-140 get_max_d_u_l: D > U > L
                int best; char dir;
                if (d < u) { if (u < l) { best = l; dir = 'L'; } else { best = u; dir = 'U'; } }
                else { if (d < l) { best = l; dir = 'L'; } else { best = d; dir = 'D'; } }
                if (dir == 'D' && seq[j + left] != lnz[i]) dir = 'd';
                m[i][j] = best;
                path[i][j] = (dir == 'D' || dir == 'd') ? pc(d_pred, dir)
                             : dir == 'U'               ? pc(u_pred, 'U')
                                                        : pc(l_pred, 'L');
                ncells += 1;
            }
            if (m[i][j] >= m[i][best_val_pos]) best_val_pos = j;
        }
        bsp[i] = best_val_pos + left;
    }
    size_t last_row = L - 2, last_col = m[last_row].size() - 1;  // :397-405
    for (size_t p : preds_of(g, L - 1)) {
        size_t tlc = (ampl[p].second - ampl[p].first) - 1;
        if (m[p][tlc] > m[last_row][last_col]) { last_row = p; last_col = tlc; }
    }
    res.score = m[last_row][last_col];
    if (cells) *cells = ncells;
    if (sc.panicked) { res.would_panic = true; return res; }
    // band_ampl_enough :428-476
    {
        size_t i = last_row, j = last_col; bool ok = true;
        while (path[i][j].dir != 'O') {
            auto [left, right] = ampl[i];
            if (i == 0 || (j == 0 && left == 0)) break;
            if ((j == 0 && left != 0) || (j == right - left - 1 && right != W)) { ok = false; break; }
            size_t pred = path[i][j].pred; size_t left_p = ampl[pred].first; size_t jp;
            if (!jpos(j, left, left_p, jp)) { res.would_panic = true; return res; }
            char dir = path[i][j].dir;
            if (dir == 'D' || dir == 'd') { if (jp == 0) { res.would_panic = true; return res; } j = jp - 1; i = pred; }
            else if (dir == 'L') j -= 1;
            else if (dir == 'U') { i = pred; j = jp; }
            else { res.would_panic = true; return res; }
            if (j >= path[i].size()) { res.would_panic = true; return res; }
        }
        if (!ok) res.out += "Band length probably too short, maybe try with larger b and f\n";
    }
    if (idx == 0) return res;
    // ---- gaf_output.rs:254-381 ----
    size_t col = last_col, row = last_row;
    std::vector<const std::string*> hia;
    std::vector<std::string> cigars;  // insert(0, ..) == push_front
    std::string cigar;
    int cm = 0, ci = 0, cd = 0;
    std::string curr_handle = "";
    char last_dir = ' ';
    size_t path_length = 0, residue = 0;
    while (path[row][col].dir != 'O') {
        size_t pred = path[row][col].pred; char dir = path[row][col].dir;
        if (g.hofp[row] != curr_handle) {
            if (!set_cigar_substring(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cigars.insert(cigars.begin(), cigar);
            cigar.clear(); cm = ci = cd = 0;
        }
        curr_handle = g.hofp[row];
        if (std::toupper(dir) != std::toupper(last_dir)) {
            if (!set_cigar_substring(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cm = ci = cd = 0;
        }
        last_dir = dir;
        size_t p_left = ampl[pred].first, jp;
        bool jp_ok;
        if (ampl[row].first < p_left) { size_t delta = p_left - ampl[row].first; jp_ok = col >= delta; jp = col - delta; }
        else { jp = col + (ampl[row].first - p_left); jp_ok = true; }
        // (j_pos is computed eagerly in the reference: a wrap only matters if it is used)
        if (dir == 'D' || dir == 'd') {
            hia.push_back(&g.hofp[row]);
            if (!jp_ok || jp == 0) { res.would_panic = true; return res; }
            row = pred; col = jp - 1; cm += 1; path_length += 1;
            if (dir == 'D') residue += 1;
        } else if (dir == 'L') {
            if (col == 0) { res.would_panic = true; return res; }
            col -= 1; cd += 1;
        } else if (dir == 'U') {
            hia.push_back(&g.hofp[row]);
            if (!jp_ok) { res.would_panic = true; return res; }
            row = pred; col = jp; ci += 1; path_length += 1;
        } else { res.would_panic = true; return res; }
        if (col >= path[row].size()) { res.would_panic = true; return res; }
    }
    if (!set_cigar_substring(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
    cigars.insert(cigars.begin(), cigar);
    std::vector<const std::string*> dd;
    for (auto* s : hia) if (dd.empty() || *dd.back() != *s) dd.push_back(s);
    std::reverse(dd.begin(), dd.end());
    GAF gaf;
    gaf.query_name = name;
    gaf.query_length = W - 1;
    gaf.query_start = col;
    gaf.query_end = last_col + ampl[last_row].first;
    gaf.strand = g.strand;
    gaf.path.clear();
    for (auto* s : dd) {
        if ((*s)[0] == '-') { res.would_panic = true; return res; }
        gaf.path.push_back(std::stoull(*s));
    }
    gaf.path_length = path_length;
    gaf.path_start = node_start(g.hofp, row);
    gaf.path_end = node_start(g.hofp, last_row);
    gaf.residue_matches_number = residue;
    gaf.alignment_block_length = "*";
    gaf.mapping_quality = "*";
    std::string comments;
    for (size_t k = 0; k + 1 < cigars.size(); ++k) { if (k) comments += ","; comments += cigars[k]; }
    gaf.comments = comments;
    res.out += gaf.to_string() + "\n";
    return res;
}

// =================================================================================
// -m 2.  src/gap_global_abpoa.rs:11-455; GAF gaf_output.rs:96-253
// =================================================================================
Result m2_gap(const std::string& seq, const std::string& name, size_t idx, const LnzGraph& g,
              const Scores& sc, int o, int e, size_t bta, uint64_t* cells) {
    Result res;
    sc.panicked = false;
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size();
    std::vector<std::vector<int>> m(L), x(L), y(L);
    std::vector<std::vector<PCell>> path(L), path_x(L), path_y(L);
    auto r_values = set_r_values(g.nwp, g.pred_hash, L);  // :38
    std::vector<size_t> bsp(L, 0);
    Ampl ampl(L, {0, 0});
    uint64_t ncells = 0;
    auto min_pred = [&](size_t i) -> size_t {
        if (!g.nwp[i]) return i - 1;
        const auto& pr = preds_of(g, i);
        return *std::min_element(pr.begin(), pr.end());
    };
    for (size_t i = 0; i + 1 < L; ++i) {
        const std::vector<size_t>& pa0 = g.nwp[i] ? preds_of(g, i) : kEmpty;
        auto lr = set_ampl_for_row(i, pa0, r_values[i], bsp, W, bta, false);
        size_t left = lr.first, right = lr.second;
        ampl[i] = lr;
        size_t best_val_pos = 0;
        if (right <= left) { res.would_panic = true; return res; }
        size_t w = right - left;
        m[i].assign(w, 0); x[i].assign(w, 0); y[i].assign(w, 0);
        path[i].assign(w, PCell{0, 'O'}); path_x[i].assign(w, PCell{0, 'O'}); path_y[i].assign(w, PCell{0, 'O'});
        std::vector<size_t> single{i ? i - 1 : 0};
        for (size_t j = 0; j < w; ++j) {
            if (i == 0 && j == 0) {
                m[i][j] = 0; path[i][j] = pc(0, 'O');
            } else if (i == 0) {
                y[i][j] = o + e * (int)(j + left);  // :74
                m[i][j] = y[i][j];
                path[i][j] = pc(i, 'L');
            } else if (j == 0 && left == 0) {
                size_t bp = min_pred(i);
                x[i][j] = o + e * (int)(bp + 1);  // :88
                m[i][j] = x[i][j];
                path[i][j] = pc(bp, 'U');
            } else {
                const std::vector<size_t>& p_arr = g.nwp[i] ? preds_of(g, i) : single;
                // get_best_l :350-368
                size_t l_pred;
                if (j > 0) {
                    int l_x = x[i][j - 1], l_m = m[i][j - 1] + o;
                    if (l_x > l_m) { x[i][j] = l_x + e; path_x[i][j] = pc(i, 'X'); }
                    else x[i][j] = l_m + e;
                    l_pred = i;
                } else {
                    size_t bp = min_pred(i);
                    x[i][j] = 2 * o + e * (int)(bp + 1) + e * (int)(j + left);  // :117
                    l_pred = bp;
                }
                // get_best_u :296-346
                size_t u_pred;
                {
                    int u_m = 0, u_y = 0; size_t u_m_idx = 0, u_y_idx = 0; bool first = true;
                    size_t left_i = left;
                    for (size_t p : p_arr) {
                        size_t left_p = ampl[p].first;
                        if (j + left_i >= ampl[p].first && j + left_i < ampl[p].second) {
                            size_t jp; jpos(j, left_i, left_p, jp);
                            int cum = m[p][jp] + o, cuy = y[p][jp];
                            if (first) { first = false; u_m = cum; u_y = cuy; u_y_idx = p; u_m_idx = p; }
                            if (cum > u_m) { u_m = cum; u_m_idx = p; }
                            if (cuy > u_y) { u_y = cuy; u_y_idx = p; }
                        }
                    }
                    if (first) {
                        size_t bp = min_pred(i);
                        y[i][j] = 2 * o + e * (int)(bp + 1) + e * (int)(j + left);  // :139
                        u_pred = bp;
                    } else if (u_y > u_m) {
                        y[i][j] = u_y + e; u_pred = u_y_idx; path_y[i][j] = pc(u_y_idx, 'Y');
                    } else {
                        y[i][j] = u_m + e; u_pred = u_m_idx;
                    }
                }
                int d = 0; size_t d_idx = 0;
                int lv = x[i][j], uv = y[i][j];
                if (best_d(p_arr, m, ampl, i, j, d, d_idx)) {  // :145-180
                    d += sc.get(lnz[i], seq[j + left]);
                    if (d < lv) {
                        if (lv < uv) {
                            if (u_pred == 0) { res.would_panic = true; return res; }  // set_path_cell(_, 'u') panics
                            path[i][j] = pc(u_pred, 'U'); m[i][j] = uv;
                        } else { path[i][j] = pc(l_pred, 'L'); m[i][j] = lv; }
                    } else {
                        if (d < uv) { path[i][j] = pc(u_pred, 'U'); m[i][j] = uv; }
                        else { path[i][j] = pc(d_idx, lnz[i] == seq[j + left] ? 'D' : 'd'); m[i][j] = d; }
                    }
                } else {  // :181-194
                    if (lv < uv) { path[i][j] = pc(u_pred, 'U'); m[i][j] = uv; }
                    else { path[i][j] = pc(l_pred, 'L'); m[i][j] = lv; }
                }
                ncells += 1;
            }
            if (m[i][j] >= m[i][best_val_pos]) best_val_pos = j;
        }
        bsp[i] = best_val_pos + left;
    }
    size_t last_row = L - 2, last_col = m[last_row].size() - 1;  // :206-214
    for (size_t p : preds_of(g, L - 1)) {
        size_t tlc = (ampl[p].second - ampl[p].first) - 1;
        if (m[p][tlc] > m[last_row][last_col]) { last_row = p; last_col = tlc; }
    }
    res.score = m[last_row][last_col];
    if (cells) *cells = ncells;
    if (sc.panicked) { res.would_panic = true; return res; }
    // band_ampl_enough :371-455
    {
        size_t i = last_row, j = last_col; bool ok = true;
        auto inb = [&](size_t r, size_t c) { return r < L - 1 && c < path[r].size(); };
        while (true) {
            if (!inb(i, j)) { res.would_panic = true; return res; }
            if (path[i][j].dir == 'O') break;
            auto [left, right] = ampl[i];
            if (i == 0 || (j == 0 && left == 0)) break;
            if ((j == 0 && left != 0) || (j == right - left - 1 && right != W)) { ok = false; break; }
            size_t pred = path[i][j].pred; char dir = path[i][j].dir;
            if (dir == 'D' || dir == 'd') {
                size_t jp; if (!jpos(j, left, ampl[pred].first, jp) || jp == 0) { res.would_panic = true; return res; }
                j = jp - 1; i = pred;
            } else if (dir == 'L') {
                if (path_x[i][j].dir == 'X') { while (inb(i, j) && path_x[i][j].dir == 'X' && j > 0) j -= 1; }
                else j -= 1;
            } else if (dir == 'U') {
                if (path_y[i][j].dir == 'Y') {
                    while (true) {
                        if (!inb(i, j)) { res.would_panic = true; return res; }
                        if (path_y[i][j].dir != 'Y') break;
                        size_t left_row = ampl[i].first, p = path_y[i][j].pred, jp;
                        if (!jpos(j, left_row, ampl[p].first, jp)) { res.would_panic = true; return res; }
                        j = jp; i = p;
                    }
                } else {
                    size_t p = path[i][j].pred, jp;
                    if (!jpos(j, left, ampl[p].first, jp)) { res.would_panic = true; return res; }
                    j = jp; i = p;
                }
            } else { ok = false; break; }
        }
        if (!ok) res.out += "Band length probably too short, maybe try with larger b and f\n";
    }
    if (idx == 0) return res;
    // ---- gaf_output.rs:96-253 ----
    size_t col = last_col, row = last_row;
    std::vector<const std::string*> hia;
    std::vector<std::string> cigars;
    std::string cigar;
    int cm = 0, ci = 0, cd = 0;
    std::string curr_handle = "";
    char last_dir = ' ';
    size_t path_length = 0, residue = 0;
    auto inb = [&](size_t r, size_t c) { return r < L - 1 && c < path[r].size(); };
    while (true) {
        if (!inb(row, col)) { res.would_panic = true; return res; }
        if (path[row][col].dir == 'O') break;
        size_t pred = path[row][col].pred; char dir = path[row][col].dir;
        if (g.hofp[row] != curr_handle) {
            if (!set_cigar_substring(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cigars.insert(cigars.begin(), cigar);
            cigar.clear(); cm = ci = cd = 0;
        }
        curr_handle = g.hofp[row];
        if (std::toupper(dir) != std::toupper(last_dir)) {
            if (!set_cigar_substring(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cm = ci = cd = 0;
        }
        last_dir = dir;
        size_t p_left = ampl[pred].first, jp; bool jp_ok = true;
        if (ampl[row].first < p_left) { size_t delta = p_left - ampl[row].first; jp_ok = col >= delta; jp = col - delta; }
        else jp = col + (ampl[row].first - p_left);
        if (dir == 'D' || dir == 'd') {
            hia.push_back(&g.hofp[row]);
            if (!jp_ok || jp == 0) { res.would_panic = true; return res; }
            row = pred; col = jp - 1; cm += 1; path_length += 1;
            if (dir == 'D') residue += 1;
        } else if (dir == 'L') {
            if (path_x[row][col].dir == 'X') {
                while (true) {
                    if (!inb(row, col)) { res.would_panic = true; return res; }
                    if (path_x[row][col].dir != 'X') break;
                    cd += 1;
                    if (col == 0) { res.would_panic = true; return res; }
                    col -= 1;
                }
            } else {
                cd += 1;
                if (col == 0) { res.would_panic = true; return res; }
                col -= 1;
            }
        } else if (dir == 'U') {
            if (path_y[row][col].dir == 'Y') {
                while (true) {
                    if (!inb(row, col)) { res.would_panic = true; return res; }
                    if (path_y[row][col].dir != 'Y') break;
                    size_t left_row = ampl[row].first, p = path_y[row][col].pred, jq;
                    if (!jpos(col, left_row, ampl[p].first, jq)) { res.would_panic = true; return res; }
                    hia.push_back(&g.hofp[row]);
                    ci += 1; path_length += 1;
                    col = jq; row = p;
                }
            } else {
                hia.push_back(&g.hofp[row]);
                ci += 1; path_length += 1;
                if (!jp_ok) { res.would_panic = true; return res; }
                row = pred; col = jp;
            }
        } else { res.would_panic = true; return res; }
    }
    if (!set_cigar_substring(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
    cigars.insert(cigars.begin(), cigar);
    std::vector<const std::string*> dd;
    for (auto* s : hia) if (dd.empty() || *dd.back() != *s) dd.push_back(s);
    std::reverse(dd.begin(), dd.end());
    GAF gaf;
    gaf.query_name = name;
    gaf.query_length = W - 1;
    gaf.query_start = col;
    gaf.query_end = last_col + ampl[last_row].first;
    gaf.strand = g.strand;
    gaf.path.clear();
    for (auto* s : dd) {
        if ((*s)[0] == '-') { res.would_panic = true; return res; }
        gaf.path.push_back(std::stoull(*s));
    }
    gaf.path_length = path_length;
    gaf.path_start = node_start(g.hofp, row);
    gaf.path_end = node_start(g.hofp, last_row);
    gaf.residue_matches_number = residue;
    gaf.alignment_block_length = "*";
    gaf.mapping_quality = "*";
    std::string comments;
    for (size_t k = 0; k + 1 < cigars.size(); ++k) { if (k) comments += ","; comments += cigars[k]; }
    gaf.comments = comments;
    res.out += gaf.to_string() + "\n";
    return res;
}

}  // namespace orc
