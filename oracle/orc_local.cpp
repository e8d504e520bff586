// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp header).
// Local POA modes: -m 1 (local_poa::exec_simd, the AVX2 f32 path, and local_poa::exec, the
// scalar i32 path) and -m 3 (gap_local_poa::exec), with their GAF walkers.  Literal restatements: full
// L x W matrices, same loop order, same clamp and tie rules (including the `first = false` quirk of
// get_best_d / get_best_u and the unclamped multi-predecessor tail of the AVX2 path).
#include <algorithm>
#include <cmath>

#include "orc_common.hpp"

namespace orc {

namespace {

// gaf_output.rs:867-874
size_t node_start_l(const std::vector<std::string>& hofp, size_t row) {
    const std::string& id = hofp[row];
    size_t i = row;
    while (hofp[i] == id && i > 0) i -= 1;
    return row - i;
}

// gaf_output.rs:876-892
bool cigar_sub(int cm, int ci, int cd, std::string& cs) {
    if (cm * ci + ci * cd + cm * cd != 0) return false;
    if (cm > 0) cs = std::to_string(cm) + "M" + cs;
    else if (ci > 0) cs = std::to_string(ci) + "I" + cs;
    else if (cd > 0) cs = std::to_string(cd) + "D" + cs;
    return true;
}

struct PC { uint16_t pred = 0; char dir = 'O'; };
inline PC pcl(size_t pred, char dir) { return PC{(uint16_t)pred, dir}; }  // bitfield_path.rs: `pred as u16`

// utils.rs:129-140
inline void max_d_u_l(int d, int u, int l, int& best, char& dir) {
    if (d < u) { if (u < l) { best = l; dir = 'L'; } else { best = u; dir = 'U'; } }
    else { if (d < l) { best = l; dir = 'L'; } else { best = d; dir = 'D'; } }
}

// shared tail of the three walkers (gaf_output.rs:455-488 and twins)
bool finish_gaf(Result& res, const LnzGraph& g, const std::string& name, size_t W, size_t row, size_t col,
                size_t last_row, size_t last_col, std::vector<const std::string*>& hia,
                const std::vector<std::string>& cigars, size_t path_length, size_t residue) {
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
    for (auto* s : dd) {
        if ((*s)[0] == '-') return false;  // "-1".parse::<usize>() fails
        gaf.path.push_back(std::stoull(*s));
    }
    gaf.path_length = path_length;
    gaf.path_start = node_start_l(g.hofp, row);
    gaf.path_end = node_start_l(g.hofp, last_row);
    gaf.residue_matches_number = residue;
    gaf.alignment_block_length = "*";
    gaf.mapping_quality = "*";
    std::string comments;
    for (size_t k = 0; k + 1 < cigars.size(); ++k) { if (k) comments += ","; comments += cigars[k]; }
    gaf.comments = comments;
    res.out += gaf.to_string() + "\n";
    return true;
}

}  // namespace

// =================================================================================
// -m 1, AVX2 path.  src/local_poa.rs:9-174; GAF gaf_output.rs:636-752
// =================================================================================
Result m1_simd(const std::string& read, const std::string& name, size_t idx, const LnzGraph& g, const Scores& sc,
               uint64_t* cells) {
    Result res;
    sc.panicked = false;
    const size_t L = g.lnz.size(), W = read.size();
    const std::string& lnz = g.lnz;
    auto S = [&](char a, char b) { return (float)sc.get(a, b); };
    std::vector<std::vector<float>> m(L, std::vector<float>(W, 0.0f)), path(L, std::vector<float>(W, 0.0f));
    const size_t max_multiple = W % 8 != 0 ? (W / 8) * 8 : W - 8;  // :19-23
    size_t best_row = 0, best_col = 0;
    uint64_t ncells = 0;
    for (size_t i = 1; i + 1 < L; ++i) {
        const float us_update = S(lnz[i], '-');
        for (size_t j = 1; j < max_multiple + 1; j += 8) {
            float ds_update[8];
            for (int k = 0; k < 8; ++k) ds_update[k] = S(lnz[i], read[j + k]);
            if (!g.nwp[i]) {
                for (int k = 0; k < 8; ++k) {
                    float us = m[i - 1][j + k] + us_update;
                    float ds = m[i - 1][j + k - 1] + ds_update[k];
                    bool bc = ds > us;
                    m[i][j + k] = bc ? ds : us;
                    path[i][j + k] = (float)(i - 1) + (bc ? 0.1f : 0.2f);
                }
            } else {
                const auto& preds = g.pred_hash.at(i);
                for (int k = 0; k < 8; ++k) {
                    float best_us = m[preds[0]][j + k], best_ds = m[preds[0]][j + k - 1];
                    float pus = (float)preds[0], pds = (float)preds[0];
                    for (size_t q = 1; q < preds.size(); ++q) {
                        float us = m[preds[q]][j + k], ds = m[preds[q]][j + k - 1];
                        if (us > best_us) { best_us = us; pus = (float)preds[q]; }
                        if (ds > best_ds) { best_ds = ds; pds = (float)preds[q]; }
                    }
                    best_us += us_update;
                    best_ds += ds_update[k];
                    bool bc = best_ds > best_us;
                    m[i][j + k] = bc ? best_ds : best_us;
                    pds += 0.1f;
                    pus += 0.2f;
                    path[i][j + k] = bc ? pds : pus;
                }
            }
            for (size_t x = j; x < std::min(j + 8, W); ++x) {  // :93-107
                float l = m[i][x - 1] + S(read[j], '-');
                if (l > m[i][x]) { m[i][x] = l; path[i][x] = (float)i + 0.3f; }
                if (m[i][x] <= 0.0f) { m[i][x] = 0.0f; path[i][x] = 0.0f; }
                if (m[i][x] >= m[best_row][best_col]) { best_row = i; best_col = x; }
            }
            ncells += 8;
        }
        for (size_t j = max_multiple + 1; j < W; ++j) {  // :109-163
            if (!g.nwp[i]) {
                float l = m[i][j - 1] + S(read[j], '-');
                float u = m[i - 1][j] + S(lnz[i], '-');
                float d = m[i - 1][j - 1] + S(lnz[i], read[j]);
                m[i][j] = std::max(std::max(l, u), d);
                if (m[i][j] < 0.0f) { m[i][j] = 0.0f; path[i][j] = 0.0f; }
                else if (m[i][j] == d) path[i][j] = (float)(i - 1) + 0.1f;
                else if (m[i][j] == u) path[i][j] = (float)(i - 1) + 0.2f;
                else path[i][j] = (float)i + 0.3f;
            } else {
                float u = 0, d = 0; size_t u_pred = 0, d_pred = 0; bool first = true;
                for (size_t p : g.pred_hash.at(i)) {
                    if (first) { u = m[p][j]; d = m[p][j - 1]; u_pred = p; d_pred = p; first = false; }
                    if (m[p][j] > u) { u = m[p][j]; u_pred = p; }
                    if (m[p][j - 1] > d) { d = m[p][j - 1]; d_pred = p; }
                }
                u += S(lnz[i], '-');
                d += S(read[j], lnz[i]);  // :147 swapped key
                float l = m[i][j - 1] + S(read[j], '-');
                m[i][j] = std::max(std::max(l, u), d);  // no clamp in this branch
                if (m[i][j] == d) path[i][j] = (float)d_pred + 0.1f;
                else if (m[i][j] == u) path[i][j] = (float)u_pred + 0.2f;
                else path[i][j] = (float)i + 0.3f;
            }
            if (m[i][j] >= m[best_row][best_col]) { best_row = i; best_col = j; }
            ncells += 1;
        }
    }
    res.score = (int)m[best_row][best_col];  // main.rs:120 `as i32`
    if (cells) *cells = ncells;
    if (sc.panicked) { res.would_panic = true; return res; }
    if (idx == 0) return res;

    // ---- gaf_output.rs:636-752 gaf_of_local_poa_simd ----
    size_t col = best_col, row = best_row;
    const size_t last_row = best_row, last_col = best_col;
    std::vector<const std::string*> hia;
    std::vector<std::string> cigars;
    std::string cigar;
    int cm = 0, ci = 0, cd = 0;
    std::string curr_handle = "";
    long last_dir = -1;
    size_t path_length = 0, residue = 0;
    while (path[row][col] != 0.0f) {
        std::string vs = f32_display(path[row][col]);  // :664-668
        size_t dot = vs.find('.');
        if (dot == std::string::npos || vs[0] == '-') { res.would_panic = true; return res; }
        if (vs.find('.', dot + 1) != std::string::npos) { res.would_panic = true; return res; }
        size_t pred = std::stoull(vs.substr(0, dot));
        std::string frac = vs.substr(dot + 1);
        if (frac.empty() || frac.size() > 9) { res.would_panic = true; return res; }
        long dir = std::stol(frac);
        if (g.hofp[row] != curr_handle) {
            if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cigars.insert(cigars.begin(), cigar);
            cigar.clear(); cm = ci = cd = 0;
        }
        curr_handle = g.hofp[row];
        if (dir != last_dir) {
            if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cm = ci = cd = 0;
        }
        last_dir = dir;
        if (dir == 1) {
            hia.push_back(&g.hofp[row]);
            row = pred;
            if (col == 0) { res.would_panic = true; return res; }
            col -= 1; cm += 1; path_length += 1; residue += 1;
        } else if (dir == 3) {
            if (col == 0) { res.would_panic = true; return res; }
            col -= 1; cd += 1;
        } else if (dir == 2) {
            hia.push_back(&g.hofp[row]);
            row = pred; ci += 1; path_length += 1;
        } else { res.would_panic = true; return res; }
        if (row >= L) { res.would_panic = true; return res; }
    }
    if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
    cigars.insert(cigars.begin(), cigar);
    if (!finish_gaf(res, g, name, W, row, col, last_row, last_col, hia, cigars, path_length, residue)) {
        res.would_panic = true; res.out.clear();
    }
    return res;
}

// get_best_d / get_best_u of local_poa.rs:263-298 and gap_local_poa.rs:126-144: `first` starts FALSE, so the
// running maximum starts from (0, row 0) instead of the first predecessor.
static void quirk_best(const std::vector<std::vector<int>>& m, const std::vector<size_t>& p_arr, size_t j, int& v,
                       size_t& idx) {
    v = 0; idx = 0;
    for (size_t p : p_arr) {
        int cur = m[p][j];
        if (cur > v) { v = cur; idx = p; }
    }
}

// =================================================================================
// -m 1 scalar.  src/local_poa.rs:176-262; GAF gaf_output.rs:383-488
// =================================================================================
Result m1_scalar(const std::string& seq, const std::string& name, size_t idx, const LnzGraph& g, const Scores& sc,
                 uint64_t* cells) {
    Result res;
    sc.panicked = false;
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size();
    std::vector<std::vector<int>> m(L, std::vector<int>(W, 0));
    std::vector<std::vector<PC>> path(L, std::vector<PC>(W));
    size_t best_row = 0, best_col = 0;
    uint64_t ncells = 0;
    for (size_t i = 0; i + 1 < L; ++i) {
        for (size_t j = 0; j < W; ++j) {
            if (i == 0 || j == 0) path[i][j] = pcl(0, 'O');
            else {
                int l = m[i][j - 1] + sc.get(seq[j], '-');
                int d, u; size_t d_idx, u_idx;
                if (!g.nwp[i]) {
                    d = m[i - 1][j - 1] + sc.get(seq[j], lnz[i]); d_idx = i - 1;
                    u = m[i - 1][j] + sc.get('-', lnz[i]); u_idx = i - 1;
                } else {
                    const auto& pr = g.pred_hash.at(i);
                    quirk_best(m, pr, j - 1, d, d_idx);
                    quirk_best(m, pr, j, u, u_idx);
                    d += sc.get(seq[j], lnz[i]);
                    u += sc.get('-', lnz[i]);
                }
                if (d < 0 && l < 0 && u < 0) { m[i][j] = 0; path[i][j] = pcl(0, 'O'); }
                else {
                    int best; char dir;
                    max_d_u_l(d, u, l, best, dir);
                    if (dir == 'D' && lnz[i] != seq[j]) dir = 'd';
                    m[i][j] = best;
                    path[i][j] = (dir == 'D' || dir == 'd') ? pcl(d_idx, dir) : dir == 'U' ? pcl(u_idx, 'U') : pcl(i, 'L');
                }
                ncells += 1;
            }
            if (m[i][j] > m[best_row][best_col]) { best_row = i; best_col = j; }
        }
    }
    res.score = m[best_row][best_col];
    if (cells) *cells = ncells;
    if (sc.panicked) { res.would_panic = true; return res; }
    if (idx == 0) return res;
    // ---- gaf_output.rs:383-488 ----
    size_t col = best_col, row = best_row;
    std::vector<const std::string*> hia;
    std::vector<std::string> cigars;
    std::string cigar;
    int cm = 0, ci = 0, cd = 0;
    std::string curr_handle = "";
    char last_dir = ' ';
    size_t path_length = 0, residue = 0;
    while (path[row][col].dir != 'O') {
        size_t pred = path[row][col].pred; char dir = path[row][col].dir;
        if (g.hofp[row] != curr_handle) {
            if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cigars.insert(cigars.begin(), cigar);
            cigar.clear(); cm = ci = cd = 0;
        }
        curr_handle = g.hofp[row];
        if (std::toupper(dir) != std::toupper(last_dir)) {
            if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cm = ci = cd = 0;
        }
        last_dir = dir;
        if (dir == 'D' || dir == 'd') {
            hia.push_back(&g.hofp[row]);
            row = pred; col -= 1; cm += 1; path_length += 1;
            if (dir == 'D') residue += 1;
        } else if (dir == 'L') { col -= 1; cd += 1; }
        else if (dir == 'U') { hia.push_back(&g.hofp[row]); row = pred; ci += 1; path_length += 1; }
        else { res.would_panic = true; return res; }
        if (row >= L - 1) { res.would_panic = true; return res; }  // u16-truncated pred (rows > 65535)
    }
    if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
    cigars.insert(cigars.begin(), cigar);
    if (!finish_gaf(res, g, name, W, row, col, best_row, best_col, hia, cigars, path_length, residue)) {
        res.would_panic = true; res.out.clear();
    }
    return res;
}

// =================================================================================
// -m 3.  src/gap_local_poa.rs:6-124 (+ get_best_u :145-183); GAF gaf_output.rs:489-635
// =================================================================================
Result m3_gap_local(const std::string& seq, const std::string& name, size_t idx, const LnzGraph& g, const Scores& sc,
                    int o, int e, uint64_t* cells) {
    Result res;
    sc.panicked = false;
    const std::string& lnz = g.lnz;
    const size_t L = lnz.size(), W = seq.size();
    std::vector<std::vector<int>> m(L, std::vector<int>(W, 0)), x = m, y = m;
    std::vector<std::vector<PC>> path(L, std::vector<PC>(W)), path_x = path, path_y = path;
    size_t best_row = 0, best_col = 0;
    uint64_t ncells = 0;
    for (size_t i = 0; i + 1 < L; ++i) {
        for (size_t j = 0; j < W; ++j) {
            if (i == 0 || j == 0) { path[i][j] = pcl(0, 'O'); path_x[i][j] = pcl(0, 'O'); path_y[i][j] = pcl(0, 'O'); }
            else {
                int l_x = x[i][j - 1] + e, l_m = m[i][j - 1] + o + e, l;
                if (l_x > l_m) { path_x[i][j] = pcl(i, 'X'); l = l_x; }
                else { path_x[i][j] = pcl(i, 'M'); l = l_m; }
                x[i][j] = l;
                int d, u; size_t d_idx, u_idx;
                if (!g.nwp[i]) {
                    d = m[i - 1][j - 1] + sc.get(seq[j], lnz[i]); d_idx = i - 1;
                    int u_y = y[i - 1][j] + e, u_m = m[i - 1][j] + o + e;
                    u_idx = i - 1;
                    if (u_y > u_m) { path_y[i][j] = pcl(u_idx, 'Y'); u = u_y; }
                    else { path_y[i][j] = pcl(u_idx, 'M'); u = u_m; }
                    y[i][j] = u;
                } else {
                    const auto& pr = g.pred_hash.at(i);
                    quirk_best(m, pr, j - 1, d, d_idx);
                    int u_m = 0, u_y = 0; size_t u_m_idx = 0, u_y_idx = 0;  // get_best_u, first = false
                    for (size_t p : pr) {
                        int cum = m[p][j] + o, cuy = y[p][j];
                        if (cum > u_m) { u_m = cum; u_m_idx = p; }
                        if (cuy > u_y) { u_y = cuy; u_y_idx = p; }
                    }
                    bool from_m;
                    if (u_m > u_y) { u = u_m; u_idx = u_m_idx; from_m = true; }
                    else { u = u_y; u_idx = u_y_idx; from_m = false; }
                    d += sc.get(seq[j], lnz[i]);
                    u += e;
                    y[i][j] = u;
                    path_y[i][j] = pcl(u_idx, from_m ? 'M' : 'Y');
                }
                if (d < 0 && l < 0 && u < 0) { m[i][j] = 0; path[i][j] = pcl(0, 'O'); }
                else {
                    int best; char dir;
                    max_d_u_l(d, u, l, best, dir);
                    if (dir == 'D' && lnz[i] != seq[j]) dir = 'd';
                    m[i][j] = best;
                    path[i][j] = (dir == 'D' || dir == 'd') ? pcl(d_idx, dir) : dir == 'U' ? pcl(u_idx, 'U') : pcl(i, 'L');
                }
                ncells += 1;
            }
            if (m[i][j] > m[best_row][best_col]) { best_row = i; best_col = j; }
        }
    }
    res.score = m[best_row][best_col];
    if (cells) *cells = ncells;
    if (sc.panicked) { res.would_panic = true; return res; }
    if (idx == 0) return res;
    // ---- gaf_output.rs:489-635 ----
    size_t col = best_col, row = best_row;
    std::vector<const std::string*> hia;
    std::vector<std::string> cigars;
    std::string cigar;
    int cm = 0, ci = 0, cd = 0;
    std::string curr_handle = "";
    char last_dir = ' ';
    size_t path_length = 0, residue = 0;
    while (path[row][col].dir != 'O') {
        size_t pred = path[row][col].pred; char dir = path[row][col].dir;
        if (g.hofp[row] != curr_handle) {
            if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cigars.insert(cigars.begin(), cigar);
            cigar.clear(); cm = ci = cd = 0;
        }
        curr_handle = g.hofp[row];
        if (std::toupper(dir) != std::toupper(last_dir)) {
            if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
            cm = ci = cd = 0;
        }
        last_dir = dir;
        if (dir == 'D' || dir == 'd') {
            hia.push_back(&g.hofp[row]);
            row = pred; col -= 1; cm += 1; path_length += 1;
            if (dir == 'D') residue += 1;
        } else if (dir == 'L') {
            if (path_x[row][col].dir == 'X') {
                while (path_x[row][col].dir == 'X') { cd += 1; col -= 1; }  // column 0 is 'O': always stops
            } else { cd += 1; col -= 1; }
        } else if (dir == 'U') {
            if (path_y[row][col].dir == 'Y') {
                while (path_y[row][col].dir == 'Y') {
                    size_t p = path_y[row][col].pred;
                    hia.push_back(&g.hofp[row]);
                    row = p; ci += 1; path_length += 1;
                    if (row >= L - 1) { res.would_panic = true; return res; }
                }
            } else { hia.push_back(&g.hofp[row]); ci += 1; path_length += 1; row = pred; }
        } else { res.would_panic = true; return res; }
        if (row >= L - 1) { res.would_panic = true; return res; }
    }
    if (!cigar_sub(cm, ci, cd, cigar)) { res.would_panic = true; return res; }
    cigars.insert(cigars.begin(), cigar);
    if (!finish_gaf(res, g, name, W, row, col, best_row, best_col, hia, cigars, path_length, residue)) {
        res.would_panic = true; res.out.clear();
    }
    return res;
}

}  // namespace orc
