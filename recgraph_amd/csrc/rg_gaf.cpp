// Host-side GAF text for librecgraph_hip: turns the packed traceback records that come back from
// the device into exactly the bytes the reference prints on stdout.
//   GAFStruct::to_string ........................ src/gaf_output.rs:70-94
//   gaf_of_global_abpoa_simd (field rules) ...... src/gaf_output.rs:818-864
//   gaf_of_global_abpoa / gaf_of_gap_abpoa ...... src/gaf_output.rs:254-381, 96-253
//   build_alignment / no_rec / rec .............. src/pathwise_alignment_output.rs:140-183,
//                                                 src/recombination_output.rs:558-630, 738-781
//   build_cigar ................................. src/pathwise_alignment_output.rs:471-556
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "rg_codes.hpp"
#include "rg_host.hpp"

namespace rg {

// Rust `{}` for f32 (core::fmt::float: flt2dec shortest digits, then fixed notation): the SHORTEST digit string that
// round-trips, laid out positionally and padded with zeros — never the exact binary value (2^100 prints as
// 1267650600000000000000000000000, not as its 31 exact digits: std::to_chars in fixed format gives the latter for values
// beyond 2^24, which round 6's independent check against numpy's Dragon4 found), never an exponent, "-0" for negative zero.
std::string f32_display(float v) {
    if (v != v) return "NaN";
    if (v == INFINITY) return "inf";
    if (v == -INFINITY) return "-inf";
    char b[64];
    auto r = std::to_chars(b, b + sizeof b - 1, v, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest round-trip digits
    *r.ptr = 0;
    const char* p = b;
    std::string out;
    if (*p == '-') { out += '-'; ++p; }
    std::string digits;
    for (; p < r.ptr && *p != 'e'; ++p) if (*p != '.') digits += *p;
    const int exp10 = (p < r.ptr) ? atoi(p + 1) : 0;          // value = 0.d1d2... x 10^(exp10 + 1)
    if (digits.find_first_not_of('0') == std::string::npos) { out += '0'; return out; }
    const int point = exp10 + 1;                              // digits in front of the decimal point
    const int nd = (int)digits.size();
    if (point <= 0) { out += "0."; out.append((size_t)(-point), '0'); out += digits; }
    else if (point >= nd) { out += digits; out.append((size_t)(point - nd), '0'); }
    else { out.append(digits, 0, (size_t)point); out += '.'; out.append(digits, (size_t)point, std::string::npos); }
    return out;
}

std::string GafFields::line() const {
    std::string s = name;
    s += '\t'; s += std::to_string(qlen);
    s += '\t'; s += std::to_string(qstart);
    s += '\t'; s += std::to_string(qend);
    s += '\t'; s += strand;
    s += "\t>";
    for (size_t i = 0; i < path.size(); ++i) { if (i) s += '>'; s += std::to_string(path[i]); }
    s += '\t'; s += std::to_string(plen);
    s += '\t'; s += std::to_string(pstart);
    s += '\t'; s += std::to_string(pend);
    s += '\t'; s += std::to_string(residues);
    s += empty ? "\t\t\t" : "\t*\t*\t";
    s += comments;
    return s;
}

namespace {

// GAFStruct::new() (gaf_output.rs:22-38): what gaf_of_global_abpoa_simd returns when the band was not enough
GafFields empty_gaf() {
    GafFields f;
    f.empty = true;
    f.strand = ' ';
    f.path.assign(1, 0);
    return f;
}

// run-length text of a D/d/U/L op string: D->M, d->X, U->I, L->D
std::string rle_cigar(const std::string& ops) {
    std::string out;
    size_t i = 0;
    while (i < ops.size()) {
        size_t j = i;
        while (j < ops.size() && ops[j] == ops[i]) ++j;
        out += std::to_string(j - i);
        out += ops[i] == 'D' ? 'M' : ops[i] == 'd' ? 'X' : ops[i] == 'U' ? 'I' : 'D';
        i = j;
    }
    return out;
}

void dedup(std::vector<uint64_t>& v) { v.erase(std::unique(v.begin(), v.end()), v.end()); }

inline char read_at(const std::string& read, int col) { return col == 0 ? '$' : read[(size_t)col - 1]; }

// rows after `node` that still belong to its segment (utils.rs:242-250)
size_t tail_in_segment(const HostGraph& g, int node) {
    size_t off = 0;
    if (node > 0) {
        uint64_t id = g.node_id[node];
        int c = node + 1;
        while (c < g.L - 1 && g.node_id[c] == id) { ++c; ++off; }
    }
    return off;
}
size_t head_in_segment(const HostGraph& g, int node) {  // utils.rs:227-235
    size_t off = 0;
    if (node > 0) {
        uint64_t id = g.node_id[node];
        int c = node - 1;
        while (c > 0 && g.node_id[c] == id) { --c; ++off; }
    }
    return off;
}

// predecessor of `row` on `path` in the forward PredHash / successor in the reverse one
int step_on_path(const HostGraph& g, int row, int path, bool fwd) {
    if (fwd) {
        if (!g.pnwp[row]) return row - 1;
        int p = -1;
        for (int e = g.eoff[row]; e < g.eoff[row + 1]; ++e) if (g.emask[e].test(path)) p = g.epred[e];
        return p < 0 ? row - 1 : p;
    }
    if (!g.rnwp[row]) return row + 1;
    int p = -1;
    for (int e = g.roff[row]; e < g.roff[row + 1]; ++e) if (g.rmask[e].test(path)) p = g.rsucc[e];
    return p < 0 ? row + 1 : p;
}

}  // namespace

// ---------------------------------------------------------------------------------
// node ids per row under `-s true`: the k-th node in row order is labelled with the id of the k-th node from the end
// (create_handle_pos_in_lnz over the reversed handle list, utils.rs:144-165 + graph.rs:128-142)
void build_rev_ids(HostGraph& g) {
    if (!g.node_id_rev.empty()) return;
    std::vector<uint64_t> runs;
    for (int i = 1; i + 1 < g.L; ++i)
        if (i == 1 || g.seg_off[i] == 1) runs.push_back(g.node_id[i]);
    g.node_id_rev.assign(g.L, 0);
    size_t k = 0;
    for (int i = 1; i + 1 < g.L; ++i) {
        if (i > 1 && g.seg_off[i] == 1) ++k;
        g.node_id_rev[i] = runs[runs.size() - 1 - k];
    }
}

// Row lists of the paths (HostGraph::pl_*), checked step by step against the PredHash lookups they replace in the formatter
void build_path_lists(HostGraph& g) {
    const int P = g.P, L = g.L;
    g.pl_ok = false;
    g.pl_off.assign((size_t)P + 1, 0);
    if (!g.has_path || P < 1) return;
    for (int i = 1; i + 1 < L; ++i)
        for (int k = 0; k < P; ++k) if (g.row_mask[(size_t)i].test(k)) ++g.pl_off[(size_t)k + 1];
    for (int k = 0; k < P; ++k) g.pl_off[(size_t)k + 1] += g.pl_off[(size_t)k];
    const size_t total = (size_t)g.pl_off[(size_t)P];
    g.pl_row.assign(total, 0); g.pl_base.assign(total, 'N'); g.pl_id.assign(total, 0);
    std::vector<int32_t> at(g.pl_off.begin(), g.pl_off.end() - 1);
    for (int i = 1; i + 1 < L; ++i)
        for (int k = 0; k < P; ++k)
            if (g.row_mask[(size_t)i].test(k)) {
                const size_t t = (size_t)at[(size_t)k]++;
                g.pl_row[t] = i; g.pl_base[t] = g.lnz[(size_t)i]; g.pl_id[t] = g.node_id[(size_t)i];
            }
    bool ok = true;
    for (int k = 0; k < P && ok; ++k) {
        const int32_t o = g.pl_off[(size_t)k], c = g.pl_off[(size_t)k + 1] - o;
        for (int32_t t = 0; t < c && ok; ++t) {
            const int row = g.pl_row[(size_t)(o + t)];
            ok = step_on_path(g, row, k, true) == (t ? g.pl_row[(size_t)(o + t - 1)] : 0) &&
                 step_on_path(g, row, k, false) == (t + 1 < c ? g.pl_row[(size_t)(o + t + 1)] : L - 1);
        }
    }
    g.pl_ok = ok;
}

GafFields fields_m0_simd(const HostGraph& g, const std::string& read, const std::string& name,
                         const ReadRecord& r, int amb) {
    const std::vector<uint64_t>& nid = (amb & 1) ? g.node_id_rev : g.node_id;
    if (r.status & RG_READ_BAND_NOT_ENOUGH) {
        GafFields e = empty_gaf();
        e.pre = "band not enough for correct output\n";
        return e;
    }
    int row = r.end_row, col = r.end_col;
    std::string ops, pseq;
    std::vector<uint64_t> ids;
    size_t plen = 0, residues = 0;
    for (int k = 0; k < r.n_ops; ++k) {
        uint8_t op = r.ops[k] & 0x7f;
        if (op == OP_D) {
            ids.push_back(nid[row]); pseq.push_back(g.lnz[row]);
            row = r.rows[k]; col -= 1;
            ops.push_back(g.lnz[row] == read_at(read, col) ? 'D' : 'd');   // tested on the destination cell
            ++plen; ++residues;
        } else if (op == OP_U) {
            ids.push_back(nid[row]); pseq.push_back(g.lnz[row]);
            row = r.rows[k];
            ops.push_back('U'); ++plen;
        } else { col -= 1; ops.push_back('L'); }
    }
    std::reverse(ops.begin(), ops.end());
    std::reverse(pseq.begin(), pseq.end());
    dedup(ids);
    std::reverse(ids.begin(), ids.end());
    GafFields f;
    f.name = name; f.qlen = read.size(); f.qstart = (size_t)col; f.qend = (size_t)r.end_col;
    f.strand = (amb & 2) ? '-' : '+';
    f.path = ids; f.plen = plen;
    f.pstart = (size_t)g.seg_off[row];          // node_start (gaf_output.rs:867-874)
    f.pend = (size_t)g.seg_off[r.end_row];
    f.residues = residues;
    f.comments = rle_cigar(ops) + ", score: " + f32_display(r.fscore) + "\t" + pseq;
    return f;
}

// ---------------------------------------------------------------------------------
// m0 scalar and m2: per-segment cigar strings (gaf_output.rs:96-381).  Ops flagged OP_CONT were
// produced inside an X/Y run of the m2 walker, which does not re-check segment/direction changes.
GafFields fields_poa_banded(const HostGraph& g, const std::string& read, const std::string& name,
                            const ReadRecord& r, int amb) {
    const std::vector<uint64_t>& nid = (amb & 1) ? g.node_id_rev : g.node_id;
    GafFields f;
    if (r.status & RG_READ_BAND_WARNING) f.pre = "Band length probably too short, maybe try with larger b and f\n";
    int row = r.end_row;
    std::vector<std::string> cigars;   // front insertion order reproduced by reversing at the end
    std::string cigar;
    long cm = 0, ci = 0, cd = 0;
    auto flush = [&]() {
        if (cm > 0) cigar = std::to_string(cm) + "M" + cigar;
        else if (ci > 0) cigar = std::to_string(ci) + "I" + cigar;
        else if (cd > 0) cigar = std::to_string(cd) + "D" + cigar;
        cm = ci = cd = 0;
    };
    bool have_handle = false;
    uint64_t curr_handle = 0;
    bool curr_is_root = false;   // hofp[0] = "-1"
    char last_dir = ' ';
    std::vector<uint64_t> ids;
    size_t plen = 0, residues = 0;
    for (int k = 0; k < r.n_ops; ++k) {
        const uint8_t raw = r.ops[k];
        const uint8_t op = raw & 0x3f;
        const bool cont = raw & OP_CONT;
        const bool mismatch = raw & 0x40;
        if (!cont) {
            bool is_root = row == 0;
            uint64_t h = nid[row];
            if (!have_handle || is_root != curr_is_root || h != curr_handle) {
                flush();
                cigars.push_back(cigar);
                cigar.clear();
            }
            have_handle = true; curr_handle = h; curr_is_root = is_root;
            char d = op == OP_D ? 'D' : op == OP_U ? 'U' : 'L';
            if (d != last_dir) flush();
            last_dir = d;
        }
        if (op == OP_D) {
            ids.push_back(nid[row]); row = r.rows[k]; cm += 1; ++plen;
            if (!mismatch) ++residues;
        } else if (op == OP_U) {
            ids.push_back(nid[row]); row = r.rows[k]; ci += 1; ++plen;
        } else cd += 1;
    }
    flush();
    cigars.push_back(cigar);
    dedup(ids);
    std::reverse(ids.begin(), ids.end());
    f.name = name; f.qlen = read.size(); f.qstart = (size_t)r.stop_col; f.qend = (size_t)r.end_col;
    f.strand = (amb & 2) ? '-' : '+';
    f.path = ids; f.plen = plen;
    f.pstart = (size_t)g.seg_off[row];
    f.pend = (size_t)g.seg_off[r.end_row];
    f.residues = residues;
    // cigars were inserted at the front: text = all but the first-created one, newest first
    std::string comments;
    for (size_t k = cigars.size(); k-- > 1;) { comments += cigars[k]; if (k > 1) comments += ","; }
    f.comments = comments;
    return f;
}

// ---------------------------------------------------------------------------------
// m4 / m8: the device returns D/U/L ops only; rows are re-derived by walking the chosen path.
//
// ONE walker (walk_pathwise) fills flat scratch buffers in their final order; two front ends read them: fields_pathwise
// builds the GAFStruct fields (rg_result_fields / rg_result_gaf), append_pathwise_text writes the line straight into the
// output buffer of format_batch (the streaming engine's formatter: no per-read strings, numbers through to_chars) — the
// 12 us per read of the string-building form were 50 CPU-ms per 4096-read tile, more than an 8-GPU node under a
// 16-CPU quota has to spare.
namespace {

// A position on one path: row + the bases / ids the walk emits, stepped towards the source (back) or the sink (ahead).
// With the path's row list (HostGraph::pl_*) a step is an index step over sequential memory; without it (a list that
// failed its check, or a row that is not on the path) it is the PredHash lookup of step_on_path.
struct PathCursor {
    const HostGraph& g;
    int path, row;
    long idx = -1, cnt = 0;
    const int32_t* rows = nullptr;
    const char* bases = nullptr;
    const uint64_t* ids = nullptr;
    bool fast = false;
    PathCursor(const HostGraph& g_, int path_, int row_) : g(g_), path(path_), row(row_) {
        if (!g.pl_ok || path < 0 || path >= g.P) return;
        const int32_t o = g.pl_off[(size_t)path];
        cnt = g.pl_off[(size_t)path + 1] - o;
        rows = g.pl_row.data() + o; bases = g.pl_base.data() + o; ids = g.pl_id.data() + o;
        const int32_t* it = std::lower_bound(rows, rows + cnt, row);
        if (it != rows + cnt && *it == row) { idx = it - rows; fast = true; }
    }
    char base() const { return fast ? bases[idx] : g.lnz[(size_t)row]; }
    uint64_t id() const { return fast ? ids[idx] : g.node_id[(size_t)row]; }
    void back() {
        if (!fast) { row = step_on_path(g, row, path, true); return; }
        if (idx > 0) { --idx; row = rows[idx]; } else { fast = false; row = 0; }              // the list's first row leads back to row 0
    }
    void ahead() {
        if (!fast) { row = step_on_path(g, row, path, false); return; }
        if (idx + 1 < cnt) { ++idx; row = rows[idx]; } else { fast = false; row = g.L - 1; }  // ... its last row on to 'F'
    }
};

struct PathwiseScratch {
    std::vector<char> ops, pseq;
    std::vector<uint64_t> ids;
};
struct PathwiseWalk {
    const char* ops = nullptr; size_t nops = 0;       // D / d / U / L in output order
    const char* pseq = nullptr; size_t npseq = 0;     // path bases in output order
    const uint64_t* ids = nullptr; size_t nids = 0;   // segment ids, consecutive duplicates removed
    size_t plen = 0, pstart = 0, pend = 0;
    bool rec = false;
    size_t rec_edge = 0;
};

inline char base_at(const uint8_t* codes, int n, int col) { return col == 0 ? '$' : "ACGTN"[codes[(size_t)col - 1]]; }

void walk_pathwise(const HostGraph& g, const uint8_t* codes, int n, const ReadRecord& r, int mode, PathwiseScratch& sc, PathwiseWalk& w) {
    const size_t cap = (size_t)std::max(r.n_ops, 1);
    if (sc.ops.size() < cap) { sc.ops.resize(cap); sc.pseq.resize(cap); sc.ids.resize(cap); }
    char* ops = sc.ops.data();
    char* pseq = sc.pseq.data();
    uint64_t* ids = sc.ids.data();
    w.rec = (mode == RG_MODE_RECOMBINATION || mode == RG_MODE_RECOMBINATION_SEMI) && r.best_path != r.rev_path;
    // semiglobal walkers stop where the read is consumed: the path may start inside the graph, and the coordinate
    // helpers receive `start = i + 1` for the row i the walk stopped on (recombination_output.rs:186,331)
    if (!w.rec) {
        PathCursor pc(g, r.best_path, r.end_row);
        int j = n;
        const size_t N = (size_t)r.n_ops;
        size_t plen = 0;
        // the walk runs from the end of the alignment to its start: ops, bases and ids are written back to front
        for (size_t k = 0; k < N; ++k) {
            const uint8_t op = r.ops[k] & 0x7f;
            if (op == OP_D) {
                const char b = pc.base();
                ops[N - 1 - k] = b != base_at(codes, n, j) ? 'd' : 'D';
                ++plen;
                ids[N - plen] = pc.id(); pseq[N - plen] = b;
                pc.back(); j -= 1;
            } else if (op == OP_U) {
                ops[N - 1 - k] = 'U';
                ++plen;
                ids[N - plen] = pc.id(); pseq[N - plen] = pc.base();
                pc.back();
            } else { ops[N - 1 - k] = 'L'; j -= 1; }
        }
        const int i = pc.row;
        w.ops = ops; w.nops = N;
        w.pseq = pseq + (N - plen); w.npseq = plen;
        uint64_t* ib = ids + (N - plen);
        w.ids = ib; w.nids = (size_t)(std::unique(ib, ib + plen) - ib);
        // utils.rs:221-254; start = 0 for the global modes (the walk pads down to row 0)
        const int start = i == 0 ? 0 : i + 1;
        w.pstart = head_in_segment(g, start);
        w.pend = plen > 0 ? w.pstart + plen - 1 : 0;
        w.plen = w.pend + tail_in_segment(g, r.end_row) + 1;
        return;
    }
    // recombination (recombination_output.rs:363-631): the forward half occupies [0, F) of the buffers (written back to
    // front), the reverse half follows it
    const int fp = r.best_path, rp = r.rev_path;
    const size_t F = (size_t)r.n_fwd_ops, N = (size_t)r.n_ops;
    size_t flen = 0, rlen = 0;
    int fwd_stop = 0;
    {   // forward half, walked backwards from (fen, rec_col)
        PathCursor pc(g, fp, r.fen);
        int j = r.rec_col;
        for (size_t k = 0; k < F; ++k) {
            const uint8_t op = r.ops[k] & 0x7f;
            if (op == OP_D) {
                const char b = pc.base();
                ops[F - 1 - k] = b != base_at(codes, n, j) ? 'd' : 'D';
                ++flen;
                ids[F - flen] = pc.id(); pseq[F - flen] = b;
                pc.back(); j -= 1;
            } else if (op == OP_U) {
                ops[F - 1 - k] = 'U';
                ++flen;
                ids[F - flen] = pc.id(); pseq[F - flen] = pc.base();
                pc.back();
            } else { ops[F - 1 - k] = 'L'; j -= 1; }
        }
        fwd_stop = pc.row;
    }
    int rev_ending = r.rsn;
    {   // reverse half, walked forwards from (rsn, rec_col); r_seq[j] = read[j+1] (get_rev_sequence)
        PathCursor pc(g, rp, r.rsn);
        int j = r.rec_col;
        for (size_t k = F; k < N; ++k) {
            const uint8_t raw = r.ops[k];
            const uint8_t op = raw & 0x3f;
            if (!(raw & OP_CONT)) rev_ending = pc.row;   // ops of the main loop (recombination_output.rs:415)
            if (op == OP_D) {
                const char rc = j + 1 <= n ? base_at(codes, n, j + 1) : 'F';
                const char b = pc.base();
                ops[k] = b != rc ? 'd' : 'D';
                ids[F + rlen] = pc.id(); pseq[F + rlen] = b;
                ++rlen;
                pc.ahead(); j += 1;
            } else if (op == OP_U) {
                ops[k] = 'U';
                ids[F + rlen] = pc.id(); pseq[F + rlen] = pc.base();
                ++rlen;
                pc.ahead();
            } else { ops[k] = 'L'; j += 1; }
        }
    }
    w.rec_edge = flen - 1;                      // (#bases of the forward half) - 1, usize arithmetic as in the reference
    w.ops = ops; w.nops = N;
    w.pseq = pseq + (F - flen); w.npseq = flen + rlen;
    uint64_t* ib = ids + (F - flen);
    w.ids = ib; w.nids = (size_t)(std::unique(ib, ib + flen + rlen) - ib);
    // utils.rs:256-323
    const int start = fwd_stop == 0 ? 0 : fwd_stop + 1;
    const size_t path_start = head_in_segment(g, start);
    const size_t forw_path_end = flen > 0 ? path_start + flen - 1 : 0;
    const size_t forw_path_len = forw_path_end + tail_in_segment(g, r.fen) + 1;
    const size_t rev_path_start = head_in_segment(g, r.rsn);
    const size_t rev_path_end = rlen > 0 ? rev_path_start + rlen - 1 : 0;
    w.pstart = path_start;
    w.pend = forw_path_len + rev_path_end;
    w.plen = forw_path_len + (rev_path_end + tail_in_segment(g, rev_ending) + 1);
}

// text cursor over a std::string grown once per record
struct Cursor {
    char* p;
    void ch(char c) { *p++ = c; }
    void str(const char* s) { while (*s) *p++ = *s++; }
    void mem(const char* s, size_t n) { memcpy(p, s, n); p += n; }
    void num(unsigned long long v) { p = std::to_chars(p, p + 24, v).ptr; }
    void snum(long long v) { p = std::to_chars(p, p + 24, v).ptr; }
    void f32(float v) { p = std::to_chars(p, p + 64, v, std::chars_format::fixed).ptr; }
    void cigar(const char* ops, size_t n) {                     // build_cigar: D->M, d->X, U->I, L->D, run-length
        size_t i = 0;
        while (i < n) {
            size_t j = i + 1;
            while (j < n && ops[j] == ops[i]) ++j;
            num(j - i);
            ch(ops[i] == 'D' ? 'M' : ops[i] == 'd' ? 'X' : ops[i] == 'U' ? 'I' : 'D');
            i = j;
        }
    }
};

thread_local PathwiseScratch t_scratch;

}  // namespace

GafFields fields_pathwise(const HostGraph& g, const std::string& read, const std::string& name,
                          const ReadRecord& r, int mode) {
    const int n = (int)read.size();
    std::vector<uint8_t> codes((size_t)n);
    for (int k = 0; k < n; ++k) { const char c = read[(size_t)k]; codes[(size_t)k] = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
    PathwiseScratch sc;
    PathwiseWalk w;
    walk_pathwise(g, codes.data(), n, r, mode, sc, w);
    GafFields f;
    f.name = name; f.qlen = (size_t)n; f.qstart = 0; f.qend = (size_t)(n - 1);
    f.path.assign(w.ids, w.ids + w.nids);
    f.pstart = w.pstart; f.pend = w.pend; f.plen = w.plen;
    const std::string ops(w.ops, w.nops), pseq(w.pseq, w.npseq);
    if (!w.rec) {
        f.comments = rle_cigar(ops) + ", best path: " + std::to_string(r.best_path) + ", score: " + std::to_string(r.score) +
                     "\t" + pseq;
        return f;
    }
    auto node_off = [&](int node) { return g.node_id[node] == 0 ? 0 : g.seg_off[node] - 1; };  // get_node_offset
    f.comments = rle_cigar(ops) + ", recombination path " + std::to_string(r.best_path) + " " + std::to_string(r.rev_path) + ", nodes " +
                 std::to_string(g.node_id[r.fen]) + "[" + std::to_string(node_off(r.fen)) + "] " +
                 std::to_string(g.node_id[r.rsn]) + "[" + std::to_string(node_off(r.rsn)) + "], score: " +
                 f32_display(r.fscore) + ", displacement: " + std::to_string(r.displacement) + "\t" + pseq + "\t" +
                 std::to_string(w.rec_edge);
    return f;
}

// fields_pathwise(..).text() written straight into `out` (GAFStruct::to_string, gaf_output.rs:70-94; comments as in
// pathwise_alignment_output.rs:161-167, recombination_output.rs:598-612, 759-765)
void append_pathwise_text(const HostGraph& g, const uint8_t* codes, int n, const char* name, const ReadRecord& r, int mode, std::string& out) {
    PathwiseWalk w;
    walk_pathwise(g, codes, n, r, mode, t_scratch, w);
    const size_t nlen = strlen(name);
    const size_t before = out.size();
    // upper bound of the line: every number <= 20 digits; a cigar run is at least one op and prints <= 21 characters only
    // when it is long, 2 per op at worst
    out.resize(before + nlen + 21 * (w.nids + 16) + 2 * w.nops + w.npseq + 160);
    Cursor c{&out[before]};
    c.mem(name, nlen);
    c.ch('\t'); c.num((unsigned long long)n);
    c.str("\t0\t"); c.num((unsigned long long)(n - 1));
    c.str("\t+\t>");
    for (size_t i = 0; i < w.nids; ++i) { if (i) c.ch('>'); c.num(w.ids[i]); }
    c.ch('\t'); c.num(w.plen);
    c.ch('\t'); c.num(w.pstart);
    c.ch('\t'); c.num(w.pend);
    c.str("\t0\t*\t*\t");
    c.cigar(w.ops, w.nops);
    if (!w.rec) {
        c.str(", best path: "); c.snum(r.best_path);
        c.str(", score: "); c.snum(r.score);
        c.ch('\t'); c.mem(w.pseq, w.npseq);
    } else {
        auto node_off = [&](int node) { return g.node_id[node] == 0 ? 0 : g.seg_off[node] - 1; };  // get_node_offset
        c.str(", recombination path "); c.snum(r.best_path); c.ch(' '); c.snum(r.rev_path);
        c.str(", nodes "); c.num(g.node_id[r.fen]); c.ch('['); c.snum(node_off(r.fen)); c.str("] ");
        c.num(g.node_id[r.rsn]); c.ch('['); c.snum(node_off(r.rsn)); c.str("], score: ");
        c.f32(r.fscore);
        c.str(", displacement: "); c.snum(r.displacement);
        c.ch('\t'); c.mem(w.pseq, w.npseq);
        c.ch('\t'); c.num((unsigned long long)w.rec_edge);
    }
    c.ch('\n');
    out.resize((size_t)(c.p - out.data()));
}

}  // namespace rg
