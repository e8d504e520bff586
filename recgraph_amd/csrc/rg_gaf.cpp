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

#include "rg_codes.hpp"
#include "rg_host.hpp"

namespace rg {

std::string f32_display(float v) {  // Rust `{}` for f32: shortest round-trip, fixed notation
    char b[128];
    auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::fixed);
    return std::string(b, r.ptr);
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
GafFields fields_pathwise(const HostGraph& g, const std::string& read, const std::string& name,
                          const ReadRecord& r, int mode) {
    const int n = (int)read.size();
    GafFields f;
    f.name = name; f.qlen = (size_t)n; f.qstart = 0; f.qend = (size_t)(n - 1);
    const bool rec = (mode == RG_MODE_RECOMBINATION || mode == RG_MODE_RECOMBINATION_SEMI) && r.best_path != r.rev_path;
    // semiglobal walkers stop where the read is consumed: the path may start inside the graph, and the coordinate
    // helpers receive `start = i + 1` for the row i the walk stopped on (recombination_output.rs:186,331)
    if (!rec) {
        const int bp = r.best_path;
        int i = r.end_row, j = n;
        std::string ops, pseq;
        std::vector<uint64_t> ids;
        size_t plen = 0;
        for (int k = 0; k < r.n_ops; ++k) {
            uint8_t op = r.ops[k] & 0x7f;
            if (op == OP_D) {
                ops.push_back(g.lnz[i] != read_at(read, j) ? 'd' : 'D');
                ids.push_back(g.node_id[i]); pseq.push_back(g.lnz[i]);
                i = step_on_path(g, i, bp, true); j -= 1; ++plen;
            } else if (op == OP_U) {
                ops.push_back('U'); ids.push_back(g.node_id[i]); pseq.push_back(g.lnz[i]);
                i = step_on_path(g, i, bp, true); ++plen;
            } else { ops.push_back('L'); j -= 1; }
        }
        std::reverse(ops.begin(), ops.end());
        std::reverse(pseq.begin(), pseq.end());
        dedup(ids);
        std::reverse(ids.begin(), ids.end());
        f.path = ids;
        // utils.rs:221-254; start = 0 for the global modes (the walk pads down to row 0)
        const int start = i == 0 ? 0 : i + 1;
        f.pstart = head_in_segment(g, start);
        f.pend = plen > 0 ? f.pstart + plen - 1 : 0;
        f.plen = f.pend + tail_in_segment(g, r.end_row) + 1;
        f.comments = rle_cigar(ops) + ", best path: " + std::to_string(bp) + ", score: " + std::to_string(r.score) +
                     "\t" + pseq;
        return f;
    }
    // recombination (recombination_output.rs:363-631)
    const int fp = r.best_path, rp = r.rev_path;
    std::string fops, fseq, rops, rseq;
    std::vector<uint64_t> fids, rids;
    size_t flen = 0, rlen = 0;
    int fwd_stop = 0;
    {   // forward half, walked backwards from (fen, rec_col)
        int i = r.fen, j = r.rec_col;
        for (int k = 0; k < r.n_fwd_ops; ++k) {
            uint8_t op = r.ops[k] & 0x7f;
            if (op == OP_D) {
                fops.push_back(g.lnz[i] != read_at(read, j) ? 'd' : 'D');
                fids.push_back(g.node_id[i]); fseq.push_back(g.lnz[i]);
                i = step_on_path(g, i, fp, true); j -= 1; ++flen;
            } else if (op == OP_U) {
                fops.push_back('U'); fids.push_back(g.node_id[i]); fseq.push_back(g.lnz[i]);
                i = step_on_path(g, i, fp, true); ++flen;
            } else { fops.push_back('L'); j -= 1; }
        }
        fwd_stop = i;
    }
    int rev_ending = r.rsn;
    {   // reverse half, walked forwards from (rsn, rec_col); r_seq[j] = read[j+1] (get_rev_sequence)
        int i = r.rsn, j = r.rec_col;
        for (int k = r.n_fwd_ops; k < r.n_ops; ++k) {
            const uint8_t raw = r.ops[k];
            const uint8_t op = raw & 0x3f;
            if (!(raw & OP_CONT)) rev_ending = i;   // ops of the main loop (recombination_output.rs:415)
            if (op == OP_D) {
                char rc = j + 1 <= n ? read_at(read, j + 1) : 'F';
                rops.push_back(g.lnz[i] != rc ? 'd' : 'D');
                rids.push_back(g.node_id[i]); rseq.push_back(g.lnz[i]);
                i = step_on_path(g, i, rp, false); j += 1; ++rlen;
            } else if (op == OP_U) {
                rops.push_back('U'); rids.push_back(g.node_id[i]); rseq.push_back(g.lnz[i]);
                i = step_on_path(g, i, rp, false); ++rlen;
            } else { rops.push_back('L'); j += 1; }
        }
    }
    const size_t rec_edge = fseq.size() - 1;
    std::reverse(fops.begin(), fops.end());
    std::reverse(fseq.begin(), fseq.end());
    std::reverse(fids.begin(), fids.end());
    std::string ops = fops + rops, pseq = fseq + rseq;
    fids.insert(fids.end(), rids.begin(), rids.end());
    dedup(fids);
    f.path = fids;
    // utils.rs:256-323
    {
        const int start = fwd_stop == 0 ? 0 : fwd_stop + 1;
        const size_t path_start = head_in_segment(g, start);
        size_t forw_path_end = flen > 0 ? path_start + flen - 1 : 0;
        size_t forw_path_len = forw_path_end + tail_in_segment(g, r.fen) + 1;
        size_t rev_path_start = head_in_segment(g, r.rsn);
        size_t rev_path_end = rlen > 0 ? rev_path_start + rlen - 1 : 0;
        f.pstart = path_start;
        f.pend = forw_path_len + rev_path_end;
        f.plen = forw_path_len + (rev_path_end + tail_in_segment(g, rev_ending) + 1);
    }
    auto node_off = [&](int node) { return g.node_id[node] == 0 ? 0 : g.seg_off[node] - 1; };  // get_node_offset
    f.comments = rle_cigar(ops) + ", recombination path " + std::to_string(fp) + " " + std::to_string(rp) + ", nodes " +
                 std::to_string(g.node_id[r.fen]) + "[" + std::to_string(node_off(r.fen)) + "] " +
                 std::to_string(g.node_id[r.rsn]) + "[" + std::to_string(node_off(r.rsn)) + "], score: " +
                 f32_display(r.fscore) + ", displacement: " + std::to_string(r.displacement) + "\t" + pseq + "\t" +
                 std::to_string(rec_edge);
    return f;
}

}  // namespace rg
