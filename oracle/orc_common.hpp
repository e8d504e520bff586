// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of AlgoLab/RecGraph's sequence-to-graph DP hot path, written from
// a reading of the reference's Rust sources (cited per function as file:line, relative
// to /root/reference).  Nothing under recgraph_amd/ may include, link or call this:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
// only as the checker / the timed CPU baseline.
//
// PARITY PIN STATUS: the reference is Rust and no Rust toolchain exists in the build
// image, so the reference itself cannot be run.  This restatement is pinned against
// the reference's own unit-test known answers (scalar POA scores, LnzGraph/PathGraph
// construction, score matrices — tests/golden/reference_unit_vectors.json).  The
// AVX2 m0 path, m4, m8 and all GAF text are NOT covered by any reference test:
// for those, parity is UNPINNED by the reference and rests on (i) this literal
// transliteration, (ii) its agreement with a structurally different second
// restatement (absolute-score formulation, orc_pathwise_abs.cpp) and (iii) hand-derived
// vectors in tests/golden.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------------
// scoring: HashMap<(char,char),i32> of the reference (src/score_matrix.rs:35-51)
// restated as a 6x6 table over "ACGTN-".  A missing key makes the reference panic on
// `.get(..).unwrap()`; MISSING marks such entries and sets the would-panic flag on use.
// ---------------------------------------------------------------------------------
constexpr int MISSING = INT32_MIN / 4;

inline int base_idx(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case 'N': return 4;
        case '-': return 5;
        default: return -1;
    }
}

struct Scores {
    int t[6][6];
    mutable bool panicked = false;
    int get(char a, char b) const {
        int ia = base_idx(a), ib = base_idx(b);
        if (ia < 0 || ib < 0 || t[ia][ib] == MISSING) {
            panicked = true;
            return 0;
        }
        return t[ia][ib];
    }
};

// src/score_matrix.rs:35-51 create_score_matrix_match_mis
Scores make_scores_match_mis(int m, int x);
// src/score_matrix.rs:52-66 create_score_matrix_match_mis_f32 (gap = x, not 2x)
Scores make_scores_match_mis_f32(int m, int x);
// src/score_matrix.rs:67-105 (.mtx text given as string; gaps -200)
Scores make_scores_from_mtx(const std::string& text);

// ---------------------------------------------------------------------------------
// GFA (what gfa 0.8.0 / handlegraph 0.5.0 give the reference)
// ---------------------------------------------------------------------------------
struct Gfa {
    std::vector<uint64_t> seg_ids;                       // file order
    std::unordered_map<uint64_t, std::string> seg_seq;   // id -> bases
    std::vector<std::pair<uint64_t, uint64_t>> links;    // (from,to) file order, '+' only
    std::vector<std::vector<uint64_t>> paths;            // P-line order = path id
    std::vector<std::string> path_names;
};
bool parse_gfa_text(const std::string& text, Gfa& out, std::string& err);

// src/graph.rs:23-27
struct LnzGraph {
    std::string lnz;                                 // '$' ... 'F'
    std::vector<uint8_t> nwp;                        // size L
    std::map<size_t, std::vector<size_t>> pred_hash;  // row -> preds (stored order)
    std::vector<std::string> hofp;                   // utils.rs:144-165, rows 0..L-2
    std::vector<std::string> hofp_rev;               // same with amb_mode = true (graph.rs:128-142: handles reversed)
    char strand = '+';                               // `if amb_mode { '-' } else { '+' }` of the GAF walkers
};
LnzGraph create_graph_struct(const Gfa& g);  // src/graph.rs:31-123 (amb_mode=false)

// src/pathwise_graph.rs:10-18, 75-125
struct PathGraph {
    std::string lnz;
    std::vector<uint8_t> nwp;
    // PredHash: row -> (pred_row -> path bitset); iteration order here is ascending
    // pred row (the reference iterates a HashMap: order unspecified)
    std::map<size_t, std::map<size_t, std::vector<uint8_t>>> pred_hash;
    std::vector<std::vector<uint8_t>> paths_nodes;  // L x P
    std::vector<size_t> alphas;
    size_t paths_number = 0;
    std::vector<uint64_t> nodes_id_pos;
};
PathGraph create_path_graph(const Gfa& g);                    // pathwise_graph.rs:135-248
PathGraph create_reverse_path_graph(const PathGraph& fwd);   // pathwise_graph.rs:250-282
std::vector<int64_t> get_distance_from_start(const PathGraph& rev);  // :306-329
std::vector<int64_t> get_distance_from_end(const PathGraph& fwd);    // :330-354

// utils.rs:103-126
std::vector<size_t> set_r_values(const std::vector<uint8_t>& nwp,
                                 const std::map<size_t, std::vector<size_t>>& pred_hash,
                                 size_t lnz_len);
// utils.rs:17-98
std::pair<size_t, size_t> set_ampl_for_row(size_t i, const std::vector<size_t>& p_arr,
                                           size_t r_val,
                                           const std::vector<size_t>& best_scoring_pos,
                                           size_t seq_len, size_t bta, bool simd_version);

// Rust `{}` for f32: shortest round-trip, never scientific, no trailing ".0"
std::string f32_display(float v);

// gaf_output.rs:6-94
struct GAF {
    std::string query_name;
    size_t query_length = 0, query_start = 0, query_end = 0;
    char strand = ' ';
    std::vector<uint64_t> path{0};
    size_t path_length = 0, path_start = 0, path_end = 0, residue_matches_number = 0;
    std::string alignment_block_length, mapping_quality, comments;
    std::string to_string() const;
};

// pathwise_alignment_output.rs:471-556
std::string build_cigar(const std::vector<char>& cigar);

// Timing probe of the FAITHFUL (unpruned) best_alignment scan of m8_literal, for bench.py's cpu_baseline leg: the scan
// is O(L^2 n) (~1e11 iterations per read at config 5), so the bench visits every `col_stride`-th column of the band,
// times that part on its own and extrapolates (said so in the bench line).  col_stride == 1 is the real thing; a
// result produced with col_stride > 1 is NOT an alignment and is never compared with anything.
struct FaithfulProbe {
    int col_stride = 1;
    double dp_secs = 0, scan_secs = 0;      // DP fill + absolute_scores / the (j, i, ri) scan incl. the per-column fp/rp
    long long cols_visited = 0, cols_total = 0;
};
extern thread_local FaithfulProbe* g_faithful_probe;

struct Result {
    int score = 0;            // what the reference's exec returns as .0 (POA modes)
    bool would_panic = false; // reference would abort (index OOB / unwrap on None / ...)
    std::string out;          // exactly what the reference prints on stdout for this read
};

// m0: src/global_abpoa.rs:10-257 + gaf_output.rs:753-865
Result m0_simd(const std::string& read /* with '$' */, const std::string& name, size_t idx,
               const LnzGraph& g, const Scores& sc, size_t bta,
               const std::vector<size_t>& r_values, uint64_t* cells = nullptr);
// m0 scalar: src/global_abpoa.rs:260-566 + gaf_output.rs:254-381
Result m0_scalar(const std::string& read, const std::string& name, size_t idx,
                 const LnzGraph& g, const Scores& sc, size_t bta, uint64_t* cells = nullptr);
// m2: src/gap_global_abpoa.rs:11-455 + gaf_output.rs:96-253
Result m2_gap(const std::string& read, const std::string& name, size_t idx, const LnzGraph& g,
              const Scores& sc, int o, int e, size_t bta, uint64_t* cells = nullptr);

// local POA, orc_local.cpp: -m 1 AVX2 path src/local_poa.rs:9-174, -m 1 scalar path :176-262,
// -m 3 src/gap_local_poa.rs:6-183; GAF walkers gaf_output.rs:383-752
Result m1_simd(const std::string& read, const std::string& name, size_t idx, const LnzGraph& g, const Scores& sc,
               uint64_t* cells = nullptr);
Result m1_scalar(const std::string& read, const std::string& name, size_t idx, const LnzGraph& g, const Scores& sc,
                 uint64_t* cells = nullptr);
Result m3_gap_local(const std::string& read, const std::string& name, size_t idx, const LnzGraph& g, const Scores& sc,
                    int o, int e, uint64_t* cells = nullptr);

// m4: src/pathwise_alignment.rs:5-340 + pathwise_alignment_output.rs:7-184 (literal,
// delta-encoded dpm)
Result m4_literal(const std::string& read, const std::string& name, const PathGraph& g,
                  const Scores& sc);
// m8: src/pathwise_alignment_recombination.rs (all) + recombination_output.rs:363-782
// (literal; `pruned` only replaces the O(L^2 n) scan of best_alignment by an exact
// equivalent)
Result m8_literal(const std::string& read, const std::string& name, const PathGraph& g,
                  const PathGraph& rev, const std::vector<int64_t>& dfs,
                  const std::vector<int64_t>& dfe, const Scores& sc, int brc, float mrc,
                  float rbw, bool pruned);

// semiglobal modes: -m 5 src/pathwise_alignment_semiglobal.rs, -m 9 the aln_mode 9 branches
Result m5_literal(const std::string& read, const std::string& name, const PathGraph& g, const Scores& sc);
Result m9_literal(const std::string& read, const std::string& name, const PathGraph& g, const PathGraph& rev,
                  const std::vector<int64_t>& dfs, const std::vector<int64_t>& dfe, const Scores& sc, int brc, float mrc,
                  float rbw, bool pruned);
Result m5_abs(const std::string& read, const std::string& name, const PathGraph& g, const Scores& sc);
Result m9_abs(const std::string& read, const std::string& name, const PathGraph& g, const PathGraph& rev,
              const std::vector<int64_t>& dfs, const std::vector<int64_t>& dfe, const Scores& sc, int brc, float mrc, float rbw);

// second restatement (absolute-score formulation), orc_pathwise_abs.cpp
Result m4_abs(const std::string& read, const std::string& name, const PathGraph& g,
              const Scores& sc);
Result m8_abs(const std::string& read, const std::string& name, const PathGraph& g,
              const PathGraph& rev, const std::vector<int64_t>& dfs,
              const std::vector<int64_t>& dfe, const Scores& sc, int brc, float mrc, float rbw);

}  // namespace orc
