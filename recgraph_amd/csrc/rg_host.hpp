// Internal host-side structures of librecgraph_hip (not part of the C ABI).
//
// Layout choices: every graph array is a flat CSR/SoA vector sized for one upload to HBM; the
// reference's HashMap/BitVec containers (graph.rs:23-27, pathwise_graph.rs:10-18,75-78) have no
// counterpart here.
#pragma once
#include "rg_codes.hpp"

#include <atomic>
#include <cstdint>
#include <deque>
#include <functional>
#include <string>
#include <vector>

#include "../../include/recgraph_hip.h"

namespace rg {

extern thread_local std::string g_last_error;

// Process-wide diagnostic switches (rg_set_option; defaults from the environment variables of the same meaning, read
// once when the library is loaded): no getenv on the run path.
struct Options {
    std::atomic<int> sweep_i32{0};      // RG_SWEEP_I32: force the i32 sweep kernel
    std::atomic<int> three_sweeps{0};   // RG_THREE_SWEEPS: force the three-sweep -m 8 pipeline
    std::atomic<int> no_frec{0};        // RG_NO_FREC: Cand-list forward emission instead of records
    std::atomic<int> debug{0};          // RG_DEBUG: candidate / record statistics on stderr
    std::atomic<int> spin_wait{0};      // RG_SPIN_WAIT: long waits for the device through hipStreamSynchronize (spins a CPU) instead of
                                        // hipEventQuery + usleep polling
    std::atomic<int> no_gather{0};      // RG_NO_GATHER: k_sweep16 without its gather runs
    std::atomic<int> no_split{0};       // RG_NO_SPLIT: k_sweep16 on the plain step tables (no TAIL records)
    std::atomic<int> no_spec{0};        // RG_NO_SPEC: -m 8 forward sweep with the provable bound (path 0) instead of the speculative one
    std::atomic<int> spec_margin{112};  // RG_SPEC_MARGIN: what the speculative bound subtracts from the picked path's score
    std::atomic<int> lb_bonus{0};       // RG_LB_BONUS (experiments only): added to the forward sweep's lower bound; > 0 may drop candidates
    std::atomic<int> stripe_c{0};       // RG_STRIPE_C: columns per lane of the striped long-read kernels (8, 16, 32; 0: 16 up to 8191 bases, else 32)
    std::atomic<int> no_retire{0};      // RG_NO_RETIRE: k_sweep16 computes every path to the end (no path retirement)
    std::atomic<int> no_order{0};       // RG_NO_ORDER: the sweeps' waves in read order (no longest-first launch order)
    std::atomic<int> dsel_edge{8};      // RG_DSEL_EDGE: the 1 / dsel_edge of the rows each sweep visits first always store their direction words
    std::atomic<int> sweep_prio{0};     // RG_SWEEP_PRIO: the pathwise sweeps on a low-priority stream of their own (measured: -1..-3 % at config 5, +2.5 % at 1.5 kbp: off)
    std::atomic<int> no_dsel{0};        // RG_NO_DSEL: every (row, group) record of the packed sweeps stores its direction word (round 6: only those with a picked path)
    std::atomic<int> no_pick2{0};       // RG_NO_PICK2: the speculative bound from one-path picks only (no two-path picks)
    std::atomic<int> layer_i32{0};      // RG_LAYER_I32: k_layer in its i32 form even when the sweep ran packed (test hook)
    std::atomic<int> lds_pad{0};        // RG_LDS_PAD (experiments only): extra dynamic LDS bytes per k_sweep16 workgroup — lowers the waves per CU
    std::atomic<int> chunk_reads{0};    // RG_CHUNK_READS: most reads one pathwise kernel launch takes (0: what the HBM budget allows, <= 8192)
};
Options& options();
// Waits for everything enqueued on `stream` so far WITHOUT spinning: records `ev` and polls it with short sleeps.  Every HIP
// wait spins by default (hipStreamSynchronize / hipEventSynchronize of a 0.5 s kernel = 0.5 s of CPU, hipEventBlockingSync
// events included: tools/probes/wait_probe.hip) — three such threads per GPU on a node under a CPU quota starve everything
// else — and the runtime's own blocking mode (hipDeviceScheduleBlockingSync) deadlocked here when two threads sat in
// hipFree's implicit device synchronisation at once (profiles/r03_notes.md).  Returns a hipError_t as int.
int wait_stream_sleeping(void* stream, void* ev, bool spin = false);   // spin: this handle asked for hipStreamSynchronize (rg_stream_opts.spin_wait)

int fail(int code, const std::string& msg);

// Path sets: the reference uses BitVec(paths_number) (pathwise_graph.rs:10-18); here a fixed 256-bit mask (4 words),
// of which the kernels see one 64-path PAGE at a time.
constexpr int RG_MAXP = 256;
constexpr int RG_PW = RG_MAXP / 64;
struct PMask {
    uint64_t w[RG_PW] = {0, 0, 0, 0};
    bool any() const { return (w[0] | w[1] | w[2] | w[3]) != 0; }
    bool test(int k) const { return k >= 0 && k < RG_MAXP && ((w[k >> 6] >> (k & 63)) & 1); }
    void set(int k) { w[k >> 6] |= 1ull << (k & 63); }
    int lowest() const { for (int i = 0; i < RG_PW; ++i) if (w[i]) return i * 64 + __builtin_ctzll(w[i]); return -1; }
    int highest() const { for (int i = RG_PW - 1; i >= 0; --i) if (w[i]) return i * 64 + 63 - __builtin_clzll(w[i]); return -1; }
    int count() const { int c = 0; for (int i = 0; i < RG_PW; ++i) c += __builtin_popcountll(w[i]); return c; }
    PMask operator&(const PMask& o) const { PMask r; for (int i = 0; i < RG_PW; ++i) r.w[i] = w[i] & o.w[i]; return r; }
    PMask& operator|=(const PMask& o) { for (int i = 0; i < RG_PW; ++i) w[i] |= o.w[i]; return *this; }
    PMask andnot(const PMask& o) const { PMask r; for (int i = 0; i < RG_PW; ++i) r.w[i] = w[i] & ~o.w[i]; return r; }
    bool operator==(const PMask& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
    static PMask first(int P) { PMask r; for (int k = 0; k < P; ++k) r.set(k); return r; }
};

// One edge group of a DP row in the pathwise modes: all paths entering `row` through the same predecessor row share the
// direction chosen by the group's alpha path.  A group whose members span several 64-path pages is listed
// once per page: the entry of the alpha's page first (it runs the alpha), then continuation entries (ga == GA_CONT) that
// only follow the directions already chosen.
struct GroupDesc {
    static constexpr uint32_t GA_CONT = 0xffffffffu;
    int32_t pred;     // predecessor row (successor row in the reverse program)
    uint32_t ga;      // group alpha: bit index inside `page` (path id = page * 64 + ga), or GA_CONT
    uint64_t mask;    // member paths of this page
    int32_t slot;     // index of this (row, group) in the direction-word store
    int32_t page;     // 64-path page of `mask`
};

struct HostGraph {
    // ---- shared linearisation (graph.rs:41-57 == pathwise_graph.rs:147-165) ----
    int32_t L = 0;                    // rows incl. '$' (row 0) and 'F' (row L-1)
    std::string lnz;
    std::vector<uint64_t> node_id;    // segment id per row, 0 for rows 0 and L-1
    std::vector<int32_t> seg_off;     // 1-based offset of the row inside its segment (0 for row 0 / F)
    std::vector<uint64_t> node_id_rev; // `-s true`: id per row when the handle order is reversed (utils.rs:144-165, amb_mode)

    // ---- LnzGraph view (m0/m2) ----
    bool has_lnz = false;
    std::vector<int32_t> pred_off, pred_rows;   // CSR; empty range = single predecessor row-1
    std::vector<int32_t> r_values;              // utils.rs:103-126
    std::vector<int32_t> min_pred;              // numerically smallest predecessor (column-0 chain)

    // ---- PathGraph view (m4/m8) ----
    bool has_path = false;
    std::string path_error;                     // why a GFA with P lines has no PathGraph view (reported for modes 4+)
    int32_t P = 0;
    std::vector<PMask> row_mask;                // paths through each row (all ones for rows 0, L-1)
    std::vector<int32_t> alphas;
    std::vector<uint8_t> pnwp, rnwp;            // forward / reverse "row has listed predecessors"
    std::vector<int32_t> eoff, epred;           // forward PredHash, CSR, ascending pred row
    std::vector<PMask> emask;
    std::vector<int32_t> roff, rsucc;           // reverse PredHash
    std::vector<PMask> rmask;
    std::vector<int32_t> dfs, dfe;              // pathwise_graph.rs:306-354; ndm[i][j] recomputed from these
    std::vector<int32_t> knm;                   // highest path id NOT through the row, -1 if none
    // DP programs: groups of row i are fgroups[fgoff[i] .. fgoff[i+1])
    std::vector<int32_t> fgoff, rgoff;
    std::vector<GroupDesc> fgroups, rgroups;
    int32_t fslots = 0, rslots = 0;
    int32_t max_path_rows = 0;
    // rows of every path in order (+ their bases and segment ids), for the host formatter's walk along a path: rows of path k
    // are pl_row[pl_off[k] .. pl_off[k + 1]); pl_ok: every step of every list equals the PredHash step it replaces
    // (build_path_lists checks that once; the formatter falls back to the PredHash lookups otherwise)
    std::vector<int32_t> pl_off, pl_row;
    std::vector<char> pl_base;
    std::vector<uint64_t> pl_id;
    bool pl_ok = false;
};

// GFA text -> HostGraph (both views when P lines exist)
int build_from_gfa(const char* text, int64_t len, HostGraph& g);
int build_from_lnz(const char* lnz, int64_t L, const int64_t* pred_off, const int64_t* pred_rows,
                   const uint64_t* node_id, HostGraph& g);
int build_from_path(const char* lnz, int64_t L, int32_t P, const uint64_t* row_mask, const int64_t* edge_off,
                    const int64_t* edge_pred, const uint64_t* edge_mask, const uint64_t* node_id, HostGraph& g);
std::string dump_graph(const HostGraph& g, int which);

// Step tables of the pathwise sweeps (rg_steps.cpp; record layout: k_sweep in rg_pathwise.hip).  `split`: the same records
// with the TAIL groups moved behind their register runs (k_sweep16, P <= 64); `lead_*`: per evaluation point of the path
// retirement and path, the members of the groups the path leads from there on ([point][64], P <= 64); `members`: the sum
// of the group sizes = the member-row updates one sweep of the reference performs per column.
struct StepRec { int32_t x, y, z, w; };
struct StepTables {
    std::vector<StepRec> plain, split;
    std::vector<unsigned long long> lead_plain, lead_split;
    unsigned long long members = 0;
    int retire_shift = RG_SWEEP16_RETIRE_SHIFT;     // the period the lead tables were built for
};
void build_step_tables(const HostGraph& h, bool forward, bool want_split, StepTables& out);
// log2 of the records between two evaluation points of the path retirement (rg_set_option "retire_shift" / RG_RETIRE_SHIFT,
// 2..12, default RG_SWEEP16_RETIRE_SHIFT = 8): read when a handle builds its step tables, carried to the kernel with them
// (StepTables::retire_shift -> SweepArgs::retire_shift).  Small graphs only retire paths with a small period: the test
// suite runs its switch families at 4.
std::atomic<int>& retire_shift_option();

// ---- alignment records as they come back from the device ----
struct ReadRecord {
    uint32_t status = 0;
    int32_t score = 0;
    float fscore = 0.f;          // m0: best f32 score; m8 rec: recombination score
    int32_t end_row = 0;         // last_row (POA) / ending node
    int32_t end_col = 0;         // last_col (absolute column)
    int32_t stop_row = 0, stop_col = 0;   // where the POA traceback stopped
    int32_t best_path = -1, rev_path = -1;
    int32_t fen = 0, rsn = 0, rec_col = 0, displacement = 0;
    int32_t rev_ending = 0;
    int32_t n_ops = 0;           // traceback ops in walk order
    const uint8_t* ops = nullptr;
    const int32_t* rows = nullptr;   // row consumed by each op (D/U), -1 for L
    int32_t n_fwd_ops = 0;       // m8 rec: ops of the forward half (walk order), the rest is the reverse half
};

// One GAFStruct (gaf_output.rs:6-20) as the host builds it from a device record, plus the warning lines the reference
// prints on stdout before it.  text() is exactly what the reference prints for the read.
struct GafFields {
    std::string pre;             // "Band length probably too short..." / "band not enough for correct output" lines
    bool empty = false;          // GAFStruct::new() (gaf_output.rs:22-38): alignment_block_length / mapping_quality are ""
    std::string name;
    size_t qlen = 0, qstart = 0, qend = 0;
    char strand = '+';
    std::vector<uint64_t> path;
    size_t plen = 0, pstart = 0, pend = 0, residues = 0;
    std::string comments;
    std::string line() const;    // GAFStruct::to_string (gaf_output.rs:70-94)
    std::string text() const { return pre + line() + "\n"; }
};
// amb: rg_params.amb_mode (bit 0 reversed handle ids, bit 1 strand '-')
GafFields fields_m0_simd(const HostGraph& g, const std::string& read, const std::string& name, const ReadRecord& r, int amb = 0);
GafFields fields_poa_banded(const HostGraph& g, const std::string& read, const std::string& name, const ReadRecord& r, int amb = 0);
GafFields fields_pathwise(const HostGraph& g, const std::string& read, const std::string& name, const ReadRecord& r, int mode);
// fields_pathwise(..).text() appended to `out` without the intermediate strings (codes: the read's base codes 0..4)
void append_pathwise_text(const HostGraph& g, const uint8_t* codes, int n, const char* name, const ReadRecord& r, int mode, std::string& out);
void build_rev_ids(HostGraph& g);
void build_path_lists(HostGraph& g);

// ---- read ingestion (rg_reads.cpp) ----
struct FastaReads {
    std::string bases;                  // concatenated canonical bases of all reads
    std::vector<int64_t> off;           // count + 1
    std::vector<std::string> names;
};
bool parse_fasta(const char* text, int64_t len, FastaReads& r, int64_t batch, const std::function<void(int64_t, int64_t)>& emit);
// sequences::get_sequences over a text that arrives in pieces (rg_stream_feed_fasta): see rg_reads.cpp
struct FastaFeeder {
    std::string carry;                       // unterminated last line of the pieces so far
    std::string cur;                         // bases of the sequence being collected
    std::deque<std::string> names, seqs;     // closed, not yet paired by index
    int64_t names_total = 0, seqs_total = 0;
    // complete reads are APPENDED to out (bases / off / names); final: the text ends with this piece
    void feed(const char* text, int64_t len, bool final, FastaReads& out);
    bool balanced() const { return names_total == seqs_total; }    // after the final piece: false = "wrong fasta file format"
private:
    void line(const char* p, const char* q, FastaReads& out);
    void pair_up(FastaReads& out);
};
void fasta_count(const char* text, int64_t len, bool final, int64_t st[4]);
int64_t canonicalise_reads(const char* reads, const int64_t* read_off, int64_t nreads, uint8_t* codes, uint8_t* bad);
std::string f32_display(float v);

}  // namespace rg
