// Argument blocks of the pathwise kernels (rg_pathwise.hip) shared with the driver.
#pragma once
#include "rg_path_args.hpp"

namespace rg {

// per-read scalar state carried between the kernels of one batch
struct ReadState {
    uint32_t status;
    int s0;            // best no-recombination score (seed of the search) / m4 best score
    int seed_path;
    int end_row;       // sink row of the chosen path
    int end_row_best;  // semiglobal: row of the overall best last-column value
    int fwd_path, rev_path, fen, rsn, rec_col, displacement;
    float fscore;
    int bound;         // integer lower bound of the final search maximum (>= s0), tightens the pruning
    int sink_val[RG_MAXP];  // A[sink row][n][k]; semiglobal: best last-column value of path k over its rows >= 1
    int trace_score;        // value of the forward layer where the traceback starts (written by k_layer)
    int path_end_row[RG_MAXP];   // semiglobal: first row attaining sink_val[k] (ending_node, pathwise_alignment_recombination.rs:885-897)
};

// one entry of the recombination candidate lists: best member of (row, col) that can still matter
struct Cand {
    int row, col, val, path;
};

// Direction word of k_sweep16 at C <= 16 (RowOps16::dir_word, LayerArgs::dir_fmt 1) -> U and L as bit-per-column masks in the
// lane's column order (bit q = column q of the lane): register r of the packed row holds columns r and H + r and contributes
// bit 7 - r of byte 0 (U, column r), byte 1 (U, column H + r), byte 2 (L, column r), byte 3 (L, column H + r)
template <int H>
__device__ __forceinline__ void dir16_decode(unsigned w, unsigned& u16, unsigned& l16) {
    constexpr unsigned LOWH = (1u << H) - 1u;
    const unsigned t = __brev(w);
    u16 = ((t >> 24) & LOWH) | (((t >> 16) & LOWH) << H);
    l16 = ((t >> 8) & LOWH) | ((t & LOWH) << H);
}

struct SweepArgs {
    PathGraphDev g;
    DevScores sc;
    const uint8_t* reads;
    const long long* read_off;
    const uint8_t* bad;
    ReadState* state;
    int* roll;                 // [reads][P][wpad] rolling rows (packed pairs for k_sweep16: half the words)
    int rev;                   // 0 forward sweep, 1 reverse sweep
    int track_best;            // maintain the best member per (row, col) (m8)
    const int* thr;            // [reads][wpad] emission thresholds by real column, or null
    const int4* fsteps;        // step tables (one record per (row, edge group), sweep order), see k_sweep
    const int4* rsteps;
    int nfsteps, nrsteps;
    const int4* fsplit;        // the same records with TAILs (rg_path_driver.hip: split tables), or null; launch_sweep16 hands
    const int4* rsplit;        // them to the record variants of k_sweep16 when `use_split` says the batch qualifies
    int use_split;
    const int* lb;             // per read lower bound of S0: when set (forward sweep only) the threshold of column
                               // j is lb + brc - (n - j) * maxmatch (no reverse information needed)
    int brc, maxmatch;
    int semi;                  // semiglobal modes (-m 5 / -m 9): border column stays 0, free end row
    float rbw;                 // -B: columns outside the recombination band never emit
    int* colmax_out;           // [reads][wpad] per-column maximum of the best members, or null
    int* colarg_out;           // [reads][wpad] (row << 8 | path) of a cell attaining that maximum
    Cand* cand;                // [reads][cand_cap] or null
    unsigned cand_cap;         // capacity of the list this sweep appends to
    unsigned* ncand_out;
    uint32_t* dirs;            // [reads][dirs_stride] direction words or null
    long long dirs_stride;
    int dir_words;             // u32 words per (row, group) slot
    unsigned long long* cells;   // [2]: member-row cell updates of the workload | cell updates the kernel performed
    int count_cells;
    int oob;                   // 1: the thresholds are tight (speculative bound): the epilogue tests the lane maximum before the columns
    // k_sweep16, two-sweep pipeline: emissions go out as one record per (row, lane) instead of one Cand per cell: (4 + C) ints = {row << 6 | lane, column mask, 0, 0, key[C]} with key = value << 16 | path
    int* frec;                 // [reads][frec_cap][4 + C] or null
    unsigned frec_cap;
    int nwv;                   // waves (column stripes of 2048) per read: > 1 for reads longer than 2047 bases (k_sweep<32, true, true>)
    int gather_ok;             // k_sweep16 gather runs: the difference of two members' packed values provably fits 16 bits
    // k_sweep16 PATH RETIREMENT (record pipelines, P <= 64): per 256 records of the step table and per path, the union of the
    // member masks of the groups the path leads from there on ([evaluation point][64]); one table per step table.
    // Evaluation points: every 2^retire_shift records (StepTables::retire_shift: what the table builder used)
    const unsigned long long* flead;
    const unsigned long long* rlead;
    const unsigned long long* fslead;   // ... of the split tables
    const unsigned long long* rslead;
    int retire;
    int retire_shift;
    const int* order;                   // launch order of the reads (block b sweeps read order[b]) or null: see launch_order
    unsigned long long table_members;   // member rows of the step table in use (k_sweep16 with path retirement counts cells from it)
    unsigned long long fmembers, rmembers;
    // k_sweep16, DIRECTION WORDS ON DEMAND: the picks of k_pick (pick[rd]; pick2[2 rd] = second path or -1) — only the (row, group)
    // records with one of those paths among their members store a direction word; k_verify sends a read whose final paths are
    // not among them to the second pass.  Null: every record stores its word.
    const int* dsel_pick;
    const int* dsel_pick2;
    // ... and every record of the rows the sweep visits first (forward: rows < dsel_lo, reverse: rows > dsel_hi) stores its word
    // whatever its members: the reference's search likes to switch to another path for the last few columns of a read (ties
    // among the paths of a shared end segment go to the highest path id), and such a path's layer is only rebuilt over those rows
    int dsel_lo, dsel_hi;
};

// expands the (row, lane) records of the forward sweep into Cand entries, keeping only cells that can still reach
// the final bound with the best reverse partner of their column
struct ExpandArgs {
    ReadState* state;
    const int* frec;           // records of this sweep
    unsigned frec_cap;
    const unsigned* nrec;
    Cand* fcand;               // out: Cand list of this sweep
    unsigned fcap;
    unsigned* nf;
    const int* wr;             // [reads][wpad] column maxima of the OTHER sweep (best possible partner per column)
    int wpad;
    int brc;
    const int* knm;            // per row: highest path id NOT through the row (-1 if none)
    int rev;                   // records of the reverse sweep: lane columns are mirrored (real column = n - c)
    const long long* read_off;
    float rbw;                 // -B
    int gcost;                 // uniform read-gap cost: the records hold z-space keys (z = A - c * gcost, rg_sweep16.hip)
};

struct SeedArgs {
    PathGraphDev g;
    ReadState* state;
    int nreads;
    int mode;
    DevScores sc;
    const uint8_t* reads;
    const long long* read_off;
};

struct ThrArgs {
    const ReadState* state;
    const int* colmax;
    int* thr;
    int wpad;
    int brc;
    int use_bound;             // 0: S0, 1: the tighter bound of k_bound
    const int* lb;             // the forward sweep's lower bound of the search maximum (or null): the base is at least that — with
                               // a two-path pick it lies above the seed; k_verify checks the speculative ones afterwards
};

struct BoundArgs {
    PathGraphDev g;
    ReadState* state;
    const long long* read_off;
    const int* mf; const int* mfarg;
    const int* wr; const int* wrarg;
    int wpad;
    int brc;
    float mrc;
    float rbw;
};

struct Opt0Args {
    PathGraphDev g;
    DevScores sc;
    const uint8_t* reads;
    const long long* read_off;
    const uint8_t* bad;
    const int* fpoff; const int* fprow;
    int* lb;                   // out: exact global alignment score of the read against path 0 (pick == null) ...
    int semi;
    const int* pick;           // ... or against path pick[rd], minus `margin`: the SPECULATIVE bound (see k_pick)
    int margin;
    int nwv;                   // column stripes per read (> 1: k_opt0_striped)
    const int* pick2;          // two-path picks {p2 or -1, X} (PickArgs), or null
    int rec_pen;               // what a two-path bound pays for its switch: base recombination cost + displacement allowance
};

// Speculative lower bound of the -m 8 search maximum.  The forward sweep of the two-sweep pipeline emits every cell that
// could still pair up to `lb`; the only PROVABLE lb before the sweep is the alignment of the read against path 0 (the one
// path that is the alpha of all its groups), which lies ~1500 below the seed at config 5 and lets 40 000 records per read
// through.  Instead: k_pick votes for the path that shares most sampled 12-mers with the read, k_opt0 aligns the read
// against THAT path, lb' = that score - margin, and k_verify checks the speculation afterwards: the pruning with lb' is
// exact iff the search maximum found is >= lb' (every pair that beats or ties it was then kept); the reads that fail
// (rs->status & ST_RETRY) are aligned again with the provable bound (rg_path_driver.hip).
struct PickArgs {
    const uint8_t* reads;
    const long long* read_off;
    const uint8_t* bad;
    const uint32_t* keys;      // open-addressing table of the 12-mers of every path: key (24 bits) or 0xffffffff
    const unsigned long long* masks;   // paths that contain the 12-mer (P <= 64)
    unsigned table_mask;       // table size - 1 (power of two)
    int P;
    int* pick;                 // out: path with the most votes (lowest id on ties; 0 when nothing matched)
    // TWO-PATH PICK (reads that switch haplotype once: half of config 5).  The votes are per sample position, so the wave
    // also finds the split t and the paths (p1, p2) that maximise votes_{< t}(p1) + votes_{>= t}(p2); when that beats the
    // one-path vote clearly, pick2[rd] = {p2, X}: X = a row both paths visit near the split.  k_opt0 then aligns the read
    // against p1's rows <= X followed by p2's rows > X — the alignment a recombination p1 -> p2 behind the shared segment
    // realises — and the bound is that score minus the recombination cost.  Speculative like the one-path pick.
    const int* fpoff; const int* fprow;
    int* pick2;                // out (null: one-path picks only): {p2 or -1, X}
};
constexpr uint32_t ST_RETRY = 0x200u;   // internal: the speculative bound of this read did not hold

struct SearchArgs {
    PathGraphDev g;
    ReadState* state;
    const Cand* fcand;
    const Cand* rcand;
    const unsigned* nf;
    const unsigned* nr;
    unsigned* ridx;
    unsigned fcap, rcap;
    const int* wr;             // [reads][wpad] reverse column maxima: forward candidates are re-filtered with the
                               // final bound before pairing (they may have been emitted with the loose lb threshold)
    int wpad;
    int brc;
    float mrc;
};

struct LayerArgs {
    PathGraphDev g;
    DevScores sc;
    const uint8_t* reads;
    const long long* read_off;
    ReadState* state;
    int rev;
    const uint32_t* dirs;
    long long dirs_stride;
    int dir_words;
    int dir_fmt;               // 0: 2 bits per column (k_sweep), 1: U mask | L mask (k_sweep16)
    int semi;
    int* layer;                // [reads][layer_stride]: traceback decisions, (rows of the path + 1) x dir_words words
    long long layer_stride;
    const int* fpoff; const int* fprow; const int* fpslot;   // rows of every path, forward order
    const int* rpoff; const int* rprow; const int* rpslot;   // rows of every path, reverse order
    int nwv;                   // column stripes per read (see SweepArgs)
};

struct TraceArgs {
    PathGraphDev g;
    DevScores sc;
    const uint8_t* reads;
    const long long* read_off;
    ReadState* state;
    DevRecord* rec;
    uint8_t* ops;
    long long ops_stride;
    const int* flayer;
    const int* rlayer;
    long long layer_stride;
    const int* fpoff; const int* fprow;
    const int* rpoff; const int* rprow;
    int nreads;
    int mode;
    int semi;
    int nwv;                   // column stripes per read (see SweepArgs)
};

void launch_sweep(const SweepArgs& a, int nreads, int C, hipStream_t s);
void launch_sweep16(const SweepArgs& a, int nreads, int C, hipStream_t s);   // packed 16-bit rows (rg_sweep16.hip)
bool sweep16_admissible(const DevScores& sc, int max_path_rows, int max_n, int C);
void launch_expand(const ExpandArgs& a, int nreads, int C, hipStream_t s);
void launch_colmax_rec(const ExpandArgs& a, int* colmax_out, int* colarg_out, int nreads, int C, hipStream_t s);
void launch_seed(const SeedArgs& a, hipStream_t s);
void launch_opt0(const Opt0Args& a, int nreads, int C, hipStream_t s);
void launch_opt0_16(const Opt0Args& a, int nreads, int C, hipStream_t s);   // packed rows (rg_sweep16.hip): batches the packed sweep admits
void launch_pick(const PickArgs& a, int nreads, hipStream_t s);
// Launch order of the sweeps' waves under path retirement: the work of a read grows with the highest id among its picked
// paths (every lower path leads it somewhere), from a few percent of a full sweep to all of it — longest first,
// so that the launch does not end on a few full-length waves (one block: counting sort by that id, descending).
void launch_order(const int* pick, const int* pick2, int* order, int nreads, hipStream_t s);
void launch_verify4(ReadState* st, const int* lb, unsigned* nretry, uint8_t* flags, int nreads, const int* dsel_pick, hipStream_t s);
void launch_verify(ReadState* st, const int* lb, unsigned* nretry, uint8_t* flags, int nreads, const int* dsel_pick, const int* dsel_pick2, int dsel_lo, int dsel_hi, hipStream_t s);
void launch_gather_reads(const uint8_t* reads, const long long* off, const int* idx, const long long* sub_off, uint8_t* out, int n, hipStream_t s);
void launch_scatter_results(const int* idx, const DevRecord* sub_rec, const uint8_t* sub_ops, DevRecord* rec, uint8_t* ops, long long ops_stride, int n, hipStream_t s);
void launch_threshold(const ThrArgs& a, int nreads, hipStream_t s);
void launch_bound(const BoundArgs& a, int nreads, hipStream_t s);
void launch_search(const SearchArgs& a, int nreads, hipStream_t s);
void launch_need(const ReadState* st, const unsigned* nf, const unsigned* nr, const unsigned* nrec, const unsigned* nrrec,
                 unsigned* need, int nreads, hipStream_t s);
void launch_layer(const LayerArgs& a, int nreads, int C, hipStream_t s);
void launch_layer16(const LayerArgs& a, int nreads, int C, hipStream_t s);   // packed rows (dir_fmt 1, one wave per read)
void launch_trace(const TraceArgs& a, int C, hipStream_t s);

}  // namespace rg
