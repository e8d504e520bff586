/*
 * recgraph_hip.h — C ABI of the MI355X-native RecGraph DP hot path.
 *
 * Drop-in boundary for the reference's per-read alignment calls.  Every entry point names the
 * reference interface it replaces (file:line relative to the RecGraph source tree).  Plain
 * pointers and sizes only; no C++/torch types cross this boundary.  All functions return
 * RG_OK (0) or a negative rg_status; none of them aborts (the reference panics instead).
 *
 * Thread-safety: a graph handle is immutable after creation and may be shared by threads and by
 * devices (its device tables are uploaded once per HIP device, under a lock, by the first batch
 * created on that device); a batch handle is bound to the device that was current when it was
 * created (every call on it selects that device) and must be used by one thread at a time.
 */
#ifndef RECGRAPH_HIP_H
#define RECGRAPH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rg_status {
    RG_OK = 0,
    RG_ERR_ARG = -1,         /* bad argument (null pointer, unsupported mode, empty read ...) */
    RG_ERR_GFA = -2,         /* GFA text not usable (non-numeric names, '-' orientations ...)  */
    RG_ERR_NO_DEVICE = -3,   /* no HIP device / HIP runtime error: the product never falls back to CPU */
    RG_ERR_GRAPH = -4,       /* graph violates a precondition of the reference (not topological, >256 paths ...) */
    RG_ERR_CAPACITY = -5,    /* internal work buffer exhausted even after regrowth */
    RG_ERR_HIP = -6
} rg_status;

/* per-read status bits (rg_result_status) */
#define RG_READ_OK 0u
#define RG_READ_BAND_WARNING 1u     /* reference prints "Band length probably too short, ..." (global_abpoa.rs:406-409, gap_global_abpoa.rs:225-227) */
#define RG_READ_BAND_NOT_ENOUGH 2u  /* reference prints "band not enough for correct output" and an empty GAF (gaf_output.rs:861-864) */
#define RG_READ_WOULD_PANIC 4u      /* reference would panic on this read (index out of range, set_path_cell('u') ...) */
#define RG_READ_BAD_BASE 8u         /* read contains a character outside ACGTN (reference: HashMap unwrap panic) */

/* Alignment modes: the `-m` values of the reference CLI (args_parser.rs:31-38) on the hot path. */
#define RG_MODE_GLOBAL_POA 0        /* global_abpoa::exec_simd          src/global_abpoa.rs:10      */
#define RG_MODE_GLOBAL_POA_SCALAR 10/* global_abpoa::exec (no-AVX2 path) src/global_abpoa.rs:260     */
#define RG_MODE_GAP_POA 2           /* gap_global_abpoa::exec           src/gap_global_abpoa.rs:11  */
#define RG_MODE_PATHWISE 4          /* pathwise_alignment::exec         src/pathwise_alignment.rs:5 */
#define RG_MODE_RECOMBINATION 8     /* pathwise_alignment_recombination::exec (aln_mode 8) src/pathwise_alignment_recombination.rs:23 */
#define RG_MODE_PATHWISE_SEMI 5     /* pathwise_alignment_semiglobal::exec  src/pathwise_alignment_semiglobal.rs:6 */
#define RG_MODE_RECOMBINATION_SEMI 9/* pathwise_alignment_recombination::exec (aln_mode 9) */
#define RG_MODE_LOCAL_POA 1         /* local_poa::exec_simd             src/local_poa.rs:9          */
#define RG_MODE_LOCAL_POA_SCALAR 11 /* local_poa::exec (no-AVX2 path)   src/local_poa.rs:176        */
#define RG_MODE_GAP_LOCAL_POA 3     /* gap_local_poa::exec              src/gap_local_poa.rs:6      */

/*
 * Scoring and banding parameters.  Replaces the HashMap<(char,char),i32|f32> score matrix arguments
 * (score_matrix.rs:35-51; api.rs:131,153) and the scalar arguments of the exec functions.
 * scores[a*6+b] is the entry for the key (ALPHA[a], ALPHA[b]) with ALPHA = "ACGTN-"; RG_SCORE_MISSING
 * marks a key the map does not hold.
 */
#define RG_SCORE_MISSING (-536870912)
typedef struct rg_params {
    int32_t mode;          /* RG_MODE_*                                                       */
    int32_t scores[36];
    int32_t gap_open;      /* o  (<= 0), gap_global_abpoa.rs:16                                */
    int32_t gap_ext;       /* e  (<= 0), gap_global_abpoa.rs:17                                */
    float band_b;          /* -b, main.rs:57: bta = (b + f * (n+1)) as usize                   */
    float band_f;          /* -f                                                              */
    int64_t bta_override;  /* >= 0: use this bases_to_add for every read (api.rs:22 callers)   */
    int32_t base_rec_cost; /* -R, pathwise_alignment_recombination.rs:29                       */
    float multi_rec_cost;  /* -r                                                              */
    float rec_band_width;  /* -B                                                              */
    int32_t amb_mode;      /* POA modes, `-s true` retry (main.rs:82-106): bit 0 = node ids of the reversed handle
                              order (utils.rs:144-165 with amb_mode), bit 1 = strand '-' (gaf_output.rs:225)       */
} rg_params;

/* Fill *p with the CLI defaults (args_parser.rs:3-147: M=2 X=4 O=4 E=2 R=4 r=0.1 B=1 b=1 f=0.01). */
void rg_params_default(rg_params* p, int32_t mode);
/* score_matrix::create_score_matrix_match_mis (score_matrix.rs:35-51); f32_variant=1 gives
 * create_score_matrix_match_mis_f32 (:52-66, gap = mismatch). */
void rg_scores_match_mis(int32_t match, int32_t mismatch, int32_t f32_variant, int32_t* scores36);

typedef struct rg_graph rg_graph;
typedef struct rg_batch rg_batch;

/*
 * Graph ingestion.  Replaces graph::read_graph / create_graph_struct (graph.rs:11-102),
 * pathwise_graph::read_graph_w_path / create_path_graph / create_reverse_path_graph /
 * nodes_displacement_matrix (pathwise_graph.rs:127-305), utils::set_r_values (utils.rs:103-126) and
 * utils::create_handle_pos_in_lnz (utils.rs:144-165).  The flattened graph is uploaded to HBM once.
 */
int32_t rg_graph_from_gfa(const char* gfa_text, int64_t len, rg_graph** out);
/* Same, from an already flattened LnzGraph (graph.rs:23-27): pred rows of row i are
 * pred_rows[pred_off[i] .. pred_off[i+1]) in pred_hash order, nwp[i] = pred_off[i+1] > pred_off[i],
 * node_id[i] = segment id of row i (0 for row 0 and the final 'F' row). */
int32_t rg_graph_create_lnz(const char* lnz, int64_t L, const int64_t* pred_off, const int64_t* pred_rows,
                            const uint64_t* node_id, rg_graph** out);
/* Same, from an already flattened PathGraph (pathwise_graph.rs:10-18).  Path sets are W = (P + 63) / 64 words each: bit
 * (k & 63) of word (k >> 6) of row_mask[i * W ..] = paths_nodes[i][k]; edges of row i are (edge_pred[e],
 * edge_mask[e * W ..]) for e in [edge_off[i], edge_off[i+1]) (PredHash, pathwise_graph.rs:75-125; listed for
 * segment-start rows and the 'F' row).  P <= 256. */
int32_t rg_graph_create_path(const char* lnz, int64_t L, int32_t P, const uint64_t* row_mask,
                             const int64_t* edge_off, const int64_t* edge_pred, const uint64_t* edge_mask,
                             const uint64_t* node_id, rg_graph** out);
void rg_graph_destroy(rg_graph* g);
/* A GFA whose P lines cannot be turned into a PathGraph (more than 256 paths, '-' path steps, a step on an unknown
 * segment, steps against the id order, a segment on no path: pathwise_graph.rs:182) still yields the LnzGraph view, which
 * is all modes 0-3 need (main.rs:29); the reason is returned here ("" when the PathGraph view exists or the GFA has no
 * P lines) and by rg_batch_create (RG_ERR_GRAPH) when a pathwise mode is requested on such a graph.
 * Limits of the pathwise kernels: at most 256 paths; reads of at most 16383 bases (reads longer than 2047 bases run
 * as column stripes of 2048 in one workgroup and need a uniform read-gap cost: every matrix the reference's CLI builds). */
const char* rg_graph_path_error(const rg_graph* g);
int64_t rg_graph_rows(const rg_graph* g);   /* lnz.len() of the LnzGraph (or PathGraph if only that exists) */
int32_t rg_graph_paths(const rg_graph* g);  /* paths_number, 0 without P lines */
/* text dumps of the flattened arrays (same `which` codes as the test oracle) for construction tests */
int64_t rg_graph_dump(const rg_graph* g, int32_t which, char* buf, int64_t cap);

/*
 * Batch alignment.  One call replaces the reference's per-read loop (main.rs:56-105, 174-213,
 * 257-261, 297-312): reads are bases without the leading '$' (sequences.rs:5-45 is applied inside:
 * upper-casing and '-' -> 'N'); read i is reads[read_off[i] .. read_off[i+1]).
 *
 *   rg_batch_create   uploads the reads and sizes every work buffer in HBM
 *   rg_batch_run      runs the DP + traceback kernels on the device (inputs resident; this is the
 *                     region bench.py times); may be called repeatedly
 *   rg_batch_fetch    copies the alignment records back (one D2H of packed records)
 *   rg_result_*       per-read accessors; rg_result_gaf formats exactly what the reference prints on
 *                     stdout for that read (warning lines + GAFStruct::to_string, gaf_output.rs:70-94)
 */
int32_t rg_batch_create(const rg_graph* g, const rg_params* p, const char* reads, const int64_t* read_off,
                        int64_t nreads, rg_batch** out);
/* Replaces the reads of an existing batch handle (same graph, same parameters) and uploads them: a streaming caller
 * keeps one or two handles and their HBM work buffers for the whole read set instead of re-creating them per chunk of
 * the reference's read loop (main.rs:56,174,257,297).  Results of the previous reads are discarded. */
int32_t rg_batch_set_reads(rg_batch* b, const char* reads, const int64_t* read_off, int64_t nreads);
int32_t rg_batch_run(rg_batch* b);
int32_t rg_batch_fetch(rg_batch* b);
void rg_batch_destroy(rg_batch* b);

int64_t rg_batch_size(const rg_batch* b);
uint32_t rg_result_status(const rg_batch* b, int64_t i);
int32_t rg_result_score(const rg_batch* b, int64_t i);   /* exec(..).0 of the reference (POA modes); best score otherwise */
/* seq_index is seq_name.1 of the reference (0 = score only: empty text). Returns bytes needed (excl. NUL). */
int64_t rg_result_gaf(const rg_batch* b, int64_t i, const char* name, int64_t seq_index, char* buf, int64_t cap);

/* Structured form of the same record: the fields of the reference's GAFStruct (gaf_output.rs:6-20) so that a caller
 * (the Rust shim under shim/) builds its GAFStruct without parsing text.  query_name is the caller's.  Strings are not
 * owned by the library: the node ids of `path` go to path_ids[0 .. n_path_ids) and `comments` (NUL-terminated) to
 * comments[] when the capacities suffice; call once with null buffers to learn the sizes.  alignment_block_length and
 * mapping_quality are "*" unless `empty` is set (GAFStruct::new(), gaf_output.rs:22-38: both "", strand ' ', path [0]). */
typedef struct rg_gaf_fields {
    int32_t has_record;          /* 0: the reference returns no GAFStruct for this read (it panics: see rg_result_status) */
    int32_t empty;               /* 1: GAFStruct::new() ("band not enough for correct output", gaf_output.rs:861-864) */
    uint32_t warning;            /* RG_READ_BAND_WARNING / RG_READ_BAND_NOT_ENOUGH: the stdout line printed before the record */
    char strand;
    uint64_t query_length, query_start, query_end;
    uint64_t path_length, path_start, path_end;
    uint64_t residue_matches_number;
    int64_t n_path_ids;
    int64_t comments_len;        /* bytes, excluding the NUL */
} rg_gaf_fields;
int32_t rg_result_fields(const rg_batch* b, int64_t i, rg_gaf_fields* out, uint64_t* path_ids, int64_t path_cap, char* comments,
                         int64_t comments_cap);

/* Formats every read of the batch (as rg_result_gaf would, name i = names[i] or "read<i>" when names is
 * NULL, seq_index = seq_index_base + i) into one buffer using `nthreads` host threads; returns bytes needed. */
int64_t rg_batch_format_all(const rg_batch* b, const char* const* names, int64_t seq_index_base, char* buf,
                            int64_t cap, int32_t nthreads);

/* Measurement hooks for bench.py: DP cell-updates of the last run (SURVEY §8d unit of work) and
 * per-kernel device time measured with HIP events on the stream the kernels were launched on. */
uint64_t rg_batch_cell_updates(const rg_batch* b);
/* The cell updates the kernels actually performed for that workload: the same number except where k_sweep16 runs a
 * segment's rows as a gather run (per row the alpha and a column map instead of one update per member path) and where
 * its -m 8 sweeps retire paths (a path that provably cannot matter any more, and leads no path that can,
 * is not updated further; the results are the same bytes).  The updates of a second pass after a failed speculation count
 * here (they are work the device performed) and not in rg_batch_cell_updates (the workload did not grow). */
uint64_t rg_batch_cell_updates_performed(const rg_batch* b);
int32_t rg_batch_kernel_count(const rg_batch* b);
const char* rg_batch_kernel_name(const rg_batch* b, int32_t k);
double rg_batch_kernel_ms(const rg_batch* b, int32_t k);       /* summed over launches of the last run */
int64_t rg_batch_kernel_launches(const rg_batch* b, int32_t k);

/* one-shot convenience: create + run + fetch */
int32_t rg_align_batch(const rg_graph* g, const rg_params* p, const char* reads, const int64_t* read_off,
                       int64_t nreads, rg_batch** out);

/*
 * The same read loop over several GPUs behind ONE call (the reference loop main.rs:56,174,257,297 has no order
 * dependence between reads): the reads go through the streaming engine below (tiles pulled from one queue by the
 * batch handles of every device in device_ids, NULL: every visible device; the graph tables are uploaded once per
 * device; nothing is exchanged between devices) and the call returns when the last tile is done.  Shard k is one tile:
 * it holds reads [begin_k, begin_{k+1}) as a results-only rg_batch, owned by the rg_multi, for the rg_result_* accessors
 * (rg_batch_run / rg_batch_set_reads refuse it); rg_multi_format_all concatenates the text of all shards in input
 * order (names[i], seq_index_base + i index the whole read set).  With names == NULL the default name of read i is
 * "read<j>", j = its index inside its shard (as rg_batch_format_all): pass names for global numbering.
 */
typedef struct rg_multi rg_multi;
int32_t rg_align_batch_multi(const rg_graph* g, const rg_params* p, const char* reads, const int64_t* read_off,
                             int64_t nreads, const int32_t* device_ids, int32_t ndev, rg_multi** out);
int32_t rg_multi_shards(const rg_multi* m);
rg_batch* rg_multi_batch(const rg_multi* m, int32_t k);
int64_t rg_multi_shard_begin(const rg_multi* m, int32_t k);      /* k == shards: nreads */
int64_t rg_multi_format_all(const rg_multi* m, const char* const* names, int64_t seq_index_base, char* buf, int64_t cap,
                            int32_t nthreads);
void rg_multi_destroy(rg_multi* m);

/*
 * FASTA ingestion.  Replaces sequences::get_sequences (sequences.rs:5-45) with the same rules: a line that starts with
 * '>' names a read (everything after the '>') and closes the read before it when that one has bases; every other
 * non-empty line is appended to the current read, upper-cased, '-' -> 'N'; lines end at '\n' with one '\r' before it
 * dropped (BufRead::lines); the reads are paired with the names BY INDEX, and a file whose counts differ is refused
 * (RG_ERR_ARG, "wrong fasta file format": the reference panics, :41-43).  The '$' the reference puts in front of
 * every read is added inside the library.  bases / offsets / names are the input form of rg_batch_create,
 * rg_stream_push and rg_align_batch_multi and stay valid until rg_reads_destroy.
 */
typedef struct rg_reads rg_reads;
int32_t rg_reads_from_fasta(const char* fasta_text, int64_t len, rg_reads** out);
int64_t rg_reads_count(const rg_reads* r);
const char* rg_reads_bases(const rg_reads* r);            /* concatenated bases of all reads */
const int64_t* rg_reads_offsets(const rg_reads* r);       /* count + 1 offsets into bases */
const char* const* rg_reads_names(const rg_reads* r);     /* count NUL-terminated names */
void rg_reads_destroy(rg_reads* r);
/* The check of sequences.rs:41-43 without keeping anything: the reference parses the whole file (and panics with "wrong
 * fasta file format" when the name and sequence counts differ) BEFORE it aligns the first read; a caller that streams
 * a file through rg_stream_feed_fasta runs this over the same blocks ahead of its first output.  state4: four words,
 * zero before the first piece.  Returns RG_ERR_ARG at the final piece of a text the reference refuses; *nreads_out:
 * reads so far. */
int32_t rg_fasta_check(const char* piece, int64_t len, int32_t final, int64_t* state4, int64_t* nreads_out);

/*
 * Streaming engine: the reference's read loop (main.rs:56,174,257,297-312) as a pipeline hidden behind the boundary.
 * Reads pushed into a stream are cut into tiles; per device `handles_per_device` batch handles (each with its own HIP
 * stream, HBM work buffers and host thread) pull tiles from ONE queue shared by all devices of the stream (a tile goes
 * to whichever handle is free first: load balance across devices without a static split), upload, align, fetch and
 * format them; the latency-bound small kernels of one tile run beside the sweeps of the others and the host work
 * (canonicalisation, upload, record fetch, GAF formatting on `format_threads` threads) overlaps the device work of
 * the other handles.  Results come back in input order, one tile per rg_stream_next call, as exactly the bytes the
 * reference prints on stdout for those reads.  Nothing is exchanged between devices.
 *
 *   rg_stream_create   starts the worker threads (the device buffers are allocated by the first tiles)
 *   rg_stream_push     copies the reads (and names) and queues their tiles; returns at once; any thread, any time
 *                      before rg_stream_finish.  names == NULL: read i of the stream is called "read<i>"
 *   rg_stream_finish   no more pushes: rg_stream_next returns RG_STREAM_END after the last tile
 *   rg_stream_next     blocks until the next tile (in input order) is done.  A tile whose device work failed returns
 *                      that error (rg_last_error has the text) and the stream moves on to the next tile
 *
 * seq_index of read i is seq_index_base + i (main.rs passes i + 1 in modes 0-3, where 0 means "score only": no text,
 * global_abpoa.rs:241; modes 4, 5, 8, 9 take no seq_name — main.rs:260,268,311 hand the 0-based i to write_gaf only —
 * and every read gets its record whatever the index).  One thread at a time may call rg_stream_next on a stream.
 *
 * Memory: by default the stream holds whatever was pushed and not yet delivered (like the reference holds its whole
 * `sequences` vector, main.rs:24).  For read sets that do not fit, bound it: `max_queued_tiles` makes rg_stream_push /
 * rg_stream_feed_fasta wait while that many tiles sit in the queue, `max_undelivered_bytes` makes the workers wait
 * before they start another tile while finished tiles that rg_stream_next has not taken yet hold more than that (the
 * tile rg_stream_next is waiting for is always started: no deadlock), rg_stream_feed_fasta takes the FASTA text in
 * pieces, and rg_stream_release returns a kept record handle early.  With a bound the pushing and the consuming side
 * must be different threads, or one thread that drains with rg_stream_next whenever rg_stream_pending says so.
 */
typedef struct rg_stream rg_stream;
typedef struct rg_stream_opts {
    int32_t handles_per_device;   /* 0: default (3)                                                            */
    int32_t tile_reads;           /* 0: default (4096 in the pathwise modes, 8192 in the POA modes)            */
    int32_t format_threads;       /* host threads that format one tile; 0: default (hardware threads / workers, 1..16) */
    int32_t keep_records;         /* 1: every tile also keeps its records (rg_stream_result.records)           */
    int64_t seq_index_base;       /* seq_index of the first read pushed                                        */
    int32_t no_text;              /* 1: skip the GAF text (records only)                                       */
    int32_t spin_wait;            /* 0: the long waits for the device (one per kernel pipeline) poll an event with short
                                     sleeps — every HIP wait spins a CPU by default, 3 threads per GPU here, which a
                                     node under a CPU quota cannot afford; 1: hipStreamSynchronize for the handles of
                                     THIS stream (rg_set_option("spin_wait", 1) is the process-wide switch)      */
    int32_t max_queued_tiles;     /* > 0: pushes wait while this many tiles are queued (not yet taken by a handle) */
    int32_t amb_strand;           /* 1: `-s true` (main.rs:82-106, 132-165, 188-212, 229-253; POA modes, ignored in the
                                     others like main.rs:254-313 ignores it): the reads that qualify (global modes: score
                                     < 0; local modes: all) are aligned again, reverse-complemented, on a second handle of
                                     the same worker (scalar exec + reversed handle labels as in the reference), the
                                     reference's per-mode comparison picks the record, and the warning lines of both
                                     exec calls come before it.  Not together with keep_records.                */
    int64_t max_undelivered_bytes;/* > 0: see "Memory" above                                                     */
} rg_stream_opts;
typedef struct rg_stream_result {
    int64_t first_read;           /* index (push order) of the tile's first read                               */
    int64_t nreads;
    const char* text;             /* GAF text of the tile's reads in input order, NUL-terminated               */
    int64_t text_len;
    const int64_t* text_off;      /* nreads + 1: the text of read i is text[text_off[i] .. text_off[i + 1])      */
    const uint32_t* status;       /* nreads: RG_READ_* bits                                                    */
    const int32_t* score;         /* nreads: rg_result_score                                                   */
    int32_t device;               /* HIP device that aligned the tile                                          */
    int32_t reserved;
    uint64_t cell_updates;        /* DP cell updates of the tile (SURVEY 8d unit of work)                       */
    rg_batch* records;            /* keep_records: results-only handle for the rg_result_* accessors, owned by the
                                     stream (valid until rg_stream_release / rg_stream_destroy), else NULL      */
    uint64_t cell_updates_performed; /* rg_batch_cell_updates_performed of the tile                              */
} rg_stream_result;
#define RG_STREAM_END 1
void rg_stream_opts_default(rg_stream_opts* o);
int32_t rg_stream_create(const rg_graph* g, const rg_params* p, const int32_t* device_ids, int32_t ndev,
                         const rg_stream_opts* opts, rg_stream** out);
int32_t rg_stream_push(rg_stream* s, const char* reads, const int64_t* read_off, int64_t nreads, const char* const* names);
/* rg_reads_from_fasta + rg_stream_push in one pass: every time tile_reads more reads are complete they are pushed, so the
 * devices start while the rest of the text is parsed.  *nreads_out: reads pushed.  A text whose name / sequence counts
 * differ at its end returns RG_ERR_ARG ("wrong fasta file format", sequences.rs:41-43) AFTER the complete reads before
 * that point were pushed: the caller drops the stream. */
int32_t rg_stream_push_fasta(rg_stream* s, const char* fasta_text, int64_t len, int64_t* nreads_out);
/* The same for a text that arrives in pieces (a file read block by block: the reference's BufReader, sequences.rs:7):
 * any split is fine, even inside a line; what a piece leaves unfinished stays inside the stream.  final != 0 closes the
 * text (the name / sequence count check happens there).  *nreads_out: reads completed and pushed by THIS call.  One
 * FASTA text per stream at a time; rg_stream_push_fasta is feed(text, final = 1). */
int32_t rg_stream_feed_fasta(rg_stream* s, const char* piece, int64_t len, int32_t final, int64_t* nreads_out);
/* Tiles pushed and not yet delivered by rg_stream_next (queued + on a device + finished). */
int64_t rg_stream_pending(rg_stream* s);
/* keep_records: gives a tile's record handle back before rg_stream_destroy (the handle is freed). */
void rg_stream_release(rg_stream* s, rg_batch* records);
int32_t rg_stream_finish(rg_stream* s);
/* Error exit of a bounded stream (the caller's counterpart of a panic inside the reference's read loop, main.rs:56-105):
 * tiles still queued are dropped, every thread blocked in rg_stream_push / rg_stream_feed_fasta (waiting for room) or in
 * rg_stream_next returns RG_ERR_ARG ("stream aborted"), later calls fail the same way; the tiles on a device finish and
 * are discarded by rg_stream_destroy, which also waits for threads still inside a push / feed call.  Any thread. */
int32_t rg_stream_abort(rg_stream* s);
/* RG_OK: *out describes the next tile (pointers valid until the next rg_stream_next / rg_stream_destroy on this stream);
 * RG_STREAM_END: finished and everything delivered; negative: that tile failed. */
int32_t rg_stream_next(rg_stream* s, rg_stream_result* out);
void rg_stream_destroy(rg_stream* s);
/* Measurement hooks (bench.py): per-kernel device time summed over every tile so far (HIP events on the handles' own
 * streams; with several handles per device a kernel's time includes the other streams' kernels), the host phases
 * ("host:set_reads", "host:run", "host:fetch", "host:format": wall milliseconds summed over the worker threads and
 * tiles; "host:first_tile_create", "host:first_tile_run": the part of them spent on each handle's first tile, where
 * the handle is created and its HBM work buffers are allocated; "host:context_warmup": device context creation in the
 * worker threads at rg_stream_create) listed after the kernels, tiles done, handles that aligned a tile. */
int32_t rg_stream_kernel_count(rg_stream* s);
const char* rg_stream_kernel_name(rg_stream* s, int32_t k);
double rg_stream_kernel_ms(rg_stream* s, int32_t k);
int64_t rg_stream_kernel_launches(rg_stream* s, int32_t k);
int64_t rg_stream_tiles_done(rg_stream* s);
int32_t rg_stream_handles(rg_stream* s);      /* batch handles that aligned at least one tile */

/* Process-wide diagnostic switches, also settable through the environment (read once when the library is loaded):
 * "sweep_i32" (RG_SWEEP_I32: i32 sweep kernel even when the packed 16-bit one is admissible), "three_sweeps"
 * (RG_THREE_SWEEPS), "no_frec" (RG_NO_FREC: Cand-list forward emission), "no_spec" (RG_NO_SPEC: -m 8 forward sweep
 * pruned with the provable bound instead of the speculative one), "spec_margin" (RG_SPEC_MARGIN, an integer: what the
 * speculative bound subtracts from the picked path's score; default 112), "chunk_reads" (RG_CHUNK_READS, an integer:
 * most reads one pathwise kernel launch takes), "no_gather" / "no_split" (RG_NO_GATHER / RG_NO_SPLIT: k_sweep16 without its gather runs / on the
 * plain step tables), "layer_i32" (RG_LAYER_I32: the layer rebuild in its i32 form), "no_retire" (RG_NO_RETIRE: k_sweep16 computes
 * every path to the end; 2 / 3: retirement in the forward / reverse sweep only), "no_pick2" (RG_NO_PICK2: the speculative
 * bound from one-path picks only), "no_order" (RG_NO_ORDER: the sweeps' waves in read order instead of longest first), "stripe_c" (RG_STRIPE_C: 8, 16 or 32 columns per lane for reads longer
 * than 2047 bases; 0 = 16 up to 8191 bases, 32 beyond), "spin_wait" (RG_SPIN_WAIT: hipStreamSynchronize instead of sleep-polling
 * for the long waits), "debug" (RG_DEBUG: list statistics on stderr; the packed k_opt0 is cross-checked against its i32 form),
 * "retire_shift" (RG_RETIRE_SHIFT, 2..12, default 8: the sweeps look for hopeless paths every 2^k step records; read when a
 * handle builds its step tables — the test suite runs its small graphs at 4), "lds_pad" (RG_LDS_PAD, bytes, experiments only:
 * extra dynamic LDS per k_sweep16 workgroup, which lowers the waves per CU).
 * The variants compute the same records byte for byte (tests/test_gpu_pathwise.py). */
int32_t rg_set_option(const char* name, int64_t value);
int64_t rg_get_option(const char* name);      /* -1: unknown option */

const char* rg_last_error(void);
int32_t rg_device_count(void);
int32_t rg_set_device(int32_t dev);

#ifdef __cplusplus
}
#endif
#endif
