// extern "C" surface of librecgraph_hip (see include/recgraph_hip.h) and the batch driver:
// uploads graph + reads, sizes the HBM work buffers, launches the DP / search / traceback kernels
// on one HIP stream with HIP-event timing per kernel, copies the packed records back.
//
// There is deliberately NO CPU fallback: without a usable HIP device every batch call returns
// RG_ERR_NO_DEVICE.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include <unistd.h>

#include <cstdio>
#include <cstdlib>

#include "rg_batch_impl.hpp"

void rg_batch_destroy_impl(rg_batch* b) {
    if (!b) return;
    DevGuard dg(b->dev);            // buffers and stream are freed on the device that owns them
    delete b;
}

namespace rg {
Options& options() {
    static Options o;
    static std::once_flag once;
    std::call_once(once, [] {
        auto env = [](const char* k) { const char* v = getenv(k); return v && *v && strcmp(v, "0") != 0 ? 1 : 0; };
        o.sweep_i32 = env("RG_SWEEP_I32");
        o.three_sweeps = env("RG_THREE_SWEEPS");
        o.no_frec = env("RG_NO_FREC");
        o.debug = env("RG_DEBUG");
        { const char* v = getenv("RG_CHUNK_READS"); o.chunk_reads = v ? atoi(v) : 0; }
        { const char* v = getenv("RG_STRIPE_C"); o.stripe_c = v ? atoi(v) : 0; }
        { const char* v = getenv("RG_LB_BONUS"); o.lb_bonus = v ? atoi(v) : 0; }
        o.no_spec = env("RG_NO_SPEC");
        o.no_gather = env("RG_NO_GATHER");
        o.no_split = env("RG_NO_SPLIT");
        o.layer_i32 = env("RG_LAYER_I32");
        { const char* v = getenv("RG_NO_RETIRE"); o.no_retire = v ? atoi(v) : 0; }
        o.no_pick2 = env("RG_NO_PICK2");
        o.no_dsel = env("RG_NO_DSEL");
        o.sweep_prio = env("RG_SWEEP_PRIO");
        if (getenv("RG_DSEL_EDGE")) o.dsel_edge = std::max(1, atoi(getenv("RG_DSEL_EDGE")));
        o.no_order = env("RG_NO_ORDER");
        o.spin_wait = env("RG_SPIN_WAIT");
        { const char* v = getenv("RG_SPEC_MARGIN"); if (v) o.spec_margin = atoi(v); }
        { const char* v = getenv("RG_LDS_PAD"); o.lds_pad = v ? atoi(v) : 0; }
    });
    return o;
}

int wait_stream_sleeping(void* stream, void* ev, bool spin) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (spin || options().spin_wait || !ev) return (int)hipStreamSynchronize(s);
    hipEvent_t e = static_cast<hipEvent_t>(ev);
    hipError_t rc = hipEventRecord(e, s);
    if (rc != hipSuccess) return (int)rc;
    for (unsigned spins = 0;; ++spins) {
        rc = hipEventQuery(e);
        if (rc != hipErrorNotReady) return (int)rc;
        (void)hipGetLastError();                         // hipErrorNotReady is sticky in hipGetLastError
        usleep(50);                                      // ~100 us per poll with the timer slack: a few % of one CPU per thread
    }
}
}  // namespace rg

namespace {

struct Timed {
    rg_batch* b;
    std::vector<std::pair<int, int>> pending;  // (stat index, event index)
    size_t used = 0;
    explicit Timed(rg_batch* b_) : b(b_) {}
    int stat(const char* name) {
        for (size_t i = 0; i < b->stats.size(); ++i) if (b->stats[i].name == name) return (int)i;
        b->stats.push_back(KernelStat{name, 0, 0});
        return (int)b->stats.size() - 1;
    }
    template <typename F>
    int run(const char* name, F&& launch) {
        if (used == b->ev_pool.size()) {
            hipEvent_t a, c;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&c));
            b->ev_pool.emplace_back(a, c);
        }
        auto& ev = b->ev_pool[used];
        HIPCHK(hipEventRecord(ev.first, b->stream));
        launch();
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ev.second, b->stream));
        pending.emplace_back(stat(name), (int)used);
        ++used;
        return RG_OK;
    }
    int collect() {
        if (!b->done_ev) HIPCHK(hipEventCreateWithFlags(&b->done_ev, hipEventDisableTiming));
        HIPCHK((hipError_t)wait_stream_sleeping(b->stream, b->done_ev, b->spin_wait));
        for (auto& pe : pending) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, b->ev_pool[pe.second].first, b->ev_pool[pe.second].second));
            b->stats[pe.first].ms += ms;
            b->stats[pe.first].launches += 1;
        }
        pending.clear();
        used = 0;
        return RG_OK;
    }
};

bool is_local(int mode) { return mode == RG_MODE_LOCAL_POA || mode == RG_MODE_LOCAL_POA_SCALAR || mode == RG_MODE_GAP_LOCAL_POA; }
bool is_poa(int mode) {
    return mode == RG_MODE_GLOBAL_POA || mode == RG_MODE_GLOBAL_POA_SCALAR || mode == RG_MODE_GAP_POA || is_local(mode);
}

// Local modes fill full (L-1) x W matrices: reads are processed in launches of as many reads as fit the free HBM.
int run_local(rg_batch* b) {
    const GraphTables* g = b->gt;
    const HostGraph& h = b->g->h;
    const int mode = b->p.mode;
    const int planes = mode == RG_MODE_GAP_LOCAL_POA ? 2 : 1;
    const int variant = mode == RG_MODE_LOCAL_POA ? 0 : mode == RG_MODE_LOCAL_POA_SCALAR ? 1 : 2;
    const char* kname = variant == 0 ? "k_m1_local_simd" : variant == 1 ? "k_m1_local_scalar" : "k_m3_gap_local";
    Timed T(b);
    for (auto& s : b->stats) { s.ms = 0; s.launches = 0; }
    b->cap_cells = (long long)(h.L - 1) * (b->max_n + 1);
    const size_t per_read = (size_t)b->cap_cells * planes * (sizeof(int) + sizeof(uint32_t));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    free_b += b->d_arena_m.bytes() + b->d_arena_pw.bytes();   // arenas of a previous run are reused
    const size_t budget = b->mem_budget ? std::min(b->mem_budget, free_b / 10 * 9) : free_b / 4 * 3;
    if (per_read > budget) return fail(RG_ERR_CAPACITY, "local POA: one read's L x W matrices exceed the free HBM");
    const long long chunk = (long long)std::min<size_t>((size_t)b->nreads, budget / per_read);
    int rc;
    if ((rc = b->d_arena_m.alloc((size_t)chunk * b->cap_cells * planes)) ||
        (rc = b->d_arena_pw.alloc((size_t)chunk * b->cap_cells * planes)))
        return rc;
    HIPCHK(hipMemsetAsync(b->d_cells.p, 0, sizeof(unsigned long long), b->stream));
    PoaArgs a;
    a.g = DevLnz{h.L, g->d_lnz.p, g->d_pred_off.p, g->d_pred_rows.p, g->d_r_values.p, g->d_min_pred.p};
    for (int i = 0; i < 36; ++i) a.sc.t[i] = b->p.scores[i];
    a.reads = b->in.reads; a.read_off = b->in.off; a.bad = b->in.bad; a.bta = b->in.bta; a.col0 = b->d_col0.p; a.rowmeta = b->d_rowmeta.p; a.rowmeta_b = b->d_rowmeta_b.p;
    a.gap_open = b->p.gap_open; a.gap_ext = b->p.gap_ext; a.max_n = b->max_n; a.lds_read = b->max_n <= 16000 ? 1 : 0;
    a.cap_cells = b->cap_cells; a.arena_m = b->d_arena_m.p; a.arena_pw = b->d_arena_pw.p; a.rinfo = b->d_rinfo.p;
    a.rec = b->d_rec.p; a.ops = b->d_ops.p; a.oprows = b->d_oprows.p; a.ops_stride = b->ops_stride;
    a.cells = b->d_cells.p;
    for (long long base = 0; base < b->nreads; base += chunk) {
        a.read_base = (int)base;
        a.nreads = (int)std::min<long long>(chunk, b->nreads - base);
        if ((rc = T.run(kname, [&] { launch_local(a, variant, b->stream); }))) return rc;
    }
    if ((rc = T.collect())) return rc;
    unsigned long long c = 0;
    HIPCHK(hipMemcpy(&c, b->d_cells.p, sizeof c, hipMemcpyDeviceToHost));
    b->cells = c;
    b->cells_performed = c;
    return RG_OK;
}

int run_poa(rg_batch* b) {
    const GraphTables* g = b->gt;
    const HostGraph& h = b->g->h;
    if (!h.has_lnz) return fail(RG_ERR_ARG, "graph has no LnzGraph view");
    const int mode = b->p.mode;
    if (is_local(mode)) return run_local(b);
    const int planes = mode == RG_MODE_GAP_POA ? 2 : 1;   // m2: m | y score planes, w0 | w1 path planes
    Timed T(b);
    for (auto& s : b->stats) { s.ms = 0; s.launches = 0; }
    int oom_shift = 0;      // the budget is halved every time an arena allocation fails (other handles / threads took the memory)
    for (int attempt = 0; attempt < 24; ++attempt) {
        int rc;
        // Reads per launch: the band arenas of one launch take at most 45 % of the HBM that is free (plus what this
        // handle already holds), so that a second handle of a streaming caller fits beside this one; a handle of the
        // streaming engine has its own share of the device instead (mem_budget).
        const size_t per_read = (size_t)b->cap_cells * planes * (sizeof(int) + sizeof(uint32_t)) + (size_t)h.L * sizeof(int4);
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        free_b += b->d_arena_m.bytes() + b->d_arena_pw.bytes() + b->d_rinfo.bytes();
        const size_t budget = (b->mem_budget ? std::min(b->mem_budget, free_b / 10 * 9) : free_b / 100 * 45) >> oom_shift;
        if (per_read > budget) return fail(RG_ERR_CAPACITY, "band arena of one read exceeds the free HBM");
        const long long maxchunk = (long long)std::min<size_t>((size_t)b->nreads, budget / per_read);
        const long long nchunks = (b->nreads + maxchunk - 1) / maxchunk;
        const long long chunk = (b->nreads + nchunks - 1) / nchunks;      // even launches
        if (options().debug) fprintf(stderr, "[rg] run_poa attempt %d: cap_cells %lld, budget %.1f GB, per read %.2f MB, chunk %lld of %lld reads\n", attempt,
                                     b->cap_cells, budget / 1e9, per_read / 1e6, chunk, (long long)b->nreads);
        if ((rc = b->d_arena_m.alloc((size_t)chunk * b->cap_cells * planes)) ||
            (rc = b->d_arena_pw.alloc((size_t)chunk * b->cap_cells * planes)) || (rc = b->d_rinfo.alloc((size_t)chunk * h.L))) {
            if (rc == RG_ERR_HIP && chunk > 1 && oom_shift < 8) { ++oom_shift; continue; }   // out of memory: smaller launches
            return rc;
        }
        HIPCHK(hipMemsetAsync(b->d_cells.p, 0, sizeof(unsigned long long), b->stream));
        PoaArgs a;
        a.g = DevLnz{h.L, g->d_lnz.p, g->d_pred_off.p, g->d_pred_rows.p, g->d_r_values.p, g->d_min_pred.p};
        for (int i = 0; i < 36; ++i) a.sc.t[i] = b->p.scores[i];
        a.reads = b->in.reads; a.read_off = b->in.off; a.bad = b->in.bad; a.bta = b->in.bta; a.col0 = b->d_col0.p; a.rowmeta = b->d_rowmeta.p; a.rowmeta_b = b->d_rowmeta_b.p;
        a.max_n = b->max_n; a.lds_read = b->max_n <= 16000 ? 1 : 0; a.gap_open = b->p.gap_open; a.gap_ext = b->p.gap_ext;
        a.cap_cells = b->cap_cells; a.arena_m = b->d_arena_m.p; a.arena_pw = b->d_arena_pw.p; a.rinfo = b->d_rinfo.p;
        a.rec = b->d_rec.p; a.ops = b->d_ops.p; a.oprows = b->d_oprows.p; a.ops_stride = b->ops_stride;
        a.cells = b->d_cells.p;
        for (long long base = 0; base < b->nreads; base += chunk) {
            a.read_base = (int)base;
            a.nreads = (int)std::min<long long>(chunk, b->nreads - base);
            if (mode == RG_MODE_GLOBAL_POA) { if ((rc = T.run("k_m0_simd", [&] { launch_m0_simd(a, b->stream); }))) return rc; }
            else if (mode == RG_MODE_GAP_POA) { if ((rc = T.run("k_m2_gap", [&] { launch_m2(a, b->stream); }))) return rc; }
            else { if ((rc = T.run("k_m0_scalar", [&] { launch_m0_scalar(a, b->stream); }))) return rc; }
        }
        if ((rc = T.collect())) return rc;
        // overflow check: a read whose band cells did not fit asks for a bigger arena
        b->rec.resize(b->nreads);
        HIPCHK(hipMemcpy(b->rec.data(), b->d_rec.p, sizeof(DevRecord) * b->nreads, hipMemcpyDeviceToHost));
        bool ovf = false;
        for (auto& r : b->rec) if (r.status & ST_OVERFLOW) { ovf = true; break; }
        if (!ovf) {
            unsigned long long c = 0;
            HIPCHK(hipMemcpy(&c, b->d_cells.p, sizeof c, hipMemcpyDeviceToHost));
            b->cells = c;
            b->cells_performed = c;      // (the POA kernels evaluate exactly the band cells they count)
            return RG_OK;
        }
        const long long full = (long long)h.L * (b->max_n + 1);
        if (b->cap_cells >= full) return fail(RG_ERR_CAPACITY, "band arena overflow at full size");
        b->cap_cells = std::min(full, b->cap_cells * 2);
        for (auto& s : b->stats) { s.ms = 0; s.launches = 0; }
    }
    return fail(RG_ERR_CAPACITY, "band arena overflow");
}

}  // namespace

// pathwise driver lives in rg_path_driver.hip
int rg_run_pathwise(rg_batch* b);
namespace rg {
int path_driver_run(const HostGraph& h, const PathGraphDev& gd, const rg_params& p, PathWork& w, const uint8_t* d_reads,
                    const long long* d_off, const uint8_t* d_bad, int nreads, int max_n, DevRecord* d_rec, uint8_t* d_ops,
                    long long ops_stride, unsigned long long* d_cells, hipStream_t stream, size_t mem_budget,
                    unsigned long long* cells_out,
                    std::vector<std::pair<std::string, std::pair<double, long long>>>& stats, int spec_level);
}

// Text of read i exactly as the reference prints it (warning lines + GAFStruct::to_string), appended to `out`.
extern "C" {
static bool build_fields(const rg_batch* b, int64_t i, const char* name, GafFields& out);
}
static uint32_t public_status(uint32_t s) { return s & 0xffu; }
// the record of read i as the host formatter takes it
static void fill_record(const rg_batch* b, int64_t i, ReadRecord& r) {
    const DevRecord& d = b->rec[i];
    r.status = public_status(d.status); r.score = d.score; r.fscore = d.fscore; r.end_row = d.end_row; r.end_col = d.end_col;
    r.stop_row = d.stop_row; r.stop_col = d.stop_col; r.best_path = d.best_path; r.rev_path = d.rev_path; r.fen = d.fen;
    r.rsn = d.rsn; r.rec_col = d.rec_col; r.displacement = d.displacement; r.n_ops = d.n_ops; r.n_fwd_ops = d.n_fwd_ops;
    r.ops = b->ops.data() + (size_t)i * b->ops_stride;
    r.rows = is_poa(b->p.mode) ? b->oprows.data() + (size_t)i * b->ops_stride : nullptr;
}
bool amb_take_rev(int mode, int32_t fwd_score, int32_t rev_score) {
    if (mode == RG_MODE_LOCAL_POA || mode == RG_MODE_LOCAL_POA_SCALAR) return !(fwd_score < rev_score);   // main.rs:160-164 (sic)
    return rev_score > fwd_score;
}
static void append_gaf(const rg_batch* b, int64_t i, const char* name, int64_t seq_index, std::string& out, const AmbRetry* amb = nullptr) {
    const DevRecord& d = b->rec[i];
    GafFields f;
    // seq_name.1 == 0 means "score only" in the POA modes alone (global_abpoa.rs:241, :411; gap_global_abpoa.rs:229;
    // local_poa.rs / gap_local_poa.rs likewise).  The pathwise modes take no seq_name: main.rs:260,268,311 hand `i` to
    // write_gaf only, and read 0 gets its record like every other read.
    const bool score_only = seq_index == 0 && is_poa(b->p.mode);
    const int64_t k = amb && amb->rb && amb->rev_index ? amb->rev_index[i] : -1;
    if (k >= 0 && !score_only && !(amb->rb->rec[(size_t)k].status & (ST_BAD_BASE | ST_WOULD_PANIC))) {
        // both exec calls print their warning lines while they run; write_gaf then prints the record the comparison picks
        GafFields r;
        if (build_fields(b, i, name, f) && build_fields(amb->rb, k, name, r)) {
            out += f.pre;
            out += r.pre;
            out += amb_take_rev(b->p.mode, d.score, amb->rb->rec[(size_t)k].score) ? r.line() : f.line();
            out += '\n';
            return;
        }
    }
    if (!score_only && !is_poa(b->p.mode) && !(d.status & (ST_BAD_BASE | ST_WOULD_PANIC))) {
        // the pathwise modes: straight into the buffer (the same bytes as build_fields(..).text(): tests/test_gpu_stream.py
        // holds the stream's text, written here, equal to rg_result_gaf's, written there)
        ReadRecord r;
        fill_record(b, i, r);
        append_pathwise_text(b->g->h, b->codes + (size_t)b->off[i], (int)(b->off[i + 1] - b->off[i]), name ? name : "", r, b->p.mode, out);
        return;
    }
    if (!score_only && build_fields(b, i, name, f)) out += f.text();
    else if ((d.status & (ST_BAD_BASE | ST_WOULD_PANIC)) == 0 && (d.status & ST_BAND_WARNING))
        out += "Band length probably too short, maybe try with larger b and f\n";
}

// All GAF text of a fetched batch in input order, formatted by `nthreads` host threads (contiguous blocks of reads per
// thread, concatenated).  Name of read i: names[i], or "read<name_base + i>".  offs (optional): nreads + 1 offsets of the
// reads' texts inside `out`.
void format_batch(const rg_batch* b, const char* const* names, int64_t name_base, int64_t seq_index_base, int nthreads,
                  std::string& out, std::vector<int64_t>* offs, const AmbRetry* amb) {
    const int64_t n = b->nreads;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n) nthreads = (int)std::max<int64_t>(1, n);
    std::vector<std::string> parts((size_t)nthreads);
    std::vector<int64_t> lens((size_t)n);
    auto work = [&](int t) {
        const int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
        std::string& o = parts[(size_t)t];
        o.reserve((size_t)(hi - lo) * 2304);
        std::string nm;
        for (int64_t i = lo; i < hi; ++i) {
            const size_t before = o.size();
            const char* name;
            if (names) name = names[i];
            else { nm = "read" + std::to_string(name_base + i); name = nm.c_str(); }
            append_gaf(b, i, name, seq_index_base + i, o, amb);
            lens[(size_t)i] = (int64_t)(o.size() - before);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    size_t total = 0;
    for (auto& s : parts) total += s.size();
    out.clear();
    out.reserve(total + 1);
    for (auto& s : parts) out += s;
    if (offs) {
        offs->resize((size_t)n + 1);
        int64_t acc = 0;
        for (int64_t i = 0; i < n; ++i) { (*offs)[(size_t)i] = acc; acc += lens[(size_t)i]; }
        (*offs)[(size_t)n] = acc;
    }
}

// Results of a fetched batch as a handle of their own (host records only): the device handle is free for the next
// read set while a caller still reads this one through the rg_result_* accessors.
rg_batch* rg_batch_detach_results(rg_batch* b) {
    auto r = std::make_unique<rg_batch>();
    r->g = b->g;
    r->p = b->p;
    r->nreads = b->nreads;
    r->dev = -1;
    r->off = b->off;
    r->codes_own.assign(b->codes, b->codes + (size_t)b->off[(size_t)b->nreads]);
    r->codes = r->codes_own.data();
    r->rec = std::move(b->rec);
    r->ops = std::move(b->ops);
    r->oprows = std::move(b->oprows);
    r->ops_stride = b->ops_stride;
    r->cells = b->cells;
    r->cells_performed = b->cells_performed;
    r->stats = b->stats;
    r->fetched = true;
    b->fetched = false;
    return r.release();
}

extern "C" {

const char* rg_last_error(void) { return g_last_error.c_str(); }

static std::atomic<int>* option_slot(const char* name) {
    if (!name) return nullptr;
    Options& o = options();
    if (!strcmp(name, "sweep_i32")) return &o.sweep_i32;
    if (!strcmp(name, "three_sweeps")) return &o.three_sweeps;
    if (!strcmp(name, "no_frec")) return &o.no_frec;
    if (!strcmp(name, "debug")) return &o.debug;
    if (!strcmp(name, "chunk_reads")) return &o.chunk_reads;
    if (!strcmp(name, "stripe_c")) return &o.stripe_c;
    if (!strcmp(name, "no_spec")) return &o.no_spec;
    if (!strcmp(name, "no_gather")) return &o.no_gather;
    if (!strcmp(name, "no_split")) return &o.no_split;
    if (!strcmp(name, "layer_i32")) return &o.layer_i32;
    if (!strcmp(name, "no_retire")) return &o.no_retire;
    if (!strcmp(name, "no_pick2")) return &o.no_pick2;
    if (!strcmp(name, "no_dsel")) return &o.no_dsel;
    if (!strcmp(name, "sweep_prio")) return &o.sweep_prio;
    if (!strcmp(name, "dsel_edge")) return &o.dsel_edge;
    if (!strcmp(name, "no_order")) return &o.no_order;
    if (!strcmp(name, "spin_wait")) return &o.spin_wait;
    if (!strcmp(name, "spec_margin")) return &o.spec_margin;
    if (!strcmp(name, "lds_pad")) return &o.lds_pad;
    if (!strcmp(name, "retire_shift")) return &retire_shift_option();
    return nullptr;
}
int32_t rg_set_option(const char* name, int64_t value) {
    std::atomic<int>* s = option_slot(name);
    if (!s) return fail(RG_ERR_ARG, std::string("unknown option ") + (name ? name : "(null)"));
    Options& o = options();
    if (s == &o.stripe_c) *s = (int)std::max<int64_t>(0, std::min<int64_t>(value, 32));
    else if (s == &o.chunk_reads) *s = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20));
    else if (s == &retire_shift_option()) *s = (int)std::max<int64_t>(2, std::min<int64_t>(value, 12));
    else if (s == &o.lds_pad) *s = (int)std::max<int64_t>(0, std::min<int64_t>(value, 40 << 10));
    else if (s == &o.dsel_edge) *s = (int)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 20));     // 1 / dsel_edge of the rows always store their direction words
    else if (s == &o.spec_margin) *s = (int)std::max<int64_t>(-(1 << 24), std::min<int64_t>(value, 1 << 24));
    else if (s == &o.no_retire) *s = (int)std::max<int64_t>(0, std::min<int64_t>(value, 3));     // 1: off, 2: forward sweep only, 3: reverse sweep only
    else *s = value ? 1 : 0;
    return RG_OK;
}
int64_t rg_get_option(const char* name) {
    std::atomic<int>* s = option_slot(name);
    return s ? (int64_t)s->load() : -1;
}

int32_t rg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int32_t rg_set_device(int32_t dev) {
    HIPCHK(hipSetDevice(dev));
    return RG_OK;
}

void rg_scores_match_mis(int32_t m, int32_t x, int32_t f32_variant, int32_t* s) {
    // score_matrix.rs:35-66
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            if (i == j) s[i * 6 + j] = m;
            else if (!f32_variant && (i == 5 || j == 5)) s[i * 6 + j] = 2 * x;
            else s[i * 6 + j] = x;
        }
    s[4 * 6 + 4] = x;
    s[5 * 6 + 5] = RG_SCORE_MISSING;
}

void rg_params_default(rg_params* p, int32_t mode) {
    memset(p, 0, sizeof *p);
    p->mode = mode;
    rg_scores_match_mis(2, -4, 0, p->scores);
    p->gap_open = -4;
    p->gap_ext = -2;
    p->band_b = 1.0f;
    p->band_f = 0.01f;
    p->bta_override = -1;
    p->base_rec_cost = 4;
    p->multi_rec_cost = 0.1f;
    p->rec_band_width = 1.0f;
}

int32_t rg_graph_from_gfa(const char* gfa_text, int64_t len, rg_graph** out) {
    if (!gfa_text || !out) return fail(RG_ERR_ARG, "null argument");
    auto g = std::make_unique<rg_graph>();
    int rc = build_from_gfa(gfa_text, len, g->h);
    if (rc) return rc;
    *out = g.release();
    return RG_OK;
}
int32_t rg_graph_create_lnz(const char* lnz, int64_t L, const int64_t* pred_off, const int64_t* pred_rows,
                            const uint64_t* node_id, rg_graph** out) {
    if (!out) return fail(RG_ERR_ARG, "null argument");
    auto g = std::make_unique<rg_graph>();
    int rc = build_from_lnz(lnz, L, pred_off, pred_rows, node_id, g->h);
    if (rc) return rc;
    *out = g.release();
    return RG_OK;
}
int32_t rg_graph_create_path(const char* lnz, int64_t L, int32_t P, const uint64_t* row_mask, const int64_t* edge_off,
                             const int64_t* edge_pred, const uint64_t* edge_mask, const uint64_t* node_id, rg_graph** out) {
    if (!out) return fail(RG_ERR_ARG, "null argument");
    auto g = std::make_unique<rg_graph>();
    int rc = build_from_path(lnz, L, P, row_mask, edge_off, edge_pred, edge_mask, node_id, g->h);
    if (rc) return rc;
    *out = g.release();
    return RG_OK;
}
void rg_graph_destroy(rg_graph* g) { delete g; }
const char* rg_graph_path_error(const rg_graph* g) { return g ? g->h.path_error.c_str() : ""; }
int64_t rg_graph_rows(const rg_graph* g) { return g ? g->h.L : 0; }
int32_t rg_graph_paths(const rg_graph* g) { return g && g->h.has_path ? g->h.P : 0; }
int64_t rg_graph_dump(const rg_graph* g, int32_t which, char* buf, int64_t cap) {
    std::string s = dump_graph(g->h, which);
    if (buf && (int64_t)s.size() + 1 <= cap) memcpy(buf, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}

// Reads of a batch: canonicalised (sequences.rs:13-22), coded, uploaded; every per-read buffer is (re)sized.
// Work buffers of a previous run are kept when they are large enough, so a streaming caller re-uses one handle.
static int load_reads(rg_batch* b, const char* reads, const int64_t* read_off, int64_t nreads) {
    const rg_graph* g = b->g;
    const rg_params* p = &b->p;
    const int mode = p->mode;
    // validation first: a rejected read set leaves the handle exactly as it was
    for (int64_t r = 0; r < nreads; ++r)
        if (read_off[r + 1] - read_off[r] < 1) return fail(RG_ERR_ARG, "empty read");
    if (read_off[nreads] - read_off[0] >= (1ll << 40)) return fail(RG_ERR_ARG, "read set too large");
    // From here on the handle describes no read set until every buffer is sized and uploaded (`valid`): rg_batch_run and
    // rg_batch_fetch refuse it after a failed allocation instead of launching the new reads against the old buffers.
    b->valid = false;
    b->fetched = false;
    const long long base = read_off[0];
    const size_t total = (size_t)(read_off[nreads] - base);
    // staging layout (8-byte aligned pieces): offsets (int64), bta (int32), codes (u8), bad (u8)
    const size_t o_off = 0, o_bta = o_off + sizeof(long long) * (size_t)(nreads + 1);
    const size_t o_codes = (o_bta + sizeof(int) * (size_t)nreads + 7) & ~(size_t)7;
    const size_t o_bad = (o_codes + total + 7) & ~(size_t)7;
    const size_t in_bytes = o_bad + (size_t)nreads;
    int rc;
    if ((rc = b->stage.alloc(in_bytes)) || (rc = b->d_in.alloc(in_bytes + in_bytes / 4)) || (rc = b->d_rec.alloc(nreads)) ||
        (rc = b->d_cells.alloc(2)))
        return rc;
    b->nreads = nreads;
    b->off.resize(nreads + 1);
    for (int64_t i = 0; i <= nreads; ++i) b->off[i] = read_off[i] - base;
    b->bad.assign(nreads, 0);
    b->bta.resize(nreads);
    uint8_t* codes = b->stage.p + o_codes;
    b->codes = codes;
    // sequences.rs:13-22 ('-' -> 'N', upper case) + base codes, one table pass over the whole blob (rg_reads.cpp)
    b->max_n = (int)std::min<int64_t>(canonicalise_reads(reads, read_off, nreads, codes, b->bad.data()), INT32_MAX);
    for (int64_t r = 0; r < nreads; ++r) {
        const long long n = b->off[r + 1] - b->off[r];
        // main.rs:57: (b + f * seq.len() as f32) as usize, seq.len() = n + 1
        float v = p->band_b + p->band_f * (float)(n + 1);
        long long bt = p->bta_override >= 0 ? p->bta_override : (v > 0 ? (long long)v : 0);
        b->bta[r] = (int)std::min<long long>(bt, 1 << 28);
    }
    memcpy(b->stage.p + o_off, b->off.data(), sizeof(long long) * (size_t)(nreads + 1));
    memcpy(b->stage.p + o_bta, b->bta.data(), sizeof(int) * (size_t)nreads);
    memcpy(b->stage.p + o_bad, b->bad.data(), (size_t)nreads);
    const HostGraph& h = g->h;
    // traceback ops per read: POA walks at most L rows + n columns; a pathwise walk stays on the rows of one path per
    // half (forward to the source, reverse to the sink): <= 2 * (rows of the longest path + n)
    b->ops_stride = is_poa(mode) ? (long long)h.L + b->max_n + 8
                                 : std::min<long long>((long long)h.L + b->max_n + 8, 2ll * (h.max_path_rows + b->max_n) + 16);
    if ((rc = b->d_ops.alloc((size_t)nreads * b->ops_stride))) return rc;
    if (is_poa(mode)) {
        if ((rc = b->d_oprows.alloc((size_t)nreads * b->ops_stride))) return rc;
        long long maxbta = 0;
        for (int v : b->bta) maxbta = std::max<long long>(maxbta, v);
        const long long per_row = std::min<long long>(b->max_n + 1, 2 * maxbta + 40);
        b->cap_cells = std::max(b->cap_cells, (long long)h.L * per_row);   // keeps an arena regrown by an earlier run
    }
    // ONE DMA from the pinned block on the batch's stream (no blit kernel that would queue behind a running sweep of
    // another handle), ordered before the kernels of the next run
    HIPCHK(hipMemcpyAsync(b->d_in.p, b->stage.p, in_bytes, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->in.off = reinterpret_cast<const long long*>(b->d_in.p + o_off);
    b->in.bta = reinterpret_cast<const int*>(b->d_in.p + o_bta);
    b->in.reads = b->d_in.p + o_codes;
    b->in.bad = b->d_in.p + o_bad;
    b->valid = true;
    return RG_OK;
}

int32_t rg_batch_create(const rg_graph* gc, const rg_params* p, const char* reads, const int64_t* read_off, int64_t nreads,
                        rg_batch** out) {
    if (!gc || !p || !reads || !read_off || !out || nreads < 1) return fail(RG_ERR_ARG, "null/empty argument");
    const rg_graph* g = gc;
    const int mode = p->mode;
    if (!(is_poa(mode) || mode == RG_MODE_PATHWISE || mode == RG_MODE_RECOMBINATION || mode == RG_MODE_PATHWISE_SEMI ||
          mode == RG_MODE_RECOMBINATION_SEMI))
        return fail(RG_ERR_ARG, "unsupported mode");
    if ((mode == RG_MODE_GAP_POA || mode == RG_MODE_GAP_LOCAL_POA) && (p->gap_open > 0 || p->gap_ext > 0))
        return fail(RG_ERR_ARG, "gap penalties must be <= 0");
    if (is_poa(mode) && !g->h.has_lnz) return fail(RG_ERR_ARG, "graph has no LnzGraph view");
    if (!is_poa(mode) && !g->h.has_path)
        return fail(g->h.path_error.empty() ? RG_ERR_ARG : RG_ERR_GRAPH,
                    g->h.path_error.empty() ? "graph has no paths (P lines)" : "graph has no PathGraph view: " + g->h.path_error);
    if ((mode == RG_MODE_RECOMBINATION || mode == RG_MODE_RECOMBINATION_SEMI) && (p->base_rec_cost < 0 || p->multi_rec_cost < 0))
        return fail(RG_ERR_ARG, "recombination costs must be non-negative");
    if ((mode == RG_MODE_GLOBAL_POA || mode == RG_MODE_LOCAL_POA) && g->h.L >= (1 << 20))
        return fail(RG_ERR_GRAPH, "rows >= 2^20 break the reference's f32 path-cell decoding (gaf_output.rs:664-668, 783-786)");
    if ((mode == RG_MODE_GAP_POA || mode == RG_MODE_GLOBAL_POA_SCALAR || mode == RG_MODE_LOCAL_POA_SCALAR ||
         mode == RG_MODE_GAP_LOCAL_POA) && g->h.L > 65536)
        return fail(RG_ERR_GRAPH, "rows >= 65536 are truncated by the reference's u16 path cells (bitfield_path.rs:41)");
    if (p->amb_mode & ~3) return fail(RG_ERR_ARG, "amb_mode: only bits 0 and 1 are defined");
    if (p->amb_mode && !is_poa(mode)) return fail(RG_ERR_ARG, "amb_mode applies to the POA modes only (main.rs:82,132,188,229)");
    GraphTables* gt = nullptr;
    int rc = upload_graph(const_cast<rg_graph*>(g), &gt);   // per-device tables, under the graph's mutex
    if (rc) return rc;
    auto b = std::make_unique<rg_batch>();
    b->g = g;
    b->gt = gt;
    b->p = *p;
    HIPCHK(hipGetDevice(&b->dev));
    {
        // the handle's stream at the highest priority: it carries the small kernels; the pathwise sweeps run on a low-priority
        // stream of their own (rg_path_driver.hip, SWEEPS ON A LOW-PRIORITY STREAM)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, greatest) != hipSuccess)
            HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    }
    if (is_poa(mode)) {
        // column-0 chain of m0 (global_abpoa.rs:36-46): depends on graph + scores only
        const HostGraph& h = g->h;
        std::vector<int> col0(h.L, 0);
        for (int i = 1; i + 1 < h.L; ++i) {
            int c = base_code(h.lnz[i]);
            col0[i] = col0[h.min_pred[i]] + p->scores[c * 6 + 5];
        }
        if ((rc = b->d_col0.upload(col0))) return rc;
        if (mode == RG_MODE_GLOBAL_POA) {
            // (k_m0_simd only) the per-row record of PoaArgs::rowmeta
            std::vector<int4> rm(h.L, make_int4(0, 0, 0, 0));
            for (int i = 0; i < h.L; ++i) {
                const int pbeg = (int)h.pred_off[i], pend = (int)h.pred_off[i + 1];
                const int p0 = pend > pbeg ? (int)h.pred_rows[pbeg] : -1;
                const int c = (i >= 1 && i + 1 < h.L) ? base_code(h.lnz[i]) : 4;
                rm[i] = make_int4(pend, (int)h.r_values[i], col0[i], (p0 + 1) | ((c < 0 ? 4 : c) << 24));
            }
            if ((rc = b->d_rowmeta.upload(rm))) return rc;
        }
        if (mode == RG_MODE_GLOBAL_POA_SCALAR || mode == RG_MODE_GAP_POA) {
            std::vector<int4> rm(h.L, make_int4(0, 0, 0, 0));
            for (int i = 0; i < h.L; ++i) {
                const int pbeg = (int)h.pred_off[i], pend = (int)h.pred_off[i + 1];
                const int p0 = pend > pbeg ? (int)h.pred_rows[pbeg] : -1;
                const int c = (i >= 1 && i + 1 < h.L) ? base_code(h.lnz[i]) : 4;
                rm[i] = make_int4(pend, (int)h.r_values[i], i > 0 ? (int)h.min_pred[i] : 0, (p0 + 1) | ((c < 0 ? 4 : c) << 24));
            }
            if ((rc = b->d_rowmeta_b.upload(rm))) return rc;
        }
    }
    if ((rc = load_reads(b.get(), reads, read_off, nreads))) return rc;
    *out = b.release();
    return RG_OK;
}

static const char* kNoDevBuffers = "results-only handle (a tile of a stream / rg_multi): it has no device buffers";
static const char* kNoReads = "the handle holds no read set (a previous rg_batch_set_reads failed)";

int32_t rg_batch_set_reads(rg_batch* b, const char* reads, const int64_t* read_off, int64_t nreads) {
    if (!b || !reads || !read_off || nreads < 1) return fail(RG_ERR_ARG, "null/empty argument");
    if (b->dev < 0) return fail(RG_ERR_ARG, kNoDevBuffers);
    DevGuard dg(b->dev);            // the handle is bound to the device it was created on; the caller's device is restored
    HIPCHK(dg.err);
    return load_reads(b, reads, read_off, nreads);
}

int32_t rg_batch_run(rg_batch* b) {
    if (!b) return fail(RG_ERR_ARG, "null batch");
    if (b->dev < 0) return fail(RG_ERR_ARG, kNoDevBuffers);
    if (!b->valid) return fail(RG_ERR_ARG, kNoReads);
    b->fetched = false;
    DevGuard dg(b->dev);
    HIPCHK(dg.err);
    if (is_poa(b->p.mode)) return run_poa(b);
    return rg_run_pathwise(b);
}

int32_t rg_batch_fetch(rg_batch* b) {
    if (!b) return fail(RG_ERR_ARG, "null batch");
    if (b->dev < 0) return fail(RG_ERR_ARG, kNoDevBuffers);
    if (!b->valid) return fail(RG_ERR_ARG, kNoReads);
    DevGuard dg(b->dev);
    HIPCHK(dg.err);
    b->rec.resize(b->nreads);
    HIPCHK(hipMemcpy(b->rec.data(), b->d_rec.p, sizeof(DevRecord) * b->nreads, hipMemcpyDeviceToHost));
    b->ops.resize((size_t)b->nreads * b->ops_stride);
    HIPCHK(hipMemcpy(b->ops.data(), b->d_ops.p, b->ops.size(), hipMemcpyDeviceToHost));
    if (is_poa(b->p.mode)) {
        b->oprows.resize((size_t)b->nreads * b->ops_stride);
        HIPCHK(hipMemcpy(b->oprows.data(), b->d_oprows.p, b->oprows.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    b->fetched = true;
    return RG_OK;
}

void rg_batch_destroy(rg_batch* b) { rg_batch_destroy_impl(b); }
int64_t rg_batch_size(const rg_batch* b) { return b ? b->nreads : 0; }

uint32_t rg_result_status(const rg_batch* b, int64_t i) {
    if (!b || !b->fetched || i < 0 || i >= b->nreads) return RG_READ_WOULD_PANIC;
    return public_status(b->rec[i].status);
}
int32_t rg_result_score(const rg_batch* b, int64_t i) {
    if (!b || !b->fetched || i < 0 || i >= b->nreads) return 0;
    return b->rec[i].score;
}

// GAFStruct of read i (false: the reference produces none — score-only call, or it panics on this read)
static bool build_fields(const rg_batch* b, int64_t i, const char* name, GafFields& out) {
    const DevRecord& d = b->rec[i];
    if (d.status & (ST_BAD_BASE | ST_WOULD_PANIC)) return false;
    ReadRecord r;
    fill_record(b, i, r);
    std::string read((size_t)(b->off[i + 1] - b->off[i]), 'N');
    for (size_t k = 0; k < read.size(); ++k) read[k] = "ACGTN"[b->codes[(size_t)b->off[i] + k]];
    std::string nm = name ? name : "";
    switch (b->p.mode) {
        case RG_MODE_GLOBAL_POA: out = fields_m0_simd(b->g->h, read, nm, r, b->p.amb_mode); break;
        case RG_MODE_GLOBAL_POA_SCALAR:
        case RG_MODE_GAP_POA:
        case RG_MODE_LOCAL_POA:
        case RG_MODE_LOCAL_POA_SCALAR:
        case RG_MODE_GAP_LOCAL_POA: out = fields_poa_banded(b->g->h, read, nm, r, b->p.amb_mode); break;
        default: out = fields_pathwise(b->g->h, read, nm, r, b->p.mode); break;
    }
    return true;
}

int64_t rg_result_gaf(const rg_batch* b, int64_t i, const char* name, int64_t seq_index, char* buf, int64_t cap) {
    if (!b || !b->fetched || i < 0 || i >= b->nreads) return fail(RG_ERR_ARG, "result not available");
    std::string out;
    append_gaf(b, i, name, seq_index, out);
    if (buf && (int64_t)out.size() + 1 <= cap) memcpy(buf, out.c_str(), out.size() + 1);
    return (int64_t)out.size();
}

int32_t rg_result_fields(const rg_batch* b, int64_t i, rg_gaf_fields* out, uint64_t* path_ids, int64_t path_cap, char* comments,
                         int64_t comments_cap) {
    if (!b || !b->fetched || i < 0 || i >= b->nreads || !out) return fail(RG_ERR_ARG, "result not available");
    memset(out, 0, sizeof *out);
    GafFields f;
    if (!build_fields(b, i, "", f)) { out->strand = ' '; return RG_OK; }
    out->has_record = 1;
    out->empty = f.empty ? 1 : 0;
    out->query_length = f.qlen; out->query_start = f.qstart; out->query_end = f.qend;
    out->strand = f.strand;
    out->path_length = f.plen; out->path_start = f.pstart; out->path_end = f.pend;
    out->residue_matches_number = f.residues;
    out->n_path_ids = (int64_t)f.path.size();
    out->comments_len = (int64_t)f.comments.size();
    out->warning = f.pre.empty() ? 0 : (f.empty ? RG_READ_BAND_NOT_ENOUGH : RG_READ_BAND_WARNING);
    if (path_ids && path_cap >= (int64_t)f.path.size()) for (size_t k = 0; k < f.path.size(); ++k) path_ids[k] = f.path[k];
    if (comments && comments_cap >= (int64_t)f.comments.size() + 1) memcpy(comments, f.comments.c_str(), f.comments.size() + 1);
    return RG_OK;
}

int64_t rg_batch_format_all(const rg_batch* b, const char* const* names, int64_t seq_index_base, char* buf, int64_t cap,
                            int32_t nthreads) {
    if (!b || !b->fetched) return fail(RG_ERR_ARG, "result not available");
    std::string out;
    format_batch(b, names, 0, seq_index_base, nthreads, out, nullptr);
    const int64_t total = (int64_t)out.size();
    if (buf && total + 1 <= cap) memcpy(buf, out.c_str(), out.size() + 1);
    return total;
}

uint64_t rg_batch_cell_updates(const rg_batch* b) { return b ? b->cells : 0; }
uint64_t rg_batch_cell_updates_performed(const rg_batch* b) { return b ? b->cells_performed : 0; }
int32_t rg_batch_kernel_count(const rg_batch* b) { return b ? (int32_t)b->stats.size() : 0; }
const char* rg_batch_kernel_name(const rg_batch* b, int32_t k) { return b->stats[k].name.c_str(); }
double rg_batch_kernel_ms(const rg_batch* b, int32_t k) { return b->stats[k].ms; }
int64_t rg_batch_kernel_launches(const rg_batch* b, int32_t k) { return b->stats[k].launches; }

int32_t rg_align_batch(const rg_graph* g, const rg_params* p, const char* reads, const int64_t* read_off, int64_t nreads,
                       rg_batch** out) {
    rg_batch* b = nullptr;
    int rc = rg_batch_create(g, p, reads, read_off, nreads, &b);
    if (rc) return rc;
    if ((rc = rg_batch_run(b)) || (rc = rg_batch_fetch(b))) { rg_batch_destroy(b); return rc; }
    *out = b;
    return RG_OK;
}

}  // extern "C"

// ---- pathwise modes: buffers are owned by PathWork, kernels by rg_path_driver.hip ----
int rg_run_pathwise(rg_batch* b) {
    const GraphTables* g = b->gt;
    const HostGraph& h = b->g->h;
    PathGraphDev gd;
    gd.L = h.L; gd.P = h.P; gd.lnz = g->d_lnz.p; gd.row_mask = g->d_row_mask.p; gd.knm = g->d_knm.p;
    gd.dfs = g->d_dfs.p; gd.dfe = g->d_dfe.p; gd.fgoff = g->d_fgoff.p; gd.rgoff = g->d_rgoff.p;
    gd.fgroups = g->d_fgroups.p; gd.rgroups = g->d_rgroups.p; gd.fslots = h.fslots; gd.rslots = h.rslots;
    gd.node_id = g->d_node_id.p; gd.segfirst = g->d_segfirst.p; gd.seglast = g->d_seglast.p;
    gd.eoff = g->d_eoff.p; gd.epred = g->d_epred.p; gd.emask = g->d_emask.p; gd.roff = g->d_roff.p; gd.rsucc = g->d_rsucc.p;
    gd.rmask = g->d_rmask.p; gd.pnwp = g->d_pnwp.p; gd.rnwp = g->d_rnwp.p;
    std::vector<std::pair<std::string, std::pair<double, long long>>> st;
    // what this handle already holds counts towards its share of the device
    unsigned long long c[2] = {0, 0};
    b->pw.spin_wait = b->spin_wait;
    int rc = path_driver_run(h, gd, b->p, b->pw, b->in.reads, b->in.off, b->in.bad, (int)b->nreads, b->max_n, b->d_rec.p,
                             b->d_ops.p, b->ops_stride, b->d_cells.p, b->stream, b->mem_budget, c, st, 0);
    b->stats.clear();
    for (auto& s : st) b->stats.push_back(KernelStat{s.first, s.second.first, s.second.second});
    if (rc) return rc;
    b->cells = c[0];
    b->cells_performed = c[1];
    return RG_OK;
}
