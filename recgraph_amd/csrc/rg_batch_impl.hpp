// Private definitions shared by rg_abi.hip (the extern "C" surface + batch driver) and rg_stream.hip (the streaming
// engine): device / pinned buffers, the per-device graph tables and the batch handle.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rg_host.hpp"
#include "rg_path_args.hpp"
#include "rg_poa_args.hpp"

using namespace rg;

#define HIPCHK(x)                                                                                       \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) {                                                                         \
            (void)hipGetLastError(); /* clears the sticky error: the handle stays usable after a failed call */ \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? RG_ERR_NO_DEVICE : RG_ERR_HIP, \
                        std::string(#x) + ": " + hipGetErrorString(e_));                                \
        }                                                                                               \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    size_t bytes() const { return p ? n * sizeof(T) : 0; }
    int alloc(size_t count) {
        if (count <= n && p) return RG_OK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        if (count == 0) count = 1;
        HIPCHK(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        return RG_OK;
    }
    int upload(const std::vector<T>& v) {
        int rc = alloc(v.size());
        if (rc) return rc;
        if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return RG_OK;
    }
};

// Device copy of the flattened graph: one per HIP device that has a batch on this graph (built on first use, under the
// graph's mutex; the host arrays are immutable after creation, so a graph handle is shareable across threads and devices).
struct GraphTables {
    int dev = 0;
    // LnzGraph view
    DevBuf<uint8_t> d_lnz;
    DevBuf<int> d_pred_off, d_pred_rows, d_r_values, d_min_pred;
    // PathGraph view
    DevBuf<uint64_t> d_row_mask;
    DevBuf<int> d_knm, d_dfs, d_dfe, d_fgoff, d_rgoff, d_segfirst, d_seglast;
    DevBuf<GroupDesc> d_fgroups, d_rgroups;
    DevBuf<unsigned long long> d_node_id;
    DevBuf<int> d_eoff, d_epred, d_roff, d_rsucc;
    DevBuf<uint64_t> d_emask, d_rmask;
    DevBuf<uint8_t> d_pnwp, d_rnwp;
};

struct rg_graph {
    HostGraph h;
    std::mutex mu;
    std::map<int, std::unique_ptr<GraphTables>> tables;   // by device id
    ~rg_graph() {
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto& kv : tables) { (void)hipSetDevice(kv.first); kv.second.reset(); }
        (void)hipSetDevice(cur);
    }
};

static int base_code(char c) {
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4;
        default: return -1;
    }
}

// tables of `g` on the CURRENT device (uploaded once per device)
static int upload_graph(rg_graph* gr, GraphTables** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RG_ERR_NO_DEVICE, "no HIP device");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(gr->mu);
    auto it = gr->tables.find(dev);
    if (it != gr->tables.end()) { *out = it->second.get(); return RG_OK; }
    auto g = std::make_unique<GraphTables>();
    g->dev = dev;
    const HostGraph& h = gr->h;
    std::vector<uint8_t> codes(h.L, 0);
    for (int i = 1; i + 1 < h.L; ++i) {
        int c = base_code(h.lnz[i]);
        if (c < 0) return fail(RG_ERR_GRAPH, "graph base outside ACGTN (the reference panics on the score lookup)");
        codes[i] = (uint8_t)c;
    }
    int rc;
    if ((rc = g->d_lnz.upload(codes))) return rc;
    if (h.has_lnz) {
        if ((rc = g->d_pred_off.upload(h.pred_off)) || (rc = g->d_pred_rows.upload(h.pred_rows)) ||
            (rc = g->d_r_values.upload(h.r_values)) || (rc = g->d_min_pred.upload(h.min_pred)))
            return rc;
    }
    if (h.has_path) {
        std::vector<unsigned long long> ids(h.node_id.begin(), h.node_id.end());
        std::vector<int> segfirst(h.L), seglast(h.L);
        for (int i = 0; i < h.L; ++i) {
            // "first row of its segment" / "last row of its segment" flags of the recombination tie rule
            // (pathwise_alignment_recombination.rs:847-851)
            segfirst[i] = i >= 1 && h.node_id[i] != h.node_id[i - 1];
            seglast[i] = (i + 1 == h.L) || h.node_id[i] != h.node_id[i + 1];
        }
        // path masks as RG_PW words per row / edge
        auto flat = [](const std::vector<PMask>& v) {
            std::vector<uint64_t> o(v.size() * RG_PW);
            for (size_t i = 0; i < v.size(); ++i) for (int w = 0; w < RG_PW; ++w) o[i * RG_PW + w] = v[i].w[w];
            return o;
        };
        if ((rc = g->d_row_mask.upload(flat(h.row_mask))) || (rc = g->d_knm.upload(h.knm)) || (rc = g->d_dfs.upload(h.dfs)) ||
            (rc = g->d_dfe.upload(h.dfe)) || (rc = g->d_fgoff.upload(h.fgoff)) || (rc = g->d_rgoff.upload(h.rgoff)) ||
            (rc = g->d_fgroups.upload(h.fgroups)) || (rc = g->d_rgroups.upload(h.rgroups)) ||
            (rc = g->d_node_id.upload(ids)) || (rc = g->d_segfirst.upload(segfirst)) ||
            (rc = g->d_seglast.upload(seglast)) || (rc = g->d_eoff.upload(h.eoff)) || (rc = g->d_epred.upload(h.epred)) ||
            (rc = g->d_emask.upload(flat(h.emask))) || (rc = g->d_roff.upload(h.roff)) || (rc = g->d_rsucc.upload(h.rsucc)) ||
            (rc = g->d_rmask.upload(flat(h.rmask))) || (rc = g->d_pnwp.upload(h.pnwp)) || (rc = g->d_rnwp.upload(h.rnwp)))
            return rc;
    }
    *out = g.get();
    gr->tables[dev] = std::move(g);
    return RG_OK;
}

// page-locked host staging buffer (H2D / D2H by DMA, no pageable bounce copy)
template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    int alloc(size_t count) {
        if (count <= n && p) return RG_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; n = 0; }
        if (count == 0) count = 1;
        count += count / 4;           // head-room: read sets of a stream differ a little in size
        HIPCHK(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
        n = count;
        return RG_OK;
    }
};

struct KernelStat {
    std::string name;
    double ms = 0;
    long long launches = 0;
};

// Selects a device for the duration of an ABI entry and restores the caller's current device afterwards (a host such as
// torch may have another device selected on the calling thread).
struct DevGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DevGuard(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct rg_batch {
    const rg_graph* g = nullptr;
    GraphTables* gt = nullptr;         // the graph's tables on this batch's device
    rg_params p;
    int64_t nreads = 0;
    const uint8_t* codes = nullptr;    // base codes 0..4 per base (inside `stage`); the canonical text of a read (upper
                                       // case, '-' -> 'N': sequences.rs:13-22) is "ACGTN"[code]
    std::vector<long long> off;
    std::vector<uint8_t> bad;
    std::vector<int> bta;
    int max_n = 0;
    hipStream_t stream = nullptr;
    int dev = 0;                       // device the handle was created on (graph tables are bound to it too); -1: a
                                       // results-only handle detached from a stream tile (no device buffers, no stream)
    bool valid = false;                // reads loaded and every per-read buffer sized (false after a failed set_reads)
    bool spin_wait = false;            // long waits for the device spin (hipStreamSynchronize) instead of sleep-polling
    size_t mem_budget = 0;             // bytes of HBM the work buffers of one run may take (0: a share of what is free);
                                       // set by the streaming engine: free memory of the device / handles on it
    std::vector<uint8_t> codes_own;    // results-only handle: its own copy of the base codes (`codes` points here)
    // device inputs
    // one device block [off | bta | codes | bad] filled by ONE DMA from the pinned block `stage` (same layout)
    DevBuf<uint8_t> d_in;
    PinBuf<uint8_t> stage;
    struct InView { const uint8_t* reads; const long long* off; const uint8_t* bad; const int* bta; } in{};
    DevBuf<int> d_col0;
    DevBuf<int4> d_rowmeta, d_rowmeta_b;   // PoaArgs::rowmeta, rowmeta_b
    // work + outputs
    DevBuf<int> d_arena_m;
    DevBuf<uint32_t> d_arena_pw;
    DevBuf<int4> d_rinfo;
    DevBuf<DevRecord> d_rec;
    DevBuf<uint8_t> d_ops;
    DevBuf<int32_t> d_oprows;
    DevBuf<unsigned long long> d_cells;
    long long cap_cells = 0, ops_stride = 0;
    PathWork pw;                       // m4/m8 buffers
    // host results
    std::vector<DevRecord> rec;
    std::vector<uint8_t> ops;
    std::vector<int32_t> oprows;
    bool fetched = false;
    uint64_t cells = 0, cells_performed = 0;
    std::vector<KernelStat> stats;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    hipEvent_t done_ev = nullptr;      // end-of-run marker polled by wait_stream_sleeping
    ~rg_batch() {
        for (auto& e : ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (done_ev) (void)hipEventDestroy(done_ev);
        if (stream) (void)hipStreamDestroy(stream);
    }
};


// ---- internals shared with the streaming engine (rg_stream.hip) ----
void rg_batch_destroy_impl(rg_batch* b);
rg_batch* rg_batch_detach_results(rg_batch* b);
// `-s true` (main.rs:82-106, 132-165, 188-212, 229-253): the second batch (reverse complements of the reads that
// qualified, aligned with the reversed handle labels) and, per read of the forward batch, its index there (-1: none)
struct AmbRetry {
    const rg_batch* rb = nullptr;
    const int64_t* rev_index = nullptr;
};
// true: the reference writes the reverse-complement record (main.rs:98-102, 206-210: rev > fwd; local_poa :160-164: unless fwd < rev)
bool amb_take_rev(int mode, int32_t fwd_score, int32_t rev_score);
void format_batch(const rg_batch* b, const char* const* names, int64_t name_base, int64_t seq_index_base, int nthreads,
                  std::string& out, std::vector<int64_t>* offs, const AmbRetry* amb = nullptr);
