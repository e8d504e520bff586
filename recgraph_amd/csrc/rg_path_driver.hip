// Batch driver of the pathwise modes (-m 4, -m 8): sizes the HBM work buffers, runs the kernel
// pipeline on one stream with HIP-event timing per kernel, regrows the candidate lists on overflow.
//
//   -m 4:  sweep(F, dirs) -> seed -> layer(F) -> trace
//   -m 8:  sweep(F1: column maxima) -> seed -> thr -> sweep(R: dirs, candidates, column maxima)
//          -> thr -> sweep(F2: dirs, candidates) -> search -> layer(F) -> layer(R) -> trace
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rg_path_kernels.hpp"

namespace rg {

#define HIPCHK(x)                                                                            \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) { (void)hipGetLastError(); /* clears the sticky error */ return fail(RG_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } \
    } while (0)

// (set by Buf::alloc when hipMalloc itself reported hipErrorOutOfMemory: the one failure the driver answers with smaller launches)
static thread_local bool g_alloc_oom = false;

template <typename T>
struct Buf {
    T* p = nullptr;
    size_t n = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    int alloc(size_t count) {
        if (p && count <= n) return RG_OK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        if (!count) count = 1;
        const hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            g_alloc_oom = e == hipErrorOutOfMemory;
            return fail(RG_ERR_HIP, std::string("hipMalloc of ") + std::to_string(count * sizeof(T)) + " bytes: " + hipGetErrorString(e));
        }
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

struct PathWorkImpl {
    bool tables = false;
    Buf<int> fpoff, fprow, fpslot, rpoff, rprow, rpslot;
    Buf<ReadState> state;
    Buf<int> roll, mf, wr, mfarg, wrarg, thr, flayer, rlayer;
    Buf<uint32_t> fdirs, rdirs;
    Buf<Cand> fcand, rcand;
    Buf<unsigned> nf, nr, ridx, nrec, nrrec;
    Buf<int> frec, rrec;
    unsigned frec_cap = 0, rrec_cap = 0;
    Buf<int> lb;
    Buf<int4> fsteps, rsteps;
    int nfsteps = 0, nrsteps = 0;
    Buf<int4> fsplit, rsplit;         // the same records with TAILs moved behind their register runs (split_tails below)
    Buf<unsigned long long> flead, rlead, fslead, rslead;    // PATH RETIREMENT tables of the four step tables (lead_table below)
    int retire_shift = RG_SWEEP16_RETIRE_SHIFT;              // the evaluation period the lead tables were built for
    unsigned long long fmembers = 0, rmembers = 0;           // member rows of the forward / reverse table (sum of the group sizes)
    bool have_split = false;
    unsigned fcap = 0, rcap = 0;
    std::vector<hipEvent_t> ev;
    hipEvent_t done_ev = nullptr;       // end-of-chunk marker polled by wait_stream_sleeping (no spinning host thread per handle)
    Buf<unsigned> need;                 // [8] per chunk: largest nf, nr, forward / reverse record count of a read (k_need); [4] reads whose
                                        // speculative bound failed (k_verify)
    // speculative bound (PickArgs): 12-mer table of the paths, per-read pick, and what aligning the failed reads again needs
    Buf<uint32_t> kmer_keys;
    Buf<unsigned long long> kmer_masks;
    unsigned kmer_mask = 0;
    Buf<int> pick, pick2, order, rt_idx;
    Buf<uint8_t> rt_flags, rt_reads, rt_bad, rt_ops;
    Buf<long long> rt_off;
    Buf<DevRecord> rt_rec;
    Buf<unsigned long long> rt_cells;
    PathWork retry;                     // work buffers of that second pass (a handful of reads)
    unsigned long long* h_sum = nullptr;  // pinned: {cell updates of the chunk, need[0..3]} read back once per chunk
    // SWEEPS ON A LOW-PRIORITY STREAM (round 6, option `sweep_prio`, OFF).  The handle's own stream has the highest priority
    // (rg_abi.hip) and carries the small kernels; with the option the sweeps go to this stream, fenced by two events.  The idea: in
    // the stream of several handles the small kernels of one tile compete with the sweep workgroups of the other handles for every
    // register slot a retiring wave frees (a 1 ms kernel takes 4-40 ms there).  Measured: the small kernels wait just as long —
    // what they wait for is a FREE slot, not their turn — config 5 118.4 / 115.0 k against 120.1 / 119.6 k reads/s, 1.5 kbp +2.6 %.
    hipStream_t sweep_stream = nullptr;
    hipEvent_t sw_before = nullptr, sw_after = nullptr;
    ~PathWorkImpl() {
        if (sweep_stream) (void)hipStreamDestroy(sweep_stream);
        if (sw_before) (void)hipEventDestroy(sw_before);
        if (sw_after) (void)hipEventDestroy(sw_after);
        for (auto e : ev) (void)hipEventDestroy(e);
        if (done_ev) (void)hipEventDestroy(done_ev);
        if (h_sum) (void)hipHostFree(h_sum);
    }
};
PathWork::~PathWork() { delete impl; }

namespace {

struct Timer {
    PathWorkImpl* w;
    hipStream_t s;
    bool spin = false;
    size_t used = 0;
    struct Pend { std::string name; size_t e0, e1; };
    std::vector<Pend> pend;
    int begin(const char* name) {
        while (w->ev.size() < used + 2) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            w->ev.push_back(e);
        }
        HIPCHK(hipEventRecord(w->ev[used], s));
        pend.push_back(Pend{name, used, used + 1});
        return RG_OK;
    }
    int end() {
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(w->ev[used + 1], s));
        used += 2;
        return RG_OK;
    }
    int collect(std::vector<std::pair<std::string, std::pair<double, long long>>>& stats) {
        if (!w->done_ev) HIPCHK(hipEventCreateWithFlags(&w->done_ev, hipEventDisableTiming));
        HIPCHK((hipError_t)wait_stream_sleeping(s, w->done_ev, spin));
        for (auto& p : pend) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, w->ev[p.e0], w->ev[p.e1]));
            bool found = false;
            for (auto& st : stats)
                if (st.first == p.name) { st.second.first += ms; st.second.second += 1; found = true; }
            if (!found) stats.push_back({p.name, {ms, 1}});
        }
        pend.clear();
        used = 0;
        return RG_OK;
    }
};

#define TIMED(T, name, call)                    \
    do {                                        \
        int rc_ = (T).begin(name);              \
        if (rc_) return rc_;                    \
        call;                                   \
        rc_ = (T).end();                        \
        if (rc_) return rc_;                    \
    } while (0)

}  // namespace

int path_driver_run(const HostGraph& h, const PathGraphDev& gd, const rg_params& p, PathWork& pw, const uint8_t* d_reads,
                    const long long* d_off, const uint8_t* d_bad, int nreads, int max_n, DevRecord* d_rec, uint8_t* d_ops,
                    long long ops_stride, unsigned long long* d_cells, hipStream_t stream, size_t mem_budget,
                    unsigned long long* cells_out /* [2]: counted | performed */,
                    std::vector<std::pair<std::string, std::pair<double, long long>>>& stats, int spec_level) {
    if (!pw.impl) pw.impl = new PathWorkImpl();
    PathWorkImpl& w = *pw.impl;
    const Options& opt = options();
    const bool debug = opt.debug != 0;
    const int pmode = p.mode;
    const bool semi = pmode == RG_MODE_PATHWISE_SEMI || pmode == RG_MODE_RECOMBINATION_SEMI;
    // pipeline selector: the semiglobal modes run the same kernels with the `semi` switches
    const int mode = pmode == RG_MODE_PATHWISE_SEMI ? RG_MODE_PATHWISE : pmode == RG_MODE_RECOMBINATION_SEMI ? RG_MODE_RECOMBINATION : pmode;
    const int P = h.P, L = h.L;
    int C = 4;
    while (C * WAVE < max_n + 1 && C < 32) C *= 2;
    // Reads longer than 2047 bases: column stripes, one wave per stripe in one workgroup of at most 8 waves (k_sweep /
    // k_layer <C, true>): i32 rows, Cand lists; needs a uniform read-gap cost (every matrix the reference's CLI builds).
    // Stripes of 1024 columns (C = 16: rows, keys and thresholds fit the 256 registers) up to 8191 bases; 2048 (C = 32,
    // which spills 800 registers) beyond; RG_STRIPE_C / rg_set_option("stripe_c") overrides (8, 16, 32).
    if (max_n + 1 > 32 * WAVE) {
        C = max_n + 1 <= 8 * 16 * WAVE ? 16 : 32;
        if ((opt.stripe_c == 8 || opt.stripe_c == 16 || opt.stripe_c == 32) && max_n + 1 <= 8 * opt.stripe_c * WAVE) C = opt.stripe_c;
    }
    const int nwv = (max_n + 1 + C * WAVE - 1) / (C * WAVE);
    if (nwv > 8) return fail(RG_ERR_ARG, "reads longer than 16383 bases are not supported by the pathwise kernels");
    if (nwv > 1)
        for (int b = 1; b < 5; ++b)
            if (p.scores[b * 6 + 5] != p.scores[5]) return fail(RG_ERR_ARG, "reads longer than 2047 bases need a uniform read-gap cost");
    const int wpad = nwv * C * WAVE;
    const int dir_words = nwv * WAVE * (C <= 16 ? 1 : 2);
    // packed 16-bit rows (rg_sweep16.hip) whenever the scores of this batch provably fit; RG_SWEEP_I32=1 forces the i32
    // kernel (test hook: the two must agree byte for byte)
    DevScores dsc;
    for (int i = 0; i < 36; ++i) dsc.t[i] = p.scores[i];
    const bool use16 = nwv == 1 && !opt.sweep_i32 && sweep16_admissible(dsc, h.max_path_rows, max_n, C);
    if (!use16) {
        // the i32 sweep packs (value, path) keys as value * 256 + path in 32 bits: |value| must stay below 2^23
        long long maxabs = 0;
        for (int i = 0; i < 36; ++i) if (i != 35 && p.scores[i] != RG_SCORE_MISSING) maxabs = std::max<long long>(maxabs, std::llabs((long long)p.scores[i]));
        if ((long long)(h.max_path_rows + max_n + 2) * maxabs >= (1ll << 23))
            return fail(RG_ERR_CAPACITY, "scores of this batch can reach 2^23 in magnitude: outside the 32-bit (value, path) keys of the pathwise kernels");
    }
    hipError_t memset_err = hipSuccess;
    auto sweep = [&](const SweepArgs& sa_, int nr) {
        // (striped sweeps advance the candidate counter of a read atomically from several waves: start it at zero)
        if (nwv > 1 && sa_.cand && sa_.ncand_out) {
            const hipError_t e = hipMemsetAsync(sa_.ncand_out, 0, sizeof(unsigned) * nr, stream);
            if (e != hipSuccess) memset_err = e;
        }
        hipStream_t ss = stream;
        if (opt.sweep_prio) {
            if (!w.sweep_stream) {
                int least = 0, greatest = 0;
                (void)hipDeviceGetStreamPriorityRange(&least, &greatest);        // (numerically: least >= greatest)
                if (hipStreamCreateWithPriority(&w.sweep_stream, hipStreamNonBlocking, least) != hipSuccess) w.sweep_stream = nullptr;
                if (w.sweep_stream && (hipEventCreateWithFlags(&w.sw_before, hipEventDisableTiming) != hipSuccess ||
                                       hipEventCreateWithFlags(&w.sw_after, hipEventDisableTiming) != hipSuccess)) {
                    (void)hipStreamDestroy(w.sweep_stream);
                    w.sweep_stream = nullptr;
                }
            }
            if (w.sweep_stream && hipEventRecord(w.sw_before, stream) == hipSuccess && hipStreamWaitEvent(w.sweep_stream, w.sw_before, 0) == hipSuccess)
                ss = w.sweep_stream;
        }
        if (use16) launch_sweep16(sa_, nr, C, ss);
        else launch_sweep(sa_, nr, C, ss);
        if (ss != stream) {
            const hipError_t e1 = hipEventRecord(w.sw_after, ss);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(stream, w.sw_after, 0) : e1;
            if (e2 != hipSuccess) memset_err = e2;      // (reported with the chunk's other asynchronous errors)
        }
    };
    int rc;
    if (!w.h_sum) HIPCHK(hipHostMalloc((void**)&w.h_sum, 8 * sizeof(unsigned long long), hipHostMallocDefault));
    if ((rc = w.need.alloc(8))) return rc;
    if (!w.tables) {
        // rows of every path in program order, with the direction-word slot of the group holding the path
        auto build = [&](const std::vector<int32_t>& goff, const std::vector<GroupDesc>& groups, bool fwd,
                         std::vector<int>& poff, std::vector<int>& prow, std::vector<int>& pslot) {
            std::vector<std::vector<std::pair<int, int>>> per(P);
            for (int step = 1; step + 1 < L; ++step) {
                const int i = fwd ? step : L - 1 - step;
                for (int gi = goff[i]; gi < goff[i + 1]; ++gi)
                    for (int b = 0; b < 64; ++b)
                        if ((groups[gi].mask >> b) & 1) per[groups[gi].page * 64 + b].push_back({i, groups[gi].slot});
            }
            poff.assign(P + 1, 0);
            prow.clear();
            pslot.clear();
            for (int k = 0; k < P; ++k) {
                for (auto& e : per[k]) { prow.push_back(e.first); pslot.push_back(e.second); }
                poff[k + 1] = (int)prow.size();
            }
        };
        std::vector<int> po, pr, ps;
        build(h.fgoff, h.fgroups, true, po, pr, ps);
        if ((rc = w.fpoff.upload(po)) || (rc = w.fprow.upload(pr)) || (rc = w.fpslot.upload(ps))) return rc;
        {
            // 12-mers of every path -> paths that contain them (k_pick votes with it): open addressing, 24-bit keys; NW words of
            // path bits per entry (one up to 64 paths)
            const size_t NW = (size_t)((P + 63) / 64);
            constexpr int K = 12;
            size_t total = 0;
            for (int k = 0; k < P; ++k) total += (size_t)std::max(0, po[k + 1] - po[k] - K + 1);
            size_t size = 64;
            while (size < 2 * total + 2) size <<= 1;
            std::vector<uint32_t> keys(size, 0xffffffffu);
            std::vector<unsigned long long> masks(size * NW, 0ull);
            for (int k = 0; k < P; ++k) {
                unsigned key = 0;
                int valid = 0;
                for (int t = po[k]; t < po[k + 1]; ++t) {
                    const size_t c = std::string("ACGT").find(h.lnz[pr[t]]);
                    if (c == std::string::npos) { valid = 0; key = 0; continue; }
                    key = ((key << 2) | (unsigned)c) & 0xffffffu;
                    if (++valid < K) continue;
                    unsigned hsh = (key * 2654435761u) >> 8;
                    for (unsigned probe = 0;; ++probe) {
                        const size_t slot = (hsh + probe) & (size - 1);
                        if (keys[slot] == 0xffffffffu) keys[slot] = key;
                        if (keys[slot] == key) { masks[slot * NW + (size_t)(k >> 6)] |= 1ull << (k & 63); break; }
                    }
                }
            }
            if ((rc = w.kmer_keys.upload(keys)) || (rc = w.kmer_masks.upload(masks))) return rc;
            w.kmer_mask = (unsigned)(size - 1);
        }
        build(h.rgoff, h.rgroups, false, po, pr, ps);
        if ((rc = w.rpoff.upload(po)) || (rc = w.rprow.upload(pr)) || (rc = w.rpslot.upload(ps))) return rc;
        // step tables of the sweeps (record layout documented in k_sweep)
        if (L >= (1 << 20) || h.fslots >= (1 << 20) || h.rslots >= (1 << 20)) return fail(RG_ERR_GRAPH, "graph too large for the sweep step table");
        // (built on the host by rg_steps.cpp: plain and split tables, the path-retirement tables of either, the member count)
        auto up_recs = [&](Buf<int4>& dst, const std::vector<StepRec>& v) -> int {
            static_assert(sizeof(StepRec) == sizeof(int4), "a step record is one int4");
            int rc2 = dst.alloc(v.size());
            if (rc2) return rc2;
            if (!v.empty()) HIPCHK(hipMemcpy(dst.p, v.data(), v.size() * sizeof(StepRec), hipMemcpyHostToDevice));
            return RG_OK;
        };
        w.have_split = true;       // (up to 64 paths: TAILs behind their register runs; more: the wide-run table, rg_steps.cpp)
        StepTables st;
        build_step_tables(h, true, w.have_split, st);
        if ((rc = up_recs(w.fsteps, st.plain))) return rc;
        w.nfsteps = (int)st.plain.size();
        w.fmembers = st.members;
        w.retire_shift = st.retire_shift;
        if ((rc = w.flead.upload(st.lead_plain))) return rc;
        if (w.have_split && ((rc = up_recs(w.fsplit, st.split)) || (rc = w.fslead.upload(st.lead_split)))) return rc;
        build_step_tables(h, false, w.have_split, st);
        if ((rc = up_recs(w.rsteps, st.plain))) return rc;
        w.nrsteps = (int)st.plain.size();
        w.rmembers = st.members;
        if ((rc = w.rlead.upload(st.lead_plain))) return rc;
        if (w.have_split && ((rc = up_recs(w.rsplit, st.split)) || (rc = w.rslead.upload(st.lead_split)))) return rc;
        w.tables = true;
    }
    const long long layer_stride = (long long)(h.max_path_rows + 2) * dir_words;   // traceback decisions: 2 bits per cell
    const long long fdirs_stride = (long long)h.fslots * dir_words;
    const long long rdirs_stride = (long long)h.rslots * dir_words;
    // reads per chunk: bounded by a memory budget for the per-read work buffers
    const size_t per_read = (size_t)(fdirs_stride + (mode == RG_MODE_RECOMBINATION ? rdirs_stride : 0)) * 4 +
                            (size_t)layer_stride * 4 * (mode == RG_MODE_RECOMBINATION ? 2 : 1) +
                            (size_t)(P + 2) * wpad * 4 + (size_t)wpad * 20 + sizeof(ReadState);
    size_t budget = (size_t)96 << 30;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget = fr / 4 * 3;
        // a handle of the streaming engine has its own share of the device (what it already holds counts towards it)
        if (mem_budget) budget = mem_budget;
    }
    // -m 8 pipeline: two sweeps (forward with a loose threshold from the exact path-0 score, then reverse) when
    // every gap entry is <= 0 (then w[.][j] <= (n - j) * max match); three sweeps otherwise / on request
    int maxmatch = 0;
    bool gaps_nonpos = true;
    for (int x = 0; x < 5; ++x) {
        gaps_nonpos = gaps_nonpos && p.scores[x * 6 + 5] <= 0 && p.scores[5 * 6 + x] <= 0;
        for (int y = 0; y < 5; ++y) maxmatch = std::max(maxmatch, p.scores[x * 6 + y]);
    }
    // (striped long reads, nwv > 1, take it too since round 4: k_opt0_striped gives them the forward bound)
    const bool two_sweep = mode == RG_MODE_RECOMBINATION && gaps_nonpos && !opt.three_sweeps;
    // forward emissions of the two-sweep pipeline are loose (threshold from the path-0 score): k_sweep16 writes them as
    // (row, lane) records that k_expand filters with the final bound; k_sweep writes plain Cand entries
    const bool use_rec = two_sweep && use16 && !opt.no_frec;
    // speculative forward bound (PickArgs in rg_path_kernels.hpp): checked by k_verify, failed reads aligned again below
    // (long reads emit Cand entries, not records: without the speculation their forward lists would hold every cell within
    // ~(seed - path-0 score) / 10 columns of a diagonal — millions per read at 5 kbp)
    // spec_level: 0 = the batch itself; 1 = the reads whose speculation failed, once more with a generous margin (their sweeps
    // still retire paths and emit few records: a second pass on the provable bound keeps one wave per read busy for two full
    // sweeps and a search over ~40 000 records); 2 = what fails even that, with the provable bound
    const bool allow_spec = spec_level < 2;
    // (round 5: the i32 sweep of reads that fit one wave takes it as well — score matrices outside the 16-bit budget, HOXD70 /
    // HOXD55 with their -200 gaps, emitted every forward cell within (seed - path-0 score) of a diagonal as a Cand)
    // (round 6: more than 64 paths too — on packed rows: k_pick votes over up to 256 paths, k_sweep16 retires over several words)
    const bool spec = two_sweep && allow_spec && !semi && (P <= 64 || (use16 && nwv == 1)) && !opt.no_spec;
    // (a follower path's sink value lies below its own NW optimum — measured up to 72 at 1 kbp — and the gap grows with the
    // read: long reads scale the margin with their length, or every read would fail the check and run again)
    // The margin is in units of the default scores (match 2): other matrices scale it with their best match (HOXD70: 100 -> x50).
    const int score_scale = std::max(1, maxmatch / 2);
    const int spec_margin = ((nwv > 1 ? opt.spec_margin * ((max_n + 999) / 1000) : (int)opt.spec_margin) +
                             (spec_level == 1 ? 320 * ((max_n + 999) / 1000) : 0)) * (opt.spec_margin > 0 ? score_scale : 1);
    // direction words on demand (k_sweep16, DIRECTION WORDS ON DEMAND): the record variants, first pass
    // only — a read that comes back because its final paths were not the picked ones stores every word the second time
    const bool pick_two_used = spec && !semi && !opt.no_pick2;
    const bool dsel = spec && use_rec && nwv == 1 && spec_level == 0 && !opt.no_dsel;
    // -m 4 on a speculative bound (round 6): k_pick's path, k_opt0's score against it minus the margin as the bound the sweep retires
    // paths against, direction words for the picked path only; k_verify4 sends a read whose best final score does not reach the bound
    // (or whose best path is not the pick) to a second pass without any of it.  Packed rows, one wave per read, the default scores' sign
    // conditions (gap entries <= 0: the hopeless bound counts on them).
    const bool spec4 = mode == RG_MODE_PATHWISE && !semi && use16 && nwv == 1 && gaps_nonpos && spec_level == 0 && !opt.no_spec;
    const int recw = 4 + C;
    if (w.fcap == 0) { w.fcap = (two_sweep && !use_rec) ? 1u << 20 : 1u << 15; w.rcap = use_rec ? 1u << 16 : 1u << 19; w.frec_cap = spec ? 1u << 14 : 1u << 16; w.rrec_cap = spec ? 1u << 13 : 1u << 15; }   // (round 6: 16 Ki / 8 Ki records of 80 B to start with instead of 64 Ki / 32 Ki — 2 MB per read instead of 7.9; a read that needs more regrows the lists and the chunk runs again, once per handle; batches WITHOUT a speculative bound keep the old sizes: their forward lists hold tens of thousands of records per read, and 128-path tiles ran twice every time a tile's largest read outgrew the last one's.  Config 5 with the two-path pick: forward mean ~11 k)
    stats.clear();
    Timer T{&w, stream, pw.spin_wait};
    int done = 0;
    unsigned long long cells_done = 0, cells_perf = 0;
    int oom_shift = 0;      // the budget is halved every time a work-buffer allocation fails (other handles / the retry pass took the memory)
    while (done < nreads) {
        HIPCHK(hipMemsetAsync(d_cells, 0, 2 * sizeof(unsigned long long), stream));  // cell updates of this chunk attempt (counted | performed)
        HIPCHK(hipMemsetAsync(w.need.p, 0, 8 * sizeof(unsigned), stream));
        const size_t per_read_all = per_read + (mode == RG_MODE_RECOMBINATION ? ((size_t)w.fcap * sizeof(Cand) + (size_t)w.rcap * (sizeof(Cand) + 4) +
                                                                                     (use_rec ? (size_t)(w.frec_cap + w.rrec_cap) * recw * 4 : 0)) : 0);
        int maxchunk = (int)std::min<size_t>(8192, std::max<size_t>(1, (budget >> oom_shift) / per_read_all));
        if (opt.chunk_reads > 0) maxchunk = std::min<int>(maxchunk, opt.chunk_reads);
        const double dbg_t0 = debug ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0;
        const int left = nreads - done;
        const int nchunks = (left + maxchunk - 1) / maxchunk;
        int chunk = (left + nchunks - 1) / nchunks;   // even chunks: no short tail launch
        auto alloc_all = [&]() -> int {
            int rc;
            if ((rc = w.state.alloc(chunk)) || (rc = w.fdirs.alloc((size_t)chunk * fdirs_stride)) ||
                (rc = w.flayer.alloc((size_t)chunk * layer_stride)))
                return rc;
            // (k_sweep16: P + 2 packed rows per read — two pseudo-rows behind the rolling rows — of wpad / 2 words)
            if ((rc = w.roll.alloc((size_t)chunk * (P + 2) * wpad))) return rc;
            if (mode == RG_MODE_RECOMBINATION) {
                if ((rc = w.rdirs.alloc((size_t)chunk * rdirs_stride)) || (rc = w.rlayer.alloc((size_t)chunk * layer_stride)) ||
                    (rc = w.mf.alloc((size_t)chunk * wpad)) || (rc = w.wr.alloc((size_t)chunk * wpad)) ||
                    (rc = w.mfarg.alloc((size_t)chunk * wpad)) || (rc = w.wrarg.alloc((size_t)chunk * wpad)) ||
                    (rc = w.thr.alloc((size_t)chunk * wpad)) || (rc = w.fcand.alloc((size_t)chunk * w.fcap)) ||
                    (rc = w.rcand.alloc((size_t)chunk * w.rcap)) || (rc = w.ridx.alloc((size_t)chunk * w.rcap)) ||
                    (rc = w.nf.alloc(chunk)) || (rc = w.nr.alloc(chunk)) || (rc = w.lb.alloc(chunk)) || (rc = w.nrec.alloc(chunk)))
                    return rc;
                if (spec && ((rc = w.pick.alloc(chunk)) || (rc = w.pick2.alloc((size_t)chunk * 2)) || (rc = w.order.alloc(chunk)) || (rc = w.rt_flags.alloc(chunk)))) return rc;
                if (use_rec && ((rc = w.frec.alloc((size_t)chunk * w.frec_cap * recw)) || (rc = w.rrec.alloc((size_t)chunk * w.rrec_cap * recw)) ||
                                (rc = w.nrrec.alloc(chunk))))
                    return rc;
            }
            if (spec4 && ((rc = w.lb.alloc(chunk)) || (rc = w.pick.alloc(chunk)) || (rc = w.order.alloc(chunk)) || (rc = w.rt_flags.alloc(chunk)))) return rc;
            return RG_OK;
        };
        g_alloc_oom = false;
        if ((rc = alloc_all())) {
            // out of memory (the share was measured before other buffers of the device existed): smaller launches, like run_poa.
            // Any OTHER HIP failure is reported at once (it used to be retried up to twelve times with halved chunks).
            if (rc == RG_ERR_HIP && g_alloc_oom && chunk > 1 && oom_shift < 12) { ++oom_shift; continue; }
            return rc;
        }
        const uint8_t* bad = d_bad + done;
        const long long* off = d_off + done;
        if (debug) fprintf(stderr, "[rg] chunk of %d reads: buffers ready after %.1f ms\n", chunk,
                           (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - dbg_t0) * 1e3);
        HIPCHK(hipMemsetAsync(w.state.p, 0, sizeof(ReadState) * chunk, stream));
        SweepArgs sa;
        memset(&sa, 0, sizeof sa);
        sa.g = gd;
        for (int i = 0; i < 36; ++i) sa.sc.t[i] = p.scores[i];
        sa.reads = d_reads; sa.read_off = off; sa.bad = bad; sa.state = w.state.p; sa.roll = w.roll.p;
        sa.fsteps = w.fsteps.p; sa.rsteps = w.rsteps.p; sa.nfsteps = w.nfsteps; sa.nrsteps = w.nrsteps;
        sa.semi = semi ? 1 : 0;
        sa.nwv = nwv;
        {
            // (gather runs carry differences of two members' stored values: sweep16_admissible bounds every such difference)
            sa.gather_ok = use16 && !opt.no_gather ? 1 : 0;
            // split tables: only where every run between the groups of a row is a register or a gather run of k_sweep16
            sa.fsplit = w.fsplit.p; sa.rsplit = w.rsplit.p;
            sa.use_split = w.have_split && sa.gather_ok && !semi && C <= 16 && !opt.no_split ? 1 : 0;
            // path retirement: the record pipelines of -m 8 (global), P <= 64
            sa.flead = w.flead.p; sa.rlead = w.rlead.p; sa.fslead = w.fslead.p; sa.rslead = w.rslead.p;
            sa.retire = (use16 || nwv == 1 || C <= 16) && (P <= 64 || (use16 && nwv == 1)) && !semi && mode == RG_MODE_RECOMBINATION && gaps_nonpos && opt.no_retire != 1 ? 1 : 0;   // (round 5: the i32 sweep too — one wave, or stripes of <= 16 columns per lane)
            sa.retire_shift = w.retire_shift;
            sa.fmembers = w.fmembers; sa.rmembers = w.rmembers;
            sa.maxmatch = maxmatch;      // (both sweeps: the retirement bound; the forward sweep's speculative thresholds)
        }
        sa.dsel_pick = dsel ? w.pick.p : nullptr; sa.dsel_pick2 = dsel && pick_two_used ? w.pick2.p : nullptr;
        { const int e = L / std::max(1, (int)opt.dsel_edge); sa.dsel_lo = e; sa.dsel_hi = L - 1 - e; }      // (an eighth of the rows at the end each sweep starts from)
        sa.rbw = p.rec_band_width; sa.cand_cap = 0; sa.dir_words = dir_words; sa.cells = d_cells;
        SeedArgs se;
        se.g = gd; se.state = w.state.p; se.nreads = chunk; se.mode = pmode; se.sc = sa.sc; se.reads = d_reads; se.read_off = off;
        if (mode == RG_MODE_PATHWISE) {
            SweepArgs f = sa;
            f.rev = 0; f.track_best = 0; f.dirs = w.fdirs.p; f.dirs_stride = fdirs_stride; f.count_cells = 1;
            f.retire = 0;
            if (spec4) {
                PickArgs pa{d_reads, off, bad, w.kmer_keys.p, w.kmer_masks.p, w.kmer_mask, P, w.pick.p, w.fpoff.p, w.fprow.p, nullptr};
                TIMED(T, "k_pick", launch_pick(pa, chunk, stream));
                // (the margin: a follower path's final score lies below its own alignment optimum — up to 180 at 1 kbp in config 4, where
                // -m 8's search maximum rarely does: 2.5 x the -m 8 margin; RG_SPEC4_MARGIN_X10 scales it for experiments)
                static const int m4x10 = getenv("RG_SPEC4_MARGIN_X10") ? atoi(getenv("RG_SPEC4_MARGIN_X10")) : 25;
                Opt0Args oa{gd, sa.sc, d_reads, off, bad, w.fpoff.p, w.fprow.p, w.lb.p, 0, w.pick.p, spec_margin * m4x10 / 10, nwv, nullptr, 0};
                TIMED(T, "k_opt0", launch_opt0_16(oa, chunk, C, stream));
                if (opt.no_retire != 1) {
                    f.retire = 1; f.lb = w.lb.p; f.maxmatch = maxmatch;
                    if (!opt.no_order) { launch_order(w.pick.p, nullptr, w.order.p, chunk, stream); f.order = w.order.p; }
                }
                if (!opt.no_dsel) { f.dsel_pick = w.pick.p; f.dsel_pick2 = nullptr; f.dsel_lo = 0; f.dsel_hi = L; }     // (no recombination: no edge rows)
            }
            TIMED(T, use16 ? "k_sweep16_fwd" : "k_sweep_fwd", sweep(f, chunk));
            TIMED(T, "k_seed", launch_seed(se, stream));
            if (spec4) launch_verify4(w.state.p, w.lb.p, w.need.p + 4, w.rt_flags.p, chunk, f.dsel_pick, stream);
        } else {
            if (two_sweep) {
                Opt0Args oa{gd, sa.sc, d_reads, off, bad, w.fpoff.p, w.fprow.p, w.lb.p, semi ? 1 : 0, nullptr, 0, nwv, nullptr, 0};
                if (spec) {
                    // (two-path picks: global mode)
                    const bool pick_two = !semi && !opt.no_pick2;
                    PickArgs pa{d_reads, off, bad, w.kmer_keys.p, w.kmer_masks.p, w.kmer_mask, P, w.pick.p, w.fpoff.p, w.fprow.p,
                                pick_two ? w.pick2.p : nullptr};
                    oa.pick2 = pa.pick2;
                    oa.rec_pen = p.base_rec_cost + (int)std::ceil(p.multi_rec_cost * 8.0f);
                    TIMED(T, "k_pick", launch_pick(pa, chunk, stream));
                    if (sa.retire && !opt.no_order) {
                        launch_order(w.pick.p, pa.pick2, w.order.p, chunk, stream);
                        sa.order = w.order.p;
                    }
                    oa.pick = w.pick.p;
                    oa.margin = spec_margin;
                }
                // (packed rows whenever the sweep runs packed: a third of the i32 form's instructions)
                const bool opt16 = use16 && nwv == 1;
                TIMED(T, "k_opt0", opt16 ? launch_opt0_16(oa, chunk, C, stream) : launch_opt0(oa, chunk, C, stream));
                if (opt16 && debug) {
                    // RG_DEBUG: the i32 form beside it — the two must agree on every read (the bound only steers the pruning, so
                    // no parity test would notice a wrong one: tests/test_gpu_pathwise.py runs this check)
                    std::vector<int> h16(chunk), h32(chunk);
                    HIPCHK(hipMemcpyAsync(h16.data(), w.lb.p, sizeof(int) * chunk, hipMemcpyDeviceToHost, stream));
                    HIPCHK(hipStreamSynchronize(stream));
                    launch_opt0(oa, chunk, C, stream);
                    HIPCHK(hipMemcpyAsync(h32.data(), w.lb.p, sizeof(int) * chunk, hipMemcpyDeviceToHost, stream));
                    HIPCHK(hipStreamSynchronize(stream));
                    for (int k = 0; k < chunk; ++k)
                        if (h16[k] != h32[k]) return fail(RG_ERR_HIP, "k_opt0_16 differs from k_opt0 at read " + std::to_string(k) + ": " + std::to_string(h16[k]) + " vs " + std::to_string(h32[k]));
                }
                SweepArgs f = sa;
                f.rev = 0; f.track_best = 1; f.lb = w.lb.p; f.brc = p.base_rec_cost + opt.lb_bonus; f.maxmatch = maxmatch;
                f.oob = spec ? 1 : 0;        // tight thresholds: most rows emit nothing (row_end tests the lane maximum first)
                f.colmax_out = w.mf.p; f.colarg_out = w.mfarg.p; f.cand = w.fcand.p; f.cand_cap = w.fcap; f.ncand_out = w.nf.p;
                if (use_rec) {
                    // records, and NO column maxima in the sweep (round 5: k_sweep16<.., 0, ..>, the reverse sweep's variant).  What
                    // the reverse thresholds and k_expand need per column is the best forward cell that can be a CANDIDATE — a
                    // usable cell at or above the forward emission threshold — and every such cell is in the records:
                    // k_colmax_rec reads maxima and cells out of them right behind the sweep (columns without a record get
                    // "no partner": no reverse emission there at all).  Round 4 tracked packed maxima of ALL cells in the sweep
                    // (8 registers, 8 v_pk_max per row) for an upper bound of the same quantity.
                    f.cand = nullptr; f.cand_cap = 0; f.frec = w.frec.p; f.frec_cap = w.frec_cap; f.ncand_out = w.nrec.p;
                    f.colmax_out = nullptr; f.colarg_out = nullptr;
                }
                f.dirs = w.fdirs.p; f.dirs_stride = fdirs_stride; f.count_cells = 1;
                if (opt.no_retire == 3) f.retire = 0;
                TIMED(T, use16 ? "k_sweep16_fwd" : "k_sweep_fwd", sweep(f, chunk));
                if (use_rec) {
                    ExpandArgs fc{w.state.p, w.frec.p, w.frec_cap, w.nrec.p, nullptr, 0, nullptr, nullptr, wpad, p.base_rec_cost, gd.knm, 0, off,
                                  p.rec_band_width, p.scores[5]};
                    TIMED(T, "k_colmax_rec_fwd", launch_colmax_rec(fc, w.mf.p, w.mfarg.p, chunk, C, stream));
                }
                TIMED(T, "k_seed", launch_seed(se, stream));
            } else {
                SweepArgs f1 = sa;
                f1.rev = 0; f1.track_best = 1; f1.colmax_out = w.mf.p; f1.colarg_out = w.mfarg.p; f1.count_cells = 0;
                TIMED(T, use16 ? "k_sweep16_fwd_colmax" : "k_sweep_fwd_colmax", sweep(f1, chunk));
                TIMED(T, "k_seed", launch_seed(se, stream));
            }
            ThrArgs t1{w.state.p, w.mf.p, w.thr.p, wpad, p.base_rec_cost, 0, two_sweep ? w.lb.p : nullptr};
            TIMED(T, "k_threshold", launch_threshold(t1, chunk, stream));
            SweepArgs r = sa;
            r.rev = 1; r.track_best = 1; r.thr = w.thr.p; r.colmax_out = w.wr.p; r.colarg_out = w.wrarg.p; r.cand = w.rcand.p; r.cand_cap = w.rcap; r.ncand_out = w.nr.p;
            if (use_rec) {
                // records instead of Cand entries; the reverse column maxima come from those records (k_colmax_rec)
                r.cand = nullptr; r.cand_cap = 0; r.frec = w.rrec.p; r.frec_cap = w.rrec_cap; r.ncand_out = w.nrrec.p;
                r.colmax_out = nullptr; r.colarg_out = nullptr;
            }
            r.dirs = w.rdirs.p; r.dirs_stride = rdirs_stride; r.count_cells = 1;
            if (opt.no_retire == 2) r.retire = 0;
            TIMED(T, use16 ? "k_sweep16_rev" : "k_sweep_rev", sweep(r, chunk));
            if (use_rec) {
                ExpandArgs ec{w.state.p, w.rrec.p, w.rrec_cap, w.nrrec.p, nullptr, 0, nullptr, nullptr, wpad, p.base_rec_cost, gd.knm, 1, off,
                              p.rec_band_width, p.scores[5]};
                TIMED(T, "k_colmax_rec", launch_colmax_rec(ec, w.wr.p, w.wrarg.p, chunk, C, stream));
            }
            BoundArgs ba{gd, w.state.p, off, w.mf.p, w.mfarg.p, w.wr.p, w.wrarg.p, wpad, p.base_rec_cost, p.multi_rec_cost,
                         p.rec_band_width};
            TIMED(T, "k_bound", launch_bound(ba, chunk, stream));
            if (!two_sweep) {
                ThrArgs t2{w.state.p, w.wr.p, w.thr.p, wpad, p.base_rec_cost, 1, nullptr};
                TIMED(T, "k_threshold", launch_threshold(t2, chunk, stream));
                SweepArgs f2 = sa;
                f2.rev = 0; f2.track_best = 1; f2.thr = w.thr.p; f2.cand = w.fcand.p; f2.cand_cap = w.fcap; f2.ncand_out = w.nf.p;
                f2.dirs = w.fdirs.p; f2.dirs_stride = fdirs_stride; f2.count_cells = 1;
                TIMED(T, use16 ? "k_sweep16_fwd" : "k_sweep_fwd", sweep(f2, chunk));
            }
            if (use_rec) {
                HIPCHK(hipMemsetAsync(w.nf.p, 0, sizeof(unsigned) * chunk, stream));
                HIPCHK(hipMemsetAsync(w.nr.p, 0, sizeof(unsigned) * chunk, stream));
                ExpandArgs ea{w.state.p, w.frec.p, w.frec_cap, w.nrec.p, w.fcand.p, w.fcap, w.nf.p, w.wr.p, wpad, p.base_rec_cost, gd.knm, 0, off, p.rec_band_width, p.scores[5]};
                TIMED(T, "k_expand", launch_expand(ea, chunk, C, stream));
                ExpandArgs er{w.state.p, w.rrec.p, w.rrec_cap, w.nrrec.p, w.rcand.p, w.rcap, w.nr.p, w.mf.p, wpad, p.base_rec_cost, gd.knm, 1, off, p.rec_band_width, p.scores[5]};
                TIMED(T, "k_expand", launch_expand(er, chunk, C, stream));
            }
            SearchArgs sr{gd, w.state.p, w.fcand.p, w.rcand.p, w.nf.p, w.nr.p, w.ridx.p, w.fcap, w.rcap, w.wr.p, wpad, p.base_rec_cost,
                          p.multi_rec_cost};
            TIMED(T, "k_search", launch_search(sr, chunk, stream));
            // largest list / record counts of the chunk, read back once after the traceback (no host round trip here)
            launch_need(w.state.p, w.nf.p, w.nr.p, use_rec ? w.nrec.p : nullptr, use_rec ? w.nrrec.p : nullptr, w.need.p, chunk, stream);
            if (spec) launch_verify(w.state.p, w.lb.p, w.need.p + 4, w.rt_flags.p, chunk, dsel ? w.pick.p : nullptr, dsel && pick_two_used ? w.pick2.p : nullptr, sa.dsel_lo, sa.dsel_hi, stream);
        }
        LayerArgs la;
        memset(&la, 0, sizeof la);
        la.semi = semi ? 1 : 0;
        la.nwv = nwv;
        la.g = gd; la.sc = sa.sc; la.reads = d_reads; la.read_off = off; la.state = w.state.p; la.dir_words = dir_words; la.dir_fmt = use16 ? 1 : 0;
        la.layer_stride = layer_stride; la.fpoff = w.fpoff.p; la.fprow = w.fprow.p; la.fpslot = w.fpslot.p;
        la.rpoff = w.rpoff.p; la.rprow = w.rprow.p; la.rpslot = w.rpslot.p;
        la.rev = 0; la.dirs = w.fdirs.p; la.dirs_stride = fdirs_stride; la.layer = w.flayer.p;
        // (packed 16-bit rows whenever the sweep ran packed: the same decisions, ~40 % fewer instructions; `layer_i32` keeps the
        // i32 form for the tests)
        bool gaps_agree = true;        // ('-', b) == (b, '-') for every base: the walkers' L key vs the sweep's
        for (int b = 0; b < 5; ++b) gaps_agree = gaps_agree && p.scores[5 * 6 + b] == p.scores[b * 6 + 5];
        const bool layer16 = use16 && nwv == 1 && gaps_agree && !opt.layer_i32;
        auto layer = [&]() { if (layer16) launch_layer16(la, chunk, C, stream); else launch_layer(la, chunk, C, stream); };
        TIMED(T, "k_layer_fwd", layer());
        if (mode == RG_MODE_RECOMBINATION) {
            la.rev = 1; la.dirs = w.rdirs.p; la.dirs_stride = rdirs_stride; la.layer = w.rlayer.p;
            TIMED(T, "k_layer_rev", layer());
        }
        TraceArgs ta;
        memset(&ta, 0, sizeof ta);
        ta.g = gd; ta.sc = sa.sc; ta.reads = d_reads; ta.read_off = off; ta.state = w.state.p; ta.rec = d_rec + done;
        ta.ops = d_ops + (long long)done * ops_stride; ta.ops_stride = ops_stride; ta.flayer = w.flayer.p;
        ta.rlayer = w.rlayer.p; ta.layer_stride = layer_stride; ta.fpoff = w.fpoff.p; ta.fprow = w.fprow.p;
        ta.rpoff = w.rpoff.p; ta.rprow = w.rprow.p; ta.nreads = chunk; ta.mode = pmode; ta.semi = semi ? 1 : 0; ta.nwv = nwv;
        TIMED(T, "k_trace", launch_trace(ta, C, stream));
        // ONE read-back per chunk: cell updates + overflow summary, through pinned memory on the batch's stream
        HIPCHK(hipMemcpyAsync(w.h_sum, d_cells, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(w.h_sum + 5, d_cells + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(w.h_sum + 1, w.need.p, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        if ((rc = T.collect(stats))) return rc;
        if (debug) fprintf(stderr, "[rg] chunk of %d reads: done after %.1f ms\n", chunk,
                           (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - dbg_t0) * 1e3);
        if (memset_err != hipSuccess) return fail(RG_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(memset_err));
        if (mode == RG_MODE_RECOMBINATION) {
            // candidate-list / record-list overflow: regrow and redo this chunk (k_layer / k_trace skipped its reads)
            const unsigned* nd = reinterpret_cast<const unsigned*>(w.h_sum + 1);
            const unsigned needf = nd[0], needr = nd[1], needrec = nd[2], needrrec = nd[3];
            if (debug) {
                std::vector<unsigned> hn(chunk), hr(chunk);
                HIPCHK(hipMemcpy(hn.data(), w.nf.p, sizeof(unsigned) * chunk, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(hr.data(), w.nr.p, sizeof(unsigned) * chunk, hipMemcpyDeviceToHost));
                unsigned long long sf = 0, sr = 0;
                for (int i = 0; i < chunk; ++i) { sf += hn[i]; sr += hr[i]; }
                fprintf(stderr, "[rg] chunk %d reads: fwd cand mean %.1f max %u (cap %u), rev cand mean %.1f max %u (cap %u)\n", chunk,
                        (double)sf / chunk, needf, w.fcap, (double)sr / chunk, needr, w.rcap);
                if (use_rec) {
                    HIPCHK(hipMemcpy(hn.data(), w.nrec.p, sizeof(unsigned) * chunk, hipMemcpyDeviceToHost));
                    HIPCHK(hipMemcpy(hr.data(), w.nrrec.p, sizeof(unsigned) * chunk, hipMemcpyDeviceToHost));
                    sf = sr = 0;
                    for (int i = 0; i < chunk; ++i) { sf += hn[i]; sr += hr[i]; }
                    fprintf(stderr, "[rg] records: forward mean %.1f max %u (cap %u), reverse mean %.1f max %u (cap %u)\n",
                            (double)sf / chunk, needrec, w.frec_cap, (double)sr / chunk, needrrec, w.rrec_cap);
                }
            }
            bool redo = false;
            if (use_rec && (needrrec > w.rrec_cap || needrec > w.frec_cap)) {
                const unsigned long long fullrec = (unsigned long long)L * WAVE;
                if ((needrrec > w.rrec_cap && w.rrec_cap >= fullrec) || (needrec > w.frec_cap && w.frec_cap >= fullrec))
                    return fail(RG_ERR_CAPACITY, "record list overflow at full size");
                while (w.rrec_cap < needrrec) w.rrec_cap = (unsigned)std::min<unsigned long long>(2ull * w.rrec_cap, fullrec);
                while (w.frec_cap < needrec) w.frec_cap = (unsigned)std::min<unsigned long long>(2ull * w.frec_cap, fullrec);
                redo = true;
            }
            if (needf > w.fcap || needr > w.rcap) {
                const unsigned long long full = (unsigned long long)L * wpad;
                if ((needf > w.fcap && w.fcap >= full) || (needr > w.rcap && w.rcap >= full) || needf > full || needr > full)
                    return fail(RG_ERR_CAPACITY, "candidate list overflow at full size");
                while (w.fcap < needf) w.fcap = (unsigned)std::min<unsigned long long>(2ull * w.fcap, full);
                while (w.rcap < needr) w.rcap = (unsigned)std::min<unsigned long long>(2ull * w.rcap, full);
                redo = true;
            }
            if (redo) continue;
        }
        const unsigned long long chunk_cells = w.h_sum[0];
        cells_perf += w.h_sum[5];
        const unsigned nretry = (spec || spec4) ? reinterpret_cast<const unsigned*>(w.h_sum + 1)[4] : 0u;
        if (debug && (spec || spec4)) fprintf(stderr, "[rg] speculative bound (margin %d): %u of %d reads did not reach it\n", spec_margin, nretry, chunk);
        if (debug && spec4 && nretry) {
            std::vector<ReadState> hs((size_t)chunk);
            std::vector<int> hp((size_t)chunk), hlb((size_t)chunk);
            HIPCHK(hipMemcpy(hs.data(), w.state.p, sizeof(ReadState) * (size_t)chunk, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(hp.data(), w.pick.p, sizeof(int) * (size_t)chunk, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(hlb.data(), w.lb.p, sizeof(int) * (size_t)chunk, hipMemcpyDeviceToHost));
            int shown = 0;
            for (int i = 0; i < chunk && shown < 16; ++i)
                if (hs[(size_t)i].status & ST_RETRY) {
                    ++shown;
                    fprintf(stderr, "[rg]   -m 4 read %d: pick %d, best path %d, best score %d, bound %d\n", i, hp[(size_t)i], hs[(size_t)i].fwd_path, hs[(size_t)i].s0, hlb[(size_t)i]);
                }
        }
        if (debug && dsel) {
            // which of the second-pass reads are there because of their paths, and what the paths were
            std::vector<ReadState> hs((size_t)chunk);
            std::vector<int> hp((size_t)chunk), hp2((size_t)chunk * 2, -1);
            std::vector<int> hlb((size_t)chunk);
            HIPCHK(hipMemcpy(hs.data(), w.state.p, sizeof(ReadState) * (size_t)chunk, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(hp.data(), w.pick.p, sizeof(int) * (size_t)chunk, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(hlb.data(), w.lb.p, sizeof(int) * (size_t)chunk, hipMemcpyDeviceToHost));
            if (pick_two_used) HIPCHK(hipMemcpy(hp2.data(), w.pick2.p, sizeof(int) * 2 * (size_t)chunk, hipMemcpyDeviceToHost));
            int nb = 0, npath = 0, shown = 0;
            for (int i = 0; i < chunk; ++i) {
                const ReadState& r = hs[(size_t)i];
                if (!(r.status & ST_RETRY)) continue;
                const bool bound_bad = r.fscore < (float)hlb[(size_t)i];
                nb += bound_bad;
                npath += !bound_bad;
                if (!bound_bad && shown < 24) {
                    ++shown;
                    fprintf(stderr, "[rg]   read %d: picks %d / %d, final paths %d -> %d (rec col %d), score %.1f, bound %d\n", i, hp[(size_t)i], hp2[2 * (size_t)i],
                            r.fwd_path, r.rev_path, r.rec_col, r.fscore, hlb[(size_t)i]);
                    fprintf(stderr, "[rg]        fen %d rsn %d of %d rows\n", r.fen, r.rsn, L);
                }
            }
            fprintf(stderr, "[rg] second pass: %d for the bound, %d for their paths\n", nb, npath);
        }
        if (nretry) {
            // The speculative bound of these reads was above what they reach: pairs may have been pruned.  Align them again,
            // as a batch of their own, with the provable bound, and put the records in place.
            std::vector<uint8_t> fl((size_t)chunk);
            std::vector<long long> ho((size_t)chunk + 1);
            HIPCHK(hipMemcpy(fl.data(), w.rt_flags.p, (size_t)chunk, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(ho.data(), d_off + done, sizeof(long long) * ((size_t)chunk + 1), hipMemcpyDeviceToHost));
            std::vector<int> idx;
            std::vector<long long> so(1, 0);
            for (int i = 0; i < chunk; ++i)
                if (fl[(size_t)i]) { idx.push_back(i); so.push_back(so.back() + (ho[(size_t)i + 1] - ho[(size_t)i])); }
            const int nr = (int)idx.size();
            if ((rc = w.rt_idx.upload(idx)) || (rc = w.rt_off.upload(so)) || (rc = w.rt_reads.alloc((size_t)so.back() + 1)) ||
                (rc = w.rt_bad.alloc((size_t)nr)) || (rc = w.rt_rec.alloc((size_t)nr)) || (rc = w.rt_ops.alloc((size_t)nr * (size_t)ops_stride)) ||
                (rc = w.rt_cells.alloc(2)))
                return rc;
            HIPCHK(hipMemsetAsync(w.rt_bad.p, 0, (size_t)nr, stream));
            launch_gather_reads(d_reads, d_off + done, w.rt_idx.p, w.rt_off.p, w.rt_reads.p, nr, stream);
            std::vector<std::pair<std::string, std::pair<double, long long>>> st2;
            unsigned long long c2[2] = {0, 0};
            if ((rc = path_driver_run(h, gd, p, w.retry, w.rt_reads.p, w.rt_off.p, w.rt_bad.p, nr, max_n, w.rt_rec.p, w.rt_ops.p, ops_stride,
                                      w.rt_cells.p, stream, 0, c2, st2, spec_level + 1)))
                return rc;
            launch_scatter_results(w.rt_idx.p, w.rt_rec.p, w.rt_ops.p, d_rec + done, d_ops + (long long)done * ops_stride, ops_stride, nr, stream);
            HIPCHK(hipStreamSynchronize(stream));
            cells_perf += c2[1];             // the second pass is work the sweeps performed too (the counted figure stays the workload's)
            for (auto& e : st2) {
                bool found = false;
                for (auto& st : stats) if (st.first == e.first) { st.second.first += e.second.first; st.second.second += e.second.second; found = true; }
                if (!found) stats.push_back(e);
            }
        }
        cells_done += chunk_cells;
        done += chunk;
        // HBM work buffers of one read of this chunk (rolling rows, direction words, layers, candidate and record lists): reported
        // as a pseudo-kernel so that callers see the footprint without another ABI entry ("ms" holds bytes, summed per chunk)
        stats.push_back({"mem:work_bytes_per_read", {(double)per_read_all, 1}});
    }
    if (cells_out) { cells_out[0] = cells_done; cells_out[1] = cells_perf; }
    return RG_OK;
}

}  // namespace rg
