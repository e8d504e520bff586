// Pathwise DP kernels for gfx950 (-m 4 and -m 8): sweep, recombination search, layer rebuild,
// traceback.  Semantics: SURVEY Appendix A.4-A.6 (absolute-score form of
// pathwise_alignment.rs:5-340 and pathwise_alignment_recombination.rs:129-873).
//
// Mapping (one wavefront per read, no inter-wave synchronisation):
//   * lane t owns the C consecutive DP columns t*C .. t*C+C-1 (C = ceil(columns/64));
//   * every path keeps ONE rolling row of absolute scores, stored as [q][lane] (q = column inside the
//     lane's chunk) so that the 64 lanes of a load/store touch 64 consecutive words: conflict-free in
//     LDS, fully coalesced in HBM.  Rows live in HBM (L2 / Infinity-Cache resident: LDS-resident rows allow one wave per CU and measured 3.4x slower);
//   * per (row, edge group): the group's alpha path runs the max-plus recurrence as a lane-local
//     serial scan + one wave-level prefix-max (the "left" dependency), giving a 2-bit direction per
//     column; every member path then follows those directions with its own values.  Runs of L are
//     resolved with a fill-forward (lane-local) + one cross-lane fetch from the nearest lane that
//     holds a non-L column;
//   * the reverse sweep is the same code on mirrored columns c' = n - j.
// Direction words (2 bits/column, one u32 per lane for C = 16) are the only per-cell data written
// to HBM; the recombination search consumes a compact candidate list instead of the two
// L x (n+1) x P matrices of the reference.
#include "rg_path_kernels.hpp"

namespace rg {

constexpr int NEG = INT32_MIN / 4;
#ifndef RG_SWEEP_WAVES
#define RG_SWEEP_WAVES 2
#endif

// rolling rows of one read in HBM (L2 / Infinity-Cache resident), [path][q][lane]
struct Rows {
    int* base;
    __device__ __forceinline__ int ld(int k, int idx, int wpad) const { return base[(long long)k * wpad + idx]; }
    __device__ __forceinline__ void st(int k, int idx, int wpad, int v) const { base[(long long)k * wpad + idx] = v; }
};
extern __shared__ __attribute__((aligned(16))) int g_lds[];

__device__ __forceinline__ int wave_excl_max(int v, int lane) {
    (void)lane;
    return dpp_shr1(dpp_incl_max(v, NEG), NEG);
}

// ---- striped long reads --------------------------------------------------------------------------------------------------
// A read longer than 2047 bases is cut into stripes of 64 * 32 columns, one wave per stripe in ONE workgroup.  Every wave
// runs the whole step table on its stripe; the recurrences only look left, so wave w needs from wave w - 1, per row
// update, the old value of its last column (the diagonal source of w's first column) and one carry (the alpha's running
// maximum / a member's last non-L value).  They travel through a single-producer single-consumer FIFO in LDS: wave w runs
// a few steps behind wave w - 1 (a systolic pipeline, no workgroup barrier in the row loop).
constexpr int FIFO_WORDS = 128;
struct StripeFifo {
    volatile int* buf;        // [FIFO_WORDS]
    unsigned* head;           // written by the producer (wave w - 1)
    unsigned* tail;           // written by the consumer (wave w)
    unsigned pos;             // this wave's own count (head as producer / tail as consumer)
    __device__ __forceinline__ void push2(int v0, int v1, int lane) {
        if (lane == 0) {
            while (pos + 2 - __hip_atomic_load(tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) > (unsigned)FIFO_WORDS) __builtin_amdgcn_s_sleep(1);
            buf[pos % FIFO_WORDS] = v0;
            buf[(pos + 1) % FIFO_WORDS] = v1;
            __hip_atomic_store(head, pos + 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        pos += 2;
    }
    __device__ __forceinline__ void pop2(int& v0, int& v1, int lane) {
        int a = 0, b = 0;
        if (lane == 0) {
            while (__hip_atomic_load(head, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < pos + 2) __builtin_amdgcn_s_sleep(1);
            a = buf[pos % FIFO_WORDS];
            b = buf[(pos + 1) % FIFO_WORDS];
            __hip_atomic_store(tail, pos + 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        pos += 2;
        v0 = __builtin_amdgcn_readfirstlane(a);
        v1 = __builtin_amdgcn_readfirstlane(b);
    }
};
struct StripeIO {            // carries of one row update: in from the stripe on the left, out to the stripe on the right
    int in_prev, in_carry;   // old value of the left stripe's last column; alpha: its running maximum (z-space), member: its last non-L value
    int out_prev, out_carry;
};

// In-place row operators.  kUni: every read base has the same gap cost (all matrices the reference CLI can
// build, score_matrix.rs:35-105), so the prefix sum of the read-gap cost up to column c is c * gcost and needs no
// registers.
// kStripes: `lane` in column arithmetic is the GLOBAL lane (stripe * 64 + lane), `wl` the lane inside the wave.
template <int C, bool kUni, bool kStripes = false>
struct RowOps {
    static __device__ __forceinline__ int gp(const int (&GP)[kUni ? 1 : C], int gcost, int lane, int q) {
        if (kUni) return (lane * C + q) * gcost;
        return GP[kUni ? 0 : q];
    }
    // group alpha: max(d, u, l) as a lane-local serial scan + one wave prefix-max; returns the direction masks and
    // the fill-forward source lane (nearest lane to the left that owns a non-L column)
    static __device__ __forceinline__ void alpha(int (&row)[C], const int (&s)[C], const int (&GP)[kUni ? 1 : C], int gcost,
                                                 int g_i, int g0, int lane, int ncols, unsigned& dmask, unsigned& lmask, int& src,
                                                 int wl = 0, StripeIO* io = nullptr) {
        int prev_old = dpp_shr1(row[C - 1], NEG);
        if (kStripes) {
            io->out_prev = __builtin_amdgcn_readlane(row[C - 1], WAVE - 1);
            if (wl == 0) prev_old = io->in_prev;
        }
        int runmax = NEG;
        unsigned dm = 0, lm = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            const int old = row[q];
            // g0: what the border column adds (the row's gap cost; 0 in the semiglobal modes)
            const int d = prev_old + s[q], u = old + (q == 0 ? (c == 0 ? g0 : g_i) : g_i);
            const bool isd = (q == 0 ? c != 0 : true) && d >= u;   // border column 0: U only; priority D > U
            int du = isd ? d : u;
            du = c < ncols ? du : NEG;
            dm |= (isd ? 1u : 0u) << q;
            prev_old = old;
            const int x = du - gp(GP, gcost, lane, q);
            row[q] = x;
            runmax = max(runmax, x);
        }
        const int incl = dpp_incl_max(runmax, NEG);
        int run = dpp_shr1(incl, NEG);
        if (kStripes) {
            run = max(run, io->in_carry);
            io->out_carry = max(io->in_carry, __builtin_amdgcn_readlane(incl, WAVE - 1));
        }
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            const int x = row[q];
            lm |= (run > x ? 1u : 0u) << q;                        // L only when strictly better (D > U > L)
            const int y = max(run, x);
            row[q] = c < ncols ? y + gp(GP, gcost, lane, q) : NEG;
            run = y;
        }
        const unsigned full = C >= 32 ? 0xffffffffu : ((1u << C) - 1u);
        // (kStripes: wave-local lanes; -1 = no non-L column to the left inside this stripe)
        src = dpp_shr1(dpp_incl_max((lm & full) != full ? (kStripes ? wl : lane) : -1, -1), kStripes ? -1 : 0);
        dmask = dm; lmask = lm;
    }
    // member: follow the alpha's directions with the path's own values
    static __device__ __forceinline__ void member(int (&row)[C], const int (&s)[C], const int (&GP)[kUni ? 1 : C], int gcost,
                                                  int g_i, int g0, int lane, int ncols, unsigned dmask, unsigned lmask, int src,
                                                  int wl = 0, StripeIO* io = nullptr) {
        int prev_old = dpp_shr1(row[C - 1], NEG);
        if (kStripes) {
            io->out_prev = __builtin_amdgcn_readlane(row[C - 1], WAVE - 1);
            if (wl == 0) prev_old = io->in_prev;
        }
        int last = NEG;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int old = row[q];
            const int base = ((dmask >> q) & 1) ? prev_old + s[q] : old + (q == 0 ? (lane == 0 ? g0 : g_i) : g_i);
            prev_old = old;
            const int y = base - gp(GP, gcost, lane, q);
            row[q] = y;
            last = ((lmask >> q) & 1) ? last : y;
        }
        int cur = __shfl(last, kStripes ? max(src, 0) : src, WAVE);   // y of the last non-L column before this lane
        if (kStripes && src < 0) cur = io->in_carry;                // ... which lies in a stripe further left
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            cur = ((lmask >> q) & 1) ? cur : row[q];
            row[q] = c < ncols ? cur + gp(GP, gcost, lane, q) : NEG;
        }
        if (kStripes) io->out_carry = __builtin_amdgcn_readlane(cur, WAVE - 1);
    }
};

// One DP sweep over the whole graph for one read.
// kStripes: a.nwv waves per read, wave w owns columns [w * 64 * C, (w + 1) * 64 * C) (see StripeFifo above).
template <int C, bool kUni, bool kStripes = false>
__global__ __launch_bounds__(kStripes ? 512 : 64, kStripes ? 1 : RG_SWEEP_WAVES) void k_sweep(SweepArgs a) {
    const int rd = (!kStripes && a.order) ? a.order[blockIdx.x] : blockIdx.x;       // (launch order: see launch_order)
    const int wl = kStripes ? (int)(threadIdx.x & (WAVE - 1)) : (int)threadIdx.x;      // lane inside the wave
    const int wv = kStripes ? (int)(threadIdx.x >> 6) : 0;                             // stripe of this wave
    const int nwv = kStripes ? a.nwv : 1;
    const int lane = wv * WAVE + wl;                                                   // global lane: column = lane * C + q
    const PathGraphDev& g = a.g;
    const int P = g.P;
    const int wpadw = C * WAVE;                 // columns of one stripe
    const int wpad = nwv * wpadw;               // columns per read (side buffers)
    ReadState* rs = a.state + rd;
    const long long ro = a.read_off[rd];
    const int n = __builtin_amdgcn_readfirstlane((int)(a.read_off[rd + 1] - ro));
    if (a.bad[rd] || n + 1 > wpad) {
        if (lane == 0 && !a.rev) { rs->status = a.bad[rd] ? ST_BAD_BASE : ST_WOULD_PANIC; }
        return;
    }
    const uint8_t* read = a.reads + ro - 1;  // read[1..n]
    const bool rev = a.rev;
    const int ncols = rev ? n : n + 1;  // mirrored columns of the reverse sweep: c' = n - j, j = n..1
    const int GAP = 5;
    // score table in LDS
    int* sct = g_lds;
    if (lane < 36) sct[lane] = a.sc.t[lane];
    // stripe FIFOs (after the score table and the semiglobal end arrays): queue w carries wave w - 1 -> wave w
    int* fifo_lds = g_lds + 64 + 2 * RG_MAXP;
    StripeFifo fin{}, fout{};
    if (kStripes) {
        constexpr int QW = FIFO_WORDS + 2;
        if (lane < nwv) { fifo_lds[lane * QW + FIFO_WORDS] = 0; fifo_lds[lane * QW + FIFO_WORDS + 1] = 0; }
        fin = StripeFifo{fifo_lds + wv * QW, (unsigned*)(fifo_lds + wv * QW + FIFO_WORDS), (unsigned*)(fifo_lds + wv * QW + FIFO_WORDS + 1), 0u};
        const int nx = wv + 1 < nwv ? wv + 1 : 0;
        fout = StripeFifo{fifo_lds + nx * QW, (unsigned*)(fifo_lds + nx * QW + FIFO_WORDS), (unsigned*)(fifo_lds + nx * QW + FIFO_WORDS + 1), 0u};
    }
    __syncthreads();

    // rolling rows of this stripe: [P][wpadw], stripes of a read back to back
    Rows rows{a.roll + ((long long)rd * nwv + wv) * P * wpadw};
    // per-column constants of this lane
    unsigned long long erp[(C + 15) / 16] = {};   // 4 bits per column: read base facing column c (forward read[c]; reverse read[n-c+1])
    int GP[kUni ? 1 : C];         // prefix sums of the read-gap cost up to column c (general matrices only)
    const int gcost = sct[GAP];
    int thr[C];                   // emission threshold per column (INT32_MAX = never)
    // recombination band (pathwise_alignment_recombination.rs:805-808)
    const int oob = max((int)((float)(n + 1) * (1.0f - a.rbw) / 2.0f), 1);
    {
        int run = 0;
        int gpl[C];
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            int code = 4;
            if (c >= 1 && c < ncols) code = rev ? read[n - c + 1] : read[c];
            erp[q / 16] |= (unsigned long long)code << (4 * (q % 16));
            run += (c >= 1 && c < ncols) ? sct[code * 6 + GAP] : 0;
            gpl[q] = run;
        }
        static_assert(!kStripes || kUni, "striped long reads: uniform read-gap cost only");
        // gap cost of the columns before this lane (stripes: uniform cost, columns 1 .. min(lane * C, ncols) - 1)
        const int pre = kStripes ? max(0, min(lane * C, ncols) - 1) * gcost : dpp_incl_sum(run) - run;
        if (kUni) GP[0] = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            gpl[q] += pre;
            if (!kUni) GP[kUni ? 0 : q] = gpl[q];
        }
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            const int j = rev ? n - c : c;
            thr[q] = INT32_MAX;
            if (c < ncols && j >= oob && j < n + 1 - oob) {
                if (a.thr) thr[q] = a.thr[(long long)rd * wpad + j];
                else if (a.lb) thr[q] = a.lb[rd] + a.brc - (n - j) * a.maxmatch;   // w[.][j] <= (n - j) * maxmatch
            }
        }
        // start rows: row 0 (forward) / row L-1 (reverse) is the gap-only row, identical for every path
        for (int k = 0; k < P; ++k) {
#pragma unroll
            for (int q = 0; q < C; ++q) rows.st(k, q * WAVE + wl, wpadw, (lane * C + q) < ncols ? gpl[q] : NEG);
        }
    }
    // PATH RETIREMENT (round 5: the i32 form of k_sweep16's; one wave per read, P <= 64, global modes).  A path
    // whose every future cell is provably below every threshold is hopeless — forward: max_j (A[j] + mmx (n - j)) < lb;
    // reverse: A[j] + mmx j < min_{j' <= j} (thr[j'] + mmx j') for every j — and stops being computed unless it still
    // leads a group with a needed member (lead tables of the plain step table).  The per-column constants wait in LDS.
    // STRIPES (stripes of <= 16 columns per lane): every stripe sees its own columns only, and the stripes of a read must
    // skip exactly the same records (their FIFOs carry one entry per row update).  So a decision is taken from what ALL
    // stripes published — at evaluation point e every stripe writes, per needed path, the maximum of (row + constant) over
    // its columns into rt_val[e][stripe][path] — and it is applied two points later, by which time the slowest stripe has
    // published (a stripe runs at most 64 row updates behind its left neighbour: the FIFO's capacity; the counters below make
    // that a wait, not an assumption).  Hopeless stays hopeless, so a late decision is still a correct one; the closure uses
    // the lead table of the point it is applied at.
    constexpr bool kRetOK = !kStripes || C <= 16;     // (k_sweep<32, true, true> keeps the control flow it had: see below)
#ifdef RG_NO_KOLD
    constexpr bool kOld = false;            // (experiment: the build that returned wrong sink values at 32 columns per lane, for the hazard scan)
#else
    constexpr bool kOld = kStripes && !kRetOK;
#endif
    constexpr int RT_SLOTS = 4, RT_LAG = 2;
    int* rt_base = fifo_lds + 8 * (FIFO_WORDS + 2);                 // stripes only: [pub 4 | app 4 | smin 8 | val 4 x 8 x 64]
    unsigned* rt_pub = (unsigned*)rt_base;
    unsigned* rt_app = rt_pub + RT_SLOTS;
    int* rt_smin = rt_base + 2 * RT_SLOTS;
    int* rt_val = rt_smin + 8;
    int* rvc = kStripes ? rt_val + RT_SLOTS * 8 * WAVE + wv * C * WAVE : g_lds + 64 + 2 * RG_MAXP;   // [C][64] of this wave
    int next_eval = INT32_MAX;
    unsigned long long needed = P >= 64 ? ~0ull : ((1ull << P) - 1ull);
    unsigned long long hop_all = 0;      // (stripes) every path found hopeless so far
    if (kRetOK && a.retire && !a.semi && P <= 64 && (rev ? a.thr != nullptr && a.rlead != nullptr : a.lb != nullptr && a.flead != nullptr)) {
        const int mmx = max(a.maxmatch, 0);
        if (kStripes && wl < 2 * RT_SLOTS && wv == 0) rt_pub[wl] = 0u;
        if (!rev) {
            const int lbv = a.lb[rd];
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const int c = lane * C + q;
                rvc[q * WAVE + wl] = c < ncols ? mmx * (n - c) - lbv : INT32_MIN / 2;
            }
        } else {
            int tq[C], ltot = INT32_MAX;
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const int c = lane * C + q;
                tq[q] = thr[q] == INT32_MAX ? INT32_MAX : thr[q] + mmx * (n - c);      // (j = n - c)
                ltot = min(ltot, tq[q]);
            }
            int suf = ltot;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) { const int o = __shfl_down(suf, d, WAVE); if (wl + d < WAVE) suf = min(suf, o); }
            int run = __shfl_down(suf, 1, WAVE);
            if (wl == WAVE - 1) run = INT32_MAX;
            if (kStripes) {
                // ... and the columns of the stripes to the right
                if (wl == 0) rt_smin[wv] = suf;
                __syncthreads();
                int right = INT32_MAX;
                for (int w2 = wv + 1; w2 < nwv; ++w2) right = min(right, rt_smin[w2]);
                run = min(run, right);
            }
#pragma unroll
            for (int q = C - 1; q >= 0; --q) {
                run = min(run, tq[q]);
                const int c = lane * C + q;
                rvc[q * WAVE + wl] = (c < ncols && run != INT32_MAX) ? mmx * (n - c) - run : INT32_MIN / 2;
            }
        }
        next_eval = 1 << a.retire_shift;
    }
    __syncthreads();

    int colmax[C], colarg[C];
#pragma unroll
    for (int q = 0; q < C; ++q) { colmax[q] = NEG; colarg[q] = 0; }
    unsigned ncand = 0;
    unsigned long long cells = 0;
    Cand* cand = a.cand ? a.cand + (long long)rd * a.cand_cap : nullptr;
    uint32_t* dirs = a.dirs ? a.dirs + (long long)rd * a.dirs_stride : nullptr;
    const bool track = a.track_best;

    // per-row epilogue: best member per column -> column maxima, candidate emission (search modes only)
    auto row_end = [&](int i, int knm, const int (&bkey)[C]) {
        unsigned emask = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            // argmax over ALL P entries where non-members hold 0: usable only if the winner is a member
            const int bv = bkey[q] >> 8, bk = bkey[q] & 255;
            const bool valid = bkey[q] != INT32_MIN && (knm < 0 || bv > 0 || (bv == 0 && bk > knm));
            if (valid) {
                if (bv > colmax[q]) { colmax[q] = bv; colarg[q] = (i << 8) | bk; }
                if (bv >= thr[q]) emask |= 1u << q;
            }
        }
        if (cand && __any(emask != 0)) {
            const int cnt = __popc(emask);
            const int incl = dpp_incl_sum(cnt);
            const int total = __shfl(incl, WAVE - 1, WAVE);
            if (kStripes) {
                // the stripes of a read append to one list: positions from the read's global counter (zeroed by the
                // driver; k_search's result does not depend on the order of the candidates)
                unsigned base = 0;
                if (wl == 0) base = atomicAdd(&a.ncand_out[rd], (unsigned)total);
                ncand = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            }
            unsigned pos = ncand + (unsigned)(incl - cnt);
#pragma unroll
            for (int q = 0; q < C; ++q) {
                if ((emask >> q) & 1) {
                    if (pos < a.cand_cap) {
                        const int c = lane * C + q;
                        Cand cd;
                        cd.row = i; cd.col = rev ? n - c : c; cd.val = bkey[q] >> 8; cd.path = bkey[q] & 255;
                        cand[pos] = cd;
                    }
                    ++pos;
                }
            }
            ncand += (unsigned)total;
        }
    };
    const int dww = WAVE * (C <= 16 ? 1 : 2);           // direction words of one stripe per (row, group) slot
    auto store_dirs = [&](int slot, unsigned dmask, unsigned lmask) {
        // 2 bits per column: 1 = D, 2 = U, 3 = L
        static_assert(C <= 16 || C == 32, "direction packing");
        const int lane = wv * dww + wl;                 // (shadows the global lane: word index inside the slot)
        if (C <= 16) {
            uint32_t wv = 0;
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const uint32_t dcode = (lmask >> q) & 1 ? 3u : ((dmask >> q) & 1 ? 1u : 2u);
                wv |= dcode << (2 * q);
            }
            dirs[(long long)slot * a.dir_words + lane] = wv;
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t wv = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int qq = h * 16 + q;
                    const uint32_t dcode = (lmask >> qq) & 1 ? 3u : ((dmask >> qq) & 1 ? 1u : 2u);
                    wv |= dcode << (2 * q);
                }
                dirs[(long long)slot * a.dir_words + h * WAVE + lane] = wv;
            }
        }
    };

    // The sweep walks a host-built step table (one 16-byte record per (row, edge group), in sweep order) instead
    // of chasing goff[] -> groups[] -> lnz[] -> knm[] with dependent uniform loads: a wave fetches 64 records with
    // one coalesced load (next batch in flight) and broadcasts record t with v_readlane.
    //   x: row (20 bits) | base code (3) << 20 | flags (3) << 23 | group alpha bit (6) << 26
    //   y: direction-word slot (20 bits) | (knm + 1) (9) << 20 | 64-path page (2) << 29 | continuation (1) << 31
    //   z, w: member mask of the page
    const int4* steps = rev ? a.rsteps : a.fsteps;
    const int nsteps = rev ? a.nrsteps : a.nfsteps;
    int4 recs = make_int4(0, 0, 0, 0), recs_next = make_int4(0, 0, 0, 0);
    if (wl < nsteps) recs = steps[wl];
    if (WAVE + wl < nsteps) recs_next = steps[WAVE + wl];
    int t = 0;
    // fetch record t (uniform) -> SGPRs; advances the double buffer at batch boundaries
    auto fetch = [&](int tt, int& w0, int& w1, unsigned long long& gmask) {
        const int idx = tt & (WAVE - 1);
        if (idx == 0 && tt > 0) {
            recs = recs_next;
            const int nb = tt + WAVE + wl;
            recs_next = nb < nsteps ? steps[nb] : make_int4(0, 0, 0, 0);
        }
        w0 = __builtin_amdgcn_readlane(recs.x, idx);
        w1 = __builtin_amdgcn_readlane(recs.y, idx);
        gmask = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(recs.w, idx) << 32) |
                (unsigned)__builtin_amdgcn_readlane(recs.z, idx);
    };
    constexpr int F_FIRST = 1, F_LAST = 2;

    // semiglobal end-row selection (forward sweep): the lane that owns column n folds every member value
    //   * per path: first row (>= 1) with the largest last-column value  (ending_node, recombination.rs:885-897)
    //   * overall: per row the lowest path id among the row maxima, across rows the first strictly larger row
    //     (best_ending_node, pathwise_alignment_semiglobal.rs:244-277; -m 9 seed :789-800 which also counts row 0)
    const bool semi_end = a.semi && !rev;
    const int ln_end = n / C, ql_end = n % C;
    int* endv = sct + 64;               // [RG_MAXP]
    int* endr = sct + 64 + RG_MAXP;     // [RG_MAXP]
    if (semi_end) for (int k = lane; k < RG_MAXP; k += WAVE) { endv[k] = INT32_MIN; endr[k] = 0; }
    __syncthreads();
    int gbest_val = INT32_MIN, gbest_row = 0, gbest_path = 0, rowkey = INT32_MIN;
    auto end_fold = [&](int k, int i, const int (&row)[C]) {
        int v = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) if (q == ql_end) v = row[q];
        if (lane == ln_end) {
            if (v > endv[k]) { endv[k] = v; endr[k] = i; }
            rowkey = max(rowkey, v * 256 + (255 - k));
        }
    };
    auto end_row_done = [&](int i) {
        if (lane == ln_end && rowkey != INT32_MIN) {
            const int rv = rowkey >> 8, rk = 255 - (rowkey & 255);
            if (rv > gbest_val) { gbest_val = rv; gbest_row = i; gbest_path = rk; }
        }
        rowkey = INT32_MIN;
    };

    int s[C];
    int bkey[C];
    unsigned dmask = 0, lmask = 0;       // directions of the current group's alpha (live across continuation entries)
    int src = 0;
    StripeIO io{NEG, NEG, NEG, NEG};
    unsigned long long performed = 0;
    int srow = -1;                       // the row `s` / bkey were set up for (a row's first record may be skipped)
    bool row_has = false;                // some group of the current row was computed: its epilogue is due at the row's last record
    auto retire_eval = [&](int e) {
        unsigned long long hop = 0;
        for (unsigned long long todo = needed; todo; todo &= todo - 1) {
            const int k = __builtin_ctzll(todo);
            int m = INT32_MIN;
#pragma unroll
            for (int q = 0; q < C; ++q) m = max(m, rows.ld(k, q * WAVE + wl, wpadw) + rvc[q * WAVE + wl]);
            if (__builtin_amdgcn_readlane(dpp_incl_max(m, INT32_MIN), WAVE - 1) < 0) hop |= 1ull << k;
        }
        const unsigned long long lead_k = wl < P ? (rev ? a.rlead : a.flead)[(long long)e * 64 + wl] : 0ull;
        unsigned long long nd = needed & ~hop;
        for (;;) {
            const unsigned long long ad = __ballot(((needed >> wl) & 1ull) && !((nd >> wl) & 1ull) && (lead_k & nd) != 0ull);
            if (!ad) break;
            nd |= ad;
        }
        needed = nd;
    };
    auto rt_wait = [&](unsigned* cnt, unsigned want) {
        if (wl == 0) while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
    };
    auto retire_eval_striped = [&](int e) {
        // (1) apply what every stripe published at point e - RT_LAG
        const int eo = e - RT_LAG;
        if (eo >= 1) {
            const int so = (eo - 1) % RT_SLOTS, go = (eo - 1) / RT_SLOTS;
            rt_wait(rt_pub + so, (unsigned)(nwv * (go + 1)));
            int v = INT32_MIN;
            for (int w2 = 0; w2 < nwv; ++w2) v = max(v, ((volatile int*)rt_val)[(so * 8 + w2) * WAVE + wl]);
            // (entries of paths that were not needed at that point are stale: `needed` only ever shrinks)
            hop_all |= __ballot(v < 0) & needed;
            if (wl == 0) __hip_atomic_fetch_add(rt_app + so, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long lead_k = wl < P ? (rev ? a.rlead : a.flead)[(long long)e * 64 + wl] : 0ull;
            unsigned long long nd = needed & ~hop_all;
            for (;;) {
                const unsigned long long ad = __ballot(((needed >> wl) & 1ull) && !((nd >> wl) & 1ull) && (lead_k & nd) != 0ull);
                if (!ad) break;
                nd |= ad;
            }
            needed = nd;
        }
        // (2) publish this point: the slot is free once every stripe has applied its previous use
        const int sl = (e - 1) % RT_SLOTS, gen = (e - 1) / RT_SLOTS;
        rt_wait(rt_app + sl, (unsigned)(nwv * gen));
        int mine = INT32_MIN;
        for (unsigned long long todo = needed; todo; todo &= todo - 1) {
            const int k = __builtin_ctzll(todo);
            int m = INT32_MIN;
#pragma unroll
            for (int q = 0; q < C; ++q) m = max(m, rows.ld(k, q * WAVE + wl, wpadw) + rvc[q * WAVE + wl]);
            const int mk = __builtin_amdgcn_readlane(dpp_incl_max(m, INT32_MIN), WAVE - 1);
            if (wl == k) mine = mk;
        }
        ((volatile int*)rt_val)[(sl * 8 + wv) * WAVE + wl] = mine;
        if (wl == 0) __hip_atomic_fetch_add(rt_pub + sl, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    while (t < nsteps) {
        int w0, w1;
        unsigned long long gmask;
        if (kRetOK && t >= next_eval) [[unlikely]] {
            if (kStripes) retire_eval_striped(t >> a.retire_shift); else retire_eval(t >> a.retire_shift);
            next_eval = (t | ((1 << a.retire_shift) - 1)) + 1;
        }
        fetch(t, w0, w1, gmask);
        int i = w0 & 0xfffff;
        int li = (w0 >> 20) & 7;
        int flags = (w0 >> 23) & 7;
        if (flags == 4) flags = 7;                      // head of a run (see the step table in rg_path_driver.hip): an ordinary one-group row here
        int slot = w1 & 0xfffff;
        const int kbase = ((w1 >> 29) & 3) * 64;        // first path id of the entry's 64-path page
        const bool cont = w1 < 0;                       // continuation entry of a group that spans pages: members only
        // (an inner row of a one-entry segment run: the alpha is the lowest member, the field holds the run length left)
        int ga = kbase + ((flags & 4) ? __builtin_ctzll(gmask | (1ull << 63)) : ((w0 >> 26) & 63));
        const int nm = __popcll(gmask);
        cells += (unsigned long long)nm;
        // (path retirement: the members still computed; a group whose members are all retired — its alpha with them — is skipped)
        const unsigned long long gm = (!kOld && next_eval != INT32_MAX) ? (gmask & needed) : gmask;
        // (kOld, stripes of 32 columns per lane: the instantiation keeps the exact control flow it had before the retirement was added — with the
        // skip, the `i != srow` set-up and `row_has` compiled in, k_sweep<32, true, true> (691 spilled VGPRs, 325 spilled SGPRs)
        // returned sink values 12 too low while every direction word stayed right: tests/test_gpu_pathwise.py::
        // test_reads_longer_than_2047_bases at stripe_c = 32; the narrower stripes were unaffected)
        if (kOld || gm != 0ull) {
        // ---- general (row, group) step ----
        const int g_i = __builtin_amdgcn_readfirstlane(sct[li * 6 + GAP]);
        const int g0 = a.semi ? 0 : g_i;
        if (kOld ? (flags & F_FIRST) != 0 : ((flags & F_FIRST) || i != srow)) {
#pragma unroll
            for (int q = 0; q < C; ++q) { s[q] = sct[li * 6 + (int)((erp[q / 16] >> (4 * (q % 16))) & 7)]; bkey[q] = INT32_MIN; }
            srow = i;
        }
        row_has = true;
        {
            unsigned long long rest = cont ? gm : gm & ~(1ull << (ga - kbase));
            performed += (unsigned long long)__popcll(gm);
            // loads: alpha row, and the first member's row in flight while the alpha recurrence runs
            int nxt[C];
            int knext = -1;
            if (rest) {
                knext = kbase + __builtin_ctzll(rest);
                rest &= rest - 1;
#pragma unroll
                for (int q = 0; q < C; ++q) nxt[q] = rows.ld(knext, q * WAVE + wl, wpadw);
            }
            if (!cont) {        // (a continuation entry keeps dmask / lmask / src of the entry that ran the group's alpha)
                int rowa[C];
#pragma unroll
                for (int q = 0; q < C; ++q) rowa[q] = rows.ld(ga, q * WAVE + wl, wpadw);
                if (kStripes) { io.in_prev = NEG; io.in_carry = NEG; if (wv > 0) fin.pop2(io.in_prev, io.in_carry, wl); }
                RowOps<C, kUni, kStripes>::alpha(rowa, s, GP, gcost, g_i, g0, lane, ncols, dmask, lmask, src, wl, &io);
                if (kStripes && wv + 1 < nwv) fout.push2(io.out_prev, io.out_carry, wl);
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    rows.st(ga, q * WAVE + wl, wpadw, rowa[q]);
                    if (track && (lane * C + q) < ncols) bkey[q] = max(bkey[q], rowa[q] * 256 + ga);
                }
                if (semi_end) end_fold(ga, i, rowa);
                if (dirs) store_dirs(slot, dmask, lmask);
            }
            // other members follow the alpha's directions; the next member's row is always in flight
            while (knext >= 0) {
                const int k = knext;
                int cur[C];
#pragma unroll
                for (int q = 0; q < C; ++q) cur[q] = nxt[q];
                if (rest) {
                    knext = kbase + __builtin_ctzll(rest);
                    rest &= rest - 1;
#pragma unroll
                    for (int q = 0; q < C; ++q) nxt[q] = rows.ld(knext, q * WAVE + wl, wpadw);
                } else knext = -1;
                if (kStripes) { io.in_prev = NEG; io.in_carry = NEG; if (wv > 0) fin.pop2(io.in_prev, io.in_carry, wl); }
                RowOps<C, kUni, kStripes>::member(cur, s, GP, gcost, g_i, g0, lane, ncols, dmask, lmask, src, wl, &io);
                if (kStripes && wv + 1 < nwv) fout.push2(io.out_prev, io.out_carry, wl);
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    rows.st(k, q * WAVE + wl, wpadw, cur[q]);
                    if (track && (lane * C + q) < ncols) bkey[q] = max(bkey[q], cur[q] * 256 + k);
                }
                if (semi_end) end_fold(k, i, cur);
            }
            // no barrier: every lane only ever re-reads the row words it wrote itself
        }
        }
        if (semi_end && (flags & F_LAST)) end_row_done(i);
        if (track && (flags & F_LAST) && (kOld || row_has)) row_end(i, ((w1 >> 20) & 511) - 1, bkey);
        if (flags & F_LAST) row_has = false;
        ++t;
    }

    // ---- outputs ----
    if (a.colmax_out) {
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            if (c < ncols) {
                a.colmax_out[(long long)rd * wpad + (rev ? n - c : c)] = colmax[q];
                if (a.colarg_out) a.colarg_out[(long long)rd * wpad + (rev ? n - c : c)] = colarg[q];
            }
        }
    }
    if (!kStripes && a.ncand_out && lane == 0) a.ncand_out[rd] = ncand;     // (stripes: the counter was advanced atomically)
    __syncthreads();
    if (!rev && !a.semi) {
        // value of every path at its sink row, column n (lane/slot that owns column n)
        const int cn = n, ql = cn % C, ln = cn / C;         // ln: global lane that owns column n
        // (a retired path's row is stale: its true final score is below the bound k_verify checks the result against)
        if (wv == ln / WAVE)
            for (int k = wl; k < P; k += WAVE)
                rs->sink_val[k] = (!kOld && next_eval != INT32_MAX && !((needed >> k) & 1ull)) ? NEG : rows.ld(k, ql * WAVE + (ln % WAVE), wpadw);
    }
    if (semi_end && wv == ln_end / WAVE) {       // the stripe that owns column n folded every value
        for (int k = wl; k < P; k += WAVE) { rs->sink_val[k] = endv[k]; rs->path_end_row[k] = endr[k]; }
        if (lane == ln_end) { rs->s0 = gbest_val; rs->end_row_best = gbest_row; rs->seed_path = gbest_path; }
    }
    if (lane == 0 && a.count_cells) {
        atomicAdd(a.cells, cells * (unsigned long long)(n + 1));
        atomicAdd(a.cells + 1, (kOld ? cells : performed) * (unsigned long long)(n + 1));       // (every member update that is not retired is performed as such here)
    }
}

// ---------------------------------------------------------------------------------
// Exact global alignment score of the read against path 0.  Path 0 has the lowest id, so it is the alpha of
// every group it belongs to: its layer is a plain NW recurrence and A[sink][n][0] <= S0 (the seed of the
// recombination search).  One wave per read, rows of path 0 only (1/P of a sweep).
template <int C>
__global__ __launch_bounds__(64) void k_opt0(Opt0Args a) {
    const int rd = blockIdx.x;
    const int lane = threadIdx.x;
    const PathGraphDev& g = a.g;
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    if (a.bad[rd] || n + 1 > C * WAVE) { if (lane == 0) a.lb[rd] = INT32_MIN / 2; return; }
    const uint8_t* read = a.reads + ro - 1;
    const int ncols = n + 1, GAP = 5;
    __shared__ int sct[36];
    if (lane < 36) sct[lane] = a.sc.t[lane];
    __syncthreads();
    int er[C], GP[C], row[C];
    {
        int run = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            int code = 4;
            if (c >= 1 && c < ncols) code = read[c];
            er[q] = code;
            run += (c >= 1 && c < ncols) ? sct[code * 6 + GAP] : 0;
            GP[q] = run;
        }
        const int pre = dpp_incl_sum(run) - run;
#pragma unroll
        for (int q = 0; q < C; ++q) { GP[q] += pre; row[q] = lane * C + q < ncols ? GP[q] : NEG; }
    }
    const int pk = a.pick ? a.pick[rd] : 0;      // (only path 0's score is a provable bound: see PickArgs)
    // (two-path pick: p1's rows <= X, then p2's rows > X)
    const int pk2 = (a.pick && a.pick2) ? a.pick2[2 * rd] : -1;
    const int X = pk2 >= 0 ? a.pick2[2 * rd + 1] : INT32_MAX;
    int beg = a.fpoff[pk], cnt = a.fpoff[pk + 1] - beg;
    int semibest = NEG;
    bool second = false;
    for (int t = 0; t < cnt; ++t) {
        const int i = a.fprow[beg + t];
        if (i > X && !second) {
            // switch lists: first row of p2 above X (its list is ascending)
            second = true;
            beg = a.fpoff[pk2]; cnt = a.fpoff[pk2 + 1] - beg;
            int lo = 0, hi = cnt;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.fprow[beg + mid] <= X) lo = mid + 1; else hi = mid; }
            t = lo - 1;
            continue;
        }
        const int li = g.lnz[i];
        const int g_i = sct[li * 6 + GAP];
        int prev_old = dpp_shr1(row[C - 1], NEG);
        int runmax = NEG;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            const int old = row[q];
            const int d = prev_old + sct[li * 6 + er[q]], u = old + ((a.semi && c == 0) ? 0 : g_i);
            int du = (c != 0 && d >= u) ? d : u;
            du = c < ncols ? du : NEG;
            prev_old = old;
            row[q] = du - GP[q];
            runmax = max(runmax, row[q]);
        }
        int run = dpp_shr1(dpp_incl_max(runmax, NEG), NEG);
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            const int x = row[q];
            const int y = max(run, x);
            row[q] = c < ncols ? y + GP[q] : NEG;
            run = y;
        }
        if (a.semi) {   // free end row: best last-column value over the rows of path 0
            int v = NEG;
#pragma unroll
            for (int q = 0; q < C; ++q) if (q == n % C) v = row[q];
            semibest = max(semibest, v);
        }
    }
    const int ln = n / C, ql = n % C;
    int v = NEG;
#pragma unroll
    for (int q = 0; q < C; ++q) if (q == ql) v = row[q];
    if (lane == ln) a.lb[rd] = (a.semi ? semibest : v) - (a.pick ? a.margin : 0) - (pk2 >= 0 ? a.rec_pen : 0);
}

// The same for reads longer than 2047 bases: a.nwv waves per read, wave w owns columns [w * 64 * C, (w + 1) * 64 * C) and runs
// the path's rows on its stripe a few steps behind wave w - 1 (StripeFifo / StripeIO as in k_sweep<C, true, true>: per row
// the old value of the left stripe's last column and the running z-maximum).  Uniform read-gap cost (checked by the
// driver for every striped batch).  This is what opens the two-sweep pipeline — and the speculative bound — for long
// reads, which ran three i32 sweeps until round 4.
template <int C>
__global__ __launch_bounds__(512) void k_opt0_striped(Opt0Args a) {
    const int rd = blockIdx.x;
    const int wl = (int)(threadIdx.x & (WAVE - 1)), wv = (int)(threadIdx.x >> 6), nwv = a.nwv;
    const int lane = wv * WAVE + wl;
    const PathGraphDev& g = a.g;
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    if (a.bad[rd] || n + 1 > nwv * C * WAVE) { if (threadIdx.x == 0) a.lb[rd] = INT32_MIN / 2; return; }
    const uint8_t* read = a.reads + ro - 1;
    const int ncols = n + 1, GAP = 5;
    __shared__ int sct[36];
    __shared__ int fifo_lds[8 * (FIFO_WORDS + 2)];
    if (lane < 36) sct[lane] = a.sc.t[lane];
    constexpr int QW = FIFO_WORDS + 2;
    if (lane < nwv) { fifo_lds[lane * QW + FIFO_WORDS] = 0; fifo_lds[lane * QW + FIFO_WORDS + 1] = 0; }
    StripeFifo fin{fifo_lds + wv * QW, (unsigned*)(fifo_lds + wv * QW + FIFO_WORDS), (unsigned*)(fifo_lds + wv * QW + FIFO_WORDS + 1), 0u};
    const int nx = wv + 1 < nwv ? wv + 1 : 0;
    StripeFifo fout{fifo_lds + nx * QW, (unsigned*)(fifo_lds + nx * QW + FIFO_WORDS), (unsigned*)(fifo_lds + nx * QW + FIFO_WORDS + 1), 0u};
    __syncthreads();
    const int gcost = sct[GAP];
    int er[C], row[C], s[C];
    const int GP[1] = {0};
#pragma unroll
    for (int q = 0; q < C; ++q) {
        const int c = lane * C + q;
        er[q] = (c >= 1 && c < ncols) ? read[c] : 4;
        row[q] = c < ncols ? c * gcost : NEG;               // the gap-only start row
    }
    const int pk = a.pick ? a.pick[rd] : 0;      // (only path 0's score is a provable bound: see PickArgs)
    // (two-path pick: p1's rows <= X, then p2's rows > X — every wave of the block takes the same rows)
    const int pk2 = (a.pick && a.pick2) ? a.pick2[2 * rd] : -1;
    const int X = pk2 >= 0 ? a.pick2[2 * rd + 1] : INT32_MAX;
    int beg = a.fpoff[pk], cnt = a.fpoff[pk + 1] - beg;
    int semibest = NEG;
    bool second = false;
    StripeIO io{NEG, NEG, NEG, NEG};
    for (int t = 0; t < cnt; ++t) {
        const int i = a.fprow[beg + t];
        if (i > X && !second) {
            second = true;
            beg = a.fpoff[pk2]; cnt = a.fpoff[pk2 + 1] - beg;
            int lo = 0, hi = cnt;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.fprow[beg + mid] <= X) lo = mid + 1; else hi = mid; }
            t = lo - 1;
            continue;
        }
        const int li = g.lnz[i];
        const int g_i = sct[li * 6 + GAP];
#pragma unroll
        for (int q = 0; q < C; ++q) s[q] = sct[li * 6 + er[q]];
        io.in_prev = NEG; io.in_carry = NEG;
        if (wv > 0) fin.pop2(io.in_prev, io.in_carry, wl);
        unsigned dmask, lmask;
        int src;
        RowOps<C, true, true>::alpha(row, s, GP, gcost, g_i, a.semi ? 0 : g_i, lane, ncols, dmask, lmask, src, wl, &io);
        if (wv + 1 < nwv) fout.push2(io.out_prev, io.out_carry, wl);
        if (a.semi) {   // free end row: best last-column value over the rows of the path
            int v = NEG;
#pragma unroll
            for (int q = 0; q < C; ++q) if (q == n % C) v = row[q];
            semibest = max(semibest, v);
        }
    }
    int v = NEG;
#pragma unroll
    for (int q = 0; q < C; ++q) if (q == n % C) v = row[q];
    if (lane == n / C) a.lb[rd] = (a.semi ? semibest : v) - (a.pick ? a.margin : 0) - (pk2 >= 0 ? a.rec_pen : 0);
}

// ---------------------------------------------------------------------------------
// Path vote for the speculative bound (PickArgs): one wave per read samples up to 256 12-mers of the read, looks each up
// in the table of the paths' 12-mers and gives one vote to every path that contains it; lane b counts the votes of path b.
__global__ __launch_bounds__(64) void k_pick(PickArgs a) {
    const int rd = blockIdx.x;
    const int lane = threadIdx.x;
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    // (up to 256 paths since round 6: NW words of path bits per table entry, lane b counts the votes of paths b, 64 + b, ...)
    __shared__ unsigned long long smask[256 * RG_PW];
    constexpr int K = 12;
    const int nw = (a.P + 63) >> 6;
    const int npos = n - K + 1;
    if (a.bad[rd] || npos < 1) { if (lane == 0) { a.pick[rd] = 0; if (a.pick2) { a.pick2[2 * rd] = -1; a.pick2[2 * rd + 1] = 0; } } return; }
    const int step = (npos + 255) / 256;
    const int nsamp = (npos + step - 1) / step;             // <= 256
    for (int t = lane; t < 256; t += WAVE) {
        unsigned long long m[RG_PW] = {0, 0, 0, 0};
        if (t < nsamp) {
            const uint8_t* p = a.reads + ro + (long long)t * step;
            unsigned key = 0;
            bool ok = true;
            for (int e = 0; e < K; ++e) { const unsigned c = p[e]; ok = ok && c < 4; key = (key << 2) | (c & 3u); }
            if (ok) {
                unsigned h = (key * 2654435761u) >> 8;
                for (int probe = 0; probe < 64; ++probe) {
                    const unsigned slot = (h + probe) & a.table_mask;
                    const unsigned kk = a.keys[slot];
                    if (kk == key) {
                        for (int w = 0; w < nw; ++w) m[w] = a.masks[(long long)slot * nw + w];
                        break;
                    }
                    if (kk == 0xffffffffu) break;
                }
            }
        }
#pragma unroll
        for (int w = 0; w < RG_PW; ++w) smask[t * RG_PW + w] = m[w];
    }
    __syncthreads();
    int votes[RG_PW];
    int key = -1;
#pragma unroll
    for (int w = 0; w < RG_PW; ++w) {
        votes[w] = 0;
        if (w < nw) for (int t = 0; t < nsamp; ++t) votes[w] += (int)((smask[t * RG_PW + w] >> lane) & 1ull);
        if (w * 64 + lane >= a.P) votes[w] = -1;
        if (votes[w] >= 0) key = max(key, (votes[w] << 8) | (255 - (w * 64 + lane)));      // most votes, lowest path id on ties
    }
    for (int d = WAVE / 2; d >= 1; d >>= 1) key = max(key, __shfl_xor(key, d, WAVE));
    const int best1 = key >> 8;
    const int p_one = best1 > 0 ? 255 - (key & 255) : 0;
    if (lane == 0) a.pick[rd] = p_one;
    if (!a.pick2) return;
    // two-path pick: split t in [1, nsamp): pre = this path's votes among samples [0, t)
    int pre[RG_PW] = {0, 0, 0, 0};
    int bestsum = -1, bt = 0, bp1 = 0, bp2 = 0;
    for (int t = 1; t < nsamp; ++t) {
        int k1 = -1, k2 = -1;
#pragma unroll
        for (int w = 0; w < RG_PW; ++w) {
            if (w < nw) pre[w] += (int)((smask[(t - 1) * RG_PW + w] >> lane) & 1ull);
            if (votes[w] >= 0) {
                k1 = max(k1, (pre[w] << 8) | (255 - (w * 64 + lane)));
                k2 = max(k2, ((votes[w] - pre[w]) << 8) | (255 - (w * 64 + lane)));
            }
        }
        k1 = __builtin_amdgcn_readlane(dpp_incl_max(k1, INT32_MIN), WAVE - 1);
        k2 = __builtin_amdgcn_readlane(dpp_incl_max(k2, INT32_MIN), WAVE - 1);
        const int sum = (k1 >> 8) + (k2 >> 8);
        if (sum > bestsum) { bestsum = sum; bt = t; bp1 = 255 - (k1 & 255); bp2 = 255 - (k2 & 255); }
    }
    int p2 = -1, X = 0;
    // worth it when the two paths explain clearly more samples than one (each sample spans K bases: a switch costs the
    // one-path alignment a mismatch or so per differing allele, i.e. per few samples)
    if (bp1 != bp2 && bestsum >= best1 + 6) {
        // a row of p1 that p2 visits too, at or behind the split (list index ~ read column: the walks are global)
        const int b1 = a.fpoff[bp1], c1 = a.fpoff[bp1 + 1] - b1, b2 = a.fpoff[bp2], c2 = a.fpoff[bp2 + 1] - b2;
        const int col = bt * step + K / 2;
        const int i0 = min(max((int)((long long)col * c1 / max(n, 1)), 0), max(c1 - 1, 0));
        unsigned long long found = 0;
        int myrow = -1;
        for (int base = i0; base < c1 && !found; base += WAVE) {
            myrow = -1;
            if (base + lane < c1) {
                myrow = a.fprow[b1 + base + lane];
                int lo = 0, hi = c2;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.fprow[b2 + mid] < myrow) lo = mid + 1; else hi = mid; }
                if (!(lo < c2 && a.fprow[b2 + lo] == myrow)) myrow = -1;
            }
            found = __ballot(myrow >= 0);
            if (base - i0 >= 3 * WAVE) break;
        }
        if (found) { X = __shfl(myrow, __builtin_ctzll(found), WAVE); p2 = bp2; if (lane == 0) a.pick[rd] = bp1; }
    }
    if (lane == 0) { a.pick2[2 * rd] = p2; a.pick2[2 * rd + 1] = X; }
}

// After k_search: the speculation held iff the maximum found is >= the bound the forward sweep pruned with — and, when the
// sweeps stored direction words for the picked paths only (SweepArgs::dsel_pick), iff the paths the traceback will walk are
// among the picks.  A read that fails either way is aligned again (rg_path_driver.hip); k_layer / k_trace skip it.
__global__ __launch_bounds__(256) void k_verify(ReadState* st, const int* lb, unsigned* nretry, uint8_t* flags, int nreads,
                                                const int* dsel_pick, const int* dsel_pick2, int dsel_lo, int dsel_hi) {
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= nreads) return;
    ReadState* rs = st + rd;
    uint8_t f = 0;
    if (!(rs->status & (ST_BAD_BASE | ST_WOULD_PANIC | ST_OVERFLOW))) {
        bool bad = rs->fscore < (float)lb[rd];
        if (dsel_pick) {
            const int p1 = dsel_pick[rd], p2 = dsel_pick2 ? dsel_pick2[2 * rd] : -1;
            const int fp = rs->fwd_path, rp = rs->rev_path;
            // k_layer rebuilds the forward layer over the rows of fp up to fen (the whole path without a recombination) and the
            // reverse layer over the rows of rp from its end back to rsn: every word it reads must have been stored
            const bool recomb = fp != rp;
            const bool fok = fp == p1 || fp == p2 || (recomb && rs->fen < dsel_lo);
            const bool rok = !recomb || rp == p1 || rp == p2 || rs->rsn > dsel_hi;
            bad = bad || !fok || !rok;
        }
        if (bad) {
            rs->status |= ST_RETRY;
            f = 1;
            atomicAdd(nretry, 1u);
        }
    }
    flags[rd] = f;
}

// -m 4 on a speculative bound (round 6): the sweep retired the paths whose final score could not reach lb[rd] and stored direction
// words for the picked path only; that was exact iff the best final score reaches the bound (every retired path then lies strictly
// below the best) and the best path is the picked one.  A read that fails is aligned again without the speculation.
__global__ __launch_bounds__(256) void k_verify4(ReadState* st, const int* lb, unsigned* nretry, uint8_t* flags, int nreads, const int* dsel_pick) {
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= nreads) return;
    ReadState* rs = st + rd;
    uint8_t f = 0;
    if (!(rs->status & (ST_BAD_BASE | ST_WOULD_PANIC | ST_OVERFLOW))) {
        const bool bad = rs->s0 < lb[rd] || (dsel_pick && rs->fwd_path != dsel_pick[rd]);
        if (bad) {
            rs->status |= ST_RETRY;
            f = 1;
            atomicAdd(nretry, 1u);
        }
    }
    flags[rd] = f;
}

// bases of the reads to align again, compacted (sub_off: their offsets in `out`)
__global__ __launch_bounds__(256) void k_gather_reads(const uint8_t* reads, const long long* off, const int* idx, const long long* sub_off,
                                                      uint8_t* out) {
    const int t = blockIdx.x;
    const long long src = off[idx[t]], len = off[idx[t] + 1] - src, dst = sub_off[t];
    for (long long i = threadIdx.x; i < len; i += blockDim.x) out[dst + i] = reads[src + i];
}
__global__ __launch_bounds__(256) void k_scatter_results(const int* idx, const DevRecord* sub_rec, const uint8_t* sub_ops, DevRecord* rec,
                                                         uint8_t* ops, long long ops_stride) {
    const int t = blockIdx.x;
    const int rd = idx[t];
    if (threadIdx.x == 0) rec[rd] = sub_rec[t];
    const int nops = sub_rec[t].n_ops;
    for (int i = threadIdx.x; i < nops; i += blockDim.x) ops[(long long)rd * ops_stride + i] = sub_ops[(long long)t * ops_stride + i];
}

// After the forward sweep: seed / best path (pathwise_alignment.rs:305-325,
// pathwise_alignment_recombination.rs:775-803).  One thread per read.
__global__ void k_seed(SeedArgs a) {
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= a.nreads) return;
    ReadState* rs = a.state + rd;
    if (rs->status & (ST_BAD_BASE | ST_WOULD_PANIC)) return;
    const PathGraphDev& g = a.g;
    const int L = g.L, P = g.P;
    if (a.mode == RG_MODE_PATHWISE_SEMI) {
        // best_ending_node (pathwise_alignment_semiglobal.rs:244-277), folded by the sweep
        rs->bound = rs->s0; rs->end_row = rs->end_row_best; rs->fwd_path = rs->seed_path; rs->rev_path = rs->seed_path;
        return;
    }
    if (a.mode == RG_MODE_RECOMBINATION_SEMI) {
        // seed over rows 0..L-2 (pathwise_alignment_recombination.rs:789-800): row 0 is the all-gap row of every path,
        // it wins ties (lowest row) with path 0 (lowest id)
        const long long ro = a.read_off[rd];
        const int n = (int)(a.read_off[rd + 1] - ro);
        int gapsum = 0;
        for (int j = 0; j < n; ++j) gapsum += a.sc.t[a.reads[ro + j] * 6 + 5];
        if (rs->s0 <= gapsum) { rs->s0 = gapsum; rs->seed_path = 0; }
        const int bp = rs->seed_path;
        rs->bound = rs->s0; rs->fwd_path = bp; rs->rev_path = bp;
        rs->end_row = rs->path_end_row[bp];       // ending_node(.., best_path, ..) :885-897 (rows >= 1 only)
        return;
    }
    if (a.mode == RG_MODE_PATHWISE) {
        // results[k] = A[sink(k)][n][k] for paths registered at F, 0 otherwise; max of (score, path id)
        int best = 0, bp = 0, bend = 0;
        bool first = true;
        for (int k = 0; k < P; ++k) {
            int v = 0, end = 0;
            for (int e = g.eoff[L - 1]; e < g.eoff[L]; ++e)
                if ((g.emask[(long long)e * RG_PW + (k >> 6)] >> (k & 63)) & 1) { v = rs->sink_val[k]; end = g.epred[e]; }
            if (first || v >= best) { best = v; bp = k; bend = end; first = false; }
        }
        rs->s0 = best; rs->bound = best; rs->seed_path = bp; rs->end_row = bend; rs->fwd_path = bp; rs->rev_path = bp;
    } else {
        // strict '<' over F's predecessors (stored order) then paths ascending: lowest id on ties
        bool have = false; int mx = 0, bp = 0;
        for (int e = g.eoff[L - 1]; e < g.eoff[L]; ++e)
            for (int k = 0; k < P; ++k)
                if ((g.emask[(long long)e * RG_PW + (k >> 6)] >> (k & 63)) & 1) {
                    const int v = rs->sink_val[k];
                    if (!have || mx < v) { mx = v; bp = k; have = true; }
                }
        int end = 0;
        for (int e = g.eoff[L - 1]; e < g.eoff[L]; ++e) if ((g.emask[(long long)e * RG_PW + (bp >> 6)] >> (bp & 63)) & 1) end = g.epred[e];
        rs->s0 = mx; rs->bound = mx; rs->seed_path = bp; rs->end_row = end; rs->fwd_path = bp; rs->rev_path = bp;
        if (!have) rs->status |= ST_WOULD_PANIC;
    }
}

// thr[j] = S0 + R - colmax[j]: a cell can be part of a winning / tying recombination only if
// value + (best partner of the column) - R >= S0 (f32 rounding is monotone)
__global__ void k_threshold(ThrArgs a) {
    const int rd = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.wpad) return;
    const ReadState* rs = a.state + rd;
    const long long o = (long long)rd * a.wpad + j;
    const int cm = a.colmax[o];
    int base = a.use_bound ? rs->bound : rs->s0;
    if (a.lb) base = max(base, a.lb[rd]);
    a.thr[o] = cm <= NEG ? INT32_MAX : base + a.brc - cm;
}

// Lower bound of the search maximum: the exact score of the pair (best forward cell, best reverse cell)
// of every column, when that pair is admissible.  Any real candidate's score is a valid bound; pairs
// whose integer part (m + w - R) is below it can neither win nor tie.
__global__ __launch_bounds__(64) void k_bound(BoundArgs a) {
    const int rd = blockIdx.x;
    const int lane = threadIdx.x;
    ReadState* rs = a.state + rd;
    if (rs->status & (ST_BAD_BASE | ST_WOULD_PANIC)) return;
    const PathGraphDev& g = a.g;
    const int n = (int)(a.read_off[rd + 1] - a.read_off[rd]);
    const int oob = max((int)((float)(n + 1) * (1.0f - a.rbw) / 2.0f), 1);
    float best = (float)rs->s0;
    for (int j = oob + lane; j < n + 1 - oob; j += WAVE) {
        const long long o = (long long)rd * a.wpad + j;
        const int m = a.mf[o], w = a.wr[o];
        if (m <= NEG || w <= NEG) continue;
        const int fa = a.mfarg[o], ra = a.wrarg[o];
        const int fi = fa >> 8, fk = fa & 255, ri = ra >> 8, rk = ra & 255;
        if (fk == rk || g.node_id[fi] == g.node_id[ri]) continue;
        // the packed sweep records column maxima without the "winner must be a member path" rule: re-check here
        const int kf = g.knm[fi], kr = g.knm[ri];
        if ((kf >= 0 && ((m << 16) | fk) <= kf) || (kr >= 0 && ((w << 16) | rk) <= kr)) continue;
        const int disp = abs(g.dfs[fi] - g.dfs[ri]) + abs(g.dfe[fi] - g.dfe[ri]);
        const float sc = __fsub_rn((float)(m + w), __fadd_rn((float)a.brc, __fmul_rn(a.mrc, (float)disp)));
        best = fmaxf(best, sc);
    }
    for (int d = WAVE / 2; d >= 1; d >>= 1) best = fmaxf(best, __shfl_xor(best, d, WAVE));
    if (lane == 0) rs->bound = max(rs->s0, (int)ceilf(best));
}

// ---------------------------------------------------------------------------------
// Recombination search over the candidate lists (pathwise_alignment_recombination.rs:804-864).
// The sequential scan of the reference, lexicographic in (column, forward row, reverse row), keeps
// "the first candidate with the maximum value for which `cond` holds, else the first candidate with
// that value; the no-recombination seed is the earliest, cond = false" (SURVEY A.5 item 5): a total
// order, so the scan is evaluated as a parallel reduction with key (score, cond, -position).
struct SearchKey {
    float score;
    int cond;
    long long order;  // smaller = earlier in the reference's scan; seed = -1
    int fi, ri;       // candidate indices
};
__device__ __forceinline__ bool key_better(const SearchKey& a, const SearchKey& b) {
    if (a.score != b.score) return a.score > b.score;
    if (a.cond != b.cond) return a.cond > b.cond;
    return a.order < b.order;
}

__global__ __launch_bounds__(64) void k_search(SearchArgs a) {
    const int rd = blockIdx.x;
    const int lane = threadIdx.x;
    ReadState* rs = a.state + rd;
    if (rs->status & (ST_BAD_BASE | ST_WOULD_PANIC)) return;
    const PathGraphDev& g = a.g;
    const long long L = g.L;
    const Cand* fc = a.fcand + (long long)rd * a.fcap;
    const Cand* rc = a.rcand + (long long)rd * a.rcap;
    const unsigned nf = a.nf[rd], nr = a.nr[rd];
    if (nf > a.fcap || nr > a.rcap) {
        if (lane == 0) rs->status |= ST_OVERFLOW;
        return;
    }
    const int wpad = a.wpad;
    // bucket the reverse candidates by column (counting sort through LDS histogram + HBM index list)
    extern __shared__ int sh[];
    int* hist = sh;            // wpad + 1
    for (int j = lane; j <= wpad; j += WAVE) hist[j] = 0;
    __syncthreads();
    for (unsigned t = lane; t < nr; t += WAVE) atomicAdd(&hist[rc[t].col + 1], 1);
    __syncthreads();
    // exclusive scan of hist (serial per wave chunk; wpad <= 2048)
    {
        int carry = 0;
        for (int base = 0; base <= wpad; base += WAVE) {
            const int j = base + lane;
            const int v = j <= wpad ? hist[j] : 0;
            const int inc = wave_incl_sum(v, lane) + carry;
            if (j <= wpad) hist[j] = inc;
            carry = __shfl(inc, WAVE - 1, WAVE);
        }
    }
    __syncthreads();
    // hist[j] = number of reverse candidates with col < j  => bucket of column j is [hist[j], hist[j+1])
    unsigned* ridx = a.ridx + (long long)rd * a.rcap;
    int* fill = sh + (wpad + 1);  // wpad + 1 cursors
    for (int j = lane; j <= wpad; j += WAVE) fill[j] = 0;
    __syncthreads();
    for (unsigned t = lane; t < nr; t += WAVE) {
        const int c = rc[t].col;
        const int p = atomicAdd(&fill[c], 1);
        ridx[hist[c] + p] = t;
    }
    __syncthreads();
    __threadfence_block();

    const float brc_f = (float)a.brc;
    const int bound = rs->bound;
    SearchKey best;
    best.score = (float)rs->s0; best.cond = 0; best.order = -1; best.fi = -1; best.ri = -1;
    // lanes take forward candidates; each scans the reverse bucket of its column
    for (unsigned base = 0; base < nf; base += WAVE) {
        const unsigned t = base + lane;
        if (t < nf) {
            const Cand f = fc[t];
            // forward candidates may come from the loose lower-bound threshold: keep only those that can reach
            // the final bound with the best reverse partner of their column
            if (f.val + a.wr[(long long)rd * wpad + f.col] - a.brc < bound) continue;
            const int b0 = hist[f.col], b1 = hist[f.col + 1];
            const unsigned long long idf = g.node_id[f.row];
            const int cond_f = g.seglast[f.row];
            for (int u = b0; u < b1; ++u) {
                const unsigned ri = ridx[u];
                const Cand r = rc[ri];
                // integer bound first: (m + w) - R >= bound is necessary to win or tie
                if (f.val + r.val - a.brc < bound) continue;
                if (g.node_id[r.row] == idf) continue;
                if (f.path == r.path) continue;
                const int disp = abs(g.dfs[f.row] - g.dfs[r.row]) + abs(g.dfe[f.row] - g.dfe[r.row]);
                // three separately rounded f32 operations, no contraction (:840-842)
                const float prod = __fmul_rn(a.mrc, (float)disp);
                const float penalty = __fadd_rn(brc_f, prod);
                const float sc = __fsub_rn((float)(f.val + r.val), penalty);
                SearchKey k;
                k.score = sc;
                k.cond = cond_f && g.segfirst[r.row];
                k.order = ((long long)f.col * L + f.row) * L + r.row;
                k.fi = (int)t; k.ri = (int)ri;
                if (key_better(k, best)) best = k;
            }
        }
    }
    // wave reduction of the key
    for (int d = WAVE / 2; d >= 1; d >>= 1) {
        SearchKey o;
        o.score = __shfl_xor(best.score, d, WAVE);
        o.cond = __shfl_xor(best.cond, d, WAVE);
        o.order = __shfl_xor(best.order, d, WAVE);
        o.fi = __shfl_xor(best.fi, d, WAVE);
        o.ri = __shfl_xor(best.ri, d, WAVE);
        if (key_better(o, best)) best = o;
    }
    if (lane == 0) {
        if (best.fi < 0) {
            rs->fwd_path = rs->seed_path; rs->rev_path = rs->seed_path; rs->fen = 0; rs->rsn = 0; rs->rec_col = 0;
            rs->fscore = (float)rs->s0; rs->displacement = 0;
        } else {
            const Cand f = fc[best.fi], r = rc[best.ri];
            rs->fwd_path = f.path; rs->rev_path = r.path; rs->fen = f.row; rs->rsn = r.row; rs->rec_col = f.col;
            rs->fscore = best.score;
            rs->displacement = abs(g.dfs[f.row] - g.dfs[r.row]) + abs(g.dfe[f.row] - g.dfe[r.row]);
        }
    }
}

// ---------------------------------------------------------------------------------
// Rebuild the absolute layer of one path from the direction words (rows of that path only) and record, per cell, the
// move the reference's traceback takes there.  The traceback re-derives d, u, l from the chosen path's own layer
// (pathwise_alignment_output.rs:32-110, recombination_output.rs:391-470, 659-736): d = A[t-1][j-1] + s(row, read[j]),
// u = A[t-1][j] + s(row, '-'), l = A[t][j-1] + s('-', read[j]), D if max == d, else U if max == u, else L.  All three
// are in registers while the row is rebuilt, so the kernel stores 2 bits per cell (tdir[t][lane] words, 1 = D, 2 = U,
// 3 = L) instead of the layer itself: 16x fewer bytes, and k_trace reads one word per step instead of three values.
template <int C, bool kStripes = false>
__global__ __launch_bounds__(kStripes ? 512 : 64) void k_layer(LayerArgs a) {
    const int rd = blockIdx.x;
    const int wl = kStripes ? (int)(threadIdx.x & (WAVE - 1)) : (int)threadIdx.x;      // lane inside the wave
    const int wv = kStripes ? (int)(threadIdx.x >> 6) : 0;                             // stripe (striped long reads, see k_sweep)
    const int nwv = kStripes ? a.nwv : 1;
    const int lane = wv * WAVE + wl;                                                   // global lane: column = lane * C + q
    const int dww = WAVE * (C <= 16 ? 1 : 2);                                          // words of one stripe per row
    const PathGraphDev& g = a.g;
    ReadState* rs = a.state + rd;
    if (rs->status & (ST_BAD_BASE | ST_WOULD_PANIC | ST_OVERFLOW | ST_RETRY)) return;      // (ST_RETRY: aligned again; its direction words may not exist)
    const bool rev = a.rev;
    const int path = rev ? rs->rev_path : rs->fwd_path;
    const bool recomb = rs->fwd_path != rs->rev_path;
    if (rev && !recomb) return;  // no recombination: reverse layer not needed
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    const uint8_t* read = a.reads + ro - 1;
    const int ncols = rev ? n : n + 1;
    const int GAP = 5;
    __shared__ int sct[36];
    __shared__ int fifo_lds[kStripes ? 8 * (FIFO_WORDS + 2) : 1];
    if (lane < 36) sct[lane] = a.sc.t[lane];
    StripeFifo fin{}, fout{};
    if (kStripes) {
        constexpr int QW = FIFO_WORDS + 2;
        if (lane < nwv) { fifo_lds[lane * QW + FIFO_WORDS] = 0; fifo_lds[lane * QW + FIFO_WORDS + 1] = 0; }
        fin = StripeFifo{fifo_lds + wv * QW, (unsigned*)(fifo_lds + wv * QW + FIFO_WORDS), (unsigned*)(fifo_lds + wv * QW + FIFO_WORDS + 1), 0u};
        const int nx = wv + 1 < nwv ? wv + 1 : 0;
        fout = StripeFifo{fifo_lds + nx * QW, (unsigned*)(fifo_lds + nx * QW + FIFO_WORDS), (unsigned*)(fifo_lds + nx * QW + FIFO_WORDS + 1), 0u};
    }
    __syncthreads();
    int er[C], GP[C];
    bool act[C];
    {
        int run = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            act[q] = c < ncols;
            int code = 4;
            if (c >= 1 && c < ncols) code = rev ? read[n - c + 1] : read[c];
            er[q] = code;
            run += (c >= 1 && c < ncols) ? sct[code * 6 + GAP] : 0;
            GP[q] = run;
        }
        // gap cost of the columns before this lane (stripes run with a uniform read-gap cost: columns 1 .. min(c0, ncols) - 1)
        const int pre = kStripes ? max(0, min(lane * C, ncols) - 1) * sct[GAP] : wave_incl_sum(run, lane) - run;
#pragma unroll
        for (int q = 0; q < C; ++q) GP[q] += pre;
    }
    int gl[C];                     // s('-', read base of the column): row-invariant
#pragma unroll
    for (int q = 0; q < C; ++q) gl[q] = sct[GAP * 6 + er[q]];
    uint32_t* tdir = reinterpret_cast<uint32_t*>(a.layer) + (long long)rd * a.layer_stride;
    int cur[C];
#pragma unroll
    for (int q = 0; q < C; ++q) cur[q] = act[q] ? GP[q] : NEG;
    const int* prow = rev ? a.rprow : a.fprow;
    const int* pslot = rev ? a.rpslot : a.fpslot;
    const int* poff = rev ? a.rpoff : a.fpoff;
    const uint32_t* dirs = a.dirs + (long long)rd * a.dirs_stride;
    const int nrows = poff[path + 1] - poff[path];
    // forward layer: the score the record reports is the layer value where the walk starts
    const int start_row = recomb ? rs->fen : rs->end_row;
    const int start_col = recomb ? rs->rec_col : n;
    // the row list, base codes and direction words of the next PF rows are fetched ahead: three dependent loads per
    // row (row/slot -> base code, direction word) would otherwise serialise at full memory latency
    constexpr int PF = 4;
    int pf_li[PF], pf_row[PF];
    uint32_t pf_w0[PF], pf_w1[PF];
    auto prefetch = [&](int tt, int& li_o, int& row_o, uint32_t& w0_o, uint32_t& w1_o) {
        li_o = 4; row_o = -1; w0_o = 0; w1_o = 0;
        if (tt < nrows) {
            const int ii = prow[poff[path] + tt];
            const int sl = pslot[poff[path] + tt];
            row_o = ii;
            li_o = g.lnz[ii];
            w0_o = dirs[(long long)sl * a.dir_words + wv * dww + wl];
            if (C > 16) w1_o = dirs[(long long)sl * a.dir_words + wv * dww + WAVE + wl];
        }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) prefetch(k, pf_li[k], pf_row[k], pf_w0[k], pf_w1[k]);
    for (int t = 0; t < nrows; ++t) {
        const int li = pf_li[0], irow = pf_row[0];
        const uint32_t word0 = pf_w0[0], word1 = pf_w1[0];
#pragma unroll
        for (int k = 0; k + 1 < PF; ++k) { pf_li[k] = pf_li[k + 1]; pf_row[k] = pf_row[k + 1]; pf_w0[k] = pf_w0[k + 1]; pf_w1[k] = pf_w1[k + 1]; }
        prefetch(t + PF, pf_li[PF - 1], pf_row[PF - 1], pf_w0[PF - 1], pf_w1[PF - 1]);
        const int g_i = sct[li * 6 + GAP];
        unsigned dmask = 0, lmask = 0;
        if (a.dir_fmt == 1) {
            if (C <= 16) { unsigned u16, l16; dir16_decode<C / 2>(word0, u16, l16); dmask = ~u16 & 0xffffu; lmask = l16; }
            else { dmask = ~word0; lmask = word1; }
        } else if (C <= 16) {
            const uint32_t wv = word0;
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const uint32_t dc = (wv >> (2 * q)) & 3u;
                if (dc == 1u) dmask |= 1u << q;
                if (dc == 3u) lmask |= 1u << q;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t wv = h == 0 ? word0 : word1;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t dc = (wv >> (2 * q)) & 3u;
                    if (dc == 1u) dmask |= 1u << (h * 16 + q);
                    if (dc == 3u) lmask |= 1u << (h * 16 + q);
                }
            }
        }
        unsigned actmask = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) if (act[q]) actmask |= 1u << q;
        const bool any_nonl = ((~lmask) & actmask) != 0;
        const int src = dpp_shr1(dpp_incl_max(any_nonl ? wl : -1, -1), kStripes ? -1 : 0);
        int pk = dpp_shr1(cur[C - 1], NEG);
        int in_carry = NEG, in_new = NEG;
        const int out_prev = kStripes ? __builtin_amdgcn_readlane(cur[C - 1], WAVE - 1) : 0;   // old last column of this stripe
        if (kStripes && wv > 0) {
            int in_prev;
            fin.pop2(in_prev, in_carry, wl);
            if (wl == 0) pk = in_prev;
        }
        int old[C];
        int y[C];
        int sq[C];
        int last = NEG;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            old[q] = cur[q];
            sq[q] = sct[li * 6 + er[q]];
            const int om1 = q == 0 ? pk : cur[q - 1];
            const int base = ((dmask >> q) & 1) ? om1 + sq[q] : cur[q] + ((a.semi && q == 0 && lane == 0) ? 0 : g_i);
            y[q] = base - GP[q];
            if (!((lmask >> q) & 1) && act[q]) last = y[q];
        }
        int cin = __shfl(last, kStripes ? max(src, 0) : src, WAVE);
        if (kStripes && src < 0) cin = in_carry;            // the last non-L column lies in a stripe further left
        int run = cin;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            if ((lmask >> q) & 1) y[q] = run; else run = y[q];
            cur[q] = act[q] ? y[q] + GP[q] : NEG;
        }
        if (kStripes) {
            if (wv + 1 < nwv) {
                fout.push2(out_prev, __builtin_amdgcn_readlane(run, WAVE - 1), wl);
                fout.push2(__builtin_amdgcn_readlane(cur[C - 1], WAVE - 1), 0, wl);      // new last column: the `l` source of the next stripe
            }
            if (wv > 0) { int pad; fin.pop2(in_new, pad, wl); }
        }
        // ---- traceback decisions of this row ----
        // the reverse matrix keeps its start row (row L-1) delta-encoded in the reference (absolute_scores skips it):
        // path 0 reads its absolute value there, every other path reads 0 (pathwise_alignment_recombination.rs:748)
        const bool zero_prev = rev && t == 0 && path != 0;
        int nk = dpp_shr1(cur[C - 1], NEG);                // new value of column c-1 for the lane's first column
        if (kStripes && wv > 0 && wl == 0) nk = in_new;
        uint32_t tw[C <= 16 ? 1 : 2] = {};
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int o1 = zero_prev ? 0 : (q == 0 ? pk : old[q - 1]);
            const int o0 = zero_prev ? 0 : old[q];
            const int d = o1 + sq[q];
            const int u = o0 + g_i;
            const int l = (q == 0 ? nk : cur[q - 1]) + gl[q];   // key ('-', seq[j])
            const int mx = max(max(d, u), l);
            const uint32_t code = mx == d ? 1u : (mx == u ? 2u : 3u);
            tw[q / 16] |= code << (2 * (q % 16));
        }
        tdir[(long long)(t + 1) * a.dir_words + wv * dww + wl] = tw[0];
        if (C > 16) tdir[(long long)(t + 1) * a.dir_words + wv * dww + WAVE + wl] = tw[C <= 16 ? 0 : 1];
        if (!rev && irow == start_row) {
            int v = 0;
#pragma unroll
            for (int q = 0; q < C; ++q) if (q == start_col % C) v = cur[q];
            if (lane == start_col / C) rs->trace_score = v;
        }
        // the walkers only ever move to earlier rows of the list: nothing behind their start row is read (every wave of a
        // striped read leaves at the same row, its FIFO traffic for that row done)
        if (recomb && irow == (rev ? rs->rsn : rs->fen)) break;
    }
}

// ---------------------------------------------------------------------------------
// Traceback over the decisions recorded by k_layer.  One lane per read walks; ops only (rows are re-derived on the
// host).
template <int C>
__global__ __launch_bounds__(64) void k_trace(TraceArgs a) {
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= a.nreads) return;
    ReadState* rs = a.state + rd;
    DevRecord* rec = a.rec + rd;
    if (rs->status & (ST_BAD_BASE | ST_WOULD_PANIC | ST_OVERFLOW | ST_RETRY)) {
        rec->status = rs->status & ~ST_RETRY; rec->n_ops = 0; rec->n_fwd_ops = 0; rec->score = 0;      // (ST_RETRY: the second pass writes the record)
        return;
    }
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    uint8_t* ops = a.ops + (long long)rd * a.ops_stride;
    int nops = 0;
    const bool recomb = (a.mode == RG_MODE_RECOMBINATION || a.mode == RG_MODE_RECOMBINATION_SEMI) && rs->fwd_path != rs->rev_path;
    const int dww = WAVE * (C <= 16 ? 1 : 2);          // words of one stripe per layer row
    const int dw = dww * a.nwv;                          // (a.nwv > 1: striped long reads, see k_sweep)
    // decision of layer row idx at (mirrored) column c
    auto move = [&](const uint32_t* td, int idx, int c) -> uint32_t {
        const int q = c % C, gl = c / C;                 // global lane that owns the column
        return (td[(long long)idx * dw + (gl / WAVE) * dww + (q / 16) * WAVE + gl % WAVE] >> (2 * (q % 16))) & 3u;
    };
    const uint32_t* fl = reinterpret_cast<const uint32_t*>(a.flayer) + (long long)rd * a.layer_stride;
    // ---- forward walk from (row, j) back to the source on path fp ----
    const int fp = rs->fwd_path;
    const int fbeg = a.fpoff[fp], fnr = a.fpoff[fp + 1] - fbeg;
    int start_row = recomb ? rs->fen : rs->end_row;
    int j = recomb ? rs->rec_col : n;
    // index of start_row among the rows of path fp (binary search; rows ascending)
    int lo = 0, hi = fnr - 1, idx = -1;
    while (lo <= hi) { int mid = (lo + hi) >> 1; int r = a.fprow[fbeg + mid]; if (r == start_row) { idx = mid; break; } if (r < start_row) lo = mid + 1; else hi = mid - 1; }
    if (idx < 0) { rec->status = ST_WOULD_PANIC; rec->n_ops = 0; return; }
    int t = idx + 1;  // layer index of the current row (0 = row 0)
    const int score = rs->trace_score;
    while (t > 0 && j > 0) {
        const uint32_t mv = move(fl, t, j);
        if (mv == 1u) { ops[nops++] = OP_D; t -= 1; j -= 1; }
        else if (mv == 2u) { ops[nops++] = OP_U; t -= 1; }
        else { ops[nops++] = OP_L; j -= 1; }
    }
    while (j > 0) { ops[nops++] = OP_L; j -= 1; }
    while (!a.semi && t > 0) { ops[nops++] = OP_U; t -= 1; }   // semiglobal: the alignment may start inside the graph
    const int nfwd = nops;
    if (recomb) {
        // ---- reverse walk from (rsn, rec_col) forward to the sink on path rp (mirrored columns c' = n - jj) ----
        const int rp = rs->rev_path;
        const uint32_t* rl = reinterpret_cast<const uint32_t*>(a.rlayer) + (long long)rd * a.layer_stride;
        const int rbeg = a.rpoff[rp], rnr = a.rpoff[rp + 1] - rbeg;
        // rows of the reverse program are descending
        int lo2 = 0, hi2 = rnr - 1, ridx = -1;
        while (lo2 <= hi2) { int mid = (lo2 + hi2) >> 1; int r = a.rprow[rbeg + mid]; if (r == rs->rsn) { ridx = mid; break; } if (r > rs->rsn) lo2 = mid + 1; else hi2 = mid - 1; }
        if (ridx < 0) { rec->status = ST_WOULD_PANIC; rec->n_ops = 0; return; }
        int tt = ridx + 1;          // layer index (0 = row L-1)
        int jj = rs->rec_col;       // real column
        while (tt > 0 && jj < n) {
            const uint32_t mv = move(rl, tt, n - jj);
            if (mv == 1u) { ops[nops++] = OP_D; tt -= 1; jj += 1; }
            else if (mv == 2u) { ops[nops++] = OP_U; tt -= 1; }
            else { ops[nops++] = OP_L; jj += 1; }
        }
        while (jj < n) { ops[nops++] = OP_L | OP_CONT; jj += 1; }
        while (!a.semi && tt > 0) { ops[nops++] = OP_U | OP_CONT; tt -= 1; }
    }
    rec->status = rs->status;
    rec->score = score;
    rec->fscore = rs->fscore;
    rec->end_row = start_row;
    rec->end_col = n;
    rec->stop_row = 0; rec->stop_col = 0;
    rec->best_path = fp;
    rec->rev_path = recomb ? rs->rev_path : fp;
    rec->fen = rs->fen; rec->rsn = rs->rsn; rec->rec_col = rs->rec_col; rec->displacement = rs->displacement;
    rec->n_ops = nops; rec->n_fwd_ops = nfwd;
}

// ---------------------------------------------------------------------------------
// Largest candidate-list / record-list length any read of the chunk asked for: the driver reads these four words back
// once per chunk (with the cell-update counter) instead of copying the per-read counters in the middle of the pipeline.
__global__ __launch_bounds__(256) void k_need(const ReadState* st, const unsigned* nf, const unsigned* nr, const unsigned* nrec,
                                              const unsigned* nrrec, unsigned* need, int nreads) {
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned v[4] = {0, 0, 0, 0};
    if (rd < nreads && !(st[rd].status & (ST_BAD_BASE | ST_WOULD_PANIC))) {
        v[0] = nf[rd]; v[1] = nr[rd];
        if (nrec) v[2] = nrec[rd];
        if (nrrec) v[3] = nrrec[rd];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        for (int d = WAVE / 2; d >= 1; d >>= 1) v[e] = max(v[e], (unsigned)__shfl_xor((int)v[e], d, WAVE));
        if ((threadIdx.x & (WAVE - 1)) == 0 && v[e]) atomicMax(&need[e], v[e]);
    }
}
void launch_need(const ReadState* st, const unsigned* nf, const unsigned* nr, const unsigned* nrec, const unsigned* nrrec,
                 unsigned* need, int nreads, hipStream_t s) {
    hipLaunchKernelGGL(k_need, dim3((nreads + 255) / 256), dim3(256), 0, s, st, nf, nr, nrec, nrrec, need, nreads);
}

// ---------------------------------------------------------------------------------
// launchers
template <int C>
static void launch_sweep_c(const SweepArgs& a, int nreads, hipStream_t s) {
    // uniform read-gap cost: every (base, '-') entry equal (reads hold ACGTN only)
    bool uni = true;
    for (int b = 1; b < 5; ++b) uni = uni && a.sc.t[b * 6 + 5] == a.sc.t[5];
    const size_t sct_bytes = (64 + 2 * RG_MAXP + C * WAVE) * sizeof(int);       // (+ the path-retirement constants)
    if (uni) hipLaunchKernelGGL((k_sweep<C, true>), dim3(nreads), dim3(64), sct_bytes, s, a);
    else hipLaunchKernelGGL((k_sweep<C, false>), dim3(nreads), dim3(64), sct_bytes, s, a);
}
void launch_sweep(const SweepArgs& a, int nreads, int C, hipStream_t s) {
    if (a.nwv > 1) {
        // striped long reads: a.nwv waves per read, C columns per lane (16: the rows and keys fit the registers; 32 spills
        // 800 of them and only serves reads beyond 8 x 1024 columns), uniform read-gap cost (checked by the driver)
        // (score table, end arrays, 8 FIFOs; path retirement across stripes: counters, suffix minima, 4 x 8 x 64 published
        // maxima, C x 64 constants per stripe)
        // (the 32-column stripes do not retire paths and get no constants: eight of them would pass the 64 KB a launch may ask for)
        const size_t bytes = (64 + 2 * RG_MAXP + 8 * (FIFO_WORDS + 2) + 16 + 4 * 8 * WAVE + (C <= 16 ? (size_t)a.nwv * C * WAVE : 0)) * sizeof(int);
        switch (C) {
            case 8: hipLaunchKernelGGL((k_sweep<8, true, true>), dim3(nreads), dim3(64 * a.nwv), bytes, s, a); break;
            case 16: hipLaunchKernelGGL((k_sweep<16, true, true>), dim3(nreads), dim3(64 * a.nwv), bytes, s, a); break;
            default: hipLaunchKernelGGL((k_sweep<32, true, true>), dim3(nreads), dim3(64 * a.nwv), bytes, s, a); break;
        }
        return;
    }
    switch (C) {
        case 4: launch_sweep_c<4>(a, nreads, s); break;
        case 8: launch_sweep_c<8>(a, nreads, s); break;
        case 16: launch_sweep_c<16>(a, nreads, s); break;
        default: launch_sweep_c<32>(a, nreads, s); break;
    }
}
void launch_seed(const SeedArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_seed, dim3((a.nreads + 63) / 64), dim3(64), 0, s, a);
}
void launch_opt0(const Opt0Args& a, int nreads, int C, hipStream_t s) {
    if (a.nwv > 1) {
        switch (C) {
            case 8: hipLaunchKernelGGL((k_opt0_striped<8>), dim3(nreads), dim3(64 * a.nwv), 0, s, a); break;
            case 16: hipLaunchKernelGGL((k_opt0_striped<16>), dim3(nreads), dim3(64 * a.nwv), 0, s, a); break;
            default: hipLaunchKernelGGL((k_opt0_striped<32>), dim3(nreads), dim3(64 * a.nwv), 0, s, a); break;
        }
        return;
    }
    switch (C) {
        case 4: hipLaunchKernelGGL((k_opt0<4>), dim3(nreads), dim3(64), 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_opt0<8>), dim3(nreads), dim3(64), 0, s, a); break;
        case 16: hipLaunchKernelGGL((k_opt0<16>), dim3(nreads), dim3(64), 0, s, a); break;
        default: hipLaunchKernelGGL((k_opt0<32>), dim3(nreads), dim3(64), 0, s, a); break;
    }
}
void launch_pick(const PickArgs& a, int nreads, hipStream_t s) { hipLaunchKernelGGL(k_pick, dim3(nreads), dim3(64), 0, s, a); }
__global__ __launch_bounds__(1024) void k_order(const int* pick, const int* pick2, int* order, int nreads) {
    __shared__ unsigned cnt[256], base[256];
    const int tid = threadIdx.x;
    if (tid < 256) cnt[tid] = 0;
    __syncthreads();
    auto key = [&](int rd) {          // bin 0 = the highest path id
        int k = pick[rd];
        if (pick2 && pick2[2 * rd] >= 0) k = max(k, pick2[2 * rd]);
        return 255 - min(max(k, 0), 255);
    };
    for (int rd = tid; rd < nreads; rd += blockDim.x) atomicAdd(&cnt[key(rd)], 1u);
    __syncthreads();
    if (tid == 0) { unsigned run = 0; for (int b = 0; b < 256; ++b) { base[b] = run; run += cnt[b]; } }
    __syncthreads();
    for (int rd = tid; rd < nreads; rd += blockDim.x) order[atomicAdd(&base[key(rd)], 1u)] = rd;
}
void launch_order(const int* pick, const int* pick2, int* order, int nreads, hipStream_t s) {
    hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, s, pick, pick2, order, nreads);
}
void launch_verify(ReadState* st, const int* lb, unsigned* nretry, uint8_t* flags, int nreads, const int* dsel_pick, const int* dsel_pick2, int dsel_lo, int dsel_hi, hipStream_t s) {
    hipLaunchKernelGGL(k_verify, dim3((nreads + 255) / 256), dim3(256), 0, s, st, lb, nretry, flags, nreads, dsel_pick, dsel_pick2, dsel_lo, dsel_hi);
}
void launch_verify4(ReadState* st, const int* lb, unsigned* nretry, uint8_t* flags, int nreads, const int* dsel_pick, hipStream_t s) {
    hipLaunchKernelGGL(k_verify4, dim3((nreads + 255) / 256), dim3(256), 0, s, st, lb, nretry, flags, nreads, dsel_pick);
}
void launch_gather_reads(const uint8_t* reads, const long long* off, const int* idx, const long long* sub_off, uint8_t* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_reads, dim3(n), dim3(256), 0, s, reads, off, idx, sub_off, out);
}
void launch_scatter_results(const int* idx, const DevRecord* sub_rec, const uint8_t* sub_ops, DevRecord* rec, uint8_t* ops, long long ops_stride,
                            int n, hipStream_t s) {
    hipLaunchKernelGGL(k_scatter_results, dim3(n), dim3(256), 0, s, idx, sub_rec, sub_ops, rec, ops, ops_stride);
}
void launch_threshold(const ThrArgs& a, int nreads, hipStream_t s) {
    hipLaunchKernelGGL(k_threshold, dim3((a.wpad + 255) / 256, nreads), dim3(256), 0, s, a);
}
void launch_bound(const BoundArgs& a, int nreads, hipStream_t s) {
    hipLaunchKernelGGL(k_bound, dim3(nreads), dim3(64), 0, s, a);
}
void launch_search(const SearchArgs& a, int nreads, hipStream_t s) {
    const size_t bytes = (size_t)(2 * (a.wpad + 1)) * sizeof(int);
    hipLaunchKernelGGL(k_search, dim3(nreads), dim3(64), bytes, s, a);
}
void launch_layer(const LayerArgs& a, int nreads, int C, hipStream_t s) {
    if (a.nwv > 1) {
        switch (C) {
            case 8: hipLaunchKernelGGL((k_layer<8, true>), dim3(nreads), dim3(64 * a.nwv), 0, s, a); break;
            case 16: hipLaunchKernelGGL((k_layer<16, true>), dim3(nreads), dim3(64 * a.nwv), 0, s, a); break;
            default: hipLaunchKernelGGL((k_layer<32, true>), dim3(nreads), dim3(64 * a.nwv), 0, s, a); break;
        }
        return;
    }
    switch (C) {
        case 4: hipLaunchKernelGGL((k_layer<4>), dim3(nreads), dim3(64), 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_layer<8>), dim3(nreads), dim3(64), 0, s, a); break;
        case 16: hipLaunchKernelGGL((k_layer<16>), dim3(nreads), dim3(64), 0, s, a); break;
        default: hipLaunchKernelGGL((k_layer<32>), dim3(nreads), dim3(64), 0, s, a); break;
    }
}
void launch_trace(const TraceArgs& a, int C, hipStream_t s) {
    const dim3 grid((a.nreads + 63) / 64), blk(64);
    switch (C) {
        case 4: hipLaunchKernelGGL((k_trace<4>), grid, blk, 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_trace<8>), grid, blk, 0, s, a); break;
        case 16: hipLaunchKernelGGL((k_trace<16>), grid, blk, 0, s, a); break;
        default: hipLaunchKernelGGL((k_trace<32>), grid, blk, 0, s, a); break;
    }
}

}  // namespace rg
