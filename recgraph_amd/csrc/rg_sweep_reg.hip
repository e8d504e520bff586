// Register-resident DP sweep for gfx950 (-m 4 / -m 8), second generation of k_sweep.
//
// The rolling rows never leave the register file: a workgroup of NW = ceil(P / 8) wavefronts handles one
// read, wave w keeps the rows of paths [8w, 8w+8) as roll[8][C] VGPRs (lane t owns columns t*C .. t*C+C-1,
// exactly as in k_sweep).  Per (row, edge group):
//   * the wave that owns the group's alpha path runs the recurrence (lane-local serial scan + one wave
//     prefix-max), updates its row and publishes the 2-bit directions + fill-forward source lane through a
//     double-buffered 512-byte LDS mailbox;
//   * one workgroup barrier;
//   * every wave applies those directions to the member paths it owns.
// Dynamic path indices select registers through an 8-way switch of fully unrolled bodies (VGPRs cannot be
// indexed by a runtime value without spilling).  In search mode the per-(row, column) best member is
// reduced across waves with LDS atomic max on a packed (value, path) key; one wave per row turns it into
// column maxima and candidate emissions.  HBM traffic: direction words and candidates only.
#include "rg_path_kernels.hpp"

namespace rg {

namespace {

constexpr int NEGR = INT32_MIN / 4;
constexpr int PW = 8;   // paths per wave

__device__ __forceinline__ int wave_excl_max_r(int v, int lane) {
    (void)lane;
    return dpp_shr1(dpp_incl_max(v, NEGR), NEGR);
}

// GPX: prefix sum of the read-gap cost up to column c.  kUni = every read base has the same gap cost
// (all matrices the reference CLI can build, score_matrix.rs:35-105): GP(c) = c * gcost needs no registers.
template <int C, bool kUni>
struct RowOps {
    static __device__ __forceinline__ int gp(const int (&GP)[kUni ? 1 : C], int gcost, int lane, int q) {
        if (kUni) return (lane * C + q) * gcost;
        return GP[kUni ? 0 : q];
    }
    // alpha recurrence on one row, in place, branch-free per column.  Returns direction masks and the
    // fill-forward source lane.
    static __device__ __forceinline__ void alpha(int (&row)[C], const int (&s)[C], const int (&GP)[kUni ? 1 : C], int gcost,
                                                 int g_i, int lane, int ncols, unsigned& dmask, unsigned& lmask, int& src) {
        int prev_old = dpp_shr1(row[C - 1], NEGR);       // old value of column c-1
        int runmax = NEGR;
        unsigned dm = 0, lm = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            const int old = row[q];
            const int d = prev_old + s[q], u = old + g_i;
            // border column 0: gap in the graph only (direction U); priority D > U elsewhere
            const bool isd = (q == 0 ? c != 0 : true) && d >= u;
            int du = isd ? d : u;
            du = c < ncols ? du : NEGR;
            dm |= (isd ? 1u : 0u) << q;
            prev_old = old;
            const int x = du - gp(GP, gcost, lane, q);     // candidate - GP
            row[q] = x;
            runmax = max(runmax, x);
        }
        int run = wave_excl_max_r(runmax, lane);
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            const int x = row[q];
            const bool isl = run > x;                      // L only when strictly better (D > U > L)
            lm |= (isl ? 1u : 0u) << q;
            const int y = max(run, x);
            row[q] = c < ncols ? y + gp(GP, gcost, lane, q) : NEGR;
            run = y;
        }
        const unsigned full = C >= 32 ? 0xffffffffu : ((1u << C) - 1u);
        const bool any_nonl = (lm & full) != full;
        src = dpp_shr1(dpp_incl_max(any_nonl ? lane : -1, -1), 0);
        dmask = dm; lmask = lm;
    }
    // member update, in place: follow the published directions with the path's own values
    static __device__ __forceinline__ void member(int (&row)[C], const int (&s)[C], const int (&GP)[kUni ? 1 : C], int gcost,
                                                  int g_i, int lane, int ncols, unsigned dmask, unsigned lmask, int src) {
        int prev_old = dpp_shr1(row[C - 1], NEGR);
        int last = NEGR;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int old = row[q];
            const int base = ((dmask >> q) & 1) ? prev_old + s[q] : old + g_i;
            prev_old = old;
            const int y = base - gp(GP, gcost, lane, q);
            row[q] = y;
            last = ((lmask >> q) & 1) ? last : y;
        }
        int cur = __shfl(last, src, WAVE);                 // y of the last non-L column before this lane
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            cur = ((lmask >> q) & 1) ? cur : row[q];
            row[q] = c < ncols ? cur + gp(GP, gcost, lane, q) : NEGR;
        }
    }
};

}  // namespace

#define RG_SWITCH8(idx, CALL)            \
    switch (idx) {                       \
        case 0: { CALL(0); } break;      \
        case 1: { CALL(1); } break;      \
        case 2: { CALL(2); } break;      \
        case 3: { CALL(3); } break;      \
        case 4: { CALL(4); } break;      \
        case 5: { CALL(5); } break;      \
        case 6: { CALL(6); } break;      \
        default: { CALL(7); } break;     \
    }

template <int C, bool kUni>
__global__ __launch_bounds__(512, 2) void k_sweep_reg(SweepArgs a) {
    const int rd = blockIdx.x;
    const int lane = threadIdx.x & 63;
    // wave-uniform values are forced into SGPRs (readfirstlane): branches on them become scalar jumps
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const PathGraphDev& g = a.g;
    const int P = g.P, L = g.L;
    const int wpad = C * WAVE;
    ReadState* rs = a.state + rd;
    const long long ro = a.read_off[rd];
    const int n = __builtin_amdgcn_readfirstlane((int)(a.read_off[rd + 1] - ro));
    if (a.bad[rd] || n + 1 > wpad) {
        if (threadIdx.x == 0 && !a.rev) rs->status = a.bad[rd] ? ST_BAD_BASE : ST_WOULD_PANIC;
        return;
    }
    const uint8_t* read = a.reads + ro - 1;
    const bool rev = a.rev;
    const int ncols = rev ? n : n + 1;
    const int GAP = 5;
    const int k0 = wave * PW;

    extern __shared__ __attribute__((aligned(16))) int smem[];
    int* sct = smem;                       // 64
    unsigned* pub = (unsigned*)(smem + 64);  // [2][64] direction masks
    int* pubsrc = smem + 64 + 128;         // [2][64]
    int* misc = smem + 64 + 256;           // [0] candidate count
    int* bestkey = smem + 64 + 256 + 64;   // [wpad]  (search mode)
    int* sthr = bestkey + wpad;            // [wpad]
    int* scolmax = sthr + wpad;            // [wpad]
    int* scolarg = scolmax + wpad;         // [wpad]
    if (threadIdx.x < 36) sct[threadIdx.x] = a.sc.t[threadIdx.x];
    if (threadIdx.x == 0) misc[0] = 0;
    const int oob = max((int)((float)(n + 1) * (1.0f - a.rbw) / 2.0f), 1);
    if (a.track_best) {
        for (int t = threadIdx.x; t < wpad; t += blockDim.x) {
            // t indexes [q][lane]
            const int q = t / WAVE, ln = t % WAVE;
            const int c = ln * C + q;
            const int j = rev ? n - c : c;
            int th = INT32_MAX;
            if (a.thr && c < ncols && j >= oob && j < n + 1 - oob) th = a.thr[(long long)rd * wpad + j];
            sthr[t] = th;
            scolmax[t] = NEGR;
            scolarg[t] = 0;
            bestkey[t] = INT32_MIN;
        }
    }
    __syncthreads();

    int GP[kUni ? 1 : C];
    const int gcost = sct[0 * 6 + GAP];   // kUni: gap cost of every read base
    unsigned long long erp = 0;   // 4 bits per column: read base facing the column
    int roll[PW][C];
    {
        int run = 0;
        int gpl[C];
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            int code = 4;
            if (c >= 1 && c < ncols) code = rev ? read[n - c + 1] : read[c];
            erp |= (unsigned long long)code << (4 * q);
            run += (c >= 1 && c < ncols) ? sct[code * 6 + GAP] : 0;
            gpl[q] = run;
        }
        const int pre = dpp_incl_sum(run) - run;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            gpl[q] += pre;
            if (!kUni) GP[kUni ? 0 : q] = gpl[q];
        }
        if (kUni) GP[0] = 0;
#pragma unroll
        for (int kk = 0; kk < PW; ++kk)
#pragma unroll
            for (int q = 0; q < C; ++q) roll[kk][q] = (lane * C + q) < ncols ? gpl[q] : NEGR;
    }

    unsigned long long cells = 0;
    Cand* cand = a.cand ? a.cand + (long long)rd * a.cand_cap : nullptr;
    const int* goff = rev ? g.rgoff : g.fgoff;
    const GroupDesc* groups = rev ? g.rgroups : g.fgroups;
    uint32_t* dirs = a.dirs ? a.dirs + (long long)rd * a.dirs_stride : nullptr;
    int parity = 0;

    for (int step = 1; step + 1 < L; ++step) {
        const int i = rev ? L - 1 - step : step;
        const int li = __builtin_amdgcn_readfirstlane((int)g.lnz[i]);
        const int g_i = __builtin_amdgcn_readfirstlane(sct[li * 6 + GAP]);
        const int gbeg = __builtin_amdgcn_readfirstlane(goff[i]), gend = __builtin_amdgcn_readfirstlane(goff[i + 1]);
        int s[C];
#pragma unroll
        for (int q = 0; q < C; ++q) s[q] = sct[li * 6 + (int)((erp >> (4 * q)) & 7)];
        unsigned rowmine = 0;   // my paths that were updated in this row
        for (int gi = gbeg; gi < gend; ++gi) {
            const GroupDesc gdv = groups[gi];
            GroupDesc gd;
            gd.pred = 0; gd.pad = 0;
            gd.ga = (uint32_t)__builtin_amdgcn_readfirstlane((int)gdv.ga);
            gd.slot = __builtin_amdgcn_readfirstlane(gdv.slot);
            gd.mask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gdv.mask >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(gdv.mask & 0xffffffffull));
            const int ga = (int)gd.ga;
            const int owner = ga / PW;
            if (wave == owner) {
                unsigned dmask = 0, lmask = 0;
                int src = 0;
                const int gl = ga - k0;
#pragma unroll
                for (int kk = 0; kk < PW; ++kk)
                    if (gl == kk) RowOps<C, kUni>::alpha(roll[kk], s, GP, gcost, g_i, lane, ncols, dmask, lmask, src);
                pub[parity * WAVE + lane] = dmask | (lmask << 16);
                pubsrc[parity * WAVE + lane] = src;
                if (dirs) {
                    uint32_t wv = 0;
#pragma unroll
                    for (int q = 0; q < C; ++q) {
                        const uint32_t dcode = (lmask >> q) & 1 ? 3u : ((dmask >> q) & 1 ? 1u : 2u);
                        wv |= dcode << (2 * q);
                    }
                    dirs[(long long)gd.slot * a.dir_words + lane] = wv;
                }
                cells += (unsigned long long)__popcll(gd.mask);
            }
            __syncthreads();
            unsigned mine = (unsigned)((gd.mask >> k0) & 0xffull);
            rowmine |= mine;
            if (wave == owner) mine &= ~(1u << (ga - k0));
            if (mine) {
                const unsigned pm = pub[parity * WAVE + lane];
                const int src = pubsrc[parity * WAVE + lane];
                const unsigned dmask = pm & 0xffffu, lmask = pm >> 16;
#pragma unroll
                for (int kk = 0; kk < PW; ++kk)
                    if ((mine >> kk) & 1) RowOps<C, kUni>::member(roll[kk], s, GP, gcost, g_i, lane, ncols, dmask, lmask, src);
            }
            parity ^= 1;
        }
        if (a.track_best) {
            // best member of the row per column across all waves: packed key (value * 64 + path)
            if (rowmine) {
                int bk[C];
#pragma unroll
                for (int q = 0; q < C; ++q) bk[q] = INT32_MIN;
#pragma unroll
                for (int kk = 0; kk < PW; ++kk)
                    if ((rowmine >> kk) & 1) {
#pragma unroll
                        for (int q = 0; q < C; ++q)
                            if (lane * C + q < ncols) bk[q] = max(bk[q], roll[kk][q] * 64 + (k0 + kk));
                    }
#pragma unroll
                for (int q = 0; q < C; ++q) atomicMax(&bestkey[q * WAVE + lane], bk[q]);
            }
            __syncthreads();
            if (wave == (step % nwaves)) {
                const int knm = __builtin_amdgcn_readfirstlane(g.knm[i]);
                unsigned emask = 0;
                int ev[C], ek[C];
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    const int c = lane * C + q;
                    const int key = bestkey[q * WAVE + lane];
                    bestkey[q * WAVE + lane] = INT32_MIN;
                    const int bv = key >> 6, bkk = key & 63;   // arithmetic shift: floor division by 64
                    ev[q] = bv; ek[q] = bkk;
                    const bool valid = c < ncols && key != INT32_MIN && (knm < 0 || bv > 0 || (bv == 0 && bkk > knm));
                    if (valid) {
                        if (bv > scolmax[q * WAVE + lane]) { scolmax[q * WAVE + lane] = bv; scolarg[q * WAVE + lane] = (i << 8) | bkk; }
                        if (bv >= sthr[q * WAVE + lane]) emask |= 1u << q;
                    }
                }
                if (cand && __any(emask != 0)) {
                    const int cnt = __popc(emask);
                    const int incl = dpp_incl_sum(cnt);
                    const int total = __shfl(incl, WAVE - 1, WAVE);
                    const unsigned base = (unsigned)misc[0];
                    unsigned pos = base + (unsigned)(incl - cnt);
#pragma unroll
                    for (int q = 0; q < C; ++q) {
                        if ((emask >> q) & 1) {
                            if (pos < a.cand_cap) {
                                const int c = lane * C + q;
                                Cand cd;
                                cd.row = i; cd.col = rev ? n - c : c; cd.val = ev[q]; cd.path = ek[q];
                                cand[pos] = cd;
                            }
                            ++pos;
                        }
                    }
                    if (lane == 0) misc[0] = (int)(base + (unsigned)total);
                }
            }
        }
    }
    __syncthreads();
    // ---- outputs ----
    if (a.colmax_out && a.track_best) {
        for (int t = threadIdx.x; t < wpad; t += blockDim.x) {
            const int q = t / WAVE, ln = t % WAVE;
            const int c = ln * C + q;
            if (c < ncols) {
                a.colmax_out[(long long)rd * wpad + (rev ? n - c : c)] = scolmax[t];
                if (a.colarg_out) a.colarg_out[(long long)rd * wpad + (rev ? n - c : c)] = scolarg[t];
            }
        }
    }
    if (a.ncand_out && threadIdx.x == 0) a.ncand_out[rd] = (unsigned)misc[0];
    if (!rev) {
        // value of every path at column n: held by lane n / C, slot n % C of the owning wave
        const int ln = n / C, ql = n % C;
#pragma unroll
        for (int kk = 0; kk < PW; ++kk) {
            int v = NEGR;
#pragma unroll
            for (int q = 0; q < C; ++q) if (q == ql) v = roll[kk][q];
            if (lane == ln && k0 + kk < P) rs->sink_val[k0 + kk] = v;
        }
    }
    if (a.count_cells) {
        cells = cells * (unsigned long long)(n + 1);
        if (lane == 0 && cells) atomicAdd(a.cells, cells);
    }
}

template <bool kUni>
static void launch_sweep_reg_u(const SweepArgs& a, int nreads, int C, hipStream_t s) {
    const int nwaves = (a.g.P + PW - 1) / PW;
    const int wpad = C * WAVE;
    const size_t bytes = (size_t)(64 + 256 + 64 + (a.track_best ? 4 * wpad : 0)) * sizeof(int);
    const dim3 grid(nreads), blk(nwaves * WAVE);
    switch (C) {
        case 4: hipLaunchKernelGGL((k_sweep_reg<4, kUni>), grid, blk, bytes, s, a); break;
        case 8: hipLaunchKernelGGL((k_sweep_reg<8, kUni>), grid, blk, bytes, s, a); break;
        default: hipLaunchKernelGGL((k_sweep_reg<16, kUni>), grid, blk, bytes, s, a); break;
    }
}
void launch_sweep_reg(const SweepArgs& a, int nreads, int C, hipStream_t s) {
    // uniform read-gap cost: every (base, '-') entry equal (reads hold ACGTN only)
    bool uni = true;
    for (int b = 1; b < 5; ++b) uni = uni && a.sc.t[b * 6 + 5] == a.sc.t[5];
    if (uni) launch_sweep_reg_u<true>(a, nreads, C, s);
    else launch_sweep_reg_u<false>(a, nreads, C, s);
}

}  // namespace rg
