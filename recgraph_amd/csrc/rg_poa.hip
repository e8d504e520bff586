// Banded POA kernels for gfx950: -m 0 (global_abpoa::exec_simd semantics) and its traceback.
//
// Mapping: one wavefront (64 lanes) per read, one workgroup = one wave, so the per-row
// __syncthreads() is a single-wave barrier that only orders this wave's own global stores before
// its next loads.  Lanes own consecutive band columns; the left-gap sweep of the reference (an
// 8-step scalar loop per AVX2 chunk, global_abpoa.rs:156-165) is evaluated as a wave-level
// max-plus prefix scan.  Rows are sequential (band of row i needs best_scoring_pos of its
// predecessors, utils.rs:31-55).  f32 values of the reference are integers below 2^24, so int32
// arithmetic is exact.
//
// HBM layout per read: score cells and 32-bit path words of each row are appended to a compact
// arena (only band cells [start,right) are stored); rinfo[row] = {arena offset, start, right,
// best_scoring_pos}.  A cell that the reference never writes reads as min_score / "-1" exactly as
// in its full-width matrices (global_abpoa.rs:20-22).
#include "rg_device.hpp"
#include "rg_poa_args.hpp"

namespace rg {

__device__ __forceinline__ int sc_at(const DevScores& sc, int a, int b) { return sc.t[a * 6 + b]; }

// utils.rs:17-98 (simd_version = true), usize arithmetic of a release build
__device__ void band_simd(int i, unsigned long long ms, unsigned long long me, int r_val, unsigned long long seq_len,
                          unsigned long long bta, unsigned long long& left, unsigned long long& right) {
    (void)i;
    int tmp_bs = min((int)ms, ((int)seq_len - r_val) - (int)bta);
    unsigned long long band_start = tmp_bs < 0 ? 0ull : (unsigned long long)tmp_bs;
    unsigned long long r64 = r_val < 0 ? ~0ull : (unsigned long long)r_val;
    unsigned long long band_end;
    if (seq_len > r64) {
        unsigned long long a = me > seq_len - r64 ? me : seq_len - r64;
        band_end = min(seq_len, a + bta);
    } else {
        band_end = min(seq_len, me + bta);
    }
    unsigned long long nr = band_end, nl = band_start;
    while ((nr - nl) % 8 != 0) {
        if ((nr - nl) % 2 == 0 && nr < seq_len) nr += 1;
        else if (nl > 0) nl -= 1;
        else break;
    }
    if (nl == 0)
        while ((nr - 1) % 8 != 0 && nr < seq_len) nr += 1;
    if (nr == seq_len)
        while ((nr - nl) % 8 != 0 && nl > 1) nl -= 1;
    left = nl;
    right = nr;
}

struct M0Ctx {
    const int* am;        // this read's score arena
    const int4* rinfo;    // this read's row info
    const int* col0;
    int min_score;
};

// value of m[p][c] in the reference's full-width matrix
__device__ __forceinline__ int m_at(const M0Ctx& x, int p, int c) {
    if (c == 0 && p > 0) return x.col0[p];
    int4 ri = x.rinfo[p];
    if (c >= ri.y && c < ri.z) return x.am[ri.x + (c - ri.y)];
    return x.min_score;
}

__global__ __launch_bounds__(64) void k_m0_simd(PoaArgs a) {
    const int rd = blockIdx.x;
    const int lane = threadIdx.x;
    const DevLnz& g = a.g;
    const int L = g.L;
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    const uint8_t* read = a.reads + ro - 1;  // read[j], j = 1..n
    DevRecord* rec = a.rec + rd;
    const int W = n + 1;
    if (a.bad[rd]) {
        if (lane == 0) { rec->status = ST_BAD_BASE; rec->n_ops = 0; rec->score = 0; }
        return;
    }
    int* am = a.arena_m + (long long)rd * a.cap_cells;
    uint32_t* apw = a.arena_pw + (long long)rd * a.cap_cells;
    int4* rinfo = a.rinfo + (long long)rd * L;
    const unsigned long long bta = (unsigned long long)a.bta[rd];
    const int GAP = 5;
    M0Ctx cx{am, rinfo, a.col0, 2 * W * sc_at(a.sc, read[1], GAP)};  // global_abpoa.rs:20
    long long off = 0;
    unsigned long long ncells = 0;
    bool overflow = false;

    // ---- row 0 (global_abpoa.rs:47-61) ----
    {
        unsigned long long left, right;
        band_simd(0, 0, 0, g.r_values[0], (unsigned long long)W, bta, left, right);
        if ((long long)right > a.cap_cells) overflow = true;
        if (!overflow) {
            int carry = 0;
            for (int cb = 0; cb < (int)right; cb += WAVE) {
                int c = cb + lane;
                int gc = (c >= 1 && c < (int)right) ? sc_at(a.sc, read[c], GAP) : 0;
                int s = wave_incl_sum(gc, lane) + carry;
                if (c < (int)right) { am[c] = s; apw[c] = (c == 0) ? 0u : 3u; }  // path 0.3 -> (pred 0, L)
                carry = __shfl(s, WAVE - 1, WAVE);
            }
        }
        if (lane == 0) rinfo[0] = make_int4(0, 0, (int)right, 0);
        off = (long long)right;
    }
    __syncthreads();

    // ---- rows 1..L-2 (global_abpoa.rs:63-226) ----
    for (int i = 1; i + 1 < L && !overflow; ++i) {
        const int pb = g.pred_off[i], pe = g.pred_off[i + 1];
        const bool nwp = pe > pb;
        unsigned long long ms, me;
        if (!nwp) {
            unsigned long long pl = (unsigned long long)rinfo[i - 1].w;
            ms = pl + 1; me = pl + 1;
        } else {
            unsigned long long pl = 0, pr = 0;
            for (int e = pb; e < pe; ++e) {
                unsigned long long cb = (unsigned long long)rinfo[g.pred_rows[e]].w;
                if (e == pb) { pl = cb; pr = cb; }
                if (cb < pl) pl = cb;
                if (cb > pr) pr = cb;
            }
            ms = pl + 1; me = pr + 1;
        }
        unsigned long long left64, right64;
        band_simd(i, ms, me, g.r_values[i], (unsigned long long)W, bta, left64, right64);
        const int left = (int)left64, right = (int)right64;
        const int start = left == 0 ? 1 : left;
        const int end = right == W ? ((right - start) / 8) * 8 + start : right;
        const int width = right - start;
        if (off + width > a.cap_cells) { overflow = true; break; }
        const int li = g.lnz[i];
        const int g_row = sc_at(a.sc, li, GAP);
        // running state of the left sweep: z = v - G (see DESIGN.md "m0 left sweep as a scan")
        int carry_z = (start - 1 == 0) ? a.col0[i] : cx.min_score;
        int carry_G = 0;
        long long best_key = left == 0 ? (((long long)a.col0[i] << 32) | 0u) : (((long long)INT32_MIN << 32) | (unsigned)left);
        for (int cb = start; cb < right; cb += WAVE) {
            const int c = cb + lane;
            const bool act = c < right;
            const bool simd = c < end;
            int b = INT32_MIN / 2, gc = 0;
            uint32_t pw = 0;
            if (act) {
                const int rc = read[c];
                int bu, bd, pu, pd;
                if (!nwp) {
                    bu = m_at(cx, i - 1, c); bd = m_at(cx, i - 1, c - 1); pu = pd = i - 1;
                } else {
                    int p0 = g.pred_rows[pb];
                    bu = m_at(cx, p0, c); bd = m_at(cx, p0, c - 1); pu = pd = p0;
                    for (int e = pb + 1; e < pe; ++e) {  // strict '>' : first predecessor wins ties (:127-139)
                        int p = g.pred_rows[e];
                        int u = m_at(cx, p, c), d = m_at(cx, p, c - 1);
                        if (u > bu) { bu = u; pu = p; }
                        if (d > bd) { bd = d; pd = p; }
                    }
                }
                const int us = bu + g_row;
                if (simd) {
                    const int ds = bd + sc_at(a.sc, li, rc);
                    const bool isd = ds > us;                      // ties -> up (:144)
                    b = isd ? ds : us;
                    pw = isd ? ((uint32_t)pd << 2 | 1u) : ((uint32_t)pu << 2 | 2u);
                    const int head = start + ((c - start) / 8) * 8;
                    gc = sc_at(a.sc, read[head], GAP);           // gap key of the chunk head (:157)
                } else {
                    const int ds = bd + (nwp ? sc_at(a.sc, rc, li) : sc_at(a.sc, li, rc));  // swapped key (:206)
                    const bool isd = ds >= us;                     // tail: D > U > L (:175-181)
                    b = isd ? ds : us;
                    pw = isd ? ((uint32_t)pd << 2 | 1u) : ((uint32_t)pu << 2 | 2u);
                    gc = sc_at(a.sc, rc, GAP);
                }
            }
            const int G = wave_incl_sum(gc, lane) + carry_G;
            const int y = act ? b - G : INT32_MIN / 2;
            const int zi = wave_incl_max(y, lane);
            int zprev = __shfl_up(zi, 1, WAVE);
            zprev = lane == 0 ? carry_z : max(zprev, carry_z);
            int v = b;
            if (act) {
                if (zprev > y) { v = zprev + G; pw = ((uint32_t)i << 2) | 3u; }  // strict '>' (:158)
                am[off + (c - start)] = v;
                apw[off + (c - start)] = pw;
            }
            // best_col: last column attaining the row maximum (:162-164, :220-222)
            long long key = act ? (((long long)v << 32) | (unsigned)c) : (((long long)INT32_MIN << 32));
            key = wave_max_ll(key);
            if ((int)(key >> 32) >= (int)(best_key >> 32)) best_key = key;
            carry_z = max(carry_z, __shfl(zi, WAVE - 1, WAVE));
            carry_G = __shfl(G, WAVE - 1, WAVE);
        }
        if (lane == 0) rinfo[i] = make_int4((int)off, start, right, (int)(best_key & 0xffffffffll));
        off += width;
        ncells += (unsigned long long)width;
        __syncthreads();
    }

    if (overflow) {
        if (lane == 0) { rec->status = ST_OVERFLOW; rec->n_ops = 0; }
        return;
    }

    // ---- end node (global_abpoa.rs:227-240) ----
    int best = 0, last_row = 0;
    {
        const int pb = g.pred_off[L - 1], pe = g.pred_off[L];
        for (int e = pb; e < pe; ++e) {
            int p = g.pred_rows[e];
            int v = m_at(cx, p, W - 1);
            if (e == pb || v > best) { best = v; last_row = p; }
        }
    }
    // ---- traceback over the stored path words (gaf_output.rs:777-817), one lane ----
    if (lane == 0) {
        uint8_t* ops = a.ops + (long long)rd * a.ops_stride;
        int32_t* orow = a.oprows + (long long)rd * a.ops_stride;
        int row = last_row, col = W - 1, nops = 0;
        uint32_t status = 0;
        while (true) {
            uint32_t pw;
            if (row == 0) {
                int4 r0 = rinfo[0];
                if (col == 0) break;                    // cell value 0.0
                pw = col < r0.z ? 3u : 0xffffffffu;
            } else if (col == 0) {
                pw = ((uint32_t)g.min_pred[row] << 2) | 2u;  // :36-46
            } else {
                int4 ri = rinfo[row];
                pw = (col >= ri.y && col < ri.z) ? apw[ri.x + (col - ri.y)] : 0xffffffffu;
            }
            if (pw == 0xffffffffu) { status |= ST_BAND_NOT_ENOUGH; break; }
            const int dir = (int)(pw & 3u), pred = (int)(pw >> 2);
            if (nops >= a.ops_stride) { status |= ST_WOULD_PANIC; break; }
            if (dir == 1) {
                if (col == 0) { status |= ST_WOULD_PANIC; break; }
                ops[nops] = OP_D; orow[nops] = pred; row = pred; col -= 1;
            } else if (dir == 2) {
                ops[nops] = OP_U; orow[nops] = pred; row = pred;
            } else {
                if (col == 0) { status |= ST_WOULD_PANIC; break; }
                ops[nops] = OP_L; orow[nops] = -1; col -= 1;
            }
            ++nops;
        }
        rec->status = status;
        rec->score = best;
        rec->fscore = (float)best;
        rec->end_row = last_row;
        rec->end_col = W - 1;
        rec->stop_row = row;
        rec->stop_col = col;
        rec->n_ops = nops;
        rec->n_fwd_ops = 0;
        atomicAdd(a.cells, ncells);
    }
}

void launch_m0_simd(const PoaArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_m0_simd, dim3(a.nreads), dim3(64), 0, s, a);
}

}  // namespace rg
