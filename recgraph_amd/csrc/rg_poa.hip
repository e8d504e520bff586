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

// utils.rs:17-98 (simd_version = true).  The reference computes in usize; every quantity here is below 2^31 (columns of a
// read of at most 2^20 bases) and the only wrap-arounds are differences taken modulo 8, which 2^32 preserves — so 32-bit
// unsigned arithmetic gives the same band with half the scalar instructions (the kernel is bound by the scalar unit).
__device__ void band_simd(int i, unsigned ms, unsigned me, int r_val, unsigned seq_len, unsigned bta, unsigned& left, unsigned& right) {
    (void)i;
    int tmp_bs = min((int)ms, ((int)seq_len - r_val) - (int)bta);
    unsigned band_start = tmp_bs < 0 ? 0u : (unsigned)tmp_bs;
    unsigned r32 = r_val < 0 ? ~0u : (unsigned)r_val;
    unsigned band_end;
    if (seq_len > r32) {
        unsigned a = me > seq_len - r32 ? me : seq_len - r32;
        band_end = min(seq_len, a + bta);
    } else {
        band_end = min(seq_len, me + bta);
    }
    unsigned nr = band_end, nl = band_start;
#ifdef RG_BAND_SIMD_LOOPS
    // (the reference's three loops as written, utils.rs:77-97: what the closed form below replaces)
    while ((nr - nl) % 8 != 0) {
        if ((nr - nl) % 2 == 0 && nr < seq_len) nr += 1;
        else if (nl > 0) nl -= 1;
        else break;
    }
    if (nl == 0)
        while ((nr - 1) % 8 != 0 && nr < seq_len) nr += 1;
    if (nr == seq_len)
        while ((nr - nl) % 8 != 0 && nl > 1) nl -= 1;
#else
    // The reference widens the band to a multiple of 8 one column at a time (utils.rs:77-85): an even width takes a column
    // on the right while there is one, an odd width (or no room on the right) one on the left, and it gives up when it needs
    // the left and the left is 0.  As scalar loops that was up to 7 iterations of ~25 instructions and branches per DP ROW
    // on the CU's one scalar unit — what bounds this kernel.  Closed form (checked against the loops for every
    // (left, right, length) up to 40 and a few long ones, profiles/r05_notes.md): k steps are needed, the steps alternate
    // E(ven), O(dd) starting with the parity of the width; E takes from the right's room R first, O from the left's room;
    // with s steps taken the left gave b(s) = #O(s) + max(0, #E(s) - R) columns, and the loop stops at the largest s <= k
    // with b(s) <= left.
    {
        const unsigned d = (nr - nl) & 7u, k = (8u - d) & 7u, p0 = d & 1u;
        const unsigned R = seq_len - nr, Lr = nl;
        const unsigned s1 = 2u * R + p0;                    // steps until the right's room is used up
        const unsigned m = min(k, s1);
        const unsigned lim = 2u * Lr + 1u - p0;             // steps until the left's room is, while the right still has room
        const unsigned st = lim < m ? lim : (m == k ? k : min(k, R + Lr));
        const unsigned nE = p0 ? st >> 1 : (st + 1u) >> 1;
        const unsigned a = min(nE, R);
        nr += a;
        nl -= st - a;
    }
    // utils.rs:86-90: a band that starts at column 0 ends at a column = 1 (mod 8) — or at the read's end
    if (nl == 0) nr = min(seq_len, nr + ((1u - nr) & 7u));
    // utils.rs:91-96: a band that ends at the read's end takes the missing columns on the left, down to column 1
    if (nr == seq_len) nl -= min((nl - nr) & 7u, nl > 1u ? nl - 1u : 0u);
#endif
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

// kUniGap: every read base has the same gap cost (all matrices the reference's CLI builds, score_matrix.rs:35-105): the
// prefix sum G of the left sweep's gap costs is then (columns so far) * g and needs neither its DPP scan nor the two
// dependent LDS lookups of the chunk head's base.
template <bool kLdsRead, bool kUniGap>
// (8 waves per SIMD instead of the 6 the compiler's default register allocation gives — `__launch_bounds__(64, 8)`: 34 VGPRs
// instead of 77 — were tried in round 4: the same 1.40 M reads/s with 16 % MORE instructions (rematerialisation): 12 %
// VALU-active per wave at 6-8 waves is a SIMD that issues most of the time; the kernel is bound by its instruction count.)
__global__ __launch_bounds__(64) void k_m0_simd(PoaArgs a) {
    const int slot = blockIdx.x;              // arena slot of this launch
    const int rd = a.read_base + slot;        // read of the batch
    const int lane = threadIdx.x;
    const DevLnz& g = a.g;
    const int L = g.L;
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    const uint8_t* gread = a.reads + ro - 1;  // read_at(j), j = 1..n
    // Score table and (when it fits) the read's base codes live in LDS: both are indexed per lane every row, and as
    // kernel-argument / global loads each lookup was a dependent memory round trip (78 % of the wave cycles waiting).
    extern __shared__ int m0_lds[];
    int* sct = m0_lds;                                       // [36]
    uint8_t* lread = reinterpret_cast<uint8_t*>(m0_lds + 36);   // [n + 1] when a.lds_read
    if (lane < 36) sct[lane] = a.sc.t[lane];
    if (kLdsRead)
        for (int j = 1 + lane; j <= n; j += WAVE) lread[j] = gread[j];
    __syncthreads();
    // (a run-time choice between the two pointers would turn every access into a FLAT load)
    auto read_at = [&](int j) -> int { return kLdsRead ? (int)lread[j] : (int)gread[j]; };
    DevRecord* rec = a.rec + rd;
    const int W = n + 1;
    if (a.bad[rd]) {
        if (lane == 0) { rec->status = ST_BAD_BASE; rec->n_ops = 0; rec->score = 0; }
        return;
    }
    int* am = a.arena_m + (long long)slot * a.cap_cells;
    uint32_t* apw = a.arena_pw + (long long)slot * a.cap_cells;
    int4* rinfo = a.rinfo + (long long)slot * L;
    const unsigned bta = (unsigned)a.bta[rd];
    const int GAP = 5;
    const int ugap = sct[GAP];            // (kUniGap: the cost of every read base)
    M0Ctx cx{am, rinfo, a.col0, 2 * W * sct[read_at(1) * 6 + GAP]};  // global_abpoa.rs:20
    long long off = 0;
    unsigned long long ncells = 0;
    bool overflow = false;

    // ---- row 0 (global_abpoa.rs:47-61) ----
    unsigned rinfo_right0 = 0;
    // m[0][64 k + lane] for the first four chunks of row 0.  (Four named registers, here and for the previous row below: as
    // `int pv[4]` selected through a loop the compiler kept the array in SCRATCH — 32 bytes per lane, one scratch store per
    // row and two dependent scratch loads per chunk in the fast path, found in the ISA in round 4.)
    int row0_v0 = 0, row0_v1 = 0, row0_v2 = 0, row0_v3 = 0;
    {
        unsigned left, right;
        band_simd(0, 0, 0, g.r_values[0], (unsigned)W, bta, left, right);
        if ((long long)right > a.cap_cells) overflow = true;
        if (!overflow) {
            int carry = 0;
            for (int cb = 0; cb < (int)right; cb += WAVE) {
                int c = cb + lane;
                int gc = (c >= 1 && c < (int)right) ? sct[read_at(c) * 6 + GAP] : 0;
                int s = wave_incl_sum(gc, lane) + carry;
                if (c < (int)right) { am[c] = s; apw[c] = (c == 0) ? 0u : 3u; }  // path 0.3 -> (pred 0, L)
                row0_v0 = cb == 0 ? s : row0_v0;
                asm volatile("" : "+v"(row0_v0));
                row0_v1 = cb == WAVE ? s : row0_v1;
                asm volatile("" : "+v"(row0_v1));
                row0_v2 = cb == 2 * WAVE ? s : row0_v2;
                asm volatile("" : "+v"(row0_v2));
                row0_v3 = cb == 3 * WAVE ? s : row0_v3;
                asm volatile("" : "+v"(row0_v3));
                carry = __shfl(s, WAVE - 1, WAVE);
            }
        }
        if (lane == 0) rinfo[0] = make_int4(0, 0, (int)right, 0);
        off = (long long)right;
        rinfo_right0 = right;
    }
    // The previous row travels in registers (chunk k, lane l holds column p_start + 64 k + l) whenever it fits KC chunks: a row whose only
    // predecessor is the row above - all rows inside a segment - then needs no load at all, its stores are fire and
    // forget, and the per-row metadata is fetched one row ahead.  Rows with listed predecessors (segment starts) and bands
    // wider than 64 columns take the memory path below; a barrier orders this wave's earlier stores before those loads.
    constexpr int KC = 4;           // up to 256 band columns per row
    static_assert(KC == 4, "four named registers per row below");
    int pv0 = row0_v0, pv1 = row0_v1, pv2 = row0_v2, pv3 = row0_v3;   // values of the previous row
    int p_start = 0, p_right = (int)rinfo_right0, p_best = 0;
    bool p_valid = (int)rinfo_right0 <= KC * WAVE;
    // chunk k of the previous row (k is wave-uniform; a select chain instead of a dynamically indexed register array)
    // (the empty asm statements keep the selects apart: left alone the compiler fuses them into a dynamically indexed
    // 4-vector, which it keeps in scratch)
    auto pv_chunk = [&](int k) -> int {
        int r = pv0;
        asm volatile("" : "+v"(r));
        r = k == 1 ? pv1 : r;
        asm volatile("" : "+v"(r));
        r = k == 2 ? pv2 : r;
        asm volatile("" : "+v"(r));
        r = k == 3 ? pv3 : r;
        asm volatile("" : "+v"(r));
        return r;
    };
    // m[i-1][col] of the reference's full-width matrix for this lane's column `col` (chunk base k0 is wave-uniform)
    auto prev_at = [&](int col, int k0, int lo, int hi) -> int {
        const int idx = col - p_start;
        const int sh_lo = __shfl(lo, idx & (WAVE - 1), WAVE), sh_hi = __shfl(hi, idx & (WAVE - 1), WAVE);
        const int v = (idx >> 6) == k0 ? sh_lo : sh_hi;
        return (col >= p_start && col < p_right) ? v : cx.min_score;
    };
    bool dirty = true;              // stores issued since the last barrier
    int c0_prev = 0;                // col0[i - 1]
    // metadata of the next row, loaded one iteration ahead
    // (PoaArgs::rowmeta: one 16-byte scalar load per row; as five loads — two of them dependent on the others — the row loop
    // waited for the scalar cache twice per row)
    int n_pb = uload(g.pred_off + 1);
    int4 n_meta = L > 2 ? uload4(a.rowmeta + 1) : make_int4(n_pb, 0, 0, 4 << 24);

    // ---- rows 1..L-2 (global_abpoa.rs:63-226) ----
    for (int i = 1; i + 1 < L && !overflow; ++i) {
        const int pb = n_pb, pe = n_meta.x, rv = n_meta.y, c0_cur = n_meta.z, p0_cur = (n_meta.w & 0xffffff) - 1, li = n_meta.w >> 24;
        if (i + 2 < L) { n_pb = pe; n_meta = uload4(a.rowmeta + i + 1); }
        const bool nwp = pe > pb;
        // a listed predecessor list that is just {i - 1} (chains of single-base segments) behaves like an inner row
        const bool only_prev = !nwp || (pe - pb == 1 && p0_cur == i - 1);
        unsigned ms, me;
        if (only_prev) {
            const unsigned pl = (unsigned)p_best;   // best_scoring_pos of row i - 1
            ms = pl + 1; me = pl + 1;
        } else {
            if (dirty) { __syncthreads(); dirty = false; }
            unsigned pl = 0, pr = 0;
            for (int e = pb; e < pe; ++e) {
                unsigned cb = (unsigned)rinfo[g.pred_rows[e]].w;
                if (e == pb) { pl = cb; pr = cb; }
                if (cb < pl) pl = cb;
                if (cb > pr) pr = cb;
            }
            ms = pl + 1; me = pr + 1;
        }
        unsigned left64, right64;
        band_simd(i, ms, me, rv, (unsigned)W, bta, left64, right64);
        const int left = (int)left64, right = (int)right64;
        const int start = left == 0 ? 1 : left;
        const int end = right == W ? ((right - start) / 8) * 8 + start : right;
        const int width = right - start;
        if (off + width > a.cap_cells) { overflow = true; break; }
        const int g_row = sct[li * 6 + GAP];
        const bool fast = only_prev && p_valid && width <= KC * WAVE;
        if (!fast && dirty) { __syncthreads(); dirty = false; }
        // running state of the left sweep: z = v - G (see DESIGN.md "m0 left sweep as a scan")
        int carry_z = (start - 1 == 0) ? c0_cur : cx.min_score;
        int carry_G = 0;
        int best_v = left == 0 ? c0_cur : INT32_MIN, best_c = left == 0 ? 0 : left;
        int vk0 = 0, vk1 = 0, vk2 = 0, vk3 = 0;
        int ci = 0;
        for (int cb = start; cb < right; cb += WAVE, ++ci) {
            const int c = cb + lane;
            const bool act = c < right;
            const bool simd = c < end;
            int b = INT32_MIN / 2, gc = 0;
            uint32_t pw = 0;
            int bu = 0, bd = 0, pu = i - 1, pd = i - 1;
            if (fast) {
                // m[i-1][c] and m[i-1][c-1] of the reference's full-width matrix, from the registers of the row above
                // columns cb-1 .. cb+63 of the row above lie in at most two of its chunks: k0 and k0 + 1
                const int k0 = (cb - 1 - p_start) >> 6;             // arithmetic shift: -1 when cb - 1 < p_start
                const int lo = pv_chunk(k0 < 0 ? 0 : k0), hi = pv_chunk(k0 + 1 < KC ? k0 + 1 : KC - 1);
                // (tried in round 4: the lane's own register / its right neighbour's through one DPP move when the band moved by
                // 0 / 1 columns — the common cases — instead of the two shuffles: 8.55 vs 8.35 ms per 10 000 reads, no gain)
                bu = prev_at(c, k0, lo, hi);                      // (the shuffles inside must run with all lanes enabled)
                // m[i-1][c-1] is the lane to the left's m[i-1][c]; lane 0 takes column cb - 1 straight from the registers
                // of the row above (a wave-uniform position: v_readlane, no second pair of shuffles)
                const int i0 = cb - 1 - p_start;
                const int e_lo = __builtin_amdgcn_readlane(lo, i0 & (WAVE - 1)), e_hi = __builtin_amdgcn_readlane(hi, i0 & (WAVE - 1));
                const int e0 = (cb - 1 >= p_start && cb - 1 < p_right) ? ((i0 >> 6) == k0 ? e_lo : e_hi) : cx.min_score;
                // (dpp_shr1 pins the move here, where every lane is enabled: see rg_device.hpp)
                const int left_bu = dpp_shr1(bu, 0);
                const int dprev = lane == 0 ? e0 : left_bu;
                bd = (c - 1 == 0 && i - 1 > 0) ? c0_prev : dprev;
            }
            // predecessor values of the rows that take the memory path (listed predecessors / wide bands)
            if (!fast && act) {
                // (opaque copies of the row and of the list start: otherwise the compiler computes the 64-bit addresses of
                // rinfo[i - 1], col0[i - 1] and pred_rows[pb] in front of the chunk loop of EVERY row — ~20 instructions on the
                // scalar unit that bounds this kernel, for a path inner rows never take)
                int io = i, pbo = pb;
                asm volatile("" : "+s"(io), "+s"(pbo));
                if (!nwp) {
                    bu = m_at(cx, io - 1, c); bd = m_at(cx, io - 1, c - 1);
                } else {
                    int p0 = g.pred_rows[pbo];
                    bu = m_at(cx, p0, c); bd = m_at(cx, p0, c - 1); pu = pd = p0;
                    for (int e = pbo + 1; e < pe; ++e) {  // strict '>' : first predecessor wins ties (:127-139)
                        int p = g.pred_rows[e];
                        int u = m_at(cx, p, c), d = m_at(cx, p, c - 1);
                        if (u > bu) { bu = u; pu = p; }
                        if (d > bd) { bd = d; pd = p; }
                    }
                }
            }
            {
                // The cell itself, WITHOUT divergent control flow: every lane computes, idle lanes (c >= right) read the
                // last read base and are masked by selects.  The kernel is bound by the scalar unit (counters_C2.json: 6 waves
                // per SIMD issue SALU 96 % of the time) and every divergent `if` costs it three instructions (save exec,
                // branch, restore): the nest this replaces — active / SIMD part or tail / strict or weak compare — was a
                // third of the row's scalar instructions.
                const int rc = read_at(min(c, W - 1));
                const int us = bu + g_row;
                // SIMD part (:144): key (row base, read base), ties -> up; tail (:175-181, :206): D > U > L and, behind a row
                // with listed predecessors, the swapped key
                const int ds = bd + sct[(simd || !nwp) ? li * 6 + rc : rc * 6 + li];
                const bool isd = simd ? ds > us : ds >= us;
                b = act ? (isd ? ds : us) : INT32_MIN / 2;
                pw = isd ? ((uint32_t)pd << 2 | 1u) : ((uint32_t)pu << 2 | 2u);
                if (!kUniGap) {
                    const int head = start + ((c - start) / 8) * 8;
                    const int gk = simd ? read_at(min(head, W - 1)) : rc;  // gap key: the chunk head's base (:157) / the cell's own
                    gc = act ? sct[gk * 6 + GAP] : 0;
                }
            }
            // active lanes are a prefix of the wave: with one gap cost for every base the inclusive sum is (lane + 1) * g
            const int G = kUniGap ? carry_G + (lane + 1) * ugap : dpp_incl_sum(gc) + carry_G;
            const int y = act ? b - G : INT32_MIN / 2;
            const int zi = dpp_incl_max(y, INT32_MIN / 2);
            int zprev = dpp_shr1(zi, INT32_MIN / 2);
            zprev = lane == 0 ? carry_z : max(zprev, carry_z);
            const bool tl = act && zprev > y;                                 // strict '>' (:158)
            const int v = tl ? zprev + G : b;
            pw = tl ? (((uint32_t)i << 2) | 3u) : pw;
            if (act) {
                am[off + (c - start)] = v;
                apw[off + (c - start)] = pw;
            }
            vk0 = ci == 0 ? v : vk0;
            asm volatile("" : "+v"(vk0));
            vk1 = ci == 1 ? v : vk1;
            asm volatile("" : "+v"(vk1));
            vk2 = ci == 2 ? v : vk2;
            asm volatile("" : "+v"(vk2));
            vk3 = ci == 3 ? v : vk3;
            asm volatile("" : "+v"(vk3));
            // best_col: last column attaining the row maximum (:162-164, :220-222)
            const int vm = act ? v : INT32_MIN;
            const int cmx = __builtin_amdgcn_readlane(dpp_incl_max(vm, INT32_MIN), WAVE - 1);
            if (cmx >= best_v) {
                const unsigned long long at = __ballot(act && v == cmx);
                best_v = cmx; best_c = cb + 63 - __clzll((long long)at);
            }
            carry_z = max(carry_z, __builtin_amdgcn_readlane(zi, WAVE - 1));
            carry_G = kUniGap ? carry_G + min(WAVE, right - cb) * ugap : __builtin_amdgcn_readlane(G, WAVE - 1);
        }
        if (lane == 0) rinfo[i] = make_int4((int)off, start, right, best_c);
        off += width;
        ncells += (unsigned long long)width;
        dirty = true;
        p_best = best_c; p_start = start; p_right = right; p_valid = width <= KC * WAVE;
        pv0 = vk0; pv1 = vk1; pv2 = vk2; pv3 = vk3;
        c0_prev = c0_cur;
    }
    __syncthreads();

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
    const size_t bytes = 36 * sizeof(int) + (a.lds_read ? (((size_t)a.max_n + 2 + 3) & ~(size_t)3) : 0);
    bool uni = true;                     // one gap cost for every read base (ACGTN)
    for (int b = 1; b < 5; ++b) uni = uni && a.sc.t[b * 6 + 5] == a.sc.t[5];
    if (a.lds_read && uni) hipLaunchKernelGGL((k_m0_simd<true, true>), dim3(a.nreads), dim3(64), bytes, s, a);
    else if (a.lds_read) hipLaunchKernelGGL((k_m0_simd<true, false>), dim3(a.nreads), dim3(64), bytes, s, a);
    else if (uni) hipLaunchKernelGGL((k_m0_simd<false, true>), dim3(a.nreads), dim3(64), bytes, s, a);
    else hipLaunchKernelGGL((k_m0_simd<false, false>), dim3(a.nreads), dim3(64), bytes, s, a);
}

}  // namespace rg
