// Local POA kernels for gfx950: -m 1 in both of the reference's flavours (local_poa::exec_simd,
// src/local_poa.rs:9-174, and local_poa::exec, :176-262) and -m 3 (gap_local_poa::exec, src/gap_local_poa.rs:6-183),
// with the traceback of their GAF walkers (gaf_output.rs:383-752).
//
// The local modes are unbanded: the reference fills full L x W matrices.  Mapping as in rg_poa.hip: one wavefront per
// read, lanes over 64 consecutive columns of a row, rows sequential, every row written to HBM once (a later row may
// name any earlier row as predecessor).  The left recurrence, including the clamp at zero, is a wave-level max-plus
// prefix scan:
//   -m 1:  v[c] = max(b'[c], v[c-1] + g[c]),  b' = max(b, 0) where the reference clamps (its AVX2 multi-predecessor
//          tail does not), b = best of diagonal / up;
//   -m 3:  x[c] = e + max(x[c-1], m[c-1] + o) = e*c + max_{k<c}(t'[k] + o - e*k)  (o <= 0),  t' = max(d, y, 0).
// The direction of each cell is then re-derived from (d, u, l) with the reference's literal tie and clamp rules.
#include "rg_device.hpp"
#include "rg_poa_args.hpp"

namespace rg {

namespace {

constexpr int NEGL = INT32_MIN / 4;
constexpr long long IDX_SPAN = 1ll << 40;   // row-major cell index < 2^40, value in the bits above

__device__ __forceinline__ int scl(const DevScores& sc, int a, int b) { return sc.t[a * 6 + b]; }

// bitfield_path.rs:3-15 direction codes used here; 0 doubles as the f32 path value 0.0 of the AVX2 flavour
enum : uint32_t { LD_O = 0, LD_D = 1, LD_d = 2, LD_L = 3, LD_U = 4 };

}  // namespace

// kVar 0: -m 1 AVX2 semantics (f32 values are integers < 2^24: int32 is exact);  1: -m 1 scalar;  2: -m 3.
// Planes per read (cap_cells each): m | y (kVar 2);  path words: w0 = pred << 3 | dir | X << 31,  w1 = predY << 1 | Y.
template <int kVar, bool kLdsRead>
__global__ __launch_bounds__(64) void k_poa_local(PoaArgs a) {
    const int slot = blockIdx.x;
    const int rd = a.read_base + slot;
    const int lane = threadIdx.x;
    const DevLnz& g = a.g;
    const int L = g.L;
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    const uint8_t* gread = a.reads + ro - 1;   // read_at(c), c = 1..n
    DevRecord* rec = a.rec + rd;
    const int W = n + 1;
    // score table and read codes in LDS, wave-uniform graph tables through the scalar cache (see rg_poa.hip)
    extern __shared__ int pl_lds[];
    int* sct = pl_lds;
    uint8_t* lread = reinterpret_cast<uint8_t*>(pl_lds + 36);
    if (lane < 36) sct[lane] = a.sc.t[lane];
    if (kLdsRead)
        for (int jj = 1 + lane; jj <= n; jj += WAVE) lread[jj] = gread[jj];
    __syncthreads();
    auto read_at = [&](int jj) -> int { return kLdsRead ? (int)lread[jj] : (int)gread[jj]; };
    if (a.bad[rd]) {
        if (lane == 0) { rec->status = ST_BAD_BASE; rec->n_ops = 0; rec->score = 0; }
        return;
    }
    if ((long long)(L - 1) * W > a.cap_cells) {
        if (lane == 0) { rec->status = ST_OVERFLOW; rec->n_ops = 0; }
        return;
    }
    constexpr int kPlanes = kVar == 2 ? 2 : 1;
    int* am = a.arena_m + (long long)slot * a.cap_cells * kPlanes;
    int* ay = am + a.cap_cells;
    uint32_t* pw0 = a.arena_pw + (long long)slot * a.cap_cells * kPlanes;
    uint32_t* pw1 = pw0 + a.cap_cells;
    const int GAP = 5;
    const int o = a.gap_open, e = a.gap_ext;
    const int max_multiple = W % 8 != 0 ? (W / 8) * 8 : W - 8;   // local_poa.rs:19-23

    // row 0: all zero, path 'O' (local_poa.rs:17-18, :190-192)
    for (int c = lane; c < W; c += WAVE) {
        am[c] = 0; pw0[c] = 0;
        if (kVar == 2) { ay[c] = 0; pw1[c] = 0; }
    }
    __syncthreads();

    // best cell: kVar 0 takes the LAST maximum in row-major order ('>=', cells i,c >= 1 only, start (0,0));
    // kVar 1/2 take the FIRST ('>', all cells, start (0,0))
    long long best_key = kVar == 0 ? 0ll : IDX_SPAN - 1;

    // The row above travels in registers (chunk k, lane l = column 64 k + l: local rows are full width, so chunks line
    // up) when W <= 64 KC: a row whose only predecessor is the row above needs no load and no barrier.
    constexpr int KC = 8;
    int pvm[KC] = {}, pvy[kVar == 2 ? KC : 1] = {};
    const bool p_fit = W <= KC * WAVE;
    bool dirty = false;

    for (int i = 1; i + 1 < L; ++i) {
        const int pb = uload(g.pred_off + i), pe = uload(g.pred_off + i + 1);
        const bool nwp = pe > pb;
        const bool fast = p_fit && !nwp;         // (rows with a listed predecessor take the reference's other code path)
        if (!fast && dirty) { __syncthreads(); dirty = false; }
        const int li = uload_u8(g.lnz, i);
        const long long rowoff = (long long)i * W;
        int keep_m[KC] = {}, keep_y[kVar == 2 ? KC : 1] = {};
        int ci = 0;
        int carry_z = NEGL, carry_G = 0;
        int carry_x = 0, carry_t = 0;          // kVar 2: x and t' of the previous chunk's last column
        for (int cb = 0; cb < W; cb += WAVE, ++ci) {
            const int c = cb + lane;
            const bool act = c < W;
            const bool cell = act && c >= 1;
            // fast path: m[i-1][c], m[i-1][c-1] (and y[i-1][c]) from the registers of the row above
            int f_u = 0, f_d = 0, f_y = 0;
            if (fast) {
                int cur_m = pvm[0], prv_m = 0, cur_y = pvy[0];
#pragma unroll
                for (int k = 1; k < KC; ++k) {
                    cur_m = ci == k ? pvm[k] : cur_m;
                    prv_m = ci == k ? pvm[k - 1] : prv_m;
                    if (kVar == 2) cur_y = ci == k ? pvy[kVar == 2 ? k : 0] : cur_y;
                }
                f_u = cur_m; f_y = cur_y;
                f_d = dpp_shr1(cur_m, 0);
                const int edge = __builtin_amdgcn_readlane(prv_m, WAVE - 1);
                if (lane == 0) f_d = edge;
            }
            const int rc = cell ? read_at(c) : 4;
            int d = 0, u = 0, dp = 0, up = 0;
            int uy = 0, uyp = 0;                 // kVar 2: y candidate
            if (cell) {
                if (fast) {
                    d = f_d; u = f_u; dp = up = i - 1;
                    if (kVar == 2) { uy = f_y; uyp = i - 1; }
                } else if (!nwp) {
                    const long long po = (long long)(i - 1) * W + c;
                    d = am[po - 1]; u = am[po]; dp = up = i - 1;
                    if (kVar == 2) { uy = ay[po]; uyp = i - 1; }
                } else if (kVar == 0) {
                    // first predecessor initialises, strict '>' afterwards (local_poa.rs:61-75, :129-143)
                    int p = g.pred_rows[pb];
                    long long po = (long long)p * W + c;
                    d = am[po - 1]; u = am[po]; dp = up = p;
                    for (int q = pb + 1; q < pe; ++q) {
                        p = g.pred_rows[q];
                        po = (long long)p * W + c;
                        const int dv = am[po - 1], uv = am[po];
                        if (uv > u) { u = uv; up = p; }
                        if (dv > d) { d = dv; dp = p; }
                    }
                } else {
                    // get_best_d / get_best_u start from (0, row 0): `first` is initialised to false
                    // (local_poa.rs:263-298, gap_local_poa.rs:126-183)
                    for (int q = pb; q < pe; ++q) {
                        const int p = g.pred_rows[q];
                        const long long po = (long long)p * W + c;
                        const int dv = am[po - 1];
                        if (dv > d) { d = dv; dp = p; }
                        if (kVar == 1) {
                            const int uv = am[po];
                            if (uv > u) { u = uv; up = p; }
                        } else {
                            const int um = am[po] + o, yv = ay[po];
                            if (um > u) { u = um; up = p; }
                            if (yv > uy) { uy = yv; uyp = p; }
                        }
                    }
                }
            }
            int mval = 0;
            uint32_t w0 = 0, w1 = 0;
            if (kVar == 0) {
                // ---------------- AVX2 flavour ----------------
                const bool simd = c <= max_multiple;
                int b = 0, gk = 0;
                bool isd = false, clamp = true;
                if (cell) {
                    const int us = u + sct[(li) * 6 + (GAP)];
                    if (simd) {
                        const int ds = d + sct[(li) * 6 + (rc)];
                        isd = ds > us;                                  // ties -> up (:47, :80)
                        b = isd ? ds : us;
                        gk = sct[(read_at(((c - 1) / 8) * 8 + 1)) * 6 + (GAP)];  // gap key of the chunk head (:94)
                    } else {
                        const int ds = d + (nwp ? sct[(rc) * 6 + (li)] : sct[(li) * 6 + (rc)]);   // swapped key (:147)
                        isd = ds >= us;                                 // D > U > L (:119-127, :150-156)
                        b = isd ? ds : us;
                        gk = sct[(rc) * 6 + (GAP)];
                        clamp = !nwp;                                   // the multi-predecessor tail never clamps
                    }
                }
                const int bsrc = !act ? NEGL : (!cell ? 0 : (clamp ? max(b, 0) : b));
                const int G = dpp_incl_sum(gk) + carry_G;
                const int y = act ? bsrc - G : NEGL;
                const int zi = dpp_incl_max(y, NEGL);
                int zprev = dpp_shr1(zi, NEGL);
                zprev = lane == 0 ? carry_z : max(zprev, carry_z);
                if (cell) {
                    const int l = zprev + G;
                    int v; uint32_t w;
                    if (l > b) { v = l; w = ((uint32_t)i << 3) | LD_L; }
                    else { v = b; w = isd ? (((uint32_t)dp << 3) | LD_D) : (((uint32_t)up << 3) | LD_U); }
                    if (clamp && (simd ? v <= 0 : v < 0)) { v = 0; w = 0; }   // '<= 0' (:99) vs '< 0' (:115)
                    mval = v; w0 = w;
                    const long long key = (long long)v * IDX_SPAN + (rowoff + c);
                    if (key > best_key) best_key = key;
                }
                carry_z = max(carry_z, __builtin_amdgcn_readlane(zi, WAVE - 1));
                carry_G = __builtin_amdgcn_readlane(G, WAVE - 1);
            } else if (kVar == 1) {
                // ---------------- scalar flavour ----------------
                int dv = 0, uv = 0, gk = 0;
                if (cell) {
                    dv = d + sct[(rc) * 6 + (li)];            // key (sequence[j], lnz[i]) (:205, :213)
                    uv = u + sct[(GAP) * 6 + (li)];           // key ('-', lnz[i])
                    gk = sct[(rc) * 6 + (GAP)];
                }
                const int bsrc = !act ? NEGL : (!cell ? 0 : max(max(dv, uv), 0));
                const int G = dpp_incl_sum(gk) + carry_G;
                const int y = act ? bsrc - G : NEGL;
                const int zi = dpp_incl_max(y, NEGL);
                int zprev = dpp_shr1(zi, NEGL);
                zprev = lane == 0 ? carry_z : max(zprev, carry_z);
                if (cell) {
                    const int l = zprev + G;
                    if (dv < 0 && l < 0 && uv < 0) { mval = 0; w0 = 0; }
                    else if (dv < uv) {                    // utils.rs:129-140
                        if (uv < l) { mval = l; w0 = ((uint32_t)(i & 0xffff) << 3) | LD_L; }
                        else { mval = uv; w0 = ((uint32_t)(up & 0xffff) << 3) | LD_U; }
                    } else {
                        if (dv < l) { mval = l; w0 = ((uint32_t)(i & 0xffff) << 3) | LD_L; }
                        else { mval = dv; w0 = ((uint32_t)(dp & 0xffff) << 3) | (li != rc ? LD_d : LD_D); }
                    }
                    const long long key = (long long)mval * IDX_SPAN + (IDX_SPAN - 1 - (rowoff + c));
                    if (key > best_key) best_key = key;
                }
                carry_z = max(carry_z, __builtin_amdgcn_readlane(zi, WAVE - 1));
                carry_G = __builtin_amdgcn_readlane(G, WAVE - 1);
            } else {
                // ---------------- -m 3 ----------------
                int dv = 0, yval = 0, ypred = 0, tcur = 0;
                bool fromy = false;
                if (cell) {
                    dv = d + sct[(rc) * 6 + (li)];
                    if (!nwp) {
                        const int u_y = uy + e, u_m = u + o + e;          // (:53-66)
                        fromy = u_y > u_m;
                        yval = fromy ? u_y : u_m;
                        ypred = i - 1;
                    } else {
                        const bool from_m = u > uy;                        // get_best_u (:177-182): u already holds m + o
                        yval = (from_m ? u : uy) + e;
                        ypred = from_m ? up : uyp;
                        fromy = !from_m;
                    }
                    tcur = max(max(dv, yval), 0);
                }
                // x[c] - e*c = max_{k<c} src[k] - e*k,  src[0] = x[0] = 0,  src[k] = t'[k] + o
                const int zsrc = !act ? NEGL : ((cell ? tcur + o : 0) - e * c);
                const int zi = dpp_incl_max(zsrc, NEGL);
                int ze = dpp_shr1(zi, NEGL);
                ze = lane == 0 ? carry_z : max(ze, carry_z);
                const int xval = cell ? ze + e * c : 0;
                int xprev = dpp_shr1(xval, 0), tprev = dpp_shr1(tcur, 0);
                if (lane == 0) { xprev = carry_x; tprev = carry_t; }
                if (cell) {
                    const bool xflag = o != 0 && xprev > tprev + o;        // path_x = 'X' iff x[c-1] + e > m[c-1] + o + e
                    const int l = xval, uu = yval;
                    if (dv < 0 && l < 0 && uu < 0) { mval = 0; w0 = 0; }
                    else if (dv < uu) {
                        if (uu < l) { mval = l; w0 = ((uint32_t)(i & 0xffff) << 3) | LD_L; }
                        else { mval = uu; w0 = ((uint32_t)(ypred & 0xffff) << 3) | LD_U; }
                    } else {
                        if (dv < l) { mval = l; w0 = ((uint32_t)(i & 0xffff) << 3) | LD_L; }
                        else { mval = dv; w0 = ((uint32_t)(dp & 0xffff) << 3) | (li != rc ? LD_d : LD_D); }
                    }
                    if (xflag) w0 |= 0x80000000u;
                    w1 = fromy ? (((uint32_t)(ypred & 0xffff) << 1) | 1u) : 0u;
                    const long long key = (long long)mval * IDX_SPAN + (IDX_SPAN - 1 - (rowoff + c));
                    if (key > best_key) best_key = key;
                }
                if (act) ay[rowoff + c] = cell ? yval : 0;
#pragma unroll
                for (int k = 0; k < KC; ++k) keep_y[kVar == 2 ? k : 0] = ci == k ? (cell ? yval : 0) : keep_y[kVar == 2 ? k : 0];
                if (act) pw1[rowoff + c] = w1;
                carry_x = __builtin_amdgcn_readlane(xval, WAVE - 1);
                carry_t = __builtin_amdgcn_readlane(tcur, WAVE - 1);
                carry_z = max(carry_z, __builtin_amdgcn_readlane(zi, WAVE - 1));
            }
            if (act) { am[rowoff + c] = mval; pw0[rowoff + c] = w0; }
#pragma unroll
            for (int k = 0; k < KC; ++k) keep_m[k] = ci == k ? mval : keep_m[k];
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) { pvm[k] = keep_m[k]; if (kVar == 2) pvy[kVar == 2 ? k : 0] = keep_y[kVar == 2 ? k : 0]; }
        dirty = true;
    }
    __syncthreads();
    best_key = wave_max_ll(best_key);
    if (lane != 0) return;

    const int bestv = (int)(best_key >> 40);   // arithmetic shift: floor division by 2^40
    long long bidx = best_key & (IDX_SPAN - 1);
    if (kVar != 0) bidx = IDX_SPAN - 1 - bidx;
    const int best_row = (int)(bidx / W), best_col = (int)(bidx % W);

    // ---- traceback (gaf_output.rs:404-453, :527-598, :662-717), one lane ----
    uint8_t* ops = a.ops + (long long)rd * a.ops_stride;
    int32_t* orow = a.oprows + (long long)rd * a.ops_stride;
    int nops = 0, row = best_row, col = best_col;
    uint32_t status = 0;
    int guard = 0;
    while (true) {
        if (++guard > 4 * (L + W) || nops + 2 >= a.ops_stride) { status |= ST_WOULD_PANIC; break; }
        if (row < 0 || row >= L - 1 || col < 0 || col >= W) { status |= ST_WOULD_PANIC; break; }
        const uint32_t w = pw0[(long long)row * W + col];
        const uint32_t dir = w & 7u;
        if (dir == LD_O) break;
        const int pred = (int)((w >> 3) & 0xfffffu);
        if (dir == LD_D || dir == LD_d) {
            if (col == 0) { status |= ST_WOULD_PANIC; break; }
            ops[nops] = OP_D | (dir == LD_d ? 0x40 : 0); orow[nops] = pred; ++nops;
            row = pred; col -= 1;
        } else if (dir == LD_L) {
            if (kVar == 2 && (w >> 31)) {
                bool first = true, bad = false;
                while (pw0[(long long)row * W + col] >> 31) {
                    if (col == 0 || nops + 2 >= a.ops_stride) { bad = true; break; }
                    ops[nops] = OP_L | (first ? 0 : OP_CONT); orow[nops] = -1; ++nops; first = false;
                    col -= 1;
                }
                if (bad) { status |= ST_WOULD_PANIC; break; }
            } else {
                if (col == 0) { status |= ST_WOULD_PANIC; break; }
                ops[nops] = OP_L; orow[nops] = -1; ++nops; col -= 1;
            }
        } else if (dir == LD_U) {
            if (kVar == 2 && (pw1[(long long)row * W + col] & 1u)) {
                bool first = true, bad = false;
                while (true) {
                    const uint32_t y1 = pw1[(long long)row * W + col];
                    if (!(y1 & 1u)) break;
                    const int p = (int)(y1 >> 1);
                    if (p >= L - 1 || nops + 2 >= a.ops_stride) { bad = true; break; }
                    ops[nops] = OP_U | (first ? 0 : OP_CONT); orow[nops] = p; ++nops; first = false;
                    row = p;
                }
                if (bad) { status |= ST_WOULD_PANIC; break; }
            } else {
                ops[nops] = OP_U; orow[nops] = pred; ++nops;
                row = pred;
            }
        } else { status |= ST_WOULD_PANIC; break; }
    }
    rec->status = status;
    rec->score = bestv;
    rec->fscore = (float)bestv;
    rec->end_row = best_row;
    rec->end_col = best_col;
    rec->stop_row = row;
    rec->stop_col = col;
    rec->n_ops = (status & ST_WOULD_PANIC) ? 0 : nops;
    rec->n_fwd_ops = 0;
    atomicAdd(a.cells, (unsigned long long)(L - 2) * (unsigned long long)(W - 1));
}

template <int kVar>
static void launch_local_v(const PoaArgs& a, hipStream_t s) {
    const size_t bytes = 36 * sizeof(int) + (a.lds_read ? (((size_t)a.max_n + 2 + 3) & ~(size_t)3) : 0);
    if (a.lds_read) hipLaunchKernelGGL((k_poa_local<kVar, true>), dim3(a.nreads), dim3(64), bytes, s, a);
    else hipLaunchKernelGGL((k_poa_local<kVar, false>), dim3(a.nreads), dim3(64), bytes, s, a);
}
void launch_local(const PoaArgs& a, int variant, hipStream_t s) {
    if (variant == 0) launch_local_v<0>(a, s);
    else if (variant == 1) launch_local_v<1>(a, s);
    else launch_local_v<2>(a, s);
}

}  // namespace rg
