// Band-relative POA kernels for gfx950: -m 2 (gap_global_abpoa::exec, src/gap_global_abpoa.rs:11-250)
// and the scalar -m 0 (global_abpoa::exec, src/global_abpoa.rs:260-427), with their band check
// (band_ampl_enough) and traceback walkers (gaf_output.rs:96-381).
//
// Same mapping as rg_poa.hip: one wavefront per read, lanes over the band columns of a row, rows
// sequential.  Storage is band-relative like the reference (cell j of row i <-> absolute column
// left_i + j); predecessor columns are translated with left_i - left_p and a predecessor only
// contributes where its band covers the column.
//
// The two "left" recurrences are evaluated as wave-level prefix scans:
//   m0 scalar:  m[j] = max(du[j], m[j-1] + g(read[c]))            (du = max(d, u), L on strict '>')
//   m2:         x[j] = e + max(x[j-1], m[j-1] + o),  m[j] = max(t[j], x[j]),  t = max(d, y)
//               => x[j] - e*j = max(boundary, max_{k<j}(t[k] + o - e*k))      (needs o <= 0)
#include "rg_device.hpp"
#include "rg_poa_args.hpp"

namespace rg {

namespace {

constexpr int NEGB = INT32_MIN / 4;

__device__ __forceinline__ int scb(const DevScores& sc, int a, int b) { return sc.t[a * 6 + b]; }

// utils.rs:17-72 with simd_version = false
__device__ void band_plain(unsigned long long ms, unsigned long long me, int r_val, unsigned long long seq_len,
                           unsigned long long bta, int& left, int& right) {
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
    left = (int)band_start;
    right = (int)band_end;
}

// direction codes of bitfield_path.rs:3-15 that these modes use
enum : uint32_t { PD_O = 0, PD_D = 1, PD_d = 2, PD_L = 3, PD_U = 4 };

}  // namespace

// kGap = false: scalar -m 0;  kGap = true: -m 2.
// Arena planes per read (cap_cells each): m | y (m2) ; path words: w0 = pred<<3 | dir | X<<31, w1 = predY<<1 | Y.
template <bool kGap, bool kLdsRead>
// (8 waves per SIMD — 64 VGPRs — were tried like in rg_poa.hip: 6 registers spill and config 3 loses 2-8 %: not bound by
// occupancy either.  profiles/r04_notes.md)
__global__ __launch_bounds__(64) void k_poa_banded(PoaArgs a) {
    const int slot = blockIdx.x;              // arena slot of this launch
    const int rd = a.read_base + slot;        // read of the batch
    const int lane = threadIdx.x;
    const DevLnz& g = a.g;
    const int L = g.L;
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    const uint8_t* gread = a.reads + ro - 1;
    DevRecord* rec = a.rec + rd;
    const int W = n + 1;
    // score table and read codes in LDS (per-lane lookups every row; as kernel-argument / global loads each one is a
    // dependent memory round trip), wave-uniform graph tables through the scalar cache (uload)
    extern __shared__ int pb_lds[];
    int* sct = pb_lds;
    uint8_t* lread = reinterpret_cast<uint8_t*>(pb_lds + 36);
    if (lane < 36) sct[lane] = a.sc.t[lane];
    if (kLdsRead)
        for (int jj = 1 + lane; jj <= n; jj += WAVE) lread[jj] = gread[jj];
    __syncthreads();
    auto read_at = [&](int jj) -> int { return kLdsRead ? (int)lread[jj] : (int)gread[jj]; };
    if (a.bad[rd]) {
        if (lane == 0) { rec->status = ST_BAD_BASE; rec->n_ops = 0; rec->score = 0; }
        return;
    }
    constexpr int planes = kGap ? 2 : 1;
    int* am = a.arena_m + (long long)slot * a.cap_cells * planes;
    int* ay = am + a.cap_cells;
    uint32_t* pw0 = a.arena_pw + (long long)slot * a.cap_cells * planes;
    uint32_t* pw1 = pw0 + a.cap_cells;
    int4* rinfo = a.rinfo + (long long)slot * L;
    const unsigned long long bta = (unsigned long long)a.bta[rd];
    const int GAP = 5;
    const int o = a.gap_open, e = a.gap_ext;
    long long off = 0;
    unsigned long long ncells = 0;
    uint32_t status = 0;
    bool overflow = false;

    // The previous row travels in registers (chunk k, lane l holds band cell 64 k + l = column p_left + 64 k + l) when it
    // fits KC chunks: a row whose only predecessor is the row above then needs no load, no barrier, and its stores are
    // fire and forget.  Rows with other predecessors take the memory path; a barrier orders the earlier stores first.
    constexpr int KC = 8;
    int pvm[KC] = {}, pvy[kGap ? KC : 1] = {};
    int p_left = 0, p_right = 0, p_best = 0, p_off = 0;
    bool p_valid = false, dirty = false;
    // (the empty asm statements keep the selects apart: left alone the compiler fuses a chain of them into a dynamically
    // indexed vector which it keeps in SCRATCH — 48 bytes per lane, stores every row and dependent loads every chunk of the
    // fast path; found in the ISA in round 4, see rg_poa.hip)
    auto chunk_of = [&](const int (&arr)[KC], int k) -> int {
        int r = arr[0];
#pragma unroll
        for (int kk = 1; kk < KC; ++kk) { r = k == kk ? arr[kk] : r; asm volatile("" : "+v"(r)); }
        return r;
    };
#ifdef RG_POA_TWO_BPERMUTE
    // stored cell of the row above at absolute column `col` (band-relative index col - p_left; chunk base k0 uniform)
    auto prev_at = [&](int col, int k0, int lo, int hi) -> int {
        const int idx = col - p_left;
        const int sh_lo = __shfl(lo, idx & (WAVE - 1), WAVE), sh_hi = __shfl(hi, idx & (WAVE - 1), WAVE);
        return (idx >> 6) == k0 ? sh_lo : sh_hi;
    };
#endif

    // per-row metadata: one 16-byte scalar load, a row ahead (PoaArgs::rowmeta_b; as six loads at the top of the row — one of
    // them dependent — every row waited for the scalar cache several times)
    int n_pb = 0;
    int4 n_meta = uload4(a.rowmeta_b);
    for (int i = 0; i + 1 < L; ++i) {
        const int pb = n_pb, pe = n_meta.x, m_rv = n_meta.y, m_minp = n_meta.z, m_p0 = (n_meta.w & 0xffffff) - 1, m_li = n_meta.w >> 24;
        n_pb = pe;
        if (i + 2 < L) n_meta = uload4(a.rowmeta_b + i + 1);
        const bool nwp = pe > pb;
        const bool only_prev = i > 0 && (!nwp || (pe - pb == 1 && m_p0 == i - 1));
        unsigned long long ms = 0, me = 0;
        if (i > 0) {
            if (only_prev) { unsigned long long pl = (unsigned long long)p_best; ms = pl + 1; me = pl + 1; }
            else {
                if (dirty) { __syncthreads(); dirty = false; }
                unsigned long long pl = 0, pr = 0;
                for (int ee = pb; ee < pe; ++ee) {
                    unsigned long long cb = (unsigned long long)rinfo[g.pred_rows[ee]].w;
                    if (ee == pb) { pl = cb; pr = cb; }
                    if (cb < pl) pl = cb;
                    if (cb > pr) pr = cb;
                }
                ms = pl + 1; me = pr + 1;
            }
        }
        int left, right;
        band_plain(ms, me, m_rv, (unsigned long long)W, bta, left, right);
        if (right <= left) { status |= ST_WOULD_PANIC; break; }   // empty row: m[i][best_val_pos] out of range
        const int width = right - left;
        if (off + width > a.cap_cells) { overflow = true; break; }
        const int li = m_li;
        const int minp = m_minp;
        const bool fast = only_prev && p_valid && width <= KC * WAVE;
        if (i > 0 && !fast && dirty) { __syncthreads(); dirty = false; }
        int keep_m[KC] = {}, keep_y[kGap ? KC : 1] = {};
        int ci = 0;
        // carries of the scans across 64-column chunks
        int carry_z = NEGB;       // running prefix max (exclusive) of the scan variable
        int carry_G = 0;          // m0 scalar: prefix sum of gap costs
        int carry_t = NEGB;       // m2: t of the last column of the previous chunk (for the X flag)
        int carry_zprev = NEGB;   // m2: exclusive prefix max at the previous chunk's last column
        long long best_key = ((long long)INT32_MIN) * 4294967296ll;
        for (int cb = 0; cb < width; cb += WAVE, ++ci) {
            const int j = cb + lane;
            const bool act = j < width;
            const int c = left + j;
            const int rc = (act && c >= 1) ? read_at(c) : 4;
            // fast path: m[i-1] at columns c-1 and c (and y[i-1] at c) from the registers of the row above
            int f_md = 0, f_mu = 0, f_yu = 0;
            if (fast) {
                const int k0 = (left + cb - 1 - p_left) >> 6;          // columns c-1 .. c of this chunk span chunks k0, k0+1
                const int ka = k0 < 0 ? 0 : (k0 < KC ? k0 : KC - 1), kb = k0 + 1 < 0 ? 0 : (k0 + 1 < KC ? k0 + 1 : KC - 1);
                const int mlo = chunk_of(pvm, ka), mhi = chunk_of(pvm, kb);
#ifdef RG_POA_TWO_BPERMUTE
                f_md = prev_at(c - 1, k0, mlo, mhi);
                f_mu = prev_at(c, k0, mlo, mhi);
#else
                // ONE cross-lane fetch per plane (round 6; two ds_bpermute per value and three values per chunk before).  The
                // lanes of a chunk read consecutive band cells: lane l reads cell base + l — a ROTATION of the lanes by
                // rot = base mod 64 over the two register chunks k0 (cells whose lane is >= rot) and k0 + 1 (the others).  So
                // the SOURCE lane selects which of its two chunk values is wanted (a lane is read exactly once), one
                // ds_bpermute rotates, and column c (the cell behind column c - 1) is the next lane's fetch: a one-lane wave
                // shift, with lane 63's value — cell base + 64: lane rot of chunk k0 + 1 — through a v_readlane.
                const int base = left + cb - 1 - p_left;               // band cell of column c - 1 for lane 0 (wave-uniform)
                const int rot = base & (WAVE - 1);
                const int rsel = ((rot + lane) & (WAVE - 1)) * 4;      // byte address of the lane ds_bpermute reads
                f_md = __builtin_amdgcn_ds_bpermute(rsel, lane >= rot ? mlo : mhi);
                {
                    const int last = __builtin_amdgcn_readlane(mhi, rot);
                    const int sh = dpp_mov<0x130, 0xf>(0, f_md);       // wave_shl:1 — lane l takes lane l + 1's value
                    f_mu = lane == WAVE - 1 ? last : sh;
                }
#endif
                if (kGap) {
                    int ylo = pvy[0], yhi = pvy[0];
#pragma unroll
                    for (int kk = 1; kk < (kGap ? KC : 1); ++kk) {
                        ylo = ka == kk ? pvy[kGap ? kk : 0] : ylo; yhi = kb == kk ? pvy[kGap ? kk : 0] : yhi;
                        asm volatile("" : "+v"(ylo), "+v"(yhi));
                    }
#ifdef RG_POA_TWO_BPERMUTE
                    f_yu = prev_at(c, k0, ylo, yhi);
#else
                    {
                        // column c of y: band cell base + 1 + l — the same rotation one cell further
                        const int rot1 = (base + 1) & (WAVE - 1);
                        const int rsel1 = ((rot1 + lane) & (WAVE - 1)) * 4;
                        const int k1 = (base + 1) >> 6;               // chunk of lane 0's cell; k1 == k0 except when rot == 63
                        // (rot == 63: base + 1 is a multiple of 64, every lane's cell lies in chunk k0 + 1 and every lane is >= rot1 = 0)
                        const int ylo1 = k1 == k0 ? ylo : yhi;
                        f_yu = __builtin_amdgcn_ds_bpermute(rsel1, lane >= rot1 ? ylo1 : yhi);
                    }
#endif
                }
            }
            const int first_prev = __builtin_amdgcn_readlane(pvm[0], 0);   // first stored cell of the row above
            // ---- candidates from the predecessor rows ----
            int d = 0, u = 0, dp = minp, up = minp;
            bool have_d = false, have_u = false;
            int uy = 0, uyp = minp;      // m2: best y of covering predecessors
            bool fixed = false;          // cell fully determined without the left chain
            int mval = 0;
            uint32_t w0 = 0, w1 = 0;
            int tval = NEGB;             // non-left candidate of the cell (max(d,u) / max(d,y))
            uint32_t tw0 = 0;            // its path word
            int yval = 0;
            if (act) {
                if (i == 0 && j == 0) { fixed = true; mval = 0; w0 = PD_O; }
                else if (i == 0) {
                    if (kGap) { fixed = true; mval = o + e * c; yval = mval; w0 = (0u << 3) | PD_L; }
                    // scalar m0 row 0 is a pure left chain: handled by the scan with tval = NEG
                } else if (j == 0 && left == 0) {
                    fixed = true;
                    if (kGap) { mval = o + e * (minp + 1); w0 = ((uint32_t)minp << 3) | PD_U; }
                    else {
                        const int first = (fast && minp == i - 1) ? first_prev : am[rinfo[minp].x];
                        mval = first + sct[(GAP) * 6 + (li)];       // m[best_p][0] band-relative (:316)
                        w0 = ((uint32_t)minp << 3) | PD_U;
                    }
                } else if (fast) {
                    const int p = i - 1;
                    if (c > p_left && c <= p_right) { d = f_md; dp = p; have_d = true; }     // diagonal: left_p < c <= right_p
                    if (c >= p_left && c < p_right) {                                         // up: left_p <= c < right_p
                        if (kGap) { u = f_mu + o; up = p; uy = f_yu; uyp = p; }
                        else { u = f_mu; up = p; }
                        have_u = true;
                    }
                } else {
                    const int np = nwp ? pe - pb : 1;
                    for (int q = 0; q < np; ++q) {
                        const int p = nwp ? g.pred_rows[pb + q] : i - 1;
                        const int4 rp = rinfo[p];
                        if (c > rp.y && c <= rp.z) {                 // diagonal: left_p < c <= right_p
                            const int v = am[rp.x + (c - rp.y) - 1];
                            if (!have_d || v > d) { d = v; dp = p; }
                            have_d = true;
                        }
                        if (c >= rp.y && c < rp.z) {                 // up: left_p <= c < right_p
                            if (kGap) {
                                const int vm = am[rp.x + (c - rp.y)] + o, vy = ay[rp.x + (c - rp.y)];
                                if (!have_u) { u = vm; up = p; uy = vy; uyp = p; }
                                if (vm > u) { u = vm; up = p; }
                                if (vy > uy) { uy = vy; uyp = p; }
                            } else {
                                const int v = am[rp.x + (c - rp.y)];
                                if (!have_u || v > u) { u = v; up = p; }
                            }
                            have_u = true;
                        }
                    }
                }
            }
            if (!kGap) {
                // ---------------- scalar m0 (global_abpoa.rs:318-390) ----------------
                int gc = 0;
                bool chain = false;           // cell takes part in the left chain
                int lfb = NEGB;               // left fallback of a band-left cell (:337-339)
                if (act && !fixed) {
                    if (i == 0) { chain = true; gc = sct[(GAP) * 6 + (rc)]; tval = NEGB; }   // key ('-', base) (:307)
                    else {
                        const int dv = have_d ? d + sct[(li) * 6 + (rc)] : sct[(li) * 6 + (GAP)] * (i + left);
                        const int uv = have_u ? u + sct[(li) * 6 + (GAP)] : sct[(li) * 6 + (GAP)] * (i + left + j);
                        const int dpp = have_d ? dp : minp, upp = have_u ? up : minp;
                        if (dv < uv) { tval = uv; tw0 = ((uint32_t)(upp & 0xffff) << 3) | PD_U; }
                        else { tval = dv; tw0 = ((uint32_t)(dpp & 0xffff) << 3) | (rc != li ? PD_d : PD_D); }
                        gc = sct[(rc) * 6 + (GAP)];
                        if (j > 0) chain = true;
                        else lfb = sct[(rc) * 6 + (GAP)] * (i + left + j);
                    }
                }
                // scan: z = value - G
                const int G = dpp_incl_sum(act ? gc : 0) + carry_G;
                int v0 = tval;                // value of a cell as a chain SOURCE
                uint32_t v0w = tw0;
                if (fixed) { v0 = mval; v0w = w0; }
                if (act && !fixed && !chain && i > 0) {   // band-left cell with left > 0: l is a constant
                    if (lfb > tval) { v0 = lfb; v0w = ((uint32_t)(minp & 0xffff) << 3) | PD_L; }
                }
                const int y = act ? v0 - G : NEGB;
                const int zi = dpp_incl_max(y, NEGB);
                int zprev = dpp_shr1(zi, NEGB);
                zprev = lane == 0 ? carry_z : max(zprev, carry_z);
                if (act) {
                    int v = v0;
                    uint32_t w = v0w;
                    if (chain && zprev > y) { v = zprev + G; w = ((uint32_t)(i & 0xffff) << 3) | PD_L; }
                    if (chain && i == 0) { v = zprev + G; w = (0u << 3) | PD_L; }
                    am[off + j] = v;
                    pw0[off + j] = w;
                    mval = v;
                    if (!fixed && i > 0) ncells += 0;  // counted below per wave
                }
                carry_z = max(carry_z, __builtin_amdgcn_readlane(zi, WAVE - 1));
                carry_G = __builtin_amdgcn_readlane(G, WAVE - 1);
#pragma unroll
                for (int k = 0; k < KC; ++k) { keep_m[k] = ci == k ? mval : keep_m[k]; asm volatile("" : "+v"(keep_m[k])); }
            } else {
                // ---------------- m2 (gap_global_abpoa.rs:67-196) ----------------
                const bool fixed0 = act && i > 0 && j == 0 && left == 0;   // first column (:78-92)
                const bool general = act && i > 0 && !fixed0;
                int dv = NEGB, up_pred = minp, tcur = NEGB, xb = NEGB;
                bool fromy = false;
                if (act && i == 0) { mval = j == 0 ? 0 : o + e * c; yval = mval; w0 = j == 0 ? (uint32_t)PD_O : ((0u << 3) | PD_L); }
                if (fixed0) { mval = o + e * (minp + 1); xb = mval; w0 = ((uint32_t)(minp & 0xffff) << 3) | PD_U; yval = 0; }
                if (general) {
                    if (have_u) {                                   // get_best_u (:296-346)
                        if (uy > u) { yval = uy + e; up_pred = uyp; fromy = true; }
                        else { yval = u + e; up_pred = up; }
                    } else { yval = 2 * o + e * (minp + 1) + e * c; up_pred = minp; }   // (:139)
                    if (have_d) dv = d + sct[(li) * 6 + (rc)];
                    tcur = max(dv, yval);
                    if (j == 0) xb = 2 * o + e * (minp + 1) + e * c;                    // (:117)
                }
                // x[j] - e*j = max over k < j of max(t[k] + o, xb[k]) - e*k   (x[j] = e + max(x[j-1], m[j-1] + o), o <= 0)
                int zsrc = NEGB;
                if (fixed0 || general) zsrc = max(tcur > NEGB ? tcur + o : NEGB, xb) - e * j;
                const int zi = dpp_incl_max(zsrc, NEGB);
                int ze = dpp_shr1(zi, NEGB);
                ze = lane == 0 ? carry_z : max(ze, carry_z);
                int xval = NEGB;
                if (fixed0 || general) xval = j == 0 ? xb : ze + e * j;
                // path_x = 'X' iff x[j-1] > m[j-1] + o, i.e. (o < 0) x[j-1] > t[j-1] + o   (:350-368)
                int xprev = dpp_shr1(xval, NEGB), tprev = dpp_shr1(tcur, NEGB);
                if (lane == 0) { xprev = carry_zprev; tprev = carry_t; }
                const bool xflag = general && j > 0 && o != 0 && xprev > (tprev > NEGB ? tprev + o : NEGB);
                if (general) {
                    const int l = xval, uu = yval;
                    const uint32_t lpred = (uint32_t)((j > 0 ? i : minp) & 0xffff);
                    int mv; uint32_t w;
                    if (have_d) {                                    // (:145-180)
                        if (dv < l) {
                            if (l < uu) {
                                if (up_pred == 0) status |= ST_WOULD_PANIC;     // set_path_cell(_, 'u') panics
                                mv = uu; w = ((uint32_t)(up_pred & 0xffff) << 3) | PD_U;
                            } else { mv = l; w = (lpred << 3) | PD_L; }
                        } else {
                            if (dv < uu) { mv = uu; w = ((uint32_t)(up_pred & 0xffff) << 3) | PD_U; }
                            else { mv = dv; w = ((uint32_t)(dp & 0xffff) << 3) | (rc == li ? PD_D : PD_d); }
                        }
                    } else {                                         // (:181-194)
                        if (l < uu) { mv = uu; w = ((uint32_t)(up_pred & 0xffff) << 3) | PD_U; }
                        else { mv = l; w = (lpred << 3) | PD_L; }
                    }
                    mval = mv;
                    w0 = w | (xflag ? 0x80000000u : 0u);
                    w1 = fromy ? (((uint32_t)(up_pred & 0xffff) << 1) | 1u) : 0u;
                }
                if (act) {
                    am[off + j] = mval;
                    ay[off + j] = yval;
                    pw0[off + j] = w0;
                    pw1[off + j] = w1;
                }
                carry_zprev = __builtin_amdgcn_readlane(xval, WAVE - 1);     // x of the chunk's last column
                carry_t = __builtin_amdgcn_readlane(tcur, WAVE - 1);
                carry_z = max(carry_z, __builtin_amdgcn_readlane(zi, WAVE - 1));
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    keep_m[k] = ci == k ? mval : keep_m[k];
                    asm volatile("" : "+v"(keep_m[k]));
                    if (kGap) { keep_y[kGap ? k : 0] = ci == k ? yval : keep_y[kGap ? k : 0]; asm volatile("" : "+v"(keep_y[kGap ? k : 0])); }
                }
            }
            // best_scoring_pos: last column attaining the row maximum
            const int cmx = __builtin_amdgcn_readlane(dpp_incl_max(act ? mval : INT32_MIN, INT32_MIN), WAVE - 1);
            if (cmx >= (int)(best_key >> 32)) {
                const unsigned long long at = __ballot(act && mval == cmx);
                best_key = (long long)cmx * 4294967296ll + (long long)(unsigned)(cb + 63 - __clzll((long long)at));
            }
        }
        // general cells of this row (reference's unit of work)
        {
            int general = 0;
            if (i > 0) general = width - (left == 0 ? 1 : 0);
            ncells += (unsigned long long)general;
        }
        if (lane == 0) rinfo[i] = make_int4((int)off, left, right, (int)(best_key & 0xffffffffll) + left);
        p_best = (int)(best_key & 0xffffffffll) + left; p_left = left; p_right = right; p_off = (int)off;
        p_valid = width <= KC * WAVE;
#pragma unroll
        for (int k = 0; k < KC; ++k) { pvm[k] = keep_m[k]; if (kGap) pvy[kGap ? k : 0] = keep_y[kGap ? k : 0]; }
        off += width;
        dirty = true;
    }
    __syncthreads();
    // combine status bits raised by any lane
    for (int dd = WAVE / 2; dd >= 1; dd >>= 1) status |= __shfl_xor(status, dd, WAVE);
    if (overflow) {
        if (lane == 0) { rec->status = ST_OVERFLOW; rec->n_ops = 0; }
        return;
    }
    if (status & ST_WOULD_PANIC) {
        if (lane == 0) { rec->status = status; rec->n_ops = 0; rec->score = 0; }
        return;
    }
    if (lane != 0) return;

    // ---- end node (global_abpoa.rs:397-405, gap_global_abpoa.rs:206-214) ----
    int last_row = L - 2;
    int4 rl = rinfo[last_row];
    int last_col = rl.z - rl.y - 1;
    int bestv = am[rl.x + last_col];
    for (int ee = g.pred_off[L - 1]; ee < g.pred_off[L]; ++ee) {
        const int p = g.pred_rows[ee];
        const int4 rp = rinfo[p];
        const int tl = rp.z - rp.y - 1;
        const int v = am[rp.x + tl];
        if (v > bestv) { bestv = v; last_row = p; last_col = tl; }
    }
    auto word0 = [&](int r, int cidx) -> uint32_t { return pw0[rinfo[r].x + cidx]; };
    auto word1 = [&](int r, int cidx) -> uint32_t { return pw1[rinfo[r].x + cidx]; };
    auto widthof = [&](int r) { const int4 q = rinfo[r]; return q.z - q.y; };
    // j_pos of the reference: column of (row, col) translated into pred's band; false = usize wrap
    auto jpos = [&](int row, int col, int pred, int& out) -> bool {
        const int lr = rinfo[row].y, lp = rinfo[pred].y;
        if (lp < lr) { out = col + (lr - lp); return true; }
        if (col < lp - lr) return false;
        out = col - (lp - lr);
        return true;
    };
    // ---- band_ampl_enough (global_abpoa.rs:428-476, gap_global_abpoa.rs:371-455) ----
    {
        int i = last_row, j = last_col;
        bool ok = true;
        int guard = 0;
        while (true) {
            if (++guard > 4 * (L + W)) { status |= ST_WOULD_PANIC; break; }
            if (i < 0 || i >= L - 1 || j < 0 || j >= widthof(i)) { status |= ST_WOULD_PANIC; break; }
            const uint32_t w = word0(i, j);
            const uint32_t dir = w & 7u;
            if (dir == PD_O) break;
            const int4 ri = rinfo[i];
            if (i == 0 || (j == 0 && ri.y == 0)) break;
            if ((j == 0 && ri.y != 0) || (j == ri.z - ri.y - 1 && ri.z != W)) { ok = false; break; }
            const int pred = (int)((w >> 3) & 0xffffu);
            if (dir == PD_D || dir == PD_d) {
                int jp;
                if (!jpos(i, j, pred, jp) || jp == 0) { status |= ST_WOULD_PANIC; break; }
                j = jp - 1; i = pred;
            } else if (dir == PD_L) {
                if (kGap && (w >> 31)) { while (j > 0 && j < widthof(i) && (word0(i, j) >> 31)) j -= 1; }
                else j -= 1;
            } else if (dir == PD_U) {
                if (kGap && (word1(i, j) & 1u)) {
                    bool bad = false;
                    while (true) {
                        if (j < 0 || j >= widthof(i)) { bad = true; break; }
                        const uint32_t y1 = word1(i, j);
                        if (!(y1 & 1u)) break;
                        const int p = (int)(y1 >> 1);
                        int jp;
                        if (!jpos(i, j, p, jp)) { bad = true; break; }
                        j = jp; i = p;
                    }
                    if (bad) { status |= ST_WOULD_PANIC; break; }
                } else {
                    int jp;
                    if (!jpos(i, j, pred, jp)) { status |= ST_WOULD_PANIC; break; }
                    j = jp; i = pred;
                }
            } else { if (kGap) { ok = false; } else { status |= ST_WOULD_PANIC; } break; }
        }
        if (!ok) status |= ST_BAND_WARNING;
    }
    // ---- traceback (gaf_output.rs:124-213 / 280-344) ----
    uint8_t* ops = a.ops + (long long)rd * a.ops_stride;
    int32_t* orow = a.oprows + (long long)rd * a.ops_stride;
    int nops = 0;
    int row = last_row, col = last_col;
    if (!(status & ST_WOULD_PANIC)) {
        int guard = 0;
        while (true) {
            if (++guard > 4 * (L + W) || nops + 2 >= a.ops_stride) { status |= ST_WOULD_PANIC; break; }
            if (row < 0 || row >= L - 1 || col < 0 || col >= widthof(row)) { status |= ST_WOULD_PANIC; break; }
            const uint32_t w = word0(row, col);
            const uint32_t dir = w & 7u;
            if (dir == PD_O) break;
            const int pred = (int)((w >> 3) & 0xffffu);
            int jp = 0;
            const bool jp_ok = jpos(row, col, pred, jp);
            if (dir == PD_D || dir == PD_d) {
                if (!jp_ok || jp == 0) { status |= ST_WOULD_PANIC; break; }
                ops[nops] = OP_D | (dir == PD_d ? 0x40 : 0); orow[nops] = pred; ++nops;
                row = pred; col = jp - 1;
            } else if (dir == PD_L) {
                if (kGap && (w >> 31)) {
                    bool first = true, bad = false;
                    while (true) {
                        if (col < 0 || col >= widthof(row)) { bad = true; break; }
                        if (!(word0(row, col) >> 31)) break;
                        if (col == 0 || nops + 2 >= a.ops_stride) { bad = true; break; }
                        ops[nops] = OP_L | (first ? 0 : OP_CONT); orow[nops] = -1; ++nops; first = false;
                        col -= 1;
                    }
                    if (bad) { status |= ST_WOULD_PANIC; break; }
                } else {
                    if (col == 0) { status |= ST_WOULD_PANIC; break; }
                    ops[nops] = OP_L; orow[nops] = -1; ++nops; col -= 1;
                }
            } else if (dir == PD_U) {
                if (kGap && (word1(row, col) & 1u)) {
                    bool first = true, bad = false;
                    while (true) {
                        if (col < 0 || col >= widthof(row)) { bad = true; break; }
                        const uint32_t y1 = word1(row, col);
                        if (!(y1 & 1u)) break;
                        const int p = (int)(y1 >> 1);
                        int jq;
                        if (!jpos(row, col, p, jq) || nops + 2 >= a.ops_stride) { bad = true; break; }
                        ops[nops] = OP_U | (first ? 0 : OP_CONT); orow[nops] = p; ++nops; first = false;
                        col = jq; row = p;
                    }
                    if (bad) { status |= ST_WOULD_PANIC; break; }
                } else {
                    if (!jp_ok) { status |= ST_WOULD_PANIC; break; }
                    ops[nops] = OP_U; orow[nops] = pred; ++nops;
                    row = pred; col = jp;
                }
            } else { status |= ST_WOULD_PANIC; break; }
        }
    }
    rec->status = status;
    rec->score = bestv;
    rec->fscore = (float)bestv;
    rec->end_row = last_row;
    rec->end_col = last_col + rinfo[last_row].y;   // query_end = last_col + left(last_row)
    rec->stop_row = row;
    rec->stop_col = col;                           // query_start = band-relative col where the walk stopped
    rec->n_ops = (status & ST_WOULD_PANIC) ? 0 : nops;
    rec->n_fwd_ops = 0;
    atomicAdd(a.cells, ncells);
}

template <bool kGap>
static void launch_banded(const PoaArgs& a, hipStream_t s) {
    const size_t bytes = 36 * sizeof(int) + (a.lds_read ? (((size_t)a.max_n + 2 + 3) & ~(size_t)3) : 0);
    if (a.lds_read) hipLaunchKernelGGL((k_poa_banded<kGap, true>), dim3(a.nreads), dim3(64), bytes, s, a);
    else hipLaunchKernelGGL((k_poa_banded<kGap, false>), dim3(a.nreads), dim3(64), bytes, s, a);
}
void launch_m2(const PoaArgs& a, hipStream_t s) { launch_banded<true>(a, s); }
void launch_m0_scalar(const PoaArgs& a, hipStream_t s) { launch_banded<false>(a, s); }

}  // namespace rg
