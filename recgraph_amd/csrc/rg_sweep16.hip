// Packed 16-bit DP sweep for gfx950 (-m 4 / -m 5 / -m 8 / -m 9): k_sweep of rg_pathwise.hip with two DP columns per
// 32-bit register (v_pk_add_i16 / v_pk_max_i16 / v_bfi_b32), selected by the driver when every absolute score of
// the batch provably fits 16 bits and the read-gap cost is uniform (every matrix the reference's CLI can build).
//
// Why: the i32 sweep is bound by the rolling rows (one row store + one row load per member-path update; at config 5
// the resident set of one launch, 2048 reads x 32 paths x 1024 columns x 4 B = 268 MB, does not fit the 256 MB
// Infinity Cache and streams through HBM) and by VALU work.  Packing halves the bytes (134 MB: cache resident) and
// roughly halves the VALU instructions per cell.
//
// Layout: lane t owns columns t*C .. t*C+C-1 as in k_sweep; register r of a row (r < H = C/2) holds column
// t*C + r in its low half and column t*C + H + r in its high half, so the lane-local serial recurrences run as TWO
// independent chains per instruction (low halves: columns 0..H-1 of the lane, high halves: H..C-1); the chains are
// stitched with 32-bit per-lane arithmetic (a dozen instructions per row update) and one wave-level DPP prefix max
// exactly like the i32 kernel.  Rows in HBM are [path][r][lane] packed words.
//
// Rows are kept in "z-space": the word of column c holds z = A - c * g (g = the uniform read-gap cost, what an L move
// adds per column).  An L run then copies z unchanged (no add in the left scan and in the members' fill-forward), the
// diagonal adds s - g, and the cross-lane stitching needs no column arithmetic.  z >= A (g <= 0): the admission test
// bounds it; every output converts back (A = z + c * g).
//
// Outputs (direction words, column maxima, candidates, sink values, semiglobal end rows) are identical to k_sweep's;
// the argument recorded for a column maximum may be a different cell with the same value (k_bound only needs a real
// cell: any candidate pair gives a valid lower bound).  Everything downstream is shared.
#include <algorithm>
#include <type_traits>

#include "rg_path_kernels.hpp"

namespace rg {

namespace {

constexpr int NEG32 = INT32_MIN / 4;
constexpr int NEG16 = -30000;                               // "minus infinity" of a 16-bit lane; real values stay > -24000
constexpr int NEGPAIR = (int)(((unsigned)(NEG16 & 0xffff) << 16) | (unsigned)(NEG16 & 0xffff));
constexpr int ONE2 = 0x00010001;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// buffer resource over [base, base + bytes) built from wave-uniform values only (the halves of the pointer go through
// readfirstlane so that the compiler can PROVE it: otherwise every buffer operation becomes a waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, unsigned bytes) {
    const unsigned long long u = (unsigned long long)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    void* p = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int pk_add(int a, int b) {
    return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ int pk_sub(int a, int b) {
    return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ int pk_max(int a, int b) {
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ int pk_min(int a, int b) {
    return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ int pk_minu(int a, int b) {
    return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// saturating packed difference (v_pk_sub_i16 ... clamp): the sign of each half is the sign of the true difference
__device__ __forceinline__ int pk_sub_sat(int a, int b) {
    return __builtin_bit_cast(int, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ int pk_add_sat(int a, int b) {
    return __builtin_bit_cast(int, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// per half: 0xffff where the half of `v` is negative, else 0  (v_pk_ashrrev_i16)
__device__ __forceinline__ int pk_sign(int v) {
    return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, v) >> (s16x2)(15)));
}
__device__ __forceinline__ int pack16(int lo, int hi) { return (int)(((unsigned)hi << 16) | ((unsigned)lo & 0xffffu)); }
__device__ __forceinline__ int lo16(int v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int hi16(int v) { return v >> 16; }
// per half: mask ? a : b   (mask halves are 0 or 0xffff)
__device__ __forceinline__ int bfi(int mask, int a, int b) {
    return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xCA);   // (mask & a) | (~mask & b) in ONE v_bitop3_b32
}

// Row operators on packed z-space rows.  MU / ML: per register, 0xffff in the halves whose column took U (not D) / L.
// s[] holds the packed pairs (s - g): z_d = z_prev + (s - g), z_u = z_old + g_i, z_l = z_left.
template <int C>
struct RowOps16 {
    static constexpr int H = C / 2;
    // bit r = column r of the lane (low half), bit 16 + r = column H + r (high half)
    static constexpr unsigned FULL = H >= 16 ? 0xffffffffu : ((((1u << H) - 1u) << 16) | ((1u << H) - 1u));

    // The alpha's row update.  Outputs the new row and the SIGN WORDS of its decisions: per half, XU[r] < 0 where the column took
    // U (not D), XL[r] < 0 where it took L.  What is made of them is the caller's business (round 6): the full 0 / 0xffff masks the
    // members select with (`masks`: sixteen v_pk_ashrrev, needed only by groups with members besides the alpha), the lane of the
    // nearest non-L column to the left (`src_lane`), the direction word that k_layer16 reads (`dir_word`: only for groups
    // that contain a path whose layer can be asked for, see DIRECTION WORDS ON DEMAND in k_sweep16).
    // lmax: per half chain the largest value of the lane's columns in the new row (a by-product of the stitching)
    static __device__ __forceinline__ void alpha(int (&row)[H], const int (&s)[H], int g_i, int g0, int lane, int (&XU)[H], int (&XL)[H], int& lmax) {
        const int GI = pack16(g_i, g_i);
        const int GI0 = lane == 0 ? pack16(g0, g_i) : GI;          // border column 0 adds g0
        int prev = __builtin_amdgcn_alignbit(row[H - 1], dpp_shr1(row[H - 1], NEGPAIR), 16);
        int m[H];
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const int old = row[r];
            const int d = pk_add(prev, s[r]), u = pk_add(old, r == 0 ? GI0 : GI);
            const int du = pk_max(d, u);                            // D on ties (d >= u); lane 0 column 0: d = -inf -> U
            XU[r] = pk_sub(d, du);                                  // d - max(d, u) < 0 where U (|d - u| is a few scores: no wrap)
            row[r] = du;
            m[r] = du;
            prev = old;
        }
        // maxima of the two half chains (a tree: a serial chain of dependent packed instructions pays a wait state per link)
#pragma unroll
        for (int w = 1; w < H; w <<= 1) {
#pragma unroll
            for (int r = 0; r + w < H; r += 2 * w) m[r] = pk_max(m[r], m[r + w]);
        }
        const int run = m[0];
        // stitch: best source inside the lane = max of both chains; left scan over the lanes in z-space is a plain
        // prefix maximum
        const int TL = lo16(run), TH = hi16(run);
        // (INT32_MIN = the identity the compiler's DPP combiner knows for a signed max: each scan step becomes ONE
        // v_max_i32_dpp instead of constant + v_mov_dpp + v_max)
        const int ze = dpp_shr1(dpp_incl_max(max(TH, TL), INT32_MIN), INT32_MIN);   // best source of the lanes to the left
        const int bl = max(ze, NEG16);                              // carry into the lane's first column
        const int bh = max(TL, bl);                                 // carry into column H of the lane
        int vprev = pack16(bl, bh);
        lmax = pk_max(run, vprev);
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const int du = row[r];
            const int v = pk_max(du, vprev);
            XL[r] = pk_sub(du, v);                                  // du - max(du, left) < 0 where L (left strictly better)
            row[r] = v;
            vprev = v;
        }
    }
    // full masks for the members; returns `lall`: per half 0xffff iff EVERY column of that half chain took L.  It stands in for
    // the L bit mask wherever the members test "does this chain have a non-L column" (~lall & FULL is non-zero exactly then)
    static __device__ __forceinline__ unsigned masks(const int (&XU)[H], const int (&XL)[H], int (&MU)[H], int (&ML)[H]) {
        int al = -1;
#pragma unroll
        for (int r = 0; r < H; ++r) { MU[r] = pk_sign(XU[r]); ML[r] = pk_sign(XL[r]); al &= ML[r]; }
        return (unsigned)al;
    }
    // nearest lane to the left that owns a non-L column: highest set bit of the ballot below this lane (0 if none)
    static __device__ __forceinline__ int src_lane(unsigned lall, int lane) {
        const unsigned long long have = __ballot(lall != 0xffffffffu) & ((1ull << lane) - 1ull);
        return have ? 63 - __clzll((long long)have) : 0;
    }
    // bit form of the decisions (bit r = column r of the lane, bit 16 + r = column H + r): what the 32-column variants store
    static __device__ __forceinline__ unsigned sign_bits(const int (&X)[H]) {
        unsigned b = 0;
#pragma unroll
        for (int r = 0; r < H; ++r) b |= ((unsigned)X[r] >> 15 & (unsigned)ONE2) << r;
        return b;
    }
    // DIRECTION WORD of a row at C <= 16 (LayerArgs::dir_fmt 1, decoded by dir16_decode, rg_path_kernels.hpp): register r contributes four
    // flags — U low column, U high column, L low column, L high column — as bit 7 - r of bytes 0 .. 3.  One v_perm collects
    // the four sign bytes of (XU[r], XL[r]), one arithmetic shift (a 2-cycle form) moves them to the register's bit, one
    // v_bitop3 (2-cycle) ORs them in under a mask: 8 + 7 + 8 instructions per row for both masks, where the bit-per-column
    // form took 16 packed shifts, 16 ANDs, 8 three-way ORs and a v_perm.
    static __device__ __forceinline__ unsigned dir_word(const int (&XU)[H], const int (&XL)[H]) {
        static_assert(H <= 8, "one byte per flag kind");
        unsigned w = 0;
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const int p = (int)__builtin_amdgcn_perm((unsigned)XL[r], (unsigned)XU[r], 0x07050301u);
            const int sh = r == 0 ? p : (p >> r);                   // (arithmetic: the copies of the top bit stay above bit 31 - r)
            w = (unsigned)__builtin_amdgcn_bitop3_b32((int)w, sh, (int)(0x80808080u >> r), 0xF8);   // w | (sh & mask)
        }
        return w;
    }

    // SEL[r]: per half what a member adds to the source the alpha chose there (g_i under the U cells, s - g under the D
    // cells); computed once per group that has members besides its alpha
    static __device__ __forceinline__ void select_steps(int (&SEL)[H], const int (&s)[H], const int (&MU)[H], int g_i, int g0, int lane) {
        const int GI = pack16(g_i, g_i);
        const int GI0 = lane == 0 ? pack16(g0, g_i) : GI;
#pragma unroll
        for (int r = 0; r < H; ++r) SEL[r] = bfi(MU[r], r == 0 ? GI0 : GI, s[r]);
    }

    // kMove: pure data movement (no step added): what a gather run applies to its column map
    // A member update is two passes over the row with ONE cross-lane fetch between them (the z of the nearest non-L column to
    // the left: a ds_bpermute, ~100 cycles before its result can be used).  The passes are separate functions so that a group
    // of several members issues every fetch before it waits for the first (the register runs, round 5).
    template <bool kMove = false>
    static __device__ __forceinline__ void member_p1(int (&row)[H], const int (&SEL)[H], const int (&MU)[H], const int (&ML)[H], unsigned lmask,
                                                     int& zl, int& v_lo) {
        int prev = __builtin_amdgcn_alignbit(row[H - 1], dpp_shr1(row[H - 1], NEGPAIR), 16);
        int lastv = NEGPAIR;                                        // z of the last non-L column of each half
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const int old = row[r];
            const int base = kMove ? bfi(MU[r], old, prev) : pk_add(bfi(MU[r], old, prev), SEL[r]);  // U: old + g_i, D: prev + (s - g)
            row[r] = base;
            lastv = bfi(ML[r], lastv, base);
            prev = old;
        }
        // z of the lane's last non-L column, fetched by the lanes to the right that start with L columns
        const unsigned nl = ~lmask & FULL;
        const unsigned nl_lo = nl & 0xffffu, nl_hi = nl >> 16;
        v_lo = lo16(lastv);
        const int v_hi = hi16(lastv);
        zl = nl_hi ? v_hi : (nl_lo ? v_lo : NEG32);
    }
    static __device__ __forceinline__ void member_p2(int (&row)[H], const int (&ML)[H], unsigned lmask, int cur, int v_lo) {
        const unsigned nl_lo = ~lmask & FULL & 0xffffu;
        const int bl = max(cur, NEG16);
        const int bh = nl_lo ? max(v_lo, NEG16) : bl;
        int vprev = pack16(bl, bh);
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const int v = bfi(ML[r], vprev, row[r]);
            row[r] = v;
            vprev = v;
        }
    }
    template <bool kMove = false>
    static __device__ __forceinline__ void member(int (&row)[H], const int (&SEL)[H], int lane,
                                                  const int (&MU)[H], const int (&ML)[H], unsigned lmask, int src) {
        int zl, v_lo;
        member_p1<kMove>(row, SEL, MU, ML, lmask, zl, v_lo);
        member_p2(row, ML, lmask, __shfl(zl, src, WAVE), v_lo);
    }
};

// run_dispatch(f, n): f(std::integral_constant<int, k>) for k = min(n, KMAX) — the register runs of k_sweep16, one loop body per
// member count
template <int K, int KMAX, class F>
__device__ __forceinline__ void run_dispatch_from(F& f, int n) {
    if constexpr (K >= KMAX) f(std::integral_constant<int, (KMAX > 0 ? KMAX : 1)>{});
    else {
        if (n <= K) f(std::integral_constant<int, K>{});
        else run_dispatch_from<K + 1, KMAX>(f, n);
    }
}

}  // namespace

#ifndef RG_SWEEP16_KRUN
#define RG_SWEEP16_KRUN 4
#endif
#ifndef RG_SWEEP16_GATHER
#define RG_SWEEP16_GATHER 1          // gather runs (see k_sweep16 and gather_pays)
#endif
#ifndef RG_SWEEP16_GATHER_FWD
#define RG_SWEEP16_GATHER_FWD 1      // the forward record variant spills 121 registers with them compiled in and still gains 3 ms (47.7 -> 44.7)
#endif

// when a gather run pays (instructions): member by member a row costs PER_MEMBER_ROW per member; a gather run PER_ROW per
// row (alpha + column map + best member per column) and PER_MEMBER_RUN per member once (a pass at each end of the run)
#ifndef RG_GATHER_PER_MEMBER_ROW
#define RG_GATHER_PER_MEMBER_ROW 84
#endif
#ifndef RG_GATHER_PER_ROW
#define RG_GATHER_PER_ROW 160
#endif
#ifndef RG_GATHER_PER_MEMBER_RUN
#define RG_GATHER_PER_MEMBER_RUN 90
#endif
#ifndef RG_SWEEP16_GATHER32
#define RG_SWEEP16_GATHER32 1        // gather runs in the record variants at 32 columns per lane (see kGather): round 6 — 32 spilled registers, and still 21.4 k -> 24.9 k reads/s at 1.5 kbp
#endif
#ifndef RG_SWEEP16_RUNWAIT
#define RG_SWEEP16_RUNWAIT 1
#endif
#ifndef RG_SWEEP16_PROFILE_AHEAD
#define RG_SWEEP16_PROFILE_AHEAD 0    // register runs: the next row's score profile is read from LDS behind this row's member steps
#endif
#ifndef RG_SWEEP16_CHAIN
// chained register runs: next run's loads before this run's stores (see CHAINED RUNS).  1: in the -m 4 / -m 5 variant only
// (config 4: 17.2 -> 14.8 ms per 4096-read sweep); 2: in the record variants of -m 8 too — measured neutral in the forward
// sweep (37.4 vs 37.5 ms, and the variant spills 48 registers with it) and 0.7 ms SLOWER in the reverse one (34.3 vs
// 33.6): their runs end in tails, whose epilogue separates the loads from the stores anyway (profiles/r04_notes.md)
#define RG_SWEEP16_CHAIN 1
#endif
#ifndef RG_SWEEP16_M4_WAVES
#define RG_SWEEP16_M4_WAVES 2
#endif
#ifndef RG_SWEEP16_KRUN_M4
#define RG_SWEEP16_KRUN_M4 3
#endif
#ifndef RG_SWEEP16_KRUN32_M4
#define RG_SWEEP16_KRUN32_M4 1       // register runs of two rows in the -m 4 / -m 5 variant at 32 columns per lane (reads of 1 024 - 2 047 bases)
#endif
#ifndef RG_SWEEP16_KRUN_REV
#define RG_SWEEP16_KRUN_REV RG_SWEEP16_KRUN     // the variant without column maxima (reverse sweep of the record pipeline)
#endif
#ifndef RG_SWEEP16_REV_WAVES
#define RG_SWEEP16_REV_WAVES 2                   // waves per SIMD the compiler must fit that variant into
#endif
#ifndef RG_SWEEP16_FWD_WAVES
#define RG_SWEEP16_FWD_WAVES 2                   // ... and the variants that track column maxima (C <= 16)
#endif
// TIMING-ONLY build variants (tools/sweep_variants.sh; the results of such a build are garbage and nothing ships them):
//   RG_SWEEP16_NOROWS   no rolling-row load / store in the main loop (rows stay whatever the registers hold)
//   RG_SWEEP16_NOKEYS   the best-member keys are not built (row_end runs on constant keys)
//   RG_SWEEP16_NOEMIT   row_end stops after the column maxima (no threshold tests, ballots, record / Cand stores)
//   RG_SWEEP16_NODIRS   no direction-word stores
//   RG_SWEEP16_KRUNNOLD / KRUNNOST   register runs without their run-start loads / run-end stores
#ifdef RG_SWEEP16_NOROWS
#define RG_ROW_LD(k, dst) ((void)0)
#define RG_ROW_ST(k, src) ((void)0)
#elif defined(RG_SWEEP16_NOROWS32)
//   RG_SWEEP16_NOROWS32 no row traffic for the steps whose group holds 16 or more paths (the rows every path visits)
#define RG_ROW_LD(k, dst) do { if (nm < 16) ld_row(k, dst); } while (0)
#define RG_ROW_ST(k, src) do { if (nm < 16) st_row(k, src); } while (0)
#else
#define RG_ROW_LD(k, dst) ld_row(k, dst)
#define RG_ROW_ST(k, src) st_row(k, src)
#endif

// kColmax = 2: per-column maxima as packed VALUES only (8 v_perm + 8 v_pk_max per row instead of 48 compare / select
// instructions and 24 fewer live registers): the forward sweep of the record pipeline — the other sweep's thresholds
// need the exact maxima, while the cell k_bound pairs per column is taken from this sweep's records (k_colmax_rec)
// kColmax = 0: no per-column maxima (the reverse sweep of the record pipeline: its maxima and their cells are
// taken from its own records by k_colmax_rec)
// kRec = true: emissions leave as (row, lane) records (a.frec); false: as Cand entries (a.cand) or not at all
// kWide = true: graphs with more than 64 paths (step entries carry a 64-path page and continuation entries exist); the
// narrow variant compiles that logic out (page 0, no continuation: it costs registers the forward sweep does not have)
// kSemi = true: the semiglobal modes (-m 5 / -m 9: zero first column, per-path end rows).  A template flag since round 5: the
// end-row bookkeeping (four per-lane registers of state, sixteen column-select masks in SGPRs) was carried — spilled — through
// the record loop of every global-mode sweep
// A path set of up to RG_PW 64-bit words as NAMED scalars: as an array (indexed by a page that is only known at run time, or
// passed by reference) the compiler kept it in scratch — a scratch load per step record (round 6, the wide variants).
struct PathWords {
    unsigned long long w0, w1, w2, w3;
    __device__ __forceinline__ unsigned long long get(int i) const { return i == 0 ? w0 : (i == 1 ? w1 : (i == 2 ? w2 : w3)); }
    __device__ __forceinline__ void set(int i, unsigned long long v) { if (i == 0) w0 = v; else if (i == 1) w1 = v; else if (i == 2) w2 = v; else w3 = v; }
};

#ifdef RG_SWEEP16_VGPR_CAP
#define RG_SWEEP16_CAP_ATTR __attribute__((amdgpu_num_vgpr(RG_SWEEP16_VGPR_CAP)))      // (experiments: a hard register budget)
#else
#define RG_SWEEP16_CAP_ATTR
#endif
template <int C, int kColmax, bool kRec, bool kWide, bool kSemi>
// (the -m 4 / -m 5 variants at <= 16 columns per lane are COMPILED for three waves per SIMD: with the path retirement of round 6
// compiled in they would otherwise take 196 registers — from 148 — and config 4 lives on the third wave)
__global__ __launch_bounds__(64, C > 16 ? 2 : ((kColmax == 0 && !kRec) ? RG_SWEEP16_M4_WAVES : (kColmax != 0 ? RG_SWEEP16_FWD_WAVES : RG_SWEEP16_REV_WAVES))) RG_SWEEP16_CAP_ATTR void k_sweep16(SweepArgs a) {
    constexpr int H = C / 2;
    constexpr bool kTrack = kColmax != 0 || kRec;      // <0, false>: the -m 4 / -m 5 sweep — no best member, no thresholds, no emission
    // gather runs: not at 32 columns per lane in the record variants (a row is 16 registers there: the run's A / G / masks / steps /
    // values / paths alone are 112, and with retirement and register runs compiled in the variant spilled 78)
    constexpr bool kGather = C <= 16 || !kRec || RG_SWEEP16_GATHER32;
    // PATH RETIREMENT: the record pipelines of -m 8 (since round 6 also beyond 64 paths) and — round 6 — the -m 4 sweep: there a path is
    // hopeless when its final score cannot reach the bound k_verify4 checks the best final score against (see retire_eval)
    constexpr bool kRet = ((kRec && kColmax != 1) || !kTrack) && !kSemi;
    constexpr int NW = kWide ? RG_PW : 1;                   // 64-bit words of a path set
    // rows kept in registers across the inner rows of a segment: groups of up to 4 paths (2 at 32 columns per lane: a row is 16 registers there)
    // (the -m 4 / -m 5 variant: 3 — with 4 the specialised run loops of round 6 need 178 registers and the variant falls from three
    // waves per SIMD to two: config 4 271 k against 296 k reads/s)
    constexpr int KRUN = C <= 16 ? (!kTrack ? (RG_SWEEP16_KRUN < RG_SWEEP16_KRUN_M4 ? RG_SWEEP16_KRUN : RG_SWEEP16_KRUN_M4) : (kColmax != 0 ? RG_SWEEP16_KRUN : RG_SWEEP16_KRUN_REV)) : (((kRec && kColmax == 0 && !kWide) || (!kTrack && !kWide && RG_SWEEP16_KRUN32_M4)) ? 2 : 0);
    const int rd = a.order ? a.order[blockIdx.x] : blockIdx.x;      // (launch order: see launch_order)
#ifdef RG_SWEEP16_STALLSTAT
    // (statistics build, tools/probes/stall_stat.py: shader-clock cycles a wave spends in the waits for row loads; the cell
    // counters carry  total >> 8 | general-path waits >> 8 << 32  and  run-start waits >> 8 | run starts << 32)
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_run = 0, st_nrun = 0, st_gen = 0;
#define RG_STALL_BEGIN() const unsigned long long st_b = __builtin_amdgcn_s_memtime()
#define RG_STALL_END(acc) do { __builtin_amdgcn_s_waitcnt(0x0F70); acc += __builtin_amdgcn_s_memtime() - st_b; } while (0)
#define RG_STALL_END_LGKM(acc) do { __builtin_amdgcn_s_waitcnt(0xC07F); acc += __builtin_amdgcn_s_memtime() - st_b; } while (0)
#else
#define RG_STALL_BEGIN() ((void)0)
#define RG_STALL_END(acc) ((void)0)
#endif
    const int lane = threadIdx.x;
    const PathGraphDev& g = a.g;
    const int P = g.P;
    const int wpad = C * WAVE;          // columns per read (i32 side buffers)
    const int wrow = H * WAVE;          // packed words per rolling row
    ReadState* rs = a.state + rd;
    const long long ro = a.read_off[rd];
    const int n = __builtin_amdgcn_readfirstlane((int)(a.read_off[rd + 1] - ro));
    if (a.bad[rd] || n + 1 > wpad) {
        if (lane == 0 && !a.rev) { rs->status = a.bad[rd] ? ST_BAD_BASE : ST_WOULD_PANIC; }
        return;
    }
    const uint8_t* read = a.reads + ro - 1;  // read[1..n]
    const bool rev = a.rev;
    const int ncols = rev ? n : n + 1;
    const int GAP = 5;
    extern __shared__ __attribute__((aligned(16))) int lds16[];
    // LDS of one wave: 13 056 bytes at C = 16 in the narrow variants — TWELVE waves (three per SIMD) fit the CU's 160 KB.
    // Round 5 took 5.6 KB out of the 18.7 KB that capped the CU at eight: the retirement constants and the profile of 'N'
    // rows live in two pseudo-rows behind the read's rolling rows (PR_RVL, PR_NPROF: read once per 256 records / per 'N'
    // row), the gather runs re-read the alpha's run-start row from its own rolling row (untouched until the run's end)
    // instead of keeping a copy, and the semiglobal end arrays are sized by the variant's path count.
    constexpr int EP = kWide ? RG_MAXP : 64;
    constexpr int GTW = C * WAVE > 5 * 64 ? C * WAVE : 5 * 64;
    int* sct = lds16;                    // [36] score table
    int* endv = lds16 + 64;              // [EP]
    int* endr = endv + EP;               // [EP]
    // gather runs (below): gT[q][lane] = best (delta << 16 | path) of the run's members at the run start; at the end of the
    // run the same words hold one packed row at a time ([r][lane]: the alpha's row at the run start, then each member's)
    int* gT = endr + EP;                 // [GTW]
    // [5][64] packed score pairs: s2[li*64 + (code_lo | code_hi << 3)] — only read while the profile below is built: it
    // shares the words of gT
    int* s2 = gT;
    // score profile of this read: sprof[(li * 64 + lane) * H + r] = the packed (s - g) pair register r of the lane adds on a
    // diagonal step into a row whose base is li (A, C, G, T) — one 16-byte LDS read per four registers and row instead of a
    // code extraction + table lookup per register (24 VALU instructions per row at C = 16)
    int* sprof = gT + GTW;               // [4][64][H]
    const int PR_RVL = P, PR_NPROF = P + 1;     // pseudo-rows behind the P rolling rows (same [lane][r] layout)
    if (lane < 36) sct[lane] = a.sc.t[lane];
    __syncthreads();
    // uniform gap cost (checked by the launcher, sweep16_admissible: score(b, '-') is the same for b = A, C, G, T, N): the
    // z-space slope — and, being the same table column, what a U move adds in EVERY row (g_i below): no per-row lookup
    const int gcost = __builtin_amdgcn_readfirstlane(sct[GAP]);
    for (int e = lane; e < 5 * 64; e += WAVE) {
        const int li = e >> 6, cl = e & 7, ch = (e >> 3) & 7;
        s2[e] = (cl < 6 && ch < 6) ? pack16(sct[li * 6 + cl] - gcost, sct[li * 6 + ch] - gcost) : 0;   // diagonal step in z-space
    }
    int* rows = a.roll + (long long)rd * (P + 2) * wrow;
    // rolling rows in HBM: [path][lane][r] — the H packed words of a lane are contiguous, so a row moves as 16-byte
    // accesses (two per lane at C = 16; the wave covers the row's 2 KB contiguously) instead of one 4-byte access per word
    // Row accesses are buffer operations: the read's row area as ONE resource descriptor in SGPRs, the row as the scalar
    // offset (k is wave-uniform), the lane's 32-bit byte offset as the only VGPR — no 64-bit pointer arithmetic in VGPRs
    // (the flat form kept `rows + lane offset` and one pointer per access as VGPR pairs and rebuilt them with v_lshl_add_u64)
    const unsigned lane_row_off = (unsigned)lane * (unsigned)(H * sizeof(int));
    const __amdgpu_buffer_rsrc_t rows_rsrc = uniform_rsrc(rows, (unsigned)((P + 2) * wrow) * 4u);
#ifndef RG_SWEEP16_FLATROWS
    auto ld_row = [&](int k, int (&dst)[H]) {
        const int so = k * (int)(wrow * sizeof(int));
        if constexpr (H >= 4) {
#pragma unroll
            for (int r4 = 0; r4 < H / 4; ++r4) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rows_rsrc, (int)lane_row_off + 16 * r4, so, 0);
                dst[4 * r4] = (int)v.x; dst[4 * r4 + 1] = (int)v.y; dst[4 * r4 + 2] = (int)v.z; dst[4 * r4 + 3] = (int)v.w;
            }
        } else {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rows_rsrc, (int)lane_row_off, so, 0);
            dst[0] = (int)v.x; dst[1] = (int)v.y;
        }
    };
    auto st_row = [&](int k, const int (&src)[H]) {
        const int so = k * (int)(wrow * sizeof(int));
        if constexpr (H >= 4) {
#pragma unroll
            for (int r4 = 0; r4 < H / 4; ++r4) {
                const u32x4 v = {(unsigned)src[4 * r4], (unsigned)src[4 * r4 + 1], (unsigned)src[4 * r4 + 2], (unsigned)src[4 * r4 + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(v, rows_rsrc, (int)lane_row_off + 16 * r4, so, 0);
                // gfx950 HAZARD the compiler does not know: a VALU write to the data registers of a buffer_store_dwordx4 in the
                // instruction right behind it changes what the store writes — also when the store takes its scalar offset from
                // an SGPR, the case the gfx9 documents (and LLVM's hazard recognizer: "only if soffset is not a register")
                // exempt from the one wait state.  Round 5's -m 4 variant had `v_and_b32 v126, …` right behind
                // `buffer_store_dwordx4 v[126:129], …, s8 offen`: with twelve waves per CU a quarter of the reads of a
                // 4096-read launch lost row words (tests/test_gpu_full_size.py::test_full_launch_every_wave_slot_vs_oracle;
                // profiles/r05_notes.md).  The asm keeps the data registers alive up to an s_nop behind the store;
                // tools/kernel_resources.py --hazards checks the ISA of every variant for the pattern.
#ifndef RG_SWEEP16_NO_STORE_NOP
                asm volatile("s_nop 1" :: "v"(v));
#endif
            }
        } else {
            const u32x2 v = {(unsigned)src[0], (unsigned)src[1]};
            __builtin_amdgcn_raw_buffer_store_b64(v, rows_rsrc, (int)lane_row_off, so, 0);
        }
    };
#else
    auto ld_row = [&](int k, int (&dst)[H]) {
        const int* p = rows + (long long)k * wrow + lane * H;
        if constexpr (H >= 4) {
#pragma unroll
            for (int r4 = 0; r4 < H / 4; ++r4) {
                const int4 v = reinterpret_cast<const int4*>(p)[r4];
                dst[4 * r4] = v.x; dst[4 * r4 + 1] = v.y; dst[4 * r4 + 2] = v.z; dst[4 * r4 + 3] = v.w;
            }
        } else {
            const int2 v = *reinterpret_cast<const int2*>(p);
            dst[0] = v.x; dst[1] = v.y;
        }
    };
    auto st_row = [&](int k, const int (&src)[H]) {
        int* p = rows + (long long)k * wrow + lane * H;
        if constexpr (H >= 4) {
#pragma unroll
            for (int r4 = 0; r4 < H / 4; ++r4) reinterpret_cast<int4*>(p)[r4] = make_int4(src[4 * r4], src[4 * r4 + 1], src[4 * r4 + 2], src[4 * r4 + 3]);
        } else {
            *reinterpret_cast<int2*>(p) = make_int2(src[0], src[1]);
        }
    };
#endif
    // per-column constants of this lane
    unsigned long long pcode[(H + 7) / 8] = {};   // 8 bits per register: code_lo | code_hi << 3
    // emission threshold << 16 per column (INT32_MAX = never; columns that do not exist)
    int thrk[(kRec || !kTrack) ? 1 : C];        // (record variants test packed values against thz: no per-column key thresholds to keep)
#define THRK(q) thrk[(kRec || !kTrack) ? 0 : (q)]
    int thz[kRec ? H : 1];               // packed (threshold >> 16) pairs, see below
    int minplain2 = 0;
    int thzmin = 0;                      // per half chain of the lane: the lowest of its packed thresholds (signed)
    int next_eval = INT32_MAX;           // PATH RETIREMENT: the record index of the next evaluation (INT32_MAX: off for this read)
    int minthrk = INT32_MAX;             // lowest threshold of the lane
    int minplain = INT32_MAX;            // lowest threshold of the lane without the member rule (rows every path visits)
    const bool tight = a.thr != nullptr || a.oob; // thresholds from the other sweep's column maxima / from the speculative bound
    const int oob = max((int)((float)(n + 1) * (1.0f - a.rbw) / 2.0f), 1);
    // Emission threshold of lane column q as a z-space key (z << 16): keys are z << 16 | path with |z| < 2^15, so a
    // threshold outside that range means always / never.  `member_rule`: additionally require a true value >= 0 with
    // key >= 1, which every cell whose winner is a member path satisfies when some path is NOT through the row (the
    // reference's non-member entries are 0: knm >= 0); rows that every path visits (knm < 0) use the plain threshold.
    auto thr_key = [&](int q, bool member_rule) -> int {
        const int c = lane * C + q;
        const int j = rev ? n - c : c;
        int th = INT32_MAX;
        if (c < ncols && j >= oob && j < n + 1 - oob) {
            if (a.thr) th = a.thr[(long long)rd * wpad + j];
            else if (a.lb) th = a.lb[rd] + a.brc - (n - j) * a.maxmatch;
        }
        if (th == INT32_MAX) return INT32_MAX;
        long long tz = (long long)th - (long long)c * gcost;       // threshold on z
        if (tz > 32767) return INT32_MAX;
        int key = tz < -32768 ? INT32_MIN : (int)tz * 65536;
        if (member_rule) key = max(key, (-c * gcost) * 65536 + 1);  // true key >= 1  <=>  z key >= (-c g) << 16 | 1
        return key;
    };
    {
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            int code = 4;
            if (c >= 1 && c < ncols) code = rev ? read[n - c + 1] : read[c];
            const int r = q % H, hi = q / H;
            pcode[r / 8] |= (unsigned long long)code << (8 * (r % 8) + 3 * hi);
            if (!kRec && kTrack) THRK(q) = thr_key(q, true);
        }
#pragma unroll
        for (int li = 0; li < 4; ++li) {
#pragma unroll
            for (int r = 0; r < H; ++r) sprof[(li * WAVE + lane) * H + r] = s2[li * 64 + (int)((pcode[r / 8] >> (8 * (r % 8))) & 63)];
        }
        {
            int np[H];
#pragma unroll
            for (int r = 0; r < H; ++r) np[r] = s2[4 * 64 + (int)((pcode[r / 8] >> (8 * (r % 8))) & 63)];
            st_row(PR_NPROF, np);
        }
        // start rows: the gap-only row, identical for every path (uniform gap cost: A = c * gcost, z = 0)
#pragma unroll
        for (int q = 0; q < (kTrack ? C : 0); ++q) { minthrk = min(minthrk, thr_key(q, true)); minplain = min(minplain, thr_key(q, false)); }
        // packed 16-bit form of the thresholds for the lazy-key pretest of the register-resident rows: a key z << 16 | path
        // can reach a threshold only if z >= threshold >> 16 (INT32_MAX -> 32767: never; INT32_MIN -> -32768: always)
#pragma unroll
        for (int r = 0; r < (kRec ? H : 1); ++r)
            thz[kRec ? r : 0] = pack16(thr_key(kRec ? r : 0, true) >> 16, thr_key(kRec ? r + H : 0, true) >> 16);
        minplain2 = pack16(minplain >> 16, minplain >> 16);
        {
            int m = thz[0];
#pragma unroll
            for (int r = 1; r < (kRec ? H : 1); ++r) m = pk_min(m, thz[kRec ? r : 0]);
            thzmin = m;
        }
#ifdef RG_SWEEP16_LANEMIN
        // (experiment: one threshold per lane — the lowest of its columns — instead of one per column: how many more records?)
        {
            int mt = 32767;
#pragma unroll
            for (int r = 0; r < (kRec ? H : 0); ++r) mt = min(mt, min(lo16(thz[kRec ? r : 0]), hi16(thz[kRec ? r : 0])));
#pragma unroll
            for (int r = 0; r < (kRec ? H : 0); ++r) thz[kRec ? r : 0] = pack16(mt, mt);
        }
#endif
        int row0[H];
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const int c0 = lane * C + r, c1 = c0 + H;
            row0[r] = pack16(c0 < ncols ? 0 : NEG16, c1 < ncols ? 0 : NEG16);
        }
        for (int k = 0; k < P; ++k) st_row(k, row0);
        // PATH RETIREMENT constants: a path whose every FUTURE cell is provably below every threshold is "hopeless".  A
        // cell of the path at a later row and column j' is reached from some cell (this row, j), j <= j' (reverse sweep:
        // j >= j'), through moves that gain at most mmx = max(max match, 0) per column (D: a substitution score, L / U: gap
        // entries, <= 0 in every batch this kernel admits): value' <= value + mmx * |j' - j|.
        //   forward: it matters only if it can reach lb (seed: its sink value; candidate: value + mmx (n - j') >= lb + R, and
        //            the test drops the R so that a retired path's final score stays below the verified lb):
        //            hopeless  <=>  max_j (A[j] + mmx (n - j)) < lb
        //   reverse: it matters only if w[j'] >= thr[j'] (the emission threshold):
        //            hopeless  <=>  for all j: A[j] + mmx j < min_{j' <= j} (thr[j'] + mmx j')
        // Both are "max over the row of (z + constant per column) < 0" on the packed z-space rows: pseudo-row PR_RVL holds the
        // constants (saturating 16-bit; a constant that would have to be ROUNDED DOWN to fit switches the retirement off for the read).
        if (kRet && a.retire && !kSemi) {
            const int mmx = max(a.maxmatch, 0);
            int tv[C];
#pragma unroll
            for (int q = 0; q < C; ++q) tv[q] = -32768;
            bool ovf = false;
            if (!rev) {
                const int lbv = a.lb ? a.lb[rd] : INT32_MIN / 2;
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    const int c = lane * C + q;
                    const long long v = (long long)c * (gcost - mmx) + (long long)mmx * n - lbv;
                    ovf = ovf || (c < ncols && v > 32767);
                    tv[q] = c < ncols ? (int)max(-32768ll, min(32767ll, v)) : -32768;
                }
            } else if constexpr (kTrack) {       // (the -m 4 variant never sweeps in reverse: the branch — thirty-odd registers of it — is not compiled there)
                // Tmin over mirrored columns c' >= c  (real columns j' <= j)
                int tq[C], ltot = INT32_MAX;
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    const int c = lane * C + q;
                    const int j = n - c;
                    int th = INT32_MAX;
                    if (c < ncols && j >= oob && j < n + 1 - oob && a.thr) th = a.thr[(long long)rd * wpad + j];
                    tq[q] = th == INT32_MAX ? INT32_MAX : th + mmx * j;
                    ltot = min(ltot, tq[q]);
                }
                int suf = ltot;
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) { const int o = __shfl_down(suf, d, WAVE); if (lane + d < WAVE) suf = min(suf, o); }
                int run = __shfl_down(suf, 1, WAVE);
                if (lane == WAVE - 1) run = INT32_MAX;
#pragma unroll
                for (int q = C - 1; q >= 0; --q) {
                    run = min(run, tq[q]);
                    const int c = lane * C + q;
                    const long long v = (long long)c * gcost + (long long)mmx * (n - c) - (long long)run;
                    ovf = ovf || (c < ncols && run != INT32_MAX && v > 32767);
                    tv[q] = (c < ncols && run != INT32_MAX) ? (int)max(-32768ll, min(32767ll, v)) : -32768;
                }
            }
            {
                int rvc[H];
#pragma unroll
                for (int r = 0; r < H; ++r) rvc[r] = pack16(tv[r], tv[r + H]);
                st_row(PR_RVL, rvc);
            }
            if ((rev ? a.thr != nullptr : a.lb != nullptr) && !__any(ovf)) next_eval = 1 << a.retire_shift;
        }
    }
    __syncthreads();

    int ckey[kColmax == 1 ? C : 1], crow[kColmax == 1 ? C : 1];   // best (value << 16 | path) per column and its row
#pragma unroll
    for (int q = 0; q < (kColmax == 1 ? C : 1); ++q) { ckey[q] = INT32_MIN; crow[q] = 0; }
    int cmv[kColmax == 2 ? H : 1];                                 // packed z-space maxima (values only)
#pragma unroll
    for (int r = 0; r < (kColmax == 2 ? H : 1); ++r) cmv[r] = NEGPAIR;
    unsigned ncand = 0;
    // cells: the workload's member-row updates (sum of |paths(row)| over the rows: the reference's unit of work, SURVEY
    // 8d); done: row operators actually applied — the same number except in gather runs, which do per row the alpha and
    // the column map, and per member one pass at each end of the run instead of one update per row
    unsigned long long cells = 0, done = 0;
#ifdef RG_SWEEP16_ROWSTAT
    // (statistics build, gpurun_tmp: rows by kind for a few reads, printed from the device)
    unsigned st_rn[5] = {0, 0, 0, 0, 0}, st_grow = 0, st_gmem = 0, st_gen = 0, st_genmem = 0, st_dirs = 0, st_skip = 0, st_tail = 0;
#define RG_ROWSTAT(x) x
#else
#define RG_ROWSTAT(x) ((void)0)
#endif
    Cand* cand = !kRec && a.cand ? a.cand + (long long)rd * a.cand_cap : nullptr;
    uint32_t* dirs = a.dirs ? a.dirs + (long long)rd * a.dirs_stride : nullptr;
    // (the record variants always track — the driver sets track_best for every record sweep —: a compile-time constant there)
    const bool track = kRec ? true : (kTrack && a.track_best);
    // (direction words of this read: at most 2^20 slots of 64 or 128 words)
    const __amdgpu_buffer_rsrc_t dirs_rsrc = uniform_rsrc(dirs, dirs ? (unsigned)min(a.dirs_stride * 4ll, 0x7fffffffll) : 0u);

    // A record slot for (row i, this lane) when the lane has a column at its threshold: the position in the read's list (one
    // ballot), the header, and where the lane's C keys go.  The callers whose keys are built on demand (rows in registers) write
    // them in two halves — columns 0..H-1, then H..C-1 — so that only H key registers are live at a time.
    auto rec_slot = [&](int i, bool lane_hit, bool& mine) -> int4* {
        const unsigned long long has = __ballot(lane_hit);
        mine = false;
        int4* rp = nullptr;
        if (has) {
            const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(has >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)has, 0u));
            const unsigned pos = ncand + before;
            mine = lane_hit && pos < a.frec_cap;      // (lane_hit IS this lane's bit of `has`: no 64-bit lane mask to keep in registers)
            rp = reinterpret_cast<int4*>(reinterpret_cast<char*>(a.frec + (long long)rd * a.frec_cap * (4 + C)) + pos * (unsigned)((4 + C) * sizeof(int)));
            if (mine) rp[0] = make_int4((i << 6) | lane, 0, 0, 0);
            ncand += (unsigned)__popcll(has);
        }
        return rp;
    };
    auto rec_half = [&](int4* rp, bool mine, int half, const int (&k)[H]) {      // keys of columns half * H .. half * H + H - 1
        if (!mine) return;
        if constexpr (H >= 4) {
#pragma unroll
            for (int q4 = 0; q4 < H / 4; ++q4) rp[1 + half * (H / 4) + q4] = make_int4(k[4 * q4], k[4 * q4 + 1], k[4 * q4 + 2], k[4 * q4 + 3]);
        } else {
            reinterpret_cast<int2*>(rp + 1)[half] = make_int2(k[0], k[1]);
        }
    };

    // Per-row epilogue on the packed keys bkey = value << 16 | path (non-members of the reference's matrices hold 0,
    // so a cell is usable iff its winner is a member: value > 0, or value == 0 and path > knm, or no non-member
    // exists -> bkey > kthr with kthr = knm (>= 0) or INT32_MIN).  ckey keeps the best usable key per column and the
    // row it came from; emission compares against thr << 16.
    // pre_done: the caller tracked the column maxima itself and already knows that some column reaches its threshold
    // hit (with pre_done): this lane's packed pretest found a column at its threshold
    auto row_end = [&](int i, int knm, const int (&bkey)[C], bool pre_done = false, bool hit = false) {
        // The column maxima ignore that rule (k_bound re-checks the recorded cell; an unusable cell has value <= 0 and
        // can only raise a non-positive maximum, which loosens thresholds, never tightens them); emissions apply it
        // (it removes most of the negative-valued cells the loose forward thresholds would let through).
        unsigned emask = 0;
        if (kColmax == 1 && !pre_done) {
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const bool better = bkey[q] > ckey[kColmax == 1 ? q : 0];
                ckey[kColmax == 1 ? q : 0] = better ? bkey[q] : ckey[kColmax == 1 ? q : 0];
                crow[kColmax == 1 ? q : 0] = better ? i : crow[kColmax == 1 ? q : 0];
            }
        }
        if (kColmax == 2 && !pre_done) {
#pragma unroll
            for (int r = 0; r < H; ++r) {
                // value halves of the two keys of register r: low column | high column << 16
                const int v2 = (int)__builtin_amdgcn_perm((unsigned)bkey[r + H], (unsigned)bkey[r], 0x07060302u);
                cmv[kColmax == 2 ? r : 0] = pk_max(cmv[kColmax == 2 ? r : 0], v2);
            }
        }
#ifdef RG_SWEEP16_NOEMIT
        return;
#endif
        if (kRec) {
            // One fixed-size record per (row, lane) with any column whose VALUE reaches its threshold (1 + C/4 16-byte
            // stores): a saturating packed compare of the value halves against thz (the key thresholds >> 16, member rule's
            // necessary condition included; the few rows every path visits — knm < 0: the rule is void there — against the
            // lane's lowest plain threshold).  A superset of `key >= threshold`: k_expand re-tests every key with the final
            // bound and the exact member-winner rule, so neither a per-lane column mask nor the key thresholds are kept.
            bool lane_hit = hit;
            if (!pre_done) {
                int acc = -1;
#pragma unroll
                for (int r = 0; r < H; ++r) {
                    const int v2 = (int)__builtin_amdgcn_perm((unsigned)bkey[r + H], (unsigned)bkey[r], 0x07060302u);
                    acc &= pk_sub_sat(v2, knm >= 0 ? thz[kRec ? r : 0] : minplain2);
                }
                lane_hit = ((unsigned)acc & 0x80008000u) != 0x80008000u;
            }
            bool mine;
            int4* rp = rec_slot(i, lane_hit, mine);
            if (mine) {
                if constexpr (C >= 4) {
#pragma unroll
                    for (int q4 = 0; q4 < C / 4; ++q4) rp[1 + q4] = make_int4(bkey[4 * q4], bkey[4 * q4 + 1], bkey[4 * q4 + 2], bkey[4 * q4 + 3]);
                }
            }
            return;
        }
        // tight thresholds (reverse sweep): most rows emit nothing; one max3 tree against the lane's lowest threshold
        // decides for the whole wave whether the per-column test is needed
        bool test = true;
        if (tight && !pre_done) {
            int mx = bkey[0];
#pragma unroll
            for (int q = 1; q < C; ++q) mx = max(mx, bkey[q]);
            test = __any(mx >= (knm >= 0 ? minthrk : minplain));
        }
        if (test) {
            // Cand entries go to k_search unfiltered: the exact member-winner rule (true key > knm) applies here
            const int gz = gcost * 65536;
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const int tkey = bkey[q] + (lane * C + q) * gz;    // true key: value << 16 | path
                // (rows every path visits: the lane's lowest plain threshold, but never a column whose own threshold is
                // "never": outside the read or outside the recombination band)
                const int tq = THRK(q);
                emask |= (knm >= 0 ? (bkey[q] >= tq && tkey > knm) : (bkey[q] >= minplain && tq != INT32_MAX)) ? (1u << q) : 0u;
            }
        }
        if (cand && __any(emask != 0)) {
            const int cnt = __popc(emask);
            const int incl = dpp_incl_sum(cnt);
            const int total = __shfl(incl, WAVE - 1, WAVE);
            unsigned pos = ncand + (unsigned)(incl - cnt);
#pragma unroll
            for (int q = 0; q < C; ++q) {
                if ((emask >> q) & 1) {
                    if (pos < a.cand_cap) {
                        const int c = lane * C + q;
                        Cand cd;
                        cd.row = i; cd.col = rev ? n - c : c; cd.val = (bkey[q] >> 16) + c * gcost; cd.path = bkey[q] & 0xffff;
                        cand[pos] = cd;
                    }
                    ++pos;
                }
            }
            ncand += (unsigned)total;
        }
    };
    // direction words, format 1 (LayerArgs::dir_fmt): C <= 16: one word per lane (RowOps16::dir_word); C = 32: word 0 = U bits,
    // word 1 = L bits (bit r = column r of the lane, bit 16 + r = column H + r)
    auto store_dirs = [&](int slot, const int (&XU)[H], const int (&XL)[H]) {
#ifdef RG_SWEEP16_NODIRS
        return;
#endif
        if constexpr (C <= 16) {
            const unsigned w = RowOps16<C>::dir_word(XU, XL);
#ifndef RG_SWEEP16_FLATDIRS
            __builtin_amdgcn_raw_buffer_store_b32(w, dirs_rsrc, lane * 4, slot * (a.dir_words * 4), 0);
#else
            dirs[(long long)slot * a.dir_words + lane] = w;
#endif
        } else {
            dirs[(long long)slot * a.dir_words + lane] = RowOps16<C>::sign_bits(XU);
            dirs[(long long)slot * a.dir_words + WAVE + lane] = RowOps16<C>::sign_bits(XL);
        }
    };
    // DIRECTION WORDS ON DEMAND (round 6).  A direction word is read by k_layer16 only when the traceback walks a path of that
    // (row, group) — the forward layer of the read's final forward path, the reverse layer of its reverse path — and a row
    // without members besides its alpha spent 40 % of its vector instructions on that word (sixteen packed subtractions for the
    // signs, the bit gathering, the store).  When the batch runs on a speculative bound the final paths are all but known
    // before the sweep: k_pick's one or two paths.  `dsel` = those paths; a (row, group) record computes and stores its word
    // only if its members include one of them.  k_verify checks afterwards that the paths k_search chose are in `dsel` and
    // sends the read to the second pass otherwise (it stores every word: a.pick is null there), exactly like a read whose
    // speculative bound failed.  Sweeps without picks (no speculation: -m 4 / 5 / 9, more than 64 paths, three sweeps) store all.
    PathWords dsel_w{~0ull, ~0ull, ~0ull, ~0ull};
    if (a.dsel_pick) {
        const int p1 = a.dsel_pick[rd], p2 = a.dsel_pick2 ? a.dsel_pick2[2 * rd] : -1;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            unsigned long long m = ((p1 >> 6) == w) ? 1ull << (p1 & 63) : 0ull;
            if (p2 >= 0 && (p2 >> 6) == w) m |= 1ull << (p2 & 63);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(m >> 32));
            dsel_w.set(w, ((unsigned long long)hi << 32) | lo);
        }
    }
    // (a wave-uniform word by a wave-uniform page: selects over named scalars; the narrow variants only have word 0)
    auto word_of = [&](const PathWords& pw, int page) -> unsigned long long { return kWide ? pw.get(page) : pw.w0; };
    // (kColmax == 1, the first sweep of the three-sweep pipeline, runs without direction words at all)
    // (the rows a sweep visits first store every word: SweepArgs::dsel_lo / dsel_hi)
    auto want_dirs = [&](unsigned long long members, int page, int row) -> bool {
        return (kColmax != 1 || dirs != nullptr) && ((members & word_of(dsel_w, page)) != 0ull || (rev ? row > a.dsel_hi : row < a.dsel_lo));
    };
    const int4* steps = rev ? a.rsteps : a.fsteps;
    const int nsteps = rev ? a.nrsteps : a.nfsteps;
    int4 recs = make_int4(0, 0, 0, 0);
    if (lane < nsteps) recs = steps[lane];
    int t = 0;
    int blk = 0;                         // the 64-record block `recs` holds (one record per lane)
    // PATH RETIREMENT: the records of the block in `recs` that still have to be looked at — a needed member, or the last group
    // of a row with several groups (it closes the row even when skipped); the record loop jumps over the others
    unsigned long long live = ~0ull;
    PathWords needed{0ull, 0ull, 0ull, 0ull};       // (one word per 64-path page; wave-uniform)
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int left = a.g.P - 64 * w;
        needed.set(w, left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull)));
    }
    auto block_live = [&]() -> unsigned long long {
        const unsigned f = ((unsigned)recs.x >> 23) & 7u, field = ((unsigned)recs.x >> 26) & 63u;
        const bool is_run = (f & 4u) && field != 0u;
        const unsigned long long m = ((unsigned long long)(unsigned)recs.w << 32) | (unsigned)recs.z;
        const unsigned long long nd = kWide ? needed.get((recs.y >> 29) & 3) : needed.w0;      // (per lane: the page of this lane's record)
        return __ballot((m & nd) != 0ull || (!is_run && (f & 3u) == 2u));
    };
    // Records are taken in order and a look-ahead never goes back: whoever first touches a record of the next block moves
    // `recs` there.  The block is waited for at once (an L2 hit once per 64 records): round 4 kept the block behind the
    // current one in flight in four more registers, and that load — pending across the back edge of every row loop, its
    // landing registers reused by the row's LDS reads — made the compiler open EVERY row with s_waitcnt vmcnt(0), which on
    // gfx9 also drains the row-before's direction-word and record stores.
    auto to_block = [&](int b) {
        blk = b;
        const int nb = b * WAVE + lane;
        recs = nb < nsteps ? steps[nb] : make_int4(0, 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (kRet) live = block_live();
    };
    auto fetch = [&](int tt, int& w0, int& w1, unsigned long long& gmask) {
        const int idx = tt & (WAVE - 1);
        if ((tt >> 6) != blk) to_block(tt >> 6);
        w0 = __builtin_amdgcn_readlane(recs.x, idx);
        w1 = __builtin_amdgcn_readlane(recs.y, idx);
        gmask = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(recs.w, idx) << 32) |
                (unsigned)__builtin_amdgcn_readlane(recs.z, idx);
    };
    constexpr int F_FIRST = 1, F_LAST = 2, F_INNER = 4;
    // flags / members of record tt without consuming it (tt = the record the next fetch will take)
    auto peek_w0 = [&](int tt) -> int {
        if ((tt >> 6) != blk) to_block(tt >> 6);
        return __builtin_amdgcn_readlane(recs.x, tt & (WAVE - 1));
    };
    auto peek_w1 = [&](int tt) -> int {
        if ((tt >> 6) != blk) to_block(tt >> 6);
        return __builtin_amdgcn_readlane(recs.y, tt & (WAVE - 1));
    };
    auto peek_gm = [&](int tt) -> unsigned long long {
        if ((tt >> 6) != blk) to_block(tt >> 6);
        const int idx = tt & (WAVE - 1);
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(recs.w, idx) << 32) | (unsigned)__builtin_amdgcn_readlane(recs.z, idx);
    };

    // semiglobal end-row selection (see k_sweep)
    const bool semi_end = kSemi && !rev;
    const int ln_end = n / C, ql_end = n % C;
    if (semi_end) for (int k = lane; k < EP; k += WAVE) { endv[k] = INT32_MIN; endr[k] = 0; }
    __syncthreads();
    int gbest_val = INT32_MIN, gbest_row = 0, gbest_path = 0, rowkey = INT32_MIN;
    auto end_fold = [&](int k, int i, const int (&row)[H]) {
        int pv = 0;
#pragma unroll
        for (int r = 0; r < H; ++r) if (r == ql_end % H) pv = row[r];
        const int v = (ql_end >= H ? hi16(pv) : lo16(pv)) + n * gcost;      // column n: A = z + n * g
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
    // The path id goes through a VGPR: a gfx9 VALU instruction takes one scalar operand, so (row & 0xffff0000) | k
    // is a single v_and_or_b32 only if the mask is the scalar and k a vector register.
    auto set_keys = [&](int (&bkey)[C], const int (&row)[H], int ks) {   // first member of a row: no reset + max
#ifdef RG_SWEEP16_NOKEYS
        return;
#endif
        int k = ks;
        asm volatile("" : "+v"(k));
#pragma unroll
        for (int r = 0; r < H; ++r) {
            bkey[r] = (int)(((unsigned)row[r] << 16) | (unsigned)k);
            bkey[r + H] = (int)(((unsigned)row[r] & 0xffff0000u) | (unsigned)k);
        }
    };
    auto fold_keys = [&](int (&bkey)[C], const int (&row)[H], int ks) {
#ifdef RG_SWEEP16_NOKEYS
        return;
#endif
        int k = ks;
        asm volatile("" : "+v"(k));
#pragma unroll
        for (int r = 0; r < H; ++r) {
            // key = value << 16 | path: one v_lshl_or_b32 / v_and_or_b32 per column
            bkey[r] = max(bkey[r], (int)(((unsigned)row[r] << 16) | (unsigned)k));
            bkey[r + H] = max(bkey[r + H], (int)(((unsigned)row[r] & 0xffff0000u) | (unsigned)k));
        }
    };

    auto load_steps = [&](int li_, int (&s)[H]) {
        if (li_ >= 4) [[unlikely]] {
            // an 'N' row of the graph: its profile is a pseudo-row.  (The empty asm keeps the two paths apart: merged, the
            // compiler selects between the LDS and the global ADDRESS and issues flat loads for every row.)
            ld_row(PR_NPROF, s);
            __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0) HERE: a load left pending makes every row of the common path wait for its stores
            asm volatile("" ::: "memory");
            return;
        }
        const int* sp = sprof + (li_ * WAVE + lane) * H;
        if constexpr (H >= 4) {
#pragma unroll
            for (int r4 = 0; r4 < H / 4; ++r4) {
                const int4 v = reinterpret_cast<const int4*>(sp)[r4];
                s[4 * r4] = v.x; s[4 * r4 + 1] = v.y; s[4 * r4 + 2] = v.z; s[4 * r4 + 3] = v.w;
            }
        } else {
            const int2 v = *reinterpret_cast<const int2*>(sp);
            s[0] = v.x; s[1] = v.y;
        }
    };
    // KEYS OF A ROW IN PROGRESS.  A row with several groups folds the (value, path) keys of its groups one record at a time
    // — and in a split table register runs of other rows lie between those records.  The 16 keys used to sit in registers
    // across the whole record loop; now they wait in LDS between the records of the row ([q][lane], the words of the gather
    // table: the step-table builder lets no gather run lie between the groups of a row, rg_steps.cpp) and are registers only
    // inside the record that folds into them: 16 VGPRs less in every run loop.
    auto keys_ld = [&](int (&key)[C]) {
#pragma unroll
        for (int q = 0; q < C; ++q) key[q] = gT[q * WAVE + lane];
    };
    auto keys_st = [&](const int (&key)[C]) {
#pragma unroll
        for (int q = 0; q < C; ++q) gT[q * WAVE + lane] = key[q];
    };
    // (wide graphs only: direction masks, L mask and fill-forward source lane of the current group's alpha, live across the
    // continuation entries of a group that spans 64-path pages.  Everywhere else these are locals of the record that
    // computes them: nothing the compiler could mistake for state of the record loop)
    int MUw[kWide ? H : 1], MLw[kWide ? H : 1];
    unsigned lmaskw = 0;
    int srcw = 0;
    // PATH RETIREMENT (kRet): `needed` = the paths whose rows are still computed.  Every 2^a.retire_shift records (at the top of the record
    // loop: every row is in memory there) the hopeless paths are found (one pass over each needed row against the constants
    // in rvl) and a hopeless path is retired unless it still LEADS a group with a needed member further down the table — its
    // decisions are that member's directions — iterated to the fixpoint (lead: per evaluation point and path, the union of
    // the member masks of the groups the path leads from there on; built beside the step table).  A retired path's row is
    // never read again: it cannot emit, it cannot be a cell's best member where something emits, its sink value is below
    // the bound k_verify checks; the needed paths see exactly the decisions they would see in the full sweep.
    // tests/c/band_experiment.cpp measures why this — not a column band — is the exact way to skip hopeless work here.
    bool row_open = false;               // some group of the current several-group row has put its keys into LDS (keys_st)
#ifdef RG_SWEEP16_RETSTAT
    // (statistics build, tools/sweep_variants.sh RETSTAT: the counters carry evaluations | needed paths << 32 and not-hopeless paths)
    unsigned long long stat_e = 0, stat_n = 0, stat_h = 0;
#endif
    auto retire_eval = [&](int e) {
        PathWords hop{0ull, 0ull, 0ull, 0ull};
        // (four rows in flight per wait: one row per wait made the evaluations ~8 % of the sweep)
        constexpr int EB = !kTrack ? 1 : (C <= 16 ? 4 : 2);      // (the -m 4 variant: one row at a time — it has to stay under 168 registers, three waves per SIMD)
        int rvc[H];
        ld_row(PR_RVL, rvc);
#pragma unroll
        for (int pg = 0; pg < NW; ++pg) {
            unsigned long long todo = needed.get(pg), hp = 0ull;
            while (todo) {
                int kq[EB];
                int tmp[EB][H];
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    kq[u] = -1;
                    if (todo) {
                        kq[u] = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        ld_row(pg * 64 + kq[u], tmp[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    if (kq[u] < 0) break;
                    int m = pk_add_sat(tmp[u][0], rvc[0]);
#pragma unroll
                    for (int r = 1; r < H; ++r) m = pk_max(m, pk_add_sat(tmp[u][r], rvc[r]));
                    const int v = max(lo16(m), hi16(m));
                    if (__builtin_amdgcn_readlane(dpp_incl_max(v, INT32_MIN), WAVE - 1) < 0) hp |= 1ull << kq[u];
                }
            }
            hop.set(pg, hp);
        }
        // lead table (rg_steps.cpp): [evaluation point][path, padded to whole pages][word]: lane l of page pg asks for path 64 pg + l
        const int nw = kWide ? (P + 63) >> 6 : 1;
        const unsigned long long* lead = (rev ? a.rlead : a.flead) + (long long)e * (64 * nw) * nw;
        PathWords nd{0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int pg = 0; pg < NW; ++pg) nd.set(pg, needed.get(pg) & ~hop.get(pg));
#ifdef RG_SWEEP16_RETSTAT
        const unsigned long long nd_first = nd.w0;
#endif
        if constexpr (!kWide) {
            const unsigned long long lead_k = lane < P ? lead[lane] : 0ull;
            for (;;) {
                const unsigned long long ad = __ballot(((needed.w0 >> lane) & 1ull) && !((nd.w0 >> lane) & 1ull) && (lead_k & nd.w0) != 0ull);
                if (!ad) break;
                nd.w0 |= ad;
            }
        } else {
            // (the lead words are read again in every round of the closure — sixteen 64-bit values per lane would otherwise sit in
            // registers across the evaluation: it runs once per 256 records, the loads are L2 hits)
            for (;;) {
                bool grown = false;
#pragma unroll
                for (int pg = 0; pg < NW; ++pg) {
                    bool leads = false;
                    if (pg < nw && pg * 64 + lane < P && ((needed.get(pg) >> lane) & 1ull) && !((nd.get(pg) >> lane) & 1ull)) {
#pragma unroll
                        for (int w = 0; w < NW; ++w)
                            if (w < nw) leads = leads || (lead[(long long)(pg * 64 + lane) * nw + w] & nd.get(w)) != 0ull;
                    }
                    const unsigned long long ad = __ballot(leads);
                    if (ad) { nd.set(pg, nd.get(pg) | ad); grown = true; }
                }
                if (!grown) break;
            }
        }
        needed = nd;
#ifdef RG_SWEEP16_RETSTAT
        stat_e += 1ull; stat_n += (unsigned long long)__popcll(nd.w0); stat_h += (unsigned long long)__popcll(nd_first);
#endif
    };
    while (t < nsteps) {
        int w0, w1;
        unsigned long long gmask;
        if (kRet && next_eval != INT32_MAX) {
            // evaluation points; then jump over the records nothing is left to do for (a skipped record costs nothing: with a
            // fetch and a test each they were a seventh of the sweep once half the member rows were retired)
            if ((t >> 6) != blk) fetch(t, w0, w1, gmask);           // (moves to t's block; the record is fetched again below)
            if (t >= next_eval) [[unlikely]] {
                retire_eval(t >> a.retire_shift);
                next_eval = (t | ((1 << a.retire_shift) - 1)) + 1;
                live = block_live();
            }
            const unsigned long long m = live >> (t & (WAVE - 1));
            if (m == 0ull) { t = (t | (WAVE - 1)) + 1; continue; }
            t += __builtin_ctzll(m);
        }
        fetch(t, w0, w1, gmask);
        const int i = w0 & 0xfffff;
        const int li = (w0 >> 20) & 7;
        // (flags == F_INNER alone: the HEAD of a run — a segment's first row that has one group led by its lowest member;
        // it starts a register / gather run exactly like an inner row, but no run continues INTO it)
        // A TAIL (bit F_INNER with a zero run field; split step tables only, see rg_path_driver.hip): one group of a row
        // that has several, placed right behind the register run of the SAME paths on its predecessor row — the run
        // continues into it (the rows stay in registers) and its keys fold into the row's bkey like a general record's.
        const int flags = (((w0 >> 23) & 7) == F_INNER && ((w0 >> 26) & 63) != 0) ? 7 : ((w0 >> 23) & 7);
        const int slot = w1 & 0xfffff;
        const int kbase = kWide ? ((w1 >> 29) & 3) * 64 : 0;   // first path id of the entry's 64-path page
        const bool cont = kWide && w1 < 0;                     // continuation entry of a group that spans pages: members only
        // (an inner row of a one-entry segment run: the alpha is the lowest member, the field holds the run length left)
        const int ga = kbase + ((flags & F_INNER) ? __builtin_ctzll(gmask | (1ull << 63)) : ((w0 >> 26) & 63));
        const int nm = __popcll(gmask);
        // (inner / head records: the alpha field holds the rows left in the run, this one included, capped at 63.)  A gather
        // run costs RG_GATHER_PER_MEMBER_RUN instructions per member once per run (two passes) + RG_GATHER_PER_ROW per row, the
        // member-by-member form RG_GATHER_PER_MEMBER_ROW per member and row (row load + store, member operator, eager keys): it
        // pays when R * (84 (nm - 1) - 160) >= 90 (nm - 1)  (16 or 32 paths: 2 rows, 8 paths: 2, 5 paths: 3; round 4 had
        // 77 / 160 / 200: 3 / 4 / 6 rows.  Config 4's lone sweep 15.2 -> 14.8 ms, config 5 unchanged)
        const int run_left = (flags & F_INNER) ? ((w0 >> 26) & 63) : 0;
        const int page = kWide ? (w1 >> 29) & 3 : 0;
        const unsigned long long gm = kRet ? (gmask & word_of(needed, page)) : gmask;     // the members still computed
        const int nme = kRet ? __popcll(gm) : nm;
        if (kRet && gm == 0ull) {
            // every member retired (and with them the alpha: it would be needed otherwise): the record is skipped; a row
            // with several groups still opens / closes around its skipped groups.  (The records of a register / gather run
            // — flags read 7 — may lie BETWEEN the groups of such a row in a split table: they leave bkey and row_open alone,
            // skipped or not.)
            if (run_left == 0) {
                if (flags & F_FIRST) row_open = false;
                if (track && (flags & F_LAST) && row_open) {
                    int key[C];
                    keys_ld(key);
                    row_end(i, ((w1 >> 20) & 511) - 1, key);
                }
                if (flags & F_LAST) row_open = false;
            }
            ++t;
            continue;
        }
        // WIDE RUNS (more than 64 paths; the wide-run table of rg_steps.cpp): the alpha entry of a row with one group led by its
        // lowest member is flagged like the narrow tables' HEAD / inner rows, continuation entries or not.  Here the members
        // that are still needed are gathered from this entry and the continuation entries behind it; when at most KRUN are left the
        // segment's rows run with those in registers (the run loop steps over the continuation entries: `went` records per row),
        // otherwise the entries take the general path one by one (this entry then must not close the row: its continuation
        // entries do).
        int went = 1, wide_n = 0, wide_total = nm, wide_needed = 0;
        PathWords wide_gw{0ull, 0ull, 0ull, 0ull};      // the needed members of the whole group, page by page
        int wide_ids[KRUN > 0 ? KRUN : 1];
        bool wide_sel = false;
        if (kWide && (flags & F_INNER) && run_left > 0) {      // (also where KRUN is 0 — the -m 4 variant at 32 columns per lane —: its gather runs iterate wide_gw)
#pragma unroll
            for (int kk = 0; kk < KRUN; ++kk) wide_ids[kk] = 0;
            auto add = [&](unsigned long long m, int base) {
                while (m) {
                    const int id = base + __builtin_ctzll(m);
                    m &= m - 1;
#pragma unroll
                    for (int kk = 0; kk < KRUN; ++kk) if (kk == wide_n) wide_ids[kk] = id;       // (no dynamically indexed array)
                    ++wide_n;
                }
            };
            add(gm, kbase);
            wide_gw.set(page, gm);
            wide_needed = nme;
            wide_sel = (gmask & word_of(dsel_w, page)) != 0ull;
            while (t + went < nsteps) {
                const int nw1 = peek_w1(t + went);
                if (nw1 >= 0) break;
                const unsigned long long mm = peek_gm(t + went);
                const int pg = (nw1 >> 29) & 3;
                wide_total += __popcll(mm);
                add(mm & word_of(needed, pg), pg * 64);
                wide_gw.set(pg, mm & word_of(needed, pg));
                wide_needed += __popcll(mm & word_of(needed, pg));
                wide_sel = wide_sel || (mm & word_of(dsel_w, pg)) != 0ull;
                ++went;
            }
        }
        const bool wide_run = kWide && KRUN > 0 && (flags & F_INNER) && run_left > 0 && wide_n >= 1 && wide_n <= KRUN;
        // (members of a run in ascending path order, the alpha left out: one word up to 64 paths, the needed members of every page of
        // the group beyond)
        struct MemberIter {
            PathWords w;
            int pg;
            __device__ __forceinline__ int next() {
                while (pg < NW) {
                    const unsigned long long m = w.get(pg);
                    if (m) { w.set(pg, m & (m - 1)); return pg * 64 + __builtin_ctzll(m); }
                    ++pg;
                }
                return -1;
            }
        };
        const int g_nm = kWide ? wide_total : nm, g_nme = kWide ? wide_needed : nme;      // members of the group | the needed ones
        if (RG_SWEEP16_GATHER && kGather && (kRec ? (kColmax == 0 || (kColmax == 2 && RG_SWEEP16_GATHER_FWD)) : !track) && a.gather_ok && !semi_end && (flags & F_INNER) && g_nm > KRUN &&
            (!kWide || run_left > 0) && (!kRet || g_nme > KRUN) && run_left * (RG_GATHER_PER_MEMBER_ROW * (g_nm - 1) - RG_GATHER_PER_ROW) >= RG_GATHER_PER_MEMBER_RUN * (g_nm - 1)) {
            // ---- GATHER RUN: R consecutive inner rows of a segment that a wide group (nm paths, one group, alpha = its lowest
            // path) runs through.  Every member follows the alpha's directions, and a direction only MOVES values (D: from
            // column c - 1 of the row above, U: from column c, L: from column c - 1 of the new row) and adds a constant that
            // does not depend on the member.  So delta_k = row_k - row_alpha is carried through the run by pure data
            // movement, the same for every k: a gather G (column -> column of the run's first row) that is itself updated like
            // a member row with zero steps.  The run therefore costs, per row, the alpha + G + one table gather for the best
            // member per column — not nm member updates — and per member ONE pass at each end of the run; the members'
            // rolling rows are read twice and written once per run instead of once each per row (for the rows every path
            // visits that traffic was 58 % of the sweep's, 10 ms of a 93 ms step: profiles/r03_notes.md).
            const int R = (w0 >> 26) & 63;
            const int ka = ga;
            int A[H], G[H];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < H; ++r) A[r] = 0;
            ld_row(ka, A);            // (the rolling row itself keeps the run-start values until phase (3) stores the new ones)
#if defined(RG_SWEEP16_STALLSTAT) && RG_SWEEP16_STALLSTAT == 2
            { RG_STALL_BEGIN(); RG_STALL_END(st_run); ++st_nrun; }
#endif
            // (1) best (delta, path) per column over the members at the run start; ties -> highest path id: members in
            // ascending order, a later one replaces on >=.  Packed: delta = row_k - A0 (saturating: |delta| fits, gather_ok)
#ifndef RG_G_NOPH1
            if (kRec && track) {         // (only the keys of the rows inside the run need it)
                int bd[H], bk[H];
#pragma unroll
                for (int r = 0; r < H; ++r) { bd[r] = 0; bk[r] = pack16(ka, ka); }
                MemberIter it{kWide ? wide_gw : PathWords{gm, 0ull, 0ull, 0ull}, 0};
                it.w.set(ka >> 6, it.w.get(ka >> 6) & ~(1ull << (ka & 63)));        // (narrow: page 0, path ids below 64)
                int nx[H];
                int kn = it.next();
                if (kn >= 0) ld_row(kn, nx);
                while (kn >= 0) {
                    const int k = kn;
                    int cur[H];
#pragma unroll
                    for (int r = 0; r < H; ++r) cur[r] = nx[r];
                    kn = it.next();
                    if (kn >= 0) ld_row(kn, nx);
                    int kk = k;
                    asm volatile("" : "+v"(kk));
                    const int K2 = (int)(((unsigned)kk << 16) | (unsigned)kk);
#pragma unroll
                    for (int r = 0; r < H; ++r) {
                        const int d = pk_sub_sat(cur[r], A[r]);
                        const int lt = pk_sign(pk_sub_sat(d, bd[r]));       // 0xffff where d < best: keep
                        bd[r] = pk_max(bd[r], d);
                        bk[r] = bfi(lt, bk[r], K2);
                    }
                }
#pragma unroll
                for (int r = 0; r < H; ++r) {
                    gT[r * WAVE + lane] = (int)(((unsigned)bd[r] << 16) | ((unsigned)bk[r] & 0xffffu));            // column r of the lane
                    gT[(r + H) * WAVE + lane] = (int)(((unsigned)bd[r] & 0xffff0000u) | ((unsigned)bk[r] >> 16)); // column H + r
                }
            }
#endif
            // (2) G: column -> code of the run-start column whose delta it carries; code = word index of the packed row
            // layout [r][lane] * 2 + half
            {
                // (an opaque copy of the lane: otherwise the H constants are computed once before the record loop and — the
                // forward variant is out of registers — spilled to scratch to be reloaded here)
                int gl = lane;
                asm volatile("" : "+v"(gl));
#pragma unroll
                for (int r = 0; r < H; ++r) { const int w = r * WAVE + gl; G[r] = pack16(2 * w, 2 * w + 1); }
            }
            __syncthreads();
            int ri = i, rli = li, rslot = slot, rw1 = w1;
            for (int step = 0;; ++step) {
                const int g_i = gcost;
                const int g0 = kSemi ? 0 : g_i;
                int s[H], MU[H], ML[H], XU[H], XL[H];
                load_steps(rli, s);
                int lmax_unused;
                RowOps16<C>::alpha(A, s, g_i, g0, lane, XU, XL, lmax_unused);
                const bool gdirs = kWide ? ((kColmax != 1 || dirs != nullptr) && (wide_sel || (rev ? ri > a.dsel_hi : ri < a.dsel_lo))) : want_dirs(gmask, 0, ri);
                RG_ROWSTAT((++st_grow, st_gmem += g_nme, st_dirs += gdirs ? 1 : 0));
                if (gdirs) store_dirs(rslot, XU, XL);
                const unsigned lmask = RowOps16<C>::masks(XU, XL, MU, ML);
                const int src = RowOps16<C>::src_lane(lmask, lane);
                RowOps16<C>::template member<true>(G, MU, lane, MU, ML, lmask, src);   // the gather follows the directions, adds nothing (SEL unused)
                cells += (unsigned long long)g_nm;
                done += 2ull;
#ifndef RG_G_NOKEYS
                if (kRec && track) {
                    // best member per column of this row: alpha value + best delta of the run-start column G points at (packed
                    // values bv + packed paths K2); the (value, path) keys only when some column can reach its threshold
                    int bv[H], K2[H];
                    constexpr int HS = 31 - __builtin_clz(H * WAVE);      // log2 of the words of one packed row
#pragma unroll
                    for (int r = 0; r < H; ++r) {
                        const int c0 = G[r] & 0xffff, c1 = (unsigned)G[r] >> 16;
                        const unsigned t0 = (unsigned)gT[(c0 >> 1) + ((c0 & 1) << HS)];
                        const unsigned t1 = (unsigned)gT[(c1 >> 1) + ((c1 & 1) << HS)];
                        bv[r] = pk_add(A[r], (int)__builtin_amdgcn_perm(t1, t0, 0x07060302u));     // delta halves
                        K2[r] = (int)__builtin_amdgcn_perm(t1, t0, 0x05040100u);                    // path halves
                    }
                    if (kColmax == 2) {
#pragma unroll
                        for (int r = 0; r < H; ++r) cmv[kColmax == 2 ? r : 0] = pk_max(cmv[kColmax == 2 ? r : 0], bv[r]);
                    }
                    const int knm_row = ((rw1 >> 20) & 511) - 1;
                    int acc = -1;
                    if (knm_row >= 0) {
#pragma unroll
                        for (int r = 0; r < H; ++r) acc &= pk_sub_sat(bv[r], thz[kRec ? r : 0]);
                    } else {
#pragma unroll
                        for (int r = 0; r < H; ++r) acc &= pk_sub_sat(bv[r], minplain2);
                    }
                    const bool lhit = ((unsigned)acc & 0x80008000u) != 0x80008000u;
                    if (__any(lhit)) {
                        bool mine;
                        int4* rp = rec_slot(ri, lhit, mine);
                        int hk[H];
#pragma unroll
                        for (int r = 0; r < H; ++r) hk[r] = (int)(((unsigned)bv[r] << 16) | ((unsigned)K2[r] & 0xffffu));
                        rec_half(rp, mine, 0, hk);
#pragma unroll
                        for (int r = 0; r < H; ++r) hk[r] = (int)(((unsigned)bv[r] & 0xffff0000u) | ((unsigned)K2[r] >> 16));
                        rec_half(rp, mine, 1, hk);
                    }
                }
#endif
                t += kWide ? went : 1;          // (wide: over the row's continuation entries)
                if (step + 1 >= R || t >= nsteps) break;
                int nw0, nw1;
                unsigned long long ngm;
                fetch(t, nw0, nw1, ngm);
                ri = nw0 & 0xfffff; rli = (nw0 >> 20) & 7; rslot = nw1 & 0xfffff; rw1 = nw1;
            }
            done += (unsigned long long)(g_nme - 1) * ((kRec && track) ? 2ull : 1ull);    // passes (1) and (3)
            // (3) every member once: row_k(end)[c] = A(end)[c] - A0[G(c)] + row_k(start)[G(c)]
#ifndef RG_G_NOPH3
            {
                int B[H];
                int* gS = gT;             // (the table of phase (1) is dead: its words hold one packed row at a time now)
                {
                    int a0[H];
                    ld_row(ka, a0);       // the alpha's row at the run start
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < H; ++r) gS[r * WAVE + lane] = a0[r];
                    __syncthreads();
                }
#pragma unroll
                for (int r = 0; r < H; ++r) {
                    const int c0 = G[r] & 0xffff, c1 = (unsigned)G[r] >> 16;
                    const int w0v = gS[c0 >> 1], w1v = gS[c1 >> 1];
                    const int a0 = (c0 & 1) ? hi16(w0v) : lo16(w0v), a1 = (c1 & 1) ? hi16(w1v) : lo16(w1v);
                    B[r] = pk_sub_sat(A[r], pack16(a0, a1));
                }
                st_row(ka, A);
                MemberIter it{kWide ? wide_gw : PathWords{gm, 0ull, 0ull, 0ull}, 0};
                it.w.set(ka >> 6, it.w.get(ka >> 6) & ~(1ull << (ka & 63)));        // (narrow: page 0, path ids below 64)
                int nx[H];
                int kn = it.next();
                if (kn >= 0) ld_row(kn, nx);
                while (kn >= 0) {
                    const int k = kn;
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < H; ++r) gS[r * WAVE + lane] = nx[r];
                    kn = it.next();
                    if (kn >= 0) ld_row(kn, nx);
                    __syncthreads();
                    int outr[H];
#pragma unroll
                    for (int r = 0; r < H; ++r) {
                        const int c0 = G[r] & 0xffff, c1 = (unsigned)G[r] >> 16;
                        const int w0v = gS[c0 >> 1], w1v = gS[c1 >> 1];
                        const int v0 = (c0 & 1) ? hi16(w0v) : lo16(w0v), v1 = (c1 & 1) ? hi16(w1v) : lo16(w1v);
                        outr[r] = pk_add(B[r], pack16(v0, v1));
                    }
                    st_row(k, outr);
                }
            }
#endif
            continue;
        }
        int e_i = i, e_w1 = w1, e_flags = flags;   // the row whose epilogue runs at the end of this iteration
        bool e_adv = true;
        if (kWide && went > 1) e_flags &= ~F_LAST;      // (a flagged alpha entry on the general path: its continuation entries close the row)
        // (PATH RETIREMENT: a wide group of which <= KRUN members are left runs here too — a gather run costs two member
        // updates per row whatever is left of the group; like a gather run it leaves bkey alone, which is what the split
        // tables count on for the runs between the groups of a row)
        if (kWide ? wide_run : (KRUN > 0 && (flags & F_INNER) && (nm <= KRUN || (kRet && nme <= KRUN && RG_SWEEP16_GATHER && kGather && a.gather_ok && !semi_end &&
                                                             run_left * (RG_GATHER_PER_MEMBER_ROW * (nm - 1) - RG_GATHER_PER_ROW) >= RG_GATHER_PER_MEMBER_RUN * (nm - 1))) && run_left > 0)) {
            // ---- inner rows of a segment with a small group: the same paths, one group, predecessor = previous row.
            // Their rows stay in registers for the whole run: no row load/store latency, no HBM traffic.  The group
            // alpha of an inner row is its lowest path (alphas[row] == alphas[pred], rg_graph.cpp) = member 0.
            // (CHAINED RUNS, RG_SWEEP16_CHAIN: when the record behind a run — behind its tail — starts another register run on
            // OTHER paths, the block goes on with it instead of returning to the record loop, and it loads the next run's rows
            // BEFORE it stores this run's: gfx9 has one vmcnt for loads and stores, so a wait for freshly loaded rows also
            // drains every store issued before them — in the old order (stores, next record, loads, wait) that was the whole
            // run's rows on every run boundary.  Here the wait covers loads only and the stores drain behind the next run's
            // arithmetic.  rnm / rgm / mk: the run in progress.)
            int rnm = kWide ? wide_n : nme;       // members computed (PATH RETIREMENT: the needed ones)
            unsigned long long rgm = gmask;       // the run's record mask (what its continuation records carry)
            int mk[KRUN > 0 ? KRUN : 1];
            if (kWide) {
#pragma unroll
                for (int kk = 0; kk < KRUN; ++kk) mk[kk] = wide_ids[kk];
            } else {
                unsigned long long tm = kRet ? gm : rgm;
#pragma unroll
                for (int kk = 0; kk < KRUN; ++kk) { mk[kk] = tm ? kbase + __builtin_ctzll(tm) : 0; tm = tm ? (tm & (tm - 1)) : 0; }
            }
            int rr[KRUN > 0 ? KRUN : 1][H];
#pragma unroll
            for (int kk = 0; kk < KRUN; ++kk)
                if (kk < rnm) {
#if defined(RG_SWEEP16_KRUNNOLD) || defined(RG_SWEEP16_NOROWS) || defined(RG_SWEEP16_NOROWS32)
#pragma unroll
                    for (int r = 0; r < H; ++r) rr[kk][r] = lane ^ (kk + t);     // (what the timing-only builds without loads keep)
#endif
#ifndef RG_SWEEP16_KRUNNOLD
                    RG_ROW_LD(mk[kk], rr[kk]);
#endif
                }
#if RG_SWEEP16_RUNWAIT
            // wait for the run's rows HERE: otherwise the compiler's wait sits at the top of the row loop as vmcnt(0) (one
            // counter for loads and stores on gfx9) and every row also waits for the direction-word store of the row before
            {
                RG_STALL_BEGIN();
                __builtin_amdgcn_s_waitcnt(0x0F70);
#if defined(RG_SWEEP16_STALLSTAT) && RG_SWEEP16_STALLSTAT == 1
                RG_STALL_END(st_run); ++st_nrun;
#endif
            }
#endif
#if defined(RG_SWEEP16_STALLSTAT) && RG_SWEEP16_STALLSTAT == 3
            const unsigned long long st_rb = __builtin_amdgcn_s_memtime();
#endif
            int ri = i, rli = li, rslot = slot, rw1 = w1, rfl = 7;
            int rleft = run_left;               // rows left in the counted run, this one included (the record's run field)
            bool tail = false;
            // The row's score profile (two 16-byte LDS reads per lane, ~100 cycles before the diagonal step can use them) is
            // fetched a row AHEAD: a record whose run field counts more rows than itself is followed by the next inner row of
            // its run, so its profile row is known — and `s` is dead — as soon as the members have their steps.
            // (not at 32 columns per lane: a row is 16 registers there and the variant is out of them)
            constexpr bool kAhead = RG_SWEEP16_PROFILE_AHEAD && C <= 16;
            int s[H];
            if (kAhead) load_steps(rli, s);
            for (;;) {          // (chained runs)
            tail = false; rfl = 7;
            // ONE LOOP BODY PER GROUP SIZE (round 6): the rows of a run with RN members, RN a compile-time constant.  With the
            // member count a run-time value the `kk < rnm` tests compiled to ~13 branches per row, the conditions re-materialised as
            // v_cndmask / v_cmp pairs, and the members' rows were shuffled through v_mov pairs where the paths of different counts
            // meet (profiles/r05_isa_sweep16.txt: 25 % of the hot loop's issue slots were SALU / branch / wait)
            auto run_rows = [&](auto rn_tag) __attribute__((always_inline)) {
            constexpr int RN = decltype(rn_tag)::value;
            // (every row of the run — and its tail — has the run's members; wide runs: a member of any page may be a picked path)
            const unsigned long long run_sel = kWide ? (wide_sel ? ~0ull : 0ull) : rgm;
            const int run_page = 0;
            unsigned nrows = 0;                 // rows of this run: the cell counters move once per run, not once per row
            while (true) {
                const int g_i = gcost;
                const int g0 = kSemi ? 0 : g_i;
                int XU[H], XL[H];
#if defined(RG_SWEEP16_STALLSTAT) && RG_SWEEP16_STALLSTAT == 3
                ++st_nrun;
#endif
                if (!kAhead) load_steps(rli, s);
                int lmax;
                RowOps16<C>::alpha(rr[0], s, g_i, g0, lane, XU, XL, lmax);
                RG_ROWSTAT((++st_rn[RN], st_dirs += (kWide ? wide_sel : want_dirs(run_sel, run_page, ri)) ? 1 : 0, st_tail += tail ? 1 : 0));
                if (kWide ? ((kColmax != 1 || dirs != nullptr) && (wide_sel || (rev ? ri > a.dsel_hi : ri < a.dsel_lo))) : want_dirs(run_sel, run_page, ri)) store_dirs(rslot, XU, XL);
                if constexpr (RN > 1) {
                    int MU[H], ML[H];
                    const unsigned lmask = RowOps16<C>::masks(XU, XL, MU, ML);
                    const int src = RowOps16<C>::src_lane(lmask, lane);
                    // (SEL is defined and used under ONE condition: with `if (rnm > 1) select_steps` beside member loops guarded
                    // by `kk < rnm` the compiler carried its eight registers around the whole record loop)
                    int SEL[H];
                    RowOps16<C>::select_steps(SEL, s, MU, g_i, g0, lane);
                    // every member's first pass and cross-lane fetch before the first wait (see member_p1)
                    int zl[KRUN > 0 ? KRUN : 1], vlo[KRUN > 0 ? KRUN : 1];
#pragma unroll
                    for (int kk = 1; kk < KRUN; ++kk)
                        if (kk < RN) { RowOps16<C>::member_p1(rr[kk], SEL, MU, ML, lmask, zl[kk], vlo[kk]); zl[kk] = __shfl(zl[kk], src, WAVE); }
#pragma unroll
                    for (int kk = 1; kk < KRUN; ++kk)
                        if (kk < RN) RowOps16<C>::member_p2(rr[kk], ML, lmask, zl[kk], vlo[kk]);
                }
                if (kAhead && !tail && rleft > 1) {
                    const int tn = t + 1;
                    if ((tn >> 6) != blk) to_block(tn >> 6);
                    load_steps((__builtin_amdgcn_readlane(recs.x, tn & (WAVE - 1)) >> 20) & 7, s);
                }
                ++nrows;
                if (kRec && kColmax != 1 && tail) {
                    if (track) {
                        if (rfl & F_FIRST) row_open = false;
                        int key[C];
                        if (row_open) { keys_ld(key); fold_keys(key, rr[0], mk[0]); } else set_keys(key, rr[0], mk[0]);
#pragma unroll
                        for (int kk = 1; kk < KRUN; ++kk) if (kk < RN) fold_keys(key, rr[kk], mk[kk]);
                        row_open = true;
                        // the tail's row: its epilogue when this was its last group
                        if (rfl & F_LAST) { row_end(ri, ((rw1 >> 20) & 511) - 1, key); row_open = false; }
                        else keys_st(key);
                    }
                } else if (track && kRec && kColmax != 1) {
                    // LAZY KEYS (rows in registers): the best VALUE per column is a packed maximum over the members (8
                    // v_pk_max per member instead of 32 key instructions); it feeds the packed column maxima directly, and
                    // the (value, path) keys are only built when some column of some lane can reach its threshold
                    // (saturating packed compare against thz: a superset of the exact test row_end applies)
                    int bv[H];
#pragma unroll
                    for (int r = 0; r < H; ++r) bv[r] = rr[0][r];
#pragma unroll
                    for (int kk = 1; kk < KRUN; ++kk)
                        if (kk < RN) {
#pragma unroll
                            for (int r = 0; r < H; ++r) bv[r] = pk_max(bv[r], rr[kk][r]);
                        }
                    if (kColmax == 2) {
#pragma unroll
                        for (int r = 0; r < H; ++r) cmv[kColmax == 2 ? r : 0] = pk_max(cmv[kColmax == 2 ? r : 0], bv[r]);
                    }
                    const int knm_row = ((rw1 >> 20) & 511) - 1;
                    int acc = -1;
                    // (a row without members besides its alpha: the lane's largest value per half chain is known from the alpha's
                    // stitching, so one packed compare against the lane's lowest threshold per half decides for the wave whether
                    // the per-column test is needed at all — in the reverse sweep it rarely is: ~800 records per read and sweep)
                    bool fine = true;
                    if constexpr (RN == 1 && kRec) {
                        if (knm_row >= 0) fine = __any(((unsigned)pk_sub_sat(lmax, thzmin) & 0x80008000u) != 0x80008000u);
                    }
                    if (!fine) {
                        // no column of any lane can reach its threshold
                    } else if (knm_row >= 0) [[likely]] {
#pragma unroll
                        for (int r = 0; r < H; ++r) acc &= pk_sub_sat(bv[r], thz[kRec ? r : 0]);
                    } else {
                        // (a row every path visits whose group retirement has thinned out to <= KRUN members: one threshold for the
                        // lane.  Written as a maximum, not as the loop above with another operand: the compiler merged the two
                        // loops into one behind eight v_mov selects of the threshold registers in EVERY row)
                        int m = bv[0];
#pragma unroll
                        for (int r = 1; r < H; ++r) m = pk_max(m, bv[r]);
                        acc = pk_sub_sat(m, minplain2);
                    }
                    const bool lhit = ((unsigned)acc & 0x80008000u) != 0x80008000u;
                    if (__any(lhit)) {      // some half >= its threshold
                        // the (value, path) keys of the row, built only now and half a lane at a time (columns 0..H-1 from the low
                        // halves of the members' words, then H..C-1 from the high halves): H key registers live, not C
                        bool mine;
                        int4* rp = rec_slot(ri, lhit, mine);
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            int hk[H];
#pragma unroll
                            for (int kk = 0; kk < KRUN; ++kk)
                                if (kk < RN) {
                                    int kv = mk[kk];
                                    asm volatile("" : "+v"(kv));      // (the path id through a VGPR: see set_keys)
#pragma unroll
                                    for (int r = 0; r < H; ++r) {
                                        const int key = half == 0 ? (int)(((unsigned)rr[kk][r] << 16) | (unsigned)kv)
                                                                  : (int)(((unsigned)rr[kk][r] & 0xffff0000u) | (unsigned)kv);
                                        hk[r] = kk == 0 ? key : max(hk[r], key);
                                    }
                                }
                            rec_half(rp, mine, half, hk);
                        }
                    }
                } else if (track) {
                    int key[C];
                    set_keys(key, rr[0], mk[0]);
#pragma unroll
                    for (int kk = 1; kk < KRUN; ++kk) if (kk < RN) fold_keys(key, rr[kk], mk[kk]);
                    row_end(ri, ((rw1 >> 20) & 511) - 1, key);
                }
                if (semi_end) {
#pragma unroll
                    for (int kk = 0; kk < KRUN; ++kk) if (kk < RN) end_fold(mk[kk], ri, rr[kk]);
                    end_row_done(ri);
                }
                t += kWide ? went : 1;                          // (wide runs: over the row's continuation entries)
                if (tail || t >= nsteps) break;                 // (a tail ends its run)
                if (rleft > 1) {
                    // the run goes on: a record whose run field counts more rows than itself is followed by the next inner row
                    // of the same segment — same group, flags 7 (rg_steps.cpp builds the field that way and moves runs whole) —
                    // so neither its flags nor its members are looked at, and only two of its four words are fetched
                    if ((t >> 6) != blk) to_block(t >> 6);
                    const int nw0 = __builtin_amdgcn_readlane(recs.x, t & (WAVE - 1)), nw1 = __builtin_amdgcn_readlane(recs.y, t & (WAVE - 1));
                    ri = nw0 & 0xfffff; rli = (nw0 >> 20) & 7; rslot = nw1 & 0xfffff; rw1 = nw1;
                    rleft = (nw0 >> 26) & 63;
                    continue;
                }
                const int pw = peek_w0(t);
                const int nf = (pw >> 23) & 7;
                const bool to_tail = kRec && kColmax != 1 && (nf & F_INNER) && ((pw >> 26) & 63) == 0;
                if (!to_tail && (nf != 7 || ((pw >> 26) & 63) == 0)) break;   // next record starts another segment (a HEAD or a general row)
                // ... or is an inner row of ANOTHER segment: in a split table the rows of a segment whose first row had all
                // its groups moved away as tails can follow an unrelated run (found by test_random_dag_graphs)
                if (peek_gm(t) != rgm) break;
                int nw0, nw1;
                unsigned long long ngm;
                fetch(t, nw0, nw1, ngm);
                ri = nw0 & 0xfffff; rli = (nw0 >> 20) & 7; rslot = nw1 & 0xfffff; rw1 = nw1;
                tail = to_tail; rfl = nf;
                rleft = to_tail ? 0 : (nw0 >> 26) & 63;
                if (kAhead) load_steps(rli, s);
            }
            cells += (unsigned long long)nrows * (unsigned long long)(kWide ? wide_total : __popcll(rgm));
            done += (unsigned long long)nrows * (unsigned long long)RN;
            };
            run_dispatch_from<1, (KRUN > 0 ? KRUN : 1)>(run_rows, rnm);
            // ---- the run (and its tail) is over: rows in rr, t = the next record
            if (tail) {
                if (semi_end && (rfl & F_LAST)) end_row_done(ri);
                if (rfl & F_LAST) row_open = false;       // (untracked sweeps: nothing folded, nothing to close)
            }
            bool chain = false;
            unsigned long long gm2 = 0;
            // (not while paths are being retired: the chained run's members would have to be masked and may all be gone)
            if ((RG_SWEEP16_CHAIN == 2 || (RG_SWEEP16_CHAIN == 1 && !kTrack)) && !kWide && !semi_end && t < nsteps && (!kRet || next_eval == INT32_MAX)) {
                const int pw = peek_w0(t);
                // a HEAD (4 alone) or an inner row (7) with rows left starts a register / gather run; <= KRUN paths: a register run
                if ((((pw >> 23) & 7) & F_INNER) && ((pw >> 26) & 63) != 0) {
                    gm2 = peek_gm(t);
                    chain = __popcll(gm2) <= KRUN && (gm2 & rgm) == 0;
                }
            }
            if (!chain) break;
            int rn[KRUN > 0 ? KRUN : 1][H];
            int mk2[KRUN > 0 ? KRUN : 1];
            {
                unsigned long long tm = gm2;
#pragma unroll
                for (int kk = 0; kk < KRUN; ++kk) { mk2[kk] = tm ? kbase + __builtin_ctzll(tm) : 0; tm = tm ? (tm & (tm - 1)) : 0; }
#pragma unroll
                for (int kk = 0; kk < KRUN; ++kk)
                    if (kk < __popcll(gm2)) RG_ROW_LD(mk2[kk], rn[kk]);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the next run's rows (every older store is long done)
#pragma unroll
            for (int kk = 0; kk < KRUN; ++kk)
                if (kk < rnm) RG_ROW_ST(mk[kk], rr[kk]);
            rnm = __popcll(gm2);
            rgm = gm2;
#pragma unroll
            for (int kk = 0; kk < KRUN; ++kk) {
                mk[kk] = mk2[kk];
#pragma unroll
                for (int r = 0; r < H; ++r) rr[kk][r] = rn[kk][r];
            }
            {
                int nw0, nw1;
                unsigned long long ngm;
                fetch(t, nw0, nw1, ngm);
                ri = nw0 & 0xfffff; rli = (nw0 >> 20) & 7; rslot = nw1 & 0xfffff; rw1 = nw1;
                rleft = (nw0 >> 26) & 63;
                if (kAhead) load_steps(rli, s);
            }
            }                   // (chained runs)
#ifdef RG_SWEEP16_KRUNNOST
#pragma unroll
            for (int kk = 0; kk < KRUN; ++kk) if (kk < rnm) { int sink = 0; for (int r = 0; r < H; ++r) sink ^= rr[kk][r]; asm volatile("" :: "v"(sink)); }      // (timing-only: no run-end stores)
#else
#pragma unroll
            for (int kk = 0; kk < KRUN; ++kk)
                if (kk < rnm) RG_ROW_ST(mk[kk], rr[kk]);
#endif
#if defined(RG_SWEEP16_STALLSTAT) && RG_SWEEP16_STALLSTAT == 3
            st_gen += __builtin_amdgcn_s_memtime() - st_rb;
#endif
            continue;           // (a tail's epilogue ran above)
        } else {
        const int g_i = gcost;
        const int g0 = kSemi ? 0 : g_i;
        int s[H];
        load_steps(li, s);       // (every record: register / gather runs of other rows may lie between the groups of one row)
        int MUl[kWide ? 1 : H], MLl[kWide ? 1 : H];
        unsigned lmaskl = 0;
        int srcl = 0;
        int (&MU)[H] = *reinterpret_cast<int (*)[H]>(kWide ? MUw : MUl);
        int (&ML)[H] = *reinterpret_cast<int (*)[H]>(kWide ? MLw : MLl);
        unsigned& lmask = kWide ? lmaskw : lmaskl;
        int& src = kWide ? srcw : srcl;
        {
            unsigned long long rest = cont ? gm : gm & ~(1ull << (ga - kbase));
            cells += (unsigned long long)nm;
            done += (unsigned long long)nme;
            int nxt[H];
#if defined(RG_SWEEP16_NOROWS) || defined(RG_SWEEP16_NOROWS32)
#pragma unroll
            for (int r = 0; r < H; ++r) nxt[r] = s[r] ^ t;
#endif
            int knext = -1;
            if (rest) {
                knext = kbase + __builtin_ctzll(rest);
                rest &= rest - 1;
                RG_ROW_LD(knext, nxt);
            }
            int key[C];           // (untracked sweeps never touch it)
            bool have = false;            // key[] holds the keys of the row's earlier records
            if (track) {
                if (flags & F_FIRST) row_open = false;
                have = row_open;
                if (have) keys_ld(key);
            }
            if (!cont) {
                int rowa[H];
#if defined(RG_SWEEP16_NOROWS) || defined(RG_SWEEP16_NOROWS32)
#pragma unroll
                for (int r = 0; r < H; ++r) rowa[r] = s[r];
#endif
                RG_ROW_LD(ga, rowa);
#if defined(RG_SWEEP16_STALLSTAT) && RG_SWEEP16_STALLSTAT == 1
                { RG_STALL_BEGIN(); RG_STALL_END(st_gen); }
#endif
                int XU[H], XL[H];
                int lmax_unused;
                RowOps16<C>::alpha(rowa, s, g_i, g0, lane, XU, XL, lmax_unused);
                RG_ROW_ST(ga, rowa);
                // (a group that spans pages — continuation entries follow — may hold a picked path in a page this entry does not
                // see: it stores its word whatever its own members are)
                bool wd = want_dirs(gmask, page, i);
                if (kWide && !wd && t + 1 < nsteps) wd = peek_w1(t + 1) < 0;
                RG_ROWSTAT((++st_gen, st_genmem += nme, st_dirs += wd ? 1 : 0));
                if (wd) store_dirs(slot, XU, XL);
                lmask = RowOps16<C>::masks(XU, XL, MU, ML);
                src = RowOps16<C>::src_lane(lmask, lane);
                if (track) {
                    if (!have) set_keys(key, rowa, ga); else fold_keys(key, rowa, ga);
                    have = true;
                }
                if (semi_end) end_fold(ga, i, rowa);
            }
            // (a continuation entry — members of another page of the group the previous entry started — takes MU / ML / lmask /
            // src from that entry's alpha.  SEL is built unconditionally: eight instructions on a rare path, and no array that
            // the compiler has to carry around the record loop because it cannot see that definition and use go together)
            int SEL[H];
            RowOps16<C>::select_steps(SEL, s, MU, g_i, g0, lane);
            while (knext >= 0) {
                const int k = knext;
                int cur[H];
#pragma unroll
                for (int r = 0; r < H; ++r) cur[r] = nxt[r];
                if (rest) {
                    knext = kbase + __builtin_ctzll(rest);
                    rest &= rest - 1;
                    RG_ROW_LD(knext, nxt);
                } else knext = -1;
                RowOps16<C>::member(cur, SEL, lane, MU, ML, lmask, src);
                RG_ROW_ST(k, cur);
                if (track) fold_keys(key, cur, k);
                if (semi_end) end_fold(k, i, cur);
            }
            if (track) {
                // (a continuation entry always follows its group's first entry: have is true by then)
                row_open = true;
                if (e_flags & F_LAST) row_end(e_i, ((e_w1 >> 20) & 511) - 1, key);
                else keys_st(key);
            }
        }
        }
        if (semi_end && (e_flags & F_LAST)) end_row_done(e_i);
        if (e_flags & F_LAST) row_open = false;       // (so that a skipped FIRST record of a later row has nothing to reset)
        if (e_adv) ++t;
    }

#ifdef RG_SWEEP16_ROWSTAT
    if (lane == 0 && (rd == 0 || rd == 1 || rd == 7 || rd == 100))
        printf("[rowstat] read %d %s: run rows by members 1:%u 2:%u 3:%u 4:%u (tails %u), gather rows %u (members %u), general records %u (members %u), direction words %u, records %d\n",
               rd, rev ? "rev" : "fwd", st_rn[1], st_rn[2], st_rn[3], st_rn[4], st_tail, st_grow, st_gmem, st_gen, st_genmem, st_dirs, nsteps);
#endif
    // ---- outputs ----
    if (kColmax == 1 && a.colmax_out) {
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = lane * C + q;
            if (c < ncols) {
                a.colmax_out[(long long)rd * wpad + (rev ? n - c : c)] = ckey[kColmax == 1 ? q : 0] == INT32_MIN ? NEG32 : (ckey[kColmax == 1 ? q : 0] >> 16) + c * gcost;
                if (a.colarg_out) a.colarg_out[(long long)rd * wpad + (rev ? n - c : c)] = (crow[kColmax == 1 ? q : 0] << 8) | (ckey[kColmax == 1 ? q : 0] & 255);
            }
        }
    }
    if (kColmax == 2 && a.colmax_out) {
        // (an opaque copy of the lane: the column indices and their `c < ncols` predicates are recomputed here instead of
        // living — spilled — across the whole record loop)
        int ol = lane;
        asm volatile("" : "+v"(ol));
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int c = ol * C + q;
            const int pv = cmv[kColmax == 2 ? q % H : 0];
            const int z = q >= H ? hi16(pv) : lo16(pv);
            if (c < ncols) a.colmax_out[(long long)rd * wpad + (rev ? n - c : c)] = z <= NEG16 ? NEG32 : z + c * gcost;
        }
    }
    if (a.ncand_out && lane == 0) a.ncand_out[rd] = ncand;
    __syncthreads();
    if (!rev && !kSemi) {
        const int ql = n % C, ln = n / C;
        for (int k = lane; k < P; k += WAVE) {
            const int pv = rows[(long long)k * wrow + ln * H + (ql % H)];
            // (a retired path's row is stale: its true final score is below the bound k_verify checks the result against)
            rs->sink_val[k] = (kRet && !((word_of(needed, k >> 6) >> (k & 63)) & 1ull)) ? NEG32 : (ql >= H ? hi16(pv) : lo16(pv)) + n * gcost;
        }
    }
    if (semi_end) {
        for (int k = lane; k < P; k += WAVE) { rs->sink_val[k] = endv[k]; rs->path_end_row[k] = endr[k]; }
        if (lane == ln_end) { rs->s0 = gbest_val; rs->end_row_best = gbest_row; rs->seed_path = gbest_path; }
    }
    if (lane == 0 && a.count_cells) {
        // counted: every member row of the table (what the reference updates) — from the table builder when records may have
        // been jumped over; performed: what this wave carried out
        const unsigned long long all = kRet && a.table_members ? a.table_members : cells;
#ifdef RG_SWEEP16_STALLSTAT
        {
            const unsigned long long tot = __builtin_amdgcn_s_memtime() - st_t0;
            atomicAdd(a.cells, (tot >> 8) | ((st_gen >> 8) << 32));
            atomicAdd(a.cells + 1, (st_run >> 8) | (st_nrun << 32));
            return;
        }
#endif
#ifdef RG_SWEEP16_RETSTAT
        if (kRet) { atomicAdd(a.cells, stat_e | (stat_n << 32)); atomicAdd(a.cells + 1, stat_h | ((unsigned long long)(rev ? 0 : 1) << 48)); return; }
#endif
        atomicAdd(a.cells, all * (unsigned long long)(n + 1));
        atomicAdd(a.cells + 1, done * (unsigned long long)(n + 1));
    }
}

// Host-side admission test: uniform gap cost and every STORED value provably inside the 16-bit budget.
// What the rows hold is z = A - c * g (c: column, g: the gap cost), not A.  A row starts at z = 0 and changes only through
//   U: z + g_i (= z + g, the gap column is uniform)        D: z_diagonal + (s - g)        L: z_left (a copy)
// — whoever chose the move (members follow their alpha), every cell is its chain's start plus one such step per move.  So after
// at most `rows` graph rows and n read bases
//   zlo = -(rows + 2) * |g| - (n + 2) * max(0, g - min s)   <=   z   <=   (n + 2) * max(0, max s - g) = zhi
// (round 4 bounded |A| <= (rows + n) * max|entry| <= 24 000 instead, which refused -X 6 at 1 kbp and every read longer than
// ~1.2 kbp: the all-gap corner of A is -(rows + n) |g|, the same corner of z only -rows |g|).  Required:
//   * zlo above the "minus infinity" of a 16-bit lane (NEG16 = -30 000) with a step of head-room, zhi below +29 000;
//   * zhi - zlo <= 32 000: every difference of two stored values (direction masks from the sign of d - max(d, u), the
//     gather runs' member deltas) fits a signed half;
//   * |A| <= (rows + n + 2) * max|entry| <= 32 000: outputs convert back (A = z + c g) inside key << 16 arithmetic;
//   * gap entries <= 0: the border column (c = 0: z = A = i * g) then holds values <= 0, so that d - max(d, u) of lane 0's
//     column 0 (d = NEG16 + s, u = the border value) stays above -32768 and the U mask keeps its sign.
bool sweep16_admissible(const DevScores& sc, int max_path_rows, int max_n, int C) {
    for (int b = 1; b < 5; ++b) if (sc.t[b * 6 + 5] != sc.t[5]) return false;
    for (int b = 0; b < 5; ++b) if (sc.t[b * 6 + 5] > 0 || sc.t[5 * 6 + b] > 0) return false;
    long long maxabs = 0, smin = INT32_MAX, smax = INT32_MIN;
    for (int x = 0; x < 6; ++x)
        for (int y = 0; y < 6; ++y) {
            if (x == 5 && y == 5) continue;
            const long long v = sc.t[x * 6 + y];
            maxabs = std::max(maxabs, v < 0 ? -v : v);
            if (x < 5 && y < 5) { smin = std::min(smin, v); smax = std::max(smax, v); }
        }
    if (maxabs > 1000) return false;
    const long long g = sc.t[5];                 // <= 0
    const long long rows = max_path_rows + 2, n = max_n + 2;
    const long long zlo = rows * g - n * std::max(0ll, g - smin);
    const long long zhi = n * std::max(0ll, smax - g);
    if (zlo < -29000 || zhi > 29000 || zhi - zlo > 32000) return false;
    if ((rows + n) * maxabs > 32000) return false;                 // |A| of every cell
    if ((long long)(C / 2 + 2) * maxabs > 2000) return false;
    return true;
}

// Forward (row, lane) records -> Cand list, filtered with the final bound (the same test k_search applies).
template <int C>
__global__ __launch_bounds__(256) void k_expand(ExpandArgs a) {
    const int rd = blockIdx.x;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE, nwv = blockDim.x / WAVE;
    ReadState* rs = a.state + rd;
    if (rs->status & (ST_BAD_BASE | ST_WOULD_PANIC)) return;
    const unsigned nrec = a.nrec[rd];
    if (nrec > a.frec_cap) { if (threadIdx.x == 0) rs->status |= ST_OVERFLOW; return; }
    const int bound = rs->bound;
    const int nread = (int)(a.read_off[rd + 1] - a.read_off[rd]);
    const int oob = max((int)((float)(nread + 1) * (1.0f - a.rbw) / 2.0f), 1);
    const int* base = a.frec + (long long)rd * a.frec_cap * (4 + C);
    Cand* out = a.fcand + (long long)rd * a.fcap;
    const int* wr = a.wr + (long long)rd * a.wpad;
    // C / 4 lanes per record, one 16-byte load of four keys each: a wave iteration covers 256 / C records
    constexpr int LPR = C / 4, RPW = WAVE / LPR;
    const int part = lane % LPR, sub = lane / LPR;
    for (unsigned t0 = wv * RPW; t0 < nrec; t0 += nwv * RPW) {      // latency-bound: several waves per read
        const unsigned t = t0 + sub;
        if (t >= nrec) continue;
        const int* rp = base + (long long)t * (4 + C);
        const int rl = rp[0];
        const int4 k4 = *reinterpret_cast<const int4*>(rp + 4 + 4 * part);
        const int keys[4] = {k4.x, k4.y, k4.z, k4.w};
        const int knm = a.knm[rl >> 6];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cc = (rl & 63) * C + 4 * part + e;
            if (cc >= (a.rev ? nread : nread + 1)) continue;               // column does not exist: the key is garbage
            const int key = keys[e] + cc * a.gcost * 65536;                // records hold z-space keys: A = z + c * g
            const int val = key >> 16;
            const int col = a.rev ? nread - cc : cc;
            if (col < oob || col >= nread + 1 - oob) continue;             // outside the recombination band (-B)
            if (val + wr[col] - a.brc < bound) continue;                   // implies the sweep's emission threshold
            if (knm >= 0 && key <= knm) continue;        // winner of the cell is not a member path: the reference's entry is 0
            const unsigned pos = atomicAdd(&a.nf[rd], 1u);
            if (pos < a.fcap) { Cand cd; cd.row = rl >> 6; cd.col = col; cd.val = val; cd.path = key & 0xffff; out[pos] = cd; }
        }
    }
}

void launch_expand(const ExpandArgs& a, int nreads, int C, hipStream_t s) {
    switch (C) {
        case 4: hipLaunchKernelGGL((k_expand<4>), dim3(nreads), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_expand<8>), dim3(nreads), dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((k_expand<16>), dim3(nreads), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((k_expand<32>), dim3(nreads), dim3(256), 0, s, a); break;
    }
}

// Column maxima of a sweep and the cells attaining them, from the sweep's own records: per column the best member-winner
// key (value << 16 | path) and its row.  Only cells above the emission threshold are in the records: for every other
// cell of that sweep no partner can reach the seed, so pruning against these maxima stays exact.
template <int C>
__global__ __launch_bounds__(256) void k_colmax_rec(ExpandArgs a, int* colmax_out, int* colarg_out) {
    const int rd = blockIdx.x;
    extern __shared__ unsigned long long cm_best[];      // [wpad]: (key + 2^31) << 32 | row
    const int nread = (int)(a.read_off[rd + 1] - a.read_off[rd]);
    for (int j = threadIdx.x; j < a.wpad; j += blockDim.x) cm_best[j] = 0ull;
    __syncthreads();
    const ReadState* rs = a.state + rd;
    const unsigned nrec = a.nrec[rd];
    if (!(rs->status & (ST_BAD_BASE | ST_WOULD_PANIC)) && nrec <= a.frec_cap) {
        const int oob = max((int)((float)(nread + 1) * (1.0f - a.rbw) / 2.0f), 1);
        const int* base = a.frec + (long long)rd * a.frec_cap * (4 + C);
        for (unsigned t = threadIdx.x / C; t < nrec; t += blockDim.x / C) {
            const int* rp = base + (long long)t * (4 + C);
            const int rl = rp[0], q = threadIdx.x % C;
            const int cc = (rl & 63) * C + q;
            if (cc >= (a.rev ? nread : nread + 1)) continue;
            const int key = rp[4 + q] + cc * a.gcost * 65536;              // z-space key -> value << 16 | path
            const int col = a.rev ? nread - cc : cc;
            if (col < oob || col >= nread + 1 - oob) continue;
            // (a plain read first: the maximum only grows, so a key that does not beat what is there now never will — most keys
            // of most records stop here, before the row lookup and the 64-bit LDS atomic)
            const unsigned long long mine = ((unsigned long long)((unsigned)key ^ 0x80000000u) << 32) | (unsigned)(rl >> 6);
            if (mine <= reinterpret_cast<volatile unsigned long long*>(cm_best)[col]) continue;
            const int knm = a.knm[rl >> 6];
            if (knm >= 0 && key <= knm) continue;
            atomicMax(&cm_best[col], mine);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < a.wpad; j += blockDim.x) {
        const unsigned long long b = cm_best[j];
        const int key = (int)((unsigned)(b >> 32) ^ 0x80000000u);
        colmax_out[(long long)rd * a.wpad + j] = b ? key >> 16 : NEG32;
        colarg_out[(long long)rd * a.wpad + j] = b ? (int)(((unsigned)b << 8) | ((unsigned)key & 255u)) : 0;
    }
}

void launch_colmax_rec(const ExpandArgs& a, int* colmax_out, int* colarg_out, int nreads, int C, hipStream_t s) {
    const size_t bytes = (size_t)a.wpad * sizeof(unsigned long long);
    switch (C) {
        case 4: hipLaunchKernelGGL((k_colmax_rec<4>), dim3(nreads), dim3(256), bytes, s, a, colmax_out, colarg_out); break;
        case 8: hipLaunchKernelGGL((k_colmax_rec<8>), dim3(nreads), dim3(256), bytes, s, a, colmax_out, colarg_out); break;
        case 16: hipLaunchKernelGGL((k_colmax_rec<16>), dim3(nreads), dim3(256), bytes, s, a, colmax_out, colarg_out); break;
        default: hipLaunchKernelGGL((k_colmax_rec<32>), dim3(nreads), dim3(256), bytes, s, a, colmax_out, colarg_out); break;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// k_layer on packed rows (k_layer of rg_pathwise.hip is the i32 form: 16 columns per lane unpacked, ~350 wave-instructions
// per row; this one ~220).  Same job: rebuild the layer of ONE path from the sweep's direction words (rows of that path
// only) and store, per cell, the move the reference's traceback takes there (pathwise_alignment_output.rs:32-110,
// recombination_output.rs:391-470, 659-736: d, u, l re-derived from the path's own layer, D > U > L).  Rows are in the
// sweep's z-space (z = A - c * g): d, u and l of one cell are compared at the same column, so the comparison is the same.
// The walkers take the L step's cost from the key ('-', read base), the sweep from (read base, '-'): the
// driver only sends batches here whose two gap tables agree (every CLI matrix); others keep the i32 form.
// k_layer16 at <= 16 columns per lane is compiled for 64 registers (8 waves per SIMD; 70 before, one register spilled now).  Not for
// its own occupancy: in the stream it runs beside the sweeps of the other handles, two sweep waves of 224 registers leave 64 of a
// SIMD's 512, and a kernel that does not fit waits for a sweep wave to retire and then holds that wave's slot (k_layer_fwd took
// 7-8 ms in the stream against 2.7 alone).  Config 5: 117.2 k -> 119.7 k reads/s (A/B/A/B/A/B on one box, profiles/r06_notes.md).
#ifndef RG_LAYER16_WAVES
#define RG_LAYER16_WAVES 8
#endif
template <int C>
__global__ __launch_bounds__(64, C <= 16 ? RG_LAYER16_WAVES : 1) void k_layer16(LayerArgs a) {
    constexpr int H = C / 2;
    const int rd = blockIdx.x;
    const int lane = threadIdx.x;
    const PathGraphDev& g = a.g;
    ReadState* rs = a.state + rd;
    if (rs->status & (ST_BAD_BASE | ST_WOULD_PANIC | ST_OVERFLOW | ST_RETRY)) return;
    const bool rev = a.rev;
    const int path = rev ? rs->rev_path : rs->fwd_path;
    const bool recomb = rs->fwd_path != rs->rev_path;
    if (rev && !recomb) return;  // no recombination: reverse layer not needed
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    const uint8_t* read = a.reads + ro - 1;
    const int ncols = rev ? n : n + 1;
    const int GAP = 5;
    extern __shared__ __attribute__((aligned(16))) int lds16[];
    int* sct = lds16;                    // [36]
    int* s2 = lds16 + 64;                // [5][64] packed (s - g) pairs by (code_lo | code_hi << 3)
    int* sprof = s2 + 5 * 64;            // [5][64][H]: the lane's packed diagonal steps per row base (see k_sweep16)
    if (lane < 36) sct[lane] = a.sc.t[lane];
    __syncthreads();
    const int gcost = __builtin_amdgcn_readfirstlane(sct[GAP]);      // (= score(b, '-') of every row base b: see k_sweep16)
    for (int e = lane; e < 5 * 64; e += WAVE) {
        const int li = e >> 6, cl = e & 7, ch = (e >> 3) & 7;
        s2[e] = (cl < 6 && ch < 6) ? pack16(sct[li * 6 + cl] - gcost, sct[li * 6 + ch] - gcost) : 0;
    }
    __syncthreads();
    int cur[H];
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const int c0 = lane * C + r, c1 = c0 + H;
        int k0 = 4, k1 = 4;
        if (c0 >= 1 && c0 < ncols) k0 = rev ? read[n - c0 + 1] : read[c0];
        if (c1 >= 1 && c1 < ncols) k1 = rev ? read[n - c1 + 1] : read[c1];
#pragma unroll
        for (int li = 0; li < 5; ++li) sprof[(li * WAVE + lane) * H + r] = s2[li * 64 + (k0 | (k1 << 3))];
        cur[r] = pack16(c0 < ncols ? 0 : NEG16, c1 < ncols ? 0 : NEG16);       // the gap-only start row: z = 0
    }
    __syncthreads();
    uint32_t* tdir = reinterpret_cast<uint32_t*>(a.layer) + (long long)rd * a.layer_stride;
    const int* prow = rev ? a.rprow : a.fprow;
    const int* pslot = rev ? a.rpslot : a.fpslot;
    const int* poff = rev ? a.rpoff : a.fpoff;
    const uint32_t* dirs = a.dirs + (long long)rd * a.dirs_stride;
    const int nrows = poff[path + 1] - poff[path];
    const int start_row = recomb ? rs->fen : rs->end_row;
    const int start_col = recomb ? rs->rec_col : n;
    constexpr unsigned FULL = RowOps16<C>::FULL;
    constexpr unsigned LOWH = H >= 16 ? 0xffffu : ((1u << H) - 1u);
#ifndef RG_LAYER16_PF
#define RG_LAYER16_PF 2      // (rows fetched ahead: 4 until round 6 — see RG_LAYER16_WAVES)
#endif
    constexpr int PF = RG_LAYER16_PF;
    int pf_li[PF], pf_row[PF];
    uint32_t pf_w0[PF], pf_w1[PF];
    // TWO-STAGE look-ahead (round 6).  A row needs its base code and its direction word, and their addresses come from the path's
    // row / slot lists: two DEPENDENT loads.  Until round 6 both were issued in one step — `poff[path]` (loop-invariant) was even
    // reloaded in front of them — and the row loop opened with two full memory round trips every iteration (the ISA: three loads
    // behind three `s_waitcnt vmcnt(0)`; 2.7 us per row, all of it latency).  Now the list entries of row t + PF + 1 are loaded
    // while row t is computed, and the loads they address are issued one iteration later, when they have landed.
    const int pbase = poff[path];
    int nx_row = -1, nx_slot = 0;               // list entries of the NEXT row to be fetched (index nx_t)
    auto fetch_idx = [&](int tt) {
        nx_row = -1; nx_slot = 0;
        if (tt < nrows) { nx_row = prow[pbase + tt]; nx_slot = pslot[pbase + tt]; }
    };
    auto prefetch = [&](int& li_o, int& row_o, uint32_t& w0_o, uint32_t& w1_o) {       // the row whose list entries are in nx_row / nx_slot
        li_o = 4; row_o = nx_row; w0_o = 0; w1_o = 0;
        if (nx_row >= 0) {
            li_o = g.lnz[nx_row];
            w0_o = dirs[(long long)nx_slot * a.dir_words + lane];
            if (C > 16) w1_o = dirs[(long long)nx_slot * a.dir_words + WAVE + lane];
        }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) { fetch_idx(k); prefetch(pf_li[k], pf_row[k], pf_w0[k], pf_w1[k]); }
    fetch_idx(PF);
    for (int t = 0; t < nrows; ++t) {
        const int li = pf_li[0], irow = pf_row[0];
        const uint32_t word0 = pf_w0[0], word1 = pf_w1[0];
#pragma unroll
        for (int k = 0; k + 1 < PF; ++k) { pf_li[k] = pf_li[k + 1]; pf_row[k] = pf_row[k + 1]; pf_w0[k] = pf_w0[k + 1]; pf_w1[k] = pf_w1[k + 1]; }
        prefetch(pf_li[PF - 1], pf_row[PF - 1], pf_w0[PF - 1], pf_w1[PF - 1]);       // row t + PF: its list entries were loaded an iteration ago
        fetch_idx(t + PF + 1);
        const int g_i = gcost;
        const int g0 = a.semi ? 0 : g_i;
        const int GI = pack16(g_i, g_i);
        const int GI0 = lane == 0 ? pack16(g0, g_i) : GI;
        // direction masks of the row in the packed-bit form of k_sweep16 (bit r: column r of the lane, bit 16 + r: column H + r)
        unsigned um2, lm2;
        if constexpr (C <= 16) {
            // (RowOps16::dir_word: after a bit reversal byte 3 = U of the low columns, bit r = register r; byte 2 = U high; 1 = L low; 0 = L high)
            const unsigned t = __brev(word0);
            um2 = ((t >> 24) & LOWH) | (t & (LOWH << 16));
            lm2 = ((t >> 8) & LOWH) | ((t & LOWH) << 16);
        } else { um2 = word0; lm2 = word1; }
        int s[H], MU[H], ML[H], SEL[H];
        {
            const int* sp = sprof + (li * WAVE + lane) * H;
#pragma unroll
            for (int r = 0; r < H; ++r) s[r] = sp[r];
        }
#pragma unroll
        for (int r = 0; r < H; ++r) {
            MU[r] = pk_sub(0, (int)((um2 >> r) & (unsigned)ONE2));       // 0 - 1 = 0xffff per half
            ML[r] = pk_sub(0, (int)((lm2 >> r) & (unsigned)ONE2));
            SEL[r] = bfi(MU[r], r == 0 ? GI0 : GI, s[r]);
        }
        const unsigned long long have = __ballot((lm2 & FULL) != FULL) & ((1ull << lane) - 1ull);
        const int src = have ? 63 - __clzll((long long)have) : 0;
        int old[H];
#pragma unroll
        for (int r = 0; r < H; ++r) old[r] = cur[r];
        RowOps16<C>::member(cur, SEL, lane, MU, ML, lm2, src);
        // ---- traceback decisions of this row ----
        // the reverse matrix keeps its start row (row L-1) delta-encoded in the reference (absolute_scores skips it): path 0
        // reads its absolute value there, every other path reads 0 (pathwise_alignment_recombination.rs:748): A = 0 is
        // z = -c * g in z-space
        const bool zero_prev = rev && t == 0 && path != 0;
        if (zero_prev) {
#pragma unroll
            for (int r = 0; r < H; ++r) {
                const int c0 = lane * C + r, c1 = c0 + H;
                old[r] = pack16(-c0 * gcost, -c1 * gcost);
            }
        }
        int o1 = __builtin_amdgcn_alignbit(old[H - 1], dpp_shr1(old[H - 1], NEGPAIR), 16);      // old row, column c - 1
        int nl = __builtin_amdgcn_alignbit(cur[H - 1], dpp_shr1(cur[H - 1], NEGPAIR), 16);      // new row, column c - 1
        if (zero_prev && lane == 0) o1 = pack16(gcost, hi16(o1));      // (k_layer reads 0 for the cell left of column 0 too: z = +g)
        unsigned tw0 = 0, tw1 = 0;
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const int d = pk_add(o1, s[r]);
            const int u = pk_add(old[r], GI);                           // (the walkers add the row's gap cost in column 0 too, semiglobal or not)
            const int mx = pk_max(pk_max(d, u), nl);
            const int nd = pk_sign(pk_sub_sat(d, mx));                  // 0xffff where D does not attain the maximum
            const int nu = pk_sign(pk_sub_sat(u, mx));
            // 1 = D, 2 = U, 3 = L per half: bit 1 = nd, bit 0 = ~nd | nu
            const int b0 = __builtin_amdgcn_bitop3_b32(nd, nu, ONE2, 0x8A);           // (~nd | nu) & 1
            const unsigned code2 = (unsigned)(b0 | (nd & 0x00020002));
            if (H <= 8) tw0 |= code2 << (2 * r);
            else { tw0 |= (code2 & 3u) << (2 * r); tw1 |= (code2 >> 16) << (2 * r); }
            o1 = old[r];
            nl = cur[r];
        }
        if (C <= 16) {
            // codes of columns 0 .. H-1 sit at bits 2r, those of H .. C-1 at bits 16 + 2r: column order wants them at 2 (H + r)
            const unsigned word = H == 8 ? tw0 : ((tw0 & 0xffffu) | ((tw0 >> 16) << (2 * H)));
            tdir[(long long)(t + 1) * a.dir_words + lane] = word;
        } else {
            tdir[(long long)(t + 1) * a.dir_words + lane] = tw0;
            tdir[(long long)(t + 1) * a.dir_words + WAVE + lane] = tw1;
        }
        if (!rev && irow == start_row) {
            int pv = 0;
#pragma unroll
            for (int r = 0; r < H; ++r) if (r == (start_col % C) % H) pv = cur[r];
            if (lane == start_col / C) rs->trace_score = ((start_col % C) >= H ? hi16(pv) : lo16(pv)) + start_col * gcost;
        }
        // the walkers start at (fen | end row, column) / (rsn, column) and only ever move to earlier rows of the list: the
        // rows behind the start row are never read (a recombined read uses about half of each path)
        if (recomb && irow == (rev ? rs->rsn : rs->fen)) break;
    }
}

void launch_layer16(const LayerArgs& a, int nreads, int C, hipStream_t s) {
    const size_t bytes = (size_t)(64 + 5 * 64 + 5 * WAVE * (C / 2)) * sizeof(int);
    switch (C) {
        case 4: hipLaunchKernelGGL((k_layer16<4>), dim3(nreads), dim3(64), bytes, s, a); break;
        case 8: hipLaunchKernelGGL((k_layer16<8>), dim3(nreads), dim3(64), bytes, s, a); break;
        case 16: hipLaunchKernelGGL((k_layer16<16>), dim3(nreads), dim3(64), bytes, s, a); break;
        default: hipLaunchKernelGGL((k_layer16<32>), dim3(nreads), dim3(64), bytes, s, a); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_opt0 on packed rows (k_opt0 of rg_pathwise.hip is the i32 form: one LDS lookup, two compares and a GP subtraction per
// column, ~170 wave-instructions per row; this one ~55).  Same job: the plain NW score of the read against the rows of one
// path — path 0 (the provable bound), the picked path, or p1's rows up to X followed by p2's (two-path picks) — in the
// sweep's z-space with the sweep's score profile.  Only the final value leaves the kernel; no masks, no direction words.
// Batches the packed sweep admits (sweep16_admissible: every stored value fits; the driver sends the others to k_opt0).
template <int C>
__global__ __launch_bounds__(64) void k_opt0_16(Opt0Args a) {
    constexpr int H = C / 2;
    const int rd = blockIdx.x;
    const int lane = threadIdx.x;
    const PathGraphDev& g = a.g;
    const long long ro = a.read_off[rd];
    const int n = (int)(a.read_off[rd + 1] - ro);
    if (a.bad[rd] || n + 1 > C * WAVE) { if (lane == 0) a.lb[rd] = INT32_MIN / 2; return; }
    const uint8_t* read = a.reads + ro - 1;
    const int ncols = n + 1, GAP = 5;
    extern __shared__ __attribute__((aligned(16))) int lds16[];
    int* sct = lds16;                    // [36]
    int* s2 = lds16 + 64;                // [5][64] packed (s - g) pairs by (code_lo | code_hi << 3)
    int* sprof = s2 + 5 * 64;            // [5][64][H]
    if (lane < 36) sct[lane] = a.sc.t[lane];
    __syncthreads();
    const int gcost = __builtin_amdgcn_readfirstlane(sct[GAP]);
    for (int e = lane; e < 5 * 64; e += WAVE) {
        const int li = e >> 6, cl = e & 7, ch = (e >> 3) & 7;
        s2[e] = (cl < 6 && ch < 6) ? pack16(sct[li * 6 + cl] - gcost, sct[li * 6 + ch] - gcost) : 0;
    }
    __syncthreads();
    int row[H];
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const int c0 = lane * C + r, c1 = c0 + H;
        int k0 = 4, k1 = 4;
        if (c0 >= 1 && c0 < ncols) k0 = read[c0];
        if (c1 >= 1 && c1 < ncols) k1 = read[c1];
#pragma unroll
        for (int li = 0; li < 5; ++li) sprof[(li * WAVE + lane) * H + r] = s2[li * 64 + (k0 | (k1 << 3))];
        row[r] = pack16(c0 < ncols ? 0 : NEG16, c1 < ncols ? 0 : NEG16);       // the gap-only start row: z = 0
    }
    __syncthreads();
    const int GI = pack16(gcost, gcost);
    const int GI0 = lane == 0 ? pack16(a.semi ? 0 : gcost, gcost) : GI;        // border column 0 (semiglobal: stays 0)
    const int pk = a.pick ? a.pick[rd] : 0;
    const int pk2 = (a.pick && a.pick2) ? a.pick2[2 * rd] : -1;
    const int X = pk2 >= 0 ? a.pick2[2 * rd + 1] : INT32_MAX;
    int beg = a.fpoff[pk], cnt = a.fpoff[pk + 1] - beg;
    const int ln = n / C, ql = n % C;
    int semibest = NEG32;
    bool second = false;
    // TWO-STAGE look-ahead (round 6): the row index of row t + 2 and the base code of row t + 1 are in flight while row t is
    // computed.  The loop used to load `fprow[beg + t]` at the top of every iteration — for the two-path switch test — and waited
    // for it: one memory round trip per row, ~0.7 of the kernel's 0.86 ms per 4096 reads.
    int i_cur = cnt > 0 ? a.fprow[beg] : 0, i_nx = cnt > 1 ? a.fprow[beg + 1] : 0;
    int li_cur = cnt > 0 ? g.lnz[i_cur] : 4;
    for (int t = 0; t < cnt; ++t) {
        const int i = i_cur;
        if (i > X && !second) {
            // switch lists: first row of p2 above X (its list is ascending)
            second = true;
            beg = a.fpoff[pk2]; cnt = a.fpoff[pk2 + 1] - beg;
            int lo = 0, hi = cnt;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.fprow[beg + mid] <= X) lo = mid + 1; else hi = mid; }
            t = lo - 1;
            i_cur = lo < cnt ? a.fprow[beg + lo] : 0;
            i_nx = lo + 1 < cnt ? a.fprow[beg + lo + 1] : 0;
            li_cur = lo < cnt ? g.lnz[i_cur] : 4;
            continue;
        }
        const int li = li_cur;
        // (issued now, consumed by the next iterations: the base code of row t + 1 — its index was loaded an iteration ago —
        // and the index of row t + 2)
        const int li_nx = t + 1 < cnt ? g.lnz[i_nx] : 4;
        const int i_nx2 = t + 2 < cnt ? a.fprow[beg + t + 2] : 0;
        i_cur = i_nx; i_nx = i_nx2; li_cur = li_nx;
        int s[H];
        {
            const int* sp = sprof + (li * WAVE + lane) * H;
#pragma unroll
            for (int r = 0; r < H; ++r) s[r] = sp[r];
        }
        int prev = __builtin_amdgcn_alignbit(row[H - 1], dpp_shr1(row[H - 1], NEGPAIR), 16);
        int run = NEGPAIR;
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const int old = row[r];
            const int du = pk_max(pk_add(prev, s[r]), pk_add(old, r == 0 ? GI0 : GI));     // (lane 0 column 0: d = -inf)
            run = pk_max(du, run);
            row[r] = du;
            prev = old;
        }
        const int TL = lo16(run), TH = hi16(run);
        const int ze = dpp_shr1(dpp_incl_max(max(TH, TL), INT32_MIN), INT32_MIN);
        const int bl = max(ze, NEG16);
        int vprev = pack16(bl, max(TL, bl));
#pragma unroll
        for (int r = 0; r < H; ++r) { vprev = pk_max(row[r], vprev); row[r] = vprev; }
        if (a.semi) {   // free end row: best last-column value over the rows of the path
            int pv = 0;
#pragma unroll
            for (int r = 0; r < H; ++r) if (r == ql % H) pv = row[r];
            semibest = max(semibest, (ql >= H ? hi16(pv) : lo16(pv)) + n * gcost);
        }
    }
    int pv = 0;
#pragma unroll
    for (int r = 0; r < H; ++r) if (r == ql % H) pv = row[r];
    const int v = (ql >= H ? hi16(pv) : lo16(pv)) + n * gcost;
    if (lane == ln) a.lb[rd] = (a.semi ? semibest : v) - (a.pick ? a.margin : 0) - (pk2 >= 0 ? a.rec_pen : 0);
}

void launch_opt0_16(const Opt0Args& a, int nreads, int C, hipStream_t s) {
    const size_t bytes = (size_t)(64 + 5 * 64 + 5 * WAVE * (C / 2)) * sizeof(int);
    switch (C) {
        case 4: hipLaunchKernelGGL((k_opt0_16<4>), dim3(nreads), dim3(64), bytes, s, a); break;
        case 8: hipLaunchKernelGGL((k_opt0_16<8>), dim3(nreads), dim3(64), bytes, s, a); break;
        case 16: hipLaunchKernelGGL((k_opt0_16<16>), dim3(nreads), dim3(64), bytes, s, a); break;
        default: hipLaunchKernelGGL((k_opt0_16<32>), dim3(nreads), dim3(64), bytes, s, a); break;
    }
}

template <int kColmax, bool kRec, bool kWide, bool kSemi>
static void launch_sweep16_s(const SweepArgs& a, int nreads, int C, hipStream_t s) {
    const size_t bytes = (size_t)(64 + 2 * (kWide ? RG_MAXP : 64) + std::max(C * WAVE, 5 * 64) + 4 * WAVE * (C / 2)) * sizeof(int) + (size_t)options().lds_pad;
    switch (C) {
        case 4: hipLaunchKernelGGL((k_sweep16<4, kColmax, kRec, kWide, kSemi>), dim3(nreads), dim3(64), bytes, s, a); break;
        case 8: hipLaunchKernelGGL((k_sweep16<8, kColmax, kRec, kWide, kSemi>), dim3(nreads), dim3(64), bytes, s, a); break;
        case 16: hipLaunchKernelGGL((k_sweep16<16, kColmax, kRec, kWide, kSemi>), dim3(nreads), dim3(64), bytes, s, a); break;
        default: hipLaunchKernelGGL((k_sweep16<32, kColmax, kRec, kWide, kSemi>), dim3(nreads), dim3(64), bytes, s, a); break;
    }
}
template <int kColmax, bool kRec, bool kWide>
static void launch_sweep16_w(const SweepArgs& a, int nreads, int C, hipStream_t s) {
    if (a.semi) launch_sweep16_s<kColmax, kRec, kWide, true>(a, nreads, C, s);
    else launch_sweep16_s<kColmax, kRec, kWide, false>(a, nreads, C, s);
}
template <int kColmax, bool kRec>
static void launch_sweep16_c(const SweepArgs& a, int nreads, int C, hipStream_t s) {
    if (a.g.P > 64) launch_sweep16_w<kColmax, kRec, true>(a, nreads, C, s);
    else launch_sweep16_w<kColmax, kRec, false>(a, nreads, C, s);
}
void launch_sweep16(const SweepArgs& a_, int nreads, int C, hipStream_t s) {
    SweepArgs a = a_;
    // split step tables (TAIL records): the record variants with lazy keys, register runs of 4 and gather runs compiled in
    constexpr bool split_built = RG_SWEEP16_KRUN == 4 && RG_SWEEP16_KRUN_REV == 4 && RG_SWEEP16_GATHER && RG_SWEEP16_GATHER_FWD;
    // (more than 64 paths: the wide-run table takes the split table's place)
    if (split_built && a.use_split && a.fsplit && a.rsplit && a.frec && !(a.colmax_out && a.colarg_out) && C <= 16) {
        a.fsteps = a.fsplit;
        a.rsteps = a.rsplit;
        a.flead = a.fslead;
        a.rlead = a.rslead;
    }
    if (!(a.rev ? a.rlead : a.flead)) a.retire = 0;
    a.table_members = a.rev ? a.rmembers : a.fmembers;
    // a sweep that writes records and is not asked for column maxima skips their tracking
    // (kColmax = 2 — packed maxima without their cells — was the forward sweep of the record pipeline until round 5; the
    // driver now reads both sweeps' maxima out of their records and nothing instantiates that form any more)
    if (a.frec && !a.colmax_out) launch_sweep16_c<0, true>(a, nreads, C, s);
    else if (a.frec) launch_sweep16_c<1, true>(a, nreads, C, s);
    // -m 4 / -m 5: no best-member tracking at all (the variant below carries the column-maxima / threshold registers it
    // would never use and spilled 57 of them)
    else if (!a.track_best && !a.colmax_out && !a.cand) launch_sweep16_c<0, false>(a, nreads, C, s);
    else launch_sweep16_c<1, false>(a, nreads, C, s);
}

}  // namespace rg
