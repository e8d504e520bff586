// Device-side shared definitions (HIP, gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rg_codes.hpp"

namespace rg {

constexpr int WAVE = 64;

// scoring table by value in kernel arguments: t[a*6+b], alphabet "ACGTN-" -> 0..5
struct DevScores {
    int t[36];
};

// flattened LnzGraph in HBM (graph.rs:23-27 restated as CSR)
struct DevLnz {
    int L;
    const uint8_t* lnz;       // base codes 0..4 per row (rows 0 and L-1 unused)
    const int* pred_off;      // L+1
    const int* pred_rows;
    const int* r_values;      // L
    const int* min_pred;      // L
};

// per-read header written by the kernels, read back by the host (fixed stride)
struct DevRecord {
    uint32_t status;
    int32_t score;
    float fscore;
    int32_t end_row, end_col;
    int32_t stop_row, stop_col;
    int32_t best_path, rev_path;
    int32_t fen, rsn, rec_col, displacement;
    int32_t n_ops, n_fwd_ops;
    int32_t pad;
};


__device__ __forceinline__ int wave_incl_sum(int v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int o = __shfl_up(v, d, WAVE);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ int wave_incl_max(int v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int o = __shfl_up(v, d, WAVE);
        if (lane >= d) v = max(v, o);
    }
    return v;
}
// ---- DPP wave-level primitives (gfx9 family: row_shr, row_bcast15/31, wave_shr) ----
// A ds_bpermute-based __shfl scan costs six dependent LDS-crossbar round trips; the DPP forms below are six
// dependent VALU instructions.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_mov(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, kCtrl, kRowMask, 0xf, false);
}
// value of lane-1 (lane 0 gets `identity`).  The empty asm pins the move where the caller wrote it: as an operand of a
// `lane == 0 ? a : dpp_shr1(x)` select the compiler sank it into the `lane != 0` region, where lane 1 reads a DISABLED
// lane 0 and silently keeps the identity (found in k_m0_simd, profiles/r03_notes.md).
__device__ __forceinline__ int dpp_shr1(int v, int identity) {
    int r = dpp_mov<0x138, 0xf>(identity, v);
    asm volatile("" : "+v"(r));
    return r;
}

// (a lane without a source keeps its own value whatever the fill is; INT32_MIN is the identity the compiler's DPP combiner
// knows for a signed max, so each step becomes ONE v_max_i32_dpp instead of constant + v_mov_dpp + v_max)
__device__ __forceinline__ int dpp_incl_max(int v, int /*identity*/) {
    v = max(v, dpp_mov<0x111, 0xf>(INT32_MIN, v));   // row_shr:1
    v = max(v, dpp_mov<0x112, 0xf>(INT32_MIN, v));   // row_shr:2
    v = max(v, dpp_mov<0x114, 0xf>(INT32_MIN, v));   // row_shr:4
    v = max(v, dpp_mov<0x118, 0xf>(INT32_MIN, v));   // row_shr:8
    v = max(v, dpp_mov<0x142, 0xa>(INT32_MIN, v));   // row_bcast:15 into rows 1 and 3
    v = max(v, dpp_mov<0x143, 0xc>(INT32_MIN, v));   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int dpp_incl_sum(int v) {
    v += dpp_mov<0x111, 0xf>(0, v);
    v += dpp_mov<0x112, 0xf>(0, v);
    v += dpp_mov<0x114, 0xf>(0, v);
    v += dpp_mov<0x118, 0xf>(0, v);
    v += dpp_mov<0x142, 0xa>(0, v);
    v += dpp_mov<0x143, 0xc>(0, v);
    return v;
}

// Wave-uniform load of read-only data through the scalar cache (s_load_dword): on gfx9 vector loads and stores share
// one in-order counter (vmcnt), so a uniform table lookup issued as a vector load makes the wave wait for all of its
// earlier stores.  Only for memory no kernel in flight writes (graph tables, score-derived tables).
__device__ __forceinline__ int uload(const int* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ int4 uload4(const int4* p) {      // (one s_load_dwordx4; p 16-byte aligned)
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = *reinterpret_cast<const __attribute__((address_space(4))) v4i*>(reinterpret_cast<uintptr_t>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int uload_u8(const uint8_t* base, int i) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base) + (uintptr_t)i;
    const int w = *reinterpret_cast<const __attribute__((address_space(4))) int*>(a & ~(uintptr_t)3);
    return (w >> (8 * (int)(a & 3))) & 0xff;
}

__device__ __forceinline__ long long wave_max_ll(long long v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) {
        long long o = __shfl_xor(v, d, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

}  // namespace rg
