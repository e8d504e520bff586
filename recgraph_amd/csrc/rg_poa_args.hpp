// Kernel argument blocks shared between the launchers (.hip) and the batch driver.
#pragma once
#include "rg_device.hpp"

namespace rg {

struct PoaArgs {
    DevLnz g;
    DevScores sc;
    const uint8_t* reads;      // base codes, all reads concatenated
    const long long* read_off; // nreads+1
    const uint8_t* bad;        // per read: 1 = contains a base outside ACGTN
    const int* bta;            // per read bases_to_add (main.rs:57)
    const int* col0;           // m0: m[i][0] per row (global_abpoa.rs:36-46), depends on scores only
    // m0 (k_m0_simd): what a row needs besides its predecessor list, in ONE 16-byte record per row — {end of the row's
    // predecessor list (pred_off[i + 1]), r_values[i], col0[i], (first listed predecessor + 1) | base code << 24} — so that
    // the row loop issues one scalar load a row ahead instead of five dependent ones (the kernel is bound by its scalar unit)
    const int4* rowmeta;
    // the same for k_poa_banded (-m 2, scalar -m 0): {pred_off[i + 1], r_values[i], min_pred[i] (0 for row 0), (first listed
    // predecessor + 1) | base code << 24}
    const int4* rowmeta_b;
    int nreads;
    int max_n;                 // longest read of the batch
    int lds_read;              // m0: the read's base codes are staged in LDS (max_n + 2 bytes per wave)
    int read_base;             // local modes: first read of this launch (arena slots are launch-relative)
    int gap_open, gap_ext;     // m2
    long long cap_cells;       // arena capacity per read (cells)
    int* arena_m;              // [nreads][cap_cells] (m2: 3 planes)
    uint32_t* arena_pw;        // [nreads][cap_cells] (m2: 3 planes)
    int4* rinfo;               // [nreads][L]  {arena offset, first stored column, right, best_scoring_pos}
    DevRecord* rec;            // [nreads]
    uint8_t* ops;              // [nreads][ops_stride]
    int32_t* oprows;           // [nreads][ops_stride]
    long long ops_stride;
    unsigned long long* cells; // DP cell-update counter
};

void launch_m0_simd(const PoaArgs& a, hipStream_t s);
void launch_m2(const PoaArgs& a, hipStream_t s);
void launch_m0_scalar(const PoaArgs& a, hipStream_t s);
void launch_local(const PoaArgs& a, int variant, hipStream_t s);   // 0: -m 1 AVX2 semantics, 1: -m 1 scalar, 2: -m 3

}  // namespace rg
