// Calibration micro-kernels for bench.py's roofline object (gfx950 / MI355X).  Not part of the product library.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_calib tools/valu_calib.hip
//   ./valu_calib valu            issue rate of the integer VALU instructions k_sweep16 is made of: streams of
//                                independent v_pk_add_u16 / v_pk_max_i16 / v_pk_sub_i16 / v_pk_ashrrev_i16 /
//                                v_bitop3_b32 / v_and_or_b32 / v_lshl_or_b32 / v_max_i32 / v_mov_b32 DPP at 1, 2 and 4
//                                waves per SIMD; prints wave-instructions per second for the whole chip (JSON)
//   ./valu_calib mem             coalesced 4 B/lane and 16 B/lane reads and writes of a known number of bytes (1 GiB,
//                                past the 256 MiB Infinity Cache): run under `rocprofv3 --pmc FETCH_SIZE` and
//                                `--pmc WRITE_SIZE` (separate passes) to calibrate the counters' byte unit on the access
//                                widths the sweep uses; prints the bytes each kernel really moved (JSON)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHK(x)                                                                      \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int NREG = 16;     // independent chains per wave (latency of one VALU op is hidden 16 deep)
constexpr int UNROLL = 8;    // NREG * UNROLL instructions per loop iteration

// one entry per instruction form: asm text with %0 = the chain register (read-modify-write), %1 / %2 = two other VGPRs
#define KINDS(X)                                                                     \
    X(PK_ADD, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1")                            \
    X(PK_MAX, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1")                            \
    X(PK_MAXU, "v_pk_max_u16", "v_pk_max_u16 %0, %0, %1")                           \
    X(PK_MIN, "v_pk_min_i16", "v_pk_min_i16 %0, %0, %1")                            \
    X(PK_SUB, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1")                            \
    X(PK_ASHR, "v_pk_ashrrev_i16", "v_pk_ashrrev_i16 %0, 1, %0")                    \
    X(PK_LSHL, "v_pk_lshlrev_b16", "v_pk_lshlrev_b16 %0, 1, %0")                    \
    X(PK_MAD, "v_pk_mad_i16", "v_pk_mad_i16 %0, %0, %1, %2")                        \
    X(BITOP3, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca")            \
    X(BFI, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2")                                 \
    X(AND_OR, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2")                        \
    X(LSHL_OR, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 16, %1")                     \
    X(LSHL_ADD, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1")                   \
    X(ADD3, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2")                              \
    X(MAX3, "v_max3_i32", "v_max3_i32 %0, %0, %1, %2")                              \
    X(MAX_I32, "v_max_i32", "v_max_i32 %0, %0, %1")                                 \
    X(MAX_U32, "v_max_u32", "v_max_u32 %0, %0, %1")                                 \
    X(MIN_I32, "v_min_i32", "v_min_i32 %0, %0, %1")                                 \
    X(MAX_I16, "v_max_i16", "v_max_i16 %0, %0, %1")                                 \
    X(ADD_U32, "v_add_u32", "v_add_u32 %0, %0, %1")                                 \
    X(SUB_U32, "v_sub_u32", "v_sub_u32 %0, %0, %1")                                 \
    X(ADD_U16, "v_add_u16", "v_add_u16 %0, %0, %1")                                 \
    X(AND_B32, "v_and_b32", "v_and_b32 %0, %0, %1")                                 \
    X(OR_B32, "v_or_b32", "v_or_b32 %0, %0, %1")                                    \
    X(XOR_B32, "v_xor_b32", "v_xor_b32 %0, %0, %1")                                 \
    X(OR3, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2")                                 \
    X(LSHLREV, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0")                          \
    X(ASHRREV, "v_ashrrev_i32", "v_ashrrev_i32 %0, 1, %0")                          \
    X(BFE_I32, "v_bfe_i32", "v_bfe_i32 %0, %0, 3, 5")                               \
    X(PERM, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2")                              \
    X(ALIGNBIT, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 16")                 \
    X(MOV, "v_mov_b32", "v_mov_b32 %0, %1")                                         \
    X(CNDMASK, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc")                    \
    X(CMP_CND, "v_cmp_gt_i32 + v_cndmask_b32 (counted as 2)", "v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc") \
    X(MAD_U24, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2")                     \
    X(MUL_LO, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                            \
    X(ADD_F32, "v_add_f32", "v_add_f32 %0, %0, %1")                                 \
    X(MAX_F32, "v_max_f32", "v_max_f32 %0, %0, %1")                                 \
    X(FMA_F32, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2")                             \
    X(PK_ADD_F16, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %1")                        \
    X(PK_MAX_F16, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %1")                        \
    X(DPP_MOV, "v_mov_b32_dpp(row_shr:1)", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(DPP_MAX, "v_max_i32_dpp(row_shr:1)", "v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(READLANE, "v_readlane_b32 (to SGPR) + v_add_u32 with it (counted as 2)", "v_readlane_b32 s20, %1, 3\n v_add_u32 %0, s20, %0")

enum Kind {
#define X(id, name, text) id,
    KINDS(X)
#undef X
    MIX, MIX_ADD, MIX_BLOCK, NKIND
};
static const char* kind_name[NKIND] = {
#define X(id, name, text) name,
    KINDS(X)
#undef X
    "sweep16 mix (pk_add, pk_max, pk_sub, pk_ashr, bitop3 x2, and_or, max_i32)",
    "alternating v_pk_max_i16 / v_add_u32",
    "blocks of 64 v_add_u32 then 64 v_pk_max_i16"};
static const int kind_count[NKIND] = {
#define X(id, name, text) (id == CMP_CND || id == READLANE) ? 2 : 1,
    KINDS(X)
#undef X
    1, 1, 1};

template <int K>
__device__ __forceinline__ void op(int& r, int a, int b) {
#define X(id, name, text) \
    if (K == id) { if (id == CNDMASK || id == CMP_CND || id == READLANE) asm volatile(text : "+v"(r) : "v"(a), "v"(b) : "vcc", "s20"); \
                   else asm volatile(text : "+v"(r) : "v"(a), "v"(b)); }
    KINDS(X)
#undef X
}

template <int K>
__global__ __launch_bounds__(1024) void k_valu(int iters, int a, int b, int* out) {
    extern __shared__ int lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = a;   // never: keeps the LDS allocation
    int r[NREG];
#pragma unroll
    for (int k = 0; k < NREG; ++k) r[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
                if (K == MIX || K == MIX_ADD || K == MIX_BLOCK) {
                    if (K == MIX_BLOCK) {
                        if (u < UNROLL / 2) op<ADD_U32>(r[k], a, b); else op<PK_MAX>(r[k], a, b);
                    } else if (K == MIX_ADD) {
                        if ((u * NREG + k) & 1) op<ADD_U32>(r[k], a, b); else op<PK_MAX>(r[k], a, b);
                    } else switch ((u * NREG + k) & 7) {
                        case 0: op<PK_ADD>(r[k], a, b); break;
                        case 1: op<PK_MAX>(r[k], a, b); break;
                        case 2: op<PK_SUB>(r[k], a, b); break;
                        case 3: op<PK_ASHR>(r[k], a, b); break;
                        case 4: op<BITOP3>(r[k], a, b); break;
                        case 5: op<BITOP3>(r[k], a, b); break;
                        case 6: op<AND_OR>(r[k], a, b); break;
                        default: op<MAX_I32>(r[k], a, b); break;
                    }
                } else {
                    op<K>(r[k], a, b);
                }
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < NREG; ++k) s ^= r[k];
    if (s == 0x7fffffff) out[0] = s;   // never true in practice: keeps the chains alive
}

template <int K>
static double run_valu(int waves_per_simd, int ncu, int iters, int* d_out) {
    // one workgroup per CU (96 KB of LDS each: two cannot share a CU) of 4 * w waves = w waves on each of the 4 SIMDs
    const int blocks = ncu;
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    (void)hipFuncSetAttribute((const void*)k_valu<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipLaunchKernelGGL((k_valu<K>), dim3(blocks), dim3(threads), 96 * 1024, 0, iters / 8, 3, 5, d_out);   // warm-up
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_valu<K>), dim3(blocks), dim3(threads), 96 * 1024, 0, iters, 3, 5, d_out);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * 4.0 * waves_per_simd * (double)iters * NREG * UNROLL * kind_count[K];
    return winstr / (ms * 1e-3);
}

template <int K>
static void valu_kind(int ncu, int iters, int* d_out, std::string& json) {
    char buf[1024];
    double v[3];
    const int w[3] = {1, 2, 4};
    for (int i = 0; i < 3; ++i) v[i] = run_valu<K>(w[i], ncu, iters, d_out);
    snprintf(buf, sizeof buf, "%s\"%s\": {\"w1\": %.4e, \"w2\": %.4e, \"w4\": %.4e}", json.size() > 1 ? ", " : "", kind_name[K], v[0], v[1], v[2]);
    json += buf;
    fprintf(stderr, "%-70s  1 wave/SIMD %.3e  2: %.3e  4: %.3e wave-instr/s\n", kind_name[K], v[0], v[1], v[2]);
}

// ---- memory-counter calibration kernels (distinct names so that the PMC CSV separates them) ----
__global__ void calib_read_4B(const int* __restrict__ p, size_t n, int* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int s = 0;
    for (; i < n; i += stride) s ^= p[i];
    if (s == 0x7fffffff) out[0] = s;
}
__global__ void calib_read_16B(const int4* __restrict__ p, size_t n, int* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int s = 0;
    for (; i < n; i += stride) { const int4 v = p[i]; s ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x7fffffff) out[0] = s;
}
__global__ void calib_write_4B(int* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (int)i;
}
__global__ void calib_write_16B(int4* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = make_int4((int)i, 1, 2, 3);
}
// the sweep's own pattern: every wave re-reads and re-writes ITS OWN 64 KB block (rows of one read) `passes` times
// with 4 B/lane coalesced accesses; total footprint = blocks * 64 KB (pick it past the Infinity Cache)
__global__ __launch_bounds__(64) void calib_rmw_rows_4B(int* p, int words_per_block, int passes) {
    int* b = p + (size_t)blockIdx.x * words_per_block;
    for (int it = 0; it < passes; ++it)
        for (int i = threadIdx.x; i < words_per_block; i += 64) b[i] = b[i] + 1;
}

int main(int argc, char** argv) {
    const std::string what = argc > 1 ? argv[1] : "valu";
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    int* d_out;
    CHK(hipMalloc((void**)&d_out, 64));
    if (what == "valu") {
        const int iters = argc > 2 ? atoi(argv[2]) : 8000;
        std::string json = "{";
#define X(id, name, text) valu_kind<id>(ncu, iters, d_out, json);
        KINDS(X)
#undef X
        valu_kind<MIX>(ncu, iters, d_out, json);
        valu_kind<MIX_ADD>(ncu, iters, d_out, json);
        valu_kind<MIX_BLOCK>(ncu, iters, d_out, json);
        char buf[256];
        snprintf(buf, sizeof buf, ", \"compute_units\": %d, \"clock_mhz\": %d, \"unit\": \"wave64 instructions per second, whole chip\"}", ncu,
                 prop.clockRate / 1000);
        json += buf;
        printf("%s\n", json.c_str());
    } else {
        const size_t bytes = (size_t)1 << 30;
        int* buf;
        CHK(hipMalloc((void**)&buf, bytes));
        CHK(hipMemset(buf, 1, bytes));
        CHK(hipDeviceSynchronize());
        const int blocks = ncu * 8;
        hipLaunchKernelGGL(calib_read_4B, dim3(blocks), dim3(256), 0, 0, buf, bytes / 4, d_out);
        hipLaunchKernelGGL(calib_read_16B, dim3(blocks), dim3(256), 0, 0, (const int4*)buf, bytes / 16, d_out);
        hipLaunchKernelGGL(calib_write_4B, dim3(blocks), dim3(256), 0, 0, buf, bytes / 4);
        hipLaunchKernelGGL(calib_write_16B, dim3(blocks), dim3(256), 0, 0, (int4*)buf, bytes / 16);
        const int wpb = 16384, passes = 8;            // 64 KB per wave, 16384 waves = 1 GiB footprint
        hipLaunchKernelGGL(calib_rmw_rows_4B, dim3((unsigned)(bytes / 4 / wpb)), dim3(64), 0, 0, buf, wpb, passes);
        CHK(hipDeviceSynchronize());
        printf("{\"calib_read_4B\": {\"read\": %zu, \"written\": 0}, \"calib_read_16B\": {\"read\": %zu, \"written\": 0}, "
               "\"calib_write_4B\": {\"read\": 0, \"written\": %zu}, \"calib_write_16B\": {\"read\": 0, \"written\": %zu}, "
               "\"calib_rmw_rows_4B\": {\"read\": %zu, \"written\": %zu}}\n",
               bytes, bytes, bytes, bytes, bytes * passes, bytes * passes);
    }
    return 0;
}
