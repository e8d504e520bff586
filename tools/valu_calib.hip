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

enum Kind { PK_ADD, PK_MAX, PK_SUB, PK_ASHR, BITOP3, AND_OR, LSHL_OR, MAX_I32, ADD_U32, DPP_MOV, MIX, NKIND };
static const char* kind_name[NKIND] = {"v_pk_add_u16", "v_pk_max_i16", "v_pk_sub_i16", "v_pk_ashrrev_i16", "v_bitop3_b32",
                                       "v_and_or_b32", "v_lshl_or_b32", "v_max_i32", "v_add_u32", "v_mov_b32_dpp(row_shr:1)",
                                       "sweep16 mix (pk_add, pk_max, pk_sub, pk_ashr, bitop3 x2, and_or, max_i32)"};

template <int K>
__device__ __forceinline__ void op(int& r, int a, int b) {
    if (K == PK_ADD) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(a));
    else if (K == PK_MAX) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(a));
    else if (K == PK_SUB) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(r) : "v"(a));
    else if (K == PK_ASHR) asm volatile("v_pk_ashrrev_i16 %0, 1, %0" : "+v"(r));
    else if (K == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca" : "+v"(r) : "v"(a), "v"(b));
    else if (K == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "s"(0xffff0000), "v"(b));
    else if (K == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(r) : "v"(b));
    else if (K == MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(a));
    else if (K == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(a));
    else if (K == DPP_MOV) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(a));
}

template <int K>
__global__ __launch_bounds__(256) void k_valu(int iters, int a, int b, int* out) {
    int r[NREG];
#pragma unroll
    for (int k = 0; k < NREG; ++k) r[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
                if (K == MIX) {
                    switch ((u * NREG + k) & 7) {
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
    const int blocks = ncu * waves_per_simd;   // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_valu<K>), dim3(blocks), dim3(256), 0, 0, iters / 8, 3, 5, d_out);   // warm-up
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_valu<K>), dim3(blocks), dim3(256), 0, 0, iters, 3, 5, d_out);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * 4.0 * (double)iters * NREG * UNROLL;
    return winstr / (ms * 1e-3);
}

template <int K>
static void valu_kind(int ncu, int iters, int* d_out, std::string& json) {
    char buf[512];
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
        const int iters = argc > 2 ? atoi(argv[2]) : 20000;
        std::string json = "{";
        valu_kind<PK_ADD>(ncu, iters, d_out, json);
        valu_kind<PK_MAX>(ncu, iters, d_out, json);
        valu_kind<PK_SUB>(ncu, iters, d_out, json);
        valu_kind<PK_ASHR>(ncu, iters, d_out, json);
        valu_kind<BITOP3>(ncu, iters, d_out, json);
        valu_kind<AND_OR>(ncu, iters, d_out, json);
        valu_kind<LSHL_OR>(ncu, iters, d_out, json);
        valu_kind<MAX_I32>(ncu, iters, d_out, json);
        valu_kind<ADD_U32>(ncu, iters, d_out, json);
        valu_kind<DPP_MOV>(ncu, iters, d_out, json);
        valu_kind<MIX>(ncu, iters, d_out, json);
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
