// Latency probe for the dependent instruction chains k_sweep16's row operators are made of (gfx950).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_dep tools/probes/valu_dep.hip && /tmp/valu_dep
// For each instruction form: NCH independent chains per wave (1 = every instruction reads the previous one's result), W waves
// per SIMD; prints shader cycles per instruction as seen by ONE wave (elapsed cycles / instructions it issued) and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(x) x x x x x x x x
template <int KIND, int NCH>
__global__ __launch_bounds__(64) void k(int* out, int iters, int seed) {
    int r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed + i * 77 + threadIdx.x;
    int a = seed ^ 0x1234, b = seed * 3 + 1;
    for (int it = 0; it < iters; ++it) {
#define STEP(i) \
        if (KIND == 0) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[(i) % NCH]) : "v"(a)); \
        else if (KIND == 1) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[(i) % NCH]) : "v"(a)); \
        else if (KIND == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca" : "+v"(r[(i) % NCH]) : "v"(a), "v"(b)); \
        else if (KIND == 3) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[(i) % NCH]) : "v"(a)); \
        else if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[(i) % NCH]) : "v"(a)); \
        else if (KIND == 5) asm volatile("v_pk_sub_i16 %0, %0, %1\n v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "+v"(r[(i) % NCH]) : "v"(a)); \
        else if (KIND == 6) asm volatile("s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[(i) % NCH])); \
        else if (KIND == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[(i) % NCH]) : "v"(a), "v"(b));
        REP8(STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7))
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s ^= r[i];
    if (s == 0x7fffffff) out[0] = s;
}

template <int KIND, int NCH>
double run(int waves_per_simd, int* d, int cus, double clk_hz) {
    const int iters = 2000;
    const int blocks = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(blocks), dim3(64), 0, 0, d, 10, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(blocks), dim3(64), 0, 0, d, iters, 1);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double per_wave = (double)iters * 64 * (KIND == 5 || KIND == 6 ? 2 : 1);
    return ms * 1e-3 * clk_hz / per_wave;      // cycles per instruction as one wave sees it
}

template <int KIND>
void kind(const char* name, int* d, int cus, double clk) {
    printf("%-34s", name);
    for (int w = 1; w <= 2; ++w) {
        printf("  W%d: nch1 %5.2f nch2 %5.2f nch4 %5.2f nch8 %5.2f |", w, run<KIND, 1>(w, d, cus, clk), run<KIND, 2>(w, d, cus, clk), run<KIND, 4>(w, d, cus, clk), run<KIND, 8>(w, d, cus, clk));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const double clk = p.clockRate * 1e3;
    int* d;
    CHK(hipMalloc(&d, 64));
    printf("CUs %d clock %.0f MHz; cycles per instruction seen by one wave (s_nop / second instruction of a pair counted as instructions where noted)\n", p.multiProcessorCount, clk / 1e6);
    kind<0>("v_pk_max_i16", d, p.multiProcessorCount, clk);
    kind<1>("v_pk_add_u16", d, p.multiProcessorCount, clk);
    kind<2>("v_bitop3_b32", d, p.multiProcessorCount, clk);
    kind<3>("v_max_i32", d, p.multiProcessorCount, clk);
    kind<4>("v_add_u32", d, p.multiProcessorCount, clk);
    kind<5>("pk_sub + pk_ashr (pair, /2)", d, p.multiProcessorCount, clk);
    kind<6>("s_nop 1 + v_max_i32_dpp (pair, /2)", d, p.multiProcessorCount, clk);
    kind<7>("v_perm_b32", d, p.multiProcessorCount, clk);
    return 0;
}
