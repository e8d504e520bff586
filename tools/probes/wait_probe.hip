#include <hip/hip_runtime.h>
#include <sys/resource.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6; }
__global__ void spin(long long cycles, int* out) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) out[0] = 1;
}
int main(int argc, char** argv) {
    (void)hipFree(nullptr);
    { void* p; (void)hipMalloc(&p, 1 << 20); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 1000, (int*)p); (void)hipDeviceSynchronize(); }
    if (argc > 1) { hipError_t e = hipSetDeviceFlags(hipDeviceScheduleBlockingSync); printf("hipSetDeviceFlags(BlockingSync) AFTER the context is active -> %d (%s)\n", (int)e, hipGetErrorString(e)); (void)hipGetLastError(); }
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreateWithFlags(&e1, hipEventBlockingSync | hipEventDisableTiming);
    const long long cyc = 100000000ll * 5 / 10;   // wall_clock64 ticks at 100 MHz: 0.5 s
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int*)nullptr);
        double t0 = now(), c0 = cpu();
        if (mode == 0) (void)hipStreamSynchronize(s);
        if (mode == 1) { (void)hipEventRecord(e0, s); (void)hipEventSynchronize(e0); }
        if (mode == 2) { (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1); }
        printf("%s: wall %.3f s, cpu %.3f s\n", mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "hipEventSynchronize(default event)" : "hipEventSynchronize(blocking event)", now() - t0, cpu() - c0);
    }
    return 0;
}
