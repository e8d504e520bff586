#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int* p) { p[0] = 1; }
int main() {
    double t0 = now();
    (void)hipFree(nullptr);
    printf("init %.3f s\n", now() - t0);
    std::vector<void*> v;
    for (size_t gb : {20, 20, 20, 12, 12, 5, 5, 33, 40, 48}) {
        void* p = nullptr;
        t0 = now();
        hipError_t e = hipMalloc(&p, gb << 30);
        double t1 = now();
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, (int*)p);
        (void)hipDeviceSynchronize();
        double t2 = now();
        printf("%zu GB fresh: malloc %.1f ms (%d), first kernel %.1f ms\n", gb, (t1 - t0) * 1e3, (int)e, (t2 - t1) * 1e3);
        v.push_back(p);
    }
    t0 = now();
    for (void* p : v) (void)hipFree(p);
    printf("free all: %.1f ms\n", (now() - t0) * 1e3);
    v.clear();
    // three threads, 20 buffers of 3 GB each, concurrently
    t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < 3; ++t) th.emplace_back([&] { (void)hipSetDevice(0); for (int i = 0; i < 20; ++i) { void* p; (void)hipMalloc(&p, (size_t)3 << 30); } });
    for (auto& t : th) t.join();
    printf("3 threads x 20 x 3 GB: %.1f ms\n", (now() - t0) * 1e3);
    return 0;
}
