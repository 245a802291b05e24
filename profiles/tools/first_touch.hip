// Dev tool: what a fresh 5 GB device allocation costs its first user (the fold slabs of config[1] are 5.2 GB: the CLI's first fold in a process ran
// 0.27 - 0.49 s against 0.066 s for the second).   hipcc -O3 --offload-arch=gfx950 profiles/tools/first_touch.hip -o profiles/tools/bin/first_touch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(unsigned* p, size_t n, unsigned v) { for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) p[k] = v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(0);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        const size_t bytes = 5200ull << 20;
        double t0 = now();
        unsigned* p = nullptr; hipMalloc((void**)&p, bytes);
        double t1 = now();
        for (int k = 0; k < 3; k++) {
            hipEventRecord(a, st);
            hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, p, bytes / 4, (unsigned)k);
            hipEventRecord(b, st); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("alloc %d: hipMalloc %.1f ms; write pass %d over 5.2 GB: %.2f ms\n", rep, 1e3 * (t1 - t0), k, ms);
        }
        double t2 = now(); hipFree(p); printf("  hipFree %.1f ms\n", 1e3 * (now() - t2));
    }
    return 0;
}
