// Dev tool: loaded latency of dependent global-memory round trips at the epilogue kernel's geometry (8 x 256-thread workgroups per CU, every
// wave one round trip in flight), as a function of the lines a round trip touches and of the footprint.  hipcc -O3 --offload-arch=gfx950 memlat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256, 8) chase(const unsigned* __restrict__ buf, size_t n_lines, int iters, int L, int per_lane, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    unsigned state = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        // wave-uniform pseudo-random base line; lane l < L reads line base + l * 977 (distinct lines), others read the base line
        state = state * 1664525u + 1013904223u;
        size_t line = ((size_t)state * 2654435761ull >> 7) & (n_lines - 1);   // n_lines: power of two
        unsigned v = 0;
        for (int p = 0; p < per_lane; p++) {
            size_t ln = (line + (size_t)(lane < L ? lane : 0) * 977 + (size_t)p * 31337) & (n_lines - 1);
            v ^= buf[ln * 32 + (lane & 31)];
        }
        // make the next address depend on the data
        v = __builtin_amdgcn_readfirstlane(v);
        state ^= v;
        acc += v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const size_t GB = 1ull << 30;
    unsigned* buf; unsigned* sink;
    size_t bytes = 8 * GB;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
    hipMemset(buf, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 300;
    for (int wgs : {2048, 256}) for (size_t fp : {(size_t)8 * GB, (size_t)128 << 20, (size_t)2 << 20}) {
        for (int per_lane : {1}) for (int L : {1, 8, 16}) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                chase<<<wgs, 256>>>(buf, fp / 128, iters, L, per_lane, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep) printf("wgs %4d footprint %6zu MB  lines/round-trip %3d x %d loads  %.2f us per round trip, %.1f G lines/s\n", wgs, fp >> 20, L, per_lane, ms * 1e3 / iters,
                                (double)wgs * 4 * iters * (L * per_lane) / (ms * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
