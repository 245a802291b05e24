// Microbenchmark: LDS read cost on gfx950 for the access shapes of the fold kernel's interior-loop phase.
// 1024 workgroups of 1024 threads (16 waves, one workgroup per CU at a time); every lane reads `iters` x 8 chunks at a lane-dependent address.
// build: hipcc -O3 --offload-arch=gfx950 lds_read.hip -o lds_read ; run: ./lds_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned* out, long long* cyc, int iters, int stride_x10, int mis) {
    extern __shared__ __align__(16) unsigned short sm[];
    for (int x = threadIdx.x; x < 60000; x += 1024) sm[x] = (unsigned short)(x * 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane start (in shorts): cells ascending with an average gap of stride_x10/10 shorts, plus `mis` shorts of misalignment
    int start = (lane * stride_x10) / 10;
    if (MODE >= 2) start = (start & ~7);            // 16-byte aligned classes
    start += mis + wave * 708;
    unsigned acc = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        const unsigned short* p = sm + start + (it & 15) * 354;
        if (MODE == 0) {            // 8 x ds_read_u16
#pragma unroll
            for (int c = 0; c < 64; c++) acc += p[c];
        } else if (MODE == 1 || MODE == 2) {   // 8 x 16-byte chunk (alignment from `mis`)
#pragma unroll
            for (int c = 0; c < 8; c++) { us8 v; __builtin_memcpy(&v, p + 8 * c, 16); acc += v[0] + v[3] + v[7]; }
        } else if (MODE == 3) {     // 16 x 8-byte chunk
#pragma unroll
            for (int c = 0; c < 16; c++) { us4 v; __builtin_memcpy(&v, p + 4 * c, 8); acc += v[0] + v[3]; }
        } else {                    // 32 x 4-byte chunk
#pragma unroll
            for (int c = 0; c < 32; c++) { us2 v; __builtin_memcpy(&v, p + 2 * c, 4); acc += v[0] + v[1]; }
        }
    }
    long long t1 = clock64();
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int stride_x10, int mis) {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(1024), 128 * 1024, 0, out, cyc, iters, stride_x10, mis);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(1024), 128 * 1024, 0, out, cyc, iters, stride_x10, mis);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // 16 waves x iters x 64 shorts x 64 lanes x 2 B
    double bytes = 16.0 * iters * 64 * 64 * 2 * 4;   // per CU: 1024 workgroups over 256 CUs
    printf("%-28s stride %.1f mis %d: %.3f ms, %.1f B/clk/CU (at 2.4 GHz), s_memtime ticks %lld\n", name, stride_x10 / 10.0, mis, ms, bytes / (ms * 1e-3 * 2.4e9), c);
}

int main() {
    for (int st : {10, 27, 80}) {
        run<0>("u16 x64", st, 0);
        run<1>("b128 x8 (lane-misaligned)", st, 1);
        run<1>("b128 x8 (mis 0, stride)", st, 0);
        run<2>("b128 x8 aligned classes", st, 0);
        run<2>("b128 x8 aligned+1 short", st, 1);
        run<2>("b128 x8 aligned+2 short", st, 2);
        run<2>("b128 x8 aligned+4 short", st, 4);
        run<3>("b64 x16 (mis 0, stride)", st, 0);
        run<4>("b32 x32 (mis 0, stride)", st, 0);
    }
    return 0;
}
