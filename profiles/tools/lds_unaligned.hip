// Dev tool: cost of 2-byte-aligned ds_read_b32 (two ring entries per read, half of the lanes off their natural alignment) against two ds_read_u16,
// in the access pattern of the fill kernel's generic rows (lane = paired cell: ascending start columns with gaps).
//   hipcc -O3 --offload-arch=gfx950 profiles/tools/lds_unaligned.hip -o profiles/tools/bin/lds_unaligned && profiles/tools/bin/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) const unsigned short* lds_u16;
typedef __attribute__((address_space(3))) const unsigned* lds_u32;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned* out, int iters) {
    __shared__ unsigned short ring[32 * 354 + 64];
    for (int x = threadIdx.x; x < 32 * 354 + 64; x += 1024) ring[x] = (unsigned short)(x * 2654435761u >> 17);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int col = (lane * 173) / 64 + ((lane * 7) & 1);          // ~37 % density, both parities
    unsigned w = 0xffffffffu, w2 = 0xffffffffu;
    for (int it = 0; it < iters; it++) {
        const int row = (it * 5) & 31;
        lds_u16 rp = (lds_u16)(ring + row * 354 + col);
        if (MODE == 0) {
#pragma unroll
            for (int n = 0; n < 22; n += 2) { const unsigned a = rp[n], b = rp[n + 1]; w = a < w ? a : w; w = b < w ? b : w; }
        } else {
            unsigned v[11];
#pragma unroll
            for (int n = 0; n < 11; n++) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[n]) : "v"((unsigned)(size_t)rp), "n"(0) : "memory"); rp += 2; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 11; n++) { u16x2 x = __builtin_bit_cast(u16x2, v[n]), y = __builtin_bit_cast(u16x2, w2); y = __builtin_elementwise_min(x, y); w2 = __builtin_bit_cast(unsigned, y); }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = w ^ w2;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, d, 20000); else hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, d, 20000);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s: %.3f ms for 20000 x 22 entries per lane\n", mode ? "11 x ds_read_b32 (2-byte aligned) + v_pk_min_u16" : "22 x ds_read_u16 + v_min_u32", ms);
        }
    return 0;
}
