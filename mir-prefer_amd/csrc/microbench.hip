// On-device micro-benchmarks that measure the two roofs of the local-fold fill kernel on the GPU the library runs on
// (the fold is integer min-plus work out of LDS, bounded by LDS reads and by integer VALU issue -- not by HBM, not by MFMA):
//   * LDS: conflict-free ds_read_b32 / ds_read_u16 wave-instructions per second with the fill kernel's geometry (1024 threads per workgroup,
//     one workgroup per CU, reads kept in flight);
//   * VALU: packed 16-bit add + min (the split loop's relaxation) and 32-bit shift-add + min (the interior-loop relaxation) per second.
// bench.py turns them into "relaxations per second" roofs (2 LDS reads or 3 integer lane-operations per relaxation).
#include <hip/hip_runtime.h>
#include "mirp_ctx.h"

namespace mirp {

#define MB_NT 1024
#define MB_UNR 16

// kind 0: ds_read_b32, 1: ds_read_u16.  Every lane reads its own bank (lane * 4 bytes), UNR reads in flight, one wait per group.
template <int KIND>
__global__ void __launch_bounds__(MB_NT) mb_lds_kernel(int iters, unsigned* __restrict__ sink) {
    __shared__ unsigned buf[MB_NT * 2 + MB_UNR * 64];
    const int tid = threadIdx.x;
    for (int x = tid; x < MB_NT * 2 + MB_UNR * 64; x += MB_NT) buf[x] = x * 2654435761u;
    __syncthreads();
    // 32 consecutive lanes cover the 32 banks once; the wave's two halves are separate LDS passes
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)(buf + (tid & 63));
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        unsigned v[MB_UNR];
#define MB_RD(k)                                                                                                   \
    if (KIND == 0) asm volatile("ds_read_b32 %0, %1 offset:" #k "*256" : "=v"(v[k]) : "v"(addr));                     \
    else asm volatile("ds_read_u16 %0, %1 offset:" #k "*256" : "=v"(v[k]) : "v"(addr));
        MB_RD(0) MB_RD(1) MB_RD(2) MB_RD(3) MB_RD(4) MB_RD(5) MB_RD(6) MB_RD(7) MB_RD(8) MB_RD(9) MB_RD(10) MB_RD(11) MB_RD(12) MB_RD(13) MB_RD(14) MB_RD(15)
#undef MB_RD
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < MB_UNR; k += 4) acc ^= v[k] ^ v[k + 1] ^ v[k + 2] ^ v[k + 3];
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;   // keeps the reads alive, practically never stores
}

// kind 0: v_pk_add_u16 (clamp) + v_pk_min_u16 pairs; 1: v_lshl_add_u32 + v_min_u32 pairs.  8 independent chains per lane.
template <int KIND>
__global__ void __launch_bounds__(MB_NT) mb_valu_kernel(int iters, unsigned* __restrict__ sink) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    unsigned b = blockIdx.x * 0x10001u + 0x00030005u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (KIND == 0) {
                asm volatile("v_pk_add_u16 %0, %0, %8 clamp\n\tv_pk_add_u16 %1, %1, %8 clamp\n\tv_pk_add_u16 %2, %2, %8 clamp\n\tv_pk_add_u16 %3, %3, %8 clamp\n\t"
                             "v_pk_add_u16 %4, %4, %8 clamp\n\tv_pk_add_u16 %5, %5, %8 clamp\n\tv_pk_add_u16 %6, %6, %8 clamp\n\tv_pk_add_u16 %7, %7, %8 clamp\n\t"
                             "v_pk_min_u16 %0, %0, %1\n\tv_pk_min_u16 %2, %2, %3\n\tv_pk_min_u16 %4, %4, %5\n\tv_pk_min_u16 %6, %6, %7\n\t"
                             "v_pk_min_u16 %1, %1, %2\n\tv_pk_min_u16 %3, %3, %4\n\tv_pk_min_u16 %5, %5, %6\n\tv_pk_min_u16 %7, %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else {
                asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n\tv_lshl_add_u32 %1, %1, 1, %8\n\tv_lshl_add_u32 %2, %2, 1, %8\n\tv_lshl_add_u32 %3, %3, 1, %8\n\t"
                             "v_lshl_add_u32 %4, %4, 1, %8\n\tv_lshl_add_u32 %5, %5, 1, %8\n\tv_lshl_add_u32 %6, %6, 1, %8\n\tv_lshl_add_u32 %7, %7, 1, %8\n\t"
                             "v_min_u32 %0, %0, %1\n\tv_min_u32 %2, %2, %3\n\tv_min_u32 %4, %4, %5\n\tv_min_u32 %6, %6, %7\n\t"
                             "v_min_u32 %1, %1, %2\n\tv_min_u32 %3, %3, %4\n\tv_min_u32 %5, %5, %6\n\tv_min_u32 %7, %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            }
        }
    }
    const unsigned acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

}  // namespace mirp

// out[0] ds_read_b32, out[1] ds_read_u16 wave-instructions per second (whole GPU); out[2] packed-16 (add+min), out[3] 32-bit (shift-add + min)
// VALU wave-instructions per second (whole GPU).  One 1024-thread workgroup per CU, like the fill kernel.
extern "C" int mirp_microbench(mirp_ctx* c, double out[4]) {
    if (!c || !out) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    TmpDevice T;
    unsigned* sink = (unsigned*)T.get(4 * (size_t)c->n_cu);
    if (!sink) return fail(c, -6, "device allocation failed (microbench)");
    const int grid = c->n_cu;
    hipEvent_t e0 = c->ev[0], e1 = c->ev[1];
    auto timed = [&](auto launch, double insts_per_thread_iter, int iters, double* res) -> int {
        launch(64);                                    // warm-up (code load, clocks)
        double best = 0;
        for (int rep = 0; rep < 3; rep++) {
            HIPCHK(c, hipEventRecord(e0, c->stream));
            launch(iters);
            HIPCHK(c, hipEventRecord(e1, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double waves = (double)grid * (MB_NT / 64);
            const double rate = waves * insts_per_thread_iter * iters / (ms * 1e-3);
            if (rate > best) best = rate;
        }
        *res = best;
        return 0;
    };
    const int it_lds = 20000, it_valu = 20000;
    int rc = 0;
    rc |= timed([&](int it) { hipLaunchKernelGGL(mirp::mb_lds_kernel<0>, dim3(grid), dim3(MB_NT), 0, c->stream, it, sink); }, MB_UNR, it_lds, &out[0]);
    rc |= timed([&](int it) { hipLaunchKernelGGL(mirp::mb_lds_kernel<1>, dim3(grid), dim3(MB_NT), 0, c->stream, it, sink); }, MB_UNR, it_lds, &out[1]);
    rc |= timed([&](int it) { hipLaunchKernelGGL(mirp::mb_valu_kernel<0>, dim3(grid), dim3(MB_NT), 0, c->stream, it, sink); }, 64, it_valu, &out[2]);
    rc |= timed([&](int it) { hipLaunchKernelGGL(mirp::mb_valu_kernel<1>, dim3(grid), dim3(MB_NT), 0, c->stream, it, sink); }, 64, it_valu, &out[3]);
    HIPCHK(c, hipGetLastError());
    return rc;
}
