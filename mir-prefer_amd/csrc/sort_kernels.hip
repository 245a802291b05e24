// Device side of the SAM ingest: the stable (tid, pos) sort of the packed alignment records that the reference gets from
// `samtools cat` + `samtools sort` on the per-sample BAMs (/root/reference/miR_PREFeR.py:667-713, 807-859), and the `samtools view -L <bed>`
// filter of its GFF options (MP:817-859) as an interval test per record.
//
// Sort: least-significant-digit radix sort over the composite key (tid << posbits | pos), 8 bits per pass, whole 16-byte records moved.  Stability
// is what makes it a drop-in: records with equal (tid, pos) keep the sample-then-file order of the concatenated input, which decides the
// "first seen" maximum of gen_loci_alignment_info (MP:1457).  Every wave owns one contiguous tile of the input and walks it in order, 64
// records at a time; the rank of a record among the records of its digit inside a chunk comes from eight ballots (lanes with the same digit
// form a peer mask), the running per-digit offsets of the wave live in LDS.  HBM-bound: 2 x 16 B read + 16 B written per record and pass.
#include <hip/hip_runtime.h>
#include "mirp_ctx.h"

namespace mirp {

#define SORT_WAVES 4                 // waves per workgroup
#define SORT_WTILE 2048              // records per wave tile

__device__ __forceinline__ unsigned sort_digit(const MirpAln& r, int posbits, int shift) {
    const unsigned long long key = ((unsigned long long)(unsigned)r.tid << posbits) | (unsigned long long)(unsigned)r.pos;
    return (unsigned)(key >> shift) & 255u;
}

// counts[digit * n_tiles + tile]
__global__ void __launch_bounds__(64 * SORT_WAVES) sort_hist_kernel(const MirpAln* __restrict__ in, long long n, int posbits, int shift, long long n_tiles,
                                                                    unsigned* __restrict__ counts) {
    __shared__ unsigned hist[SORT_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * SORT_WAVES + wave;
    for (int x = lane; x < 256; x += 64) hist[wave][x] = 0;
    __syncthreads();
    if (tile < n_tiles) {
        const long long b = tile * SORT_WTILE, e = b + SORT_WTILE < n ? b + SORT_WTILE : n;
        for (long long k = b + lane; k < e; k += 64) atomicAdd(&hist[wave][sort_digit(in[k], posbits, shift)], 1u);
    }
    __syncthreads();
    if (tile < n_tiles)
        for (int x = lane; x < 256; x += 64) counts[(long long)x * n_tiles + tile] = hist[wave][x];
}

// exclusive scan of a 32-bit array in place (one workgroup: thread-sequential segments + one block scan); returns the total in *total
__global__ void __launch_bounds__(1024) sort_scan_kernel(unsigned* __restrict__ a, long long n, unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const long long per = (n + 1023) / 1024;
    const long long b = (long long)t * per, e = b + per < n ? b + per : n;
    unsigned long long s = 0;
    for (long long k = b; k < e; k++) s += a[k];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        unsigned long long v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = part[t] - s;
    for (long long k = b; k < e; k++) { const unsigned v = a[k]; a[k] = (unsigned)run; run += v; }
    if (t == 1023 && total) *total = part[1023];
}

__global__ void __launch_bounds__(64 * SORT_WAVES) sort_scatter_kernel(const MirpAln* __restrict__ in, MirpAln* __restrict__ out, long long n, int posbits,
                                                                       int shift, long long n_tiles, const unsigned* __restrict__ offsets) {
    __shared__ unsigned run[SORT_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * SORT_WAVES + wave;
    if (tile < n_tiles)
        for (int x = lane; x < 256; x += 64) run[wave][x] = offsets[(long long)x * n_tiles + tile];
    __syncthreads();
    if (tile >= n_tiles) return;
    const long long b = tile * SORT_WTILE, e = b + SORT_WTILE < n ? b + SORT_WTILE : n;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (long long c = b; c < e; c += 64) {
        const long long k = c + lane;
        const bool act = k < e;
        MirpAln r;
        unsigned dg = 0;
        if (act) { r = in[k]; dg = sort_digit(r, posbits, shift); }
        // lanes with the same digit (in lane order = input order)
        unsigned long long peers = __ballot(act);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const unsigned long long m = __ballot((dg >> bit) & 1u);
            peers &= ((dg >> bit) & 1u) ? m : ~m;
        }
        const unsigned rank = (unsigned)__popcll(peers & lt);
        unsigned base = 0;
        if (act) base = run[wave][dg];
        __builtin_amdgcn_wave_barrier();          // every lane has read its digit's running offset before any leader advances it
        if (act) {
            out[(unsigned long long)base + rank] = r;
            if (rank == 0) run[wave][dg] = base + (unsigned)__popcll(peers);     // one leader per digit
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// `samtools view -L bed` on the record array: keep[k] = 1 iff record k overlaps one of the (merged, per-contig sorted) regions.
// Regions: starts[r], ends[r] 0-based half-open, emax[r] = running maximum of the ends inside the contig, rfirst[tid] .. rfirst[tid+1] the contig's slice.
__global__ void mask_keep_kernel(const MirpAln* __restrict__ alns, long long n, const long long* __restrict__ rfirst, const int* __restrict__ rstart,
                                 const int* __restrict__ remax, const int* __restrict__ rspan, int* __restrict__ keep) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const MirpAln r = alns[k];
        // `samtools view -L` tests [POS - 1, bam_calend): the reference span (M + D + N) of a gapped alignment, len(SEQ) of a plain one
        const long long a0 = (long long)r.pos - 1, a1 = a0 + ((rspan && rspan[k]) ? (long long)rspan[k] - 1 : (long long)r.len);
        long long lo = rfirst[r.tid], hi = rfirst[r.tid + 1];
        const long long f = lo;
        while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)rstart[mid] < a1) lo = mid + 1; else hi = mid; }   // regions with start < a1
        keep[k] = (lo > f && (long long)remax[lo - 1] > a0) ? 1 : 0;
    }
}
__global__ void mask_compact_kernel(const MirpAln* __restrict__ in, const int* __restrict__ keep, const long long* __restrict__ kscan, long long n,
                                    MirpAln* __restrict__ out) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x)
        if (keep[k]) out[kscan[k]] = in[k];
}
// reference span + 1 of the gapped records: carried by their subtract segment (seg_span != 0), scattered to the record's slot
__global__ void mask_span_kernel(const int* __restrict__ owner, const int* __restrict__ seg_span, long long n, int* __restrict__ rspan) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x)
        if (seg_span[k]) rspan[owner[k]] = seg_span[k];
}
// coverage segments of gapped alignments follow their owner (index into the unfiltered record array)
__global__ void mask_seg_keep_kernel(const int* __restrict__ owner, long long n, const int* __restrict__ keep_rec, int* __restrict__ keep) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) keep[k] = keep_rec[owner[k]];
}

}  // namespace mirp

static inline int grid_for(long long n, int block, int cap) {
    long long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

// Stable sort of d_alns[0, n) by (tid, pos) into place; d_tmp: a second record buffer of the same size; d_counts: 256 * n_tiles + 8 unsigned.
int mirp_device_sort_alns(mirp_ctx* c, MirpAln* d_alns, MirpAln* d_tmp, long long n, int posbits, int tidbits) {
    if (n <= 1) return 0;
    const long long n_tiles = (n + SORT_WTILE - 1) / SORT_WTILE;
    if (n_tiles * SORT_WTILE > 0xffffffffll) return fail(c, -5, "device sort: more than 2^32 records in one call");
    if (c->sort_counts.ensure(4 * (size_t)(256 * n_tiles + 8))) return fail(c, -6, "device allocation failed (sort)");
    unsigned* counts = (unsigned*)c->sort_counts.p;
    const int blocks = (int)((n_tiles + SORT_WAVES - 1) / SORT_WAVES);
    MirpAln* src = d_alns;
    MirpAln* dst = d_tmp;
    for (int shift = 0; shift < posbits + tidbits; shift += 8) {
        hipLaunchKernelGGL(mirp::sort_hist_kernel, dim3(blocks), dim3(64 * SORT_WAVES), 0, c->stream, (const MirpAln*)src, n, posbits, shift, n_tiles, counts);
        hipLaunchKernelGGL(mirp::sort_scan_kernel, dim3(1), dim3(1024), 0, c->stream, counts, 256 * n_tiles, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(mirp::sort_scatter_kernel, dim3(blocks), dim3(64 * SORT_WAVES), 0, c->stream, (const MirpAln*)src, dst, n, posbits, shift, n_tiles,
                           (const unsigned*)counts);
        MirpAln* t = src; src = dst; dst = t;
    }
    if (src != d_alns) HIPCHK(c, hipMemcpyAsync(d_alns, src, sizeof(MirpAln) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipGetLastError());
    return 0;
}

// keep[] + stable compaction of records (and of the coverage segments through their owners).  Regions arrive per contig, sorted by start.
int mirp_device_mask_alns(mirp_ctx* c, MirpAln* d_alns, MirpAln* d_tmp, long long* n_io, MirpAln* d_segs, MirpAln* d_segtmp, const int* d_owner, const int* d_seg_span,
                          long long* nseg_io,
                          const long long* d_rfirst, const int* d_rstart, const int* d_remax) {
    const long long n = *n_io, ns = *nseg_io;
    if (n <= 0) return 0;
    if (c->keep.ensure(4 * (size_t)std::max<long long>(n, 1)) || c->kscan.ensure(8 * (size_t)(n + 1))) return fail(c, -6, "device allocation failed (mask)");
    int* keep = (int*)c->keep.p;
    long long* kscan = (long long*)c->kscan.p;
    TmpDevice TS;
    int* rspan = nullptr;
    if (ns > 0 && d_seg_span) {
        rspan = (int*)TS.get(4 * (size_t)n);
        if (!rspan) return fail(c, -6, "device allocation failed (mask)");
        HIPCHK(c, hipMemsetAsync(rspan, 0, 4 * (size_t)n, c->stream));
        hipLaunchKernelGGL(mirp::mask_span_kernel, dim3(grid_for(ns, 256, 8192)), dim3(256), 0, c->stream, d_owner, d_seg_span, ns, rspan);
    }
    hipLaunchKernelGGL(mirp::mask_keep_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, c->stream, (const MirpAln*)d_alns, n, d_rfirst, d_rstart, d_remax,
                       (const int*)rspan, keep);
    mirp::launch_excl_scan(c->stream, keep, kscan, n);
    long long kept = 0;
    HIPCHK(c, hipMemcpyAsync(&kept, kscan + n, 8, hipMemcpyDeviceToHost, c->stream));
    hipLaunchKernelGGL(mirp::mask_compact_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, c->stream, (const MirpAln*)d_alns, (const int*)keep,
                       (const long long*)kscan, n, d_tmp);
    if (ns > 0) {
        TmpDevice T;
        int* skeep = (int*)T.get(4 * (size_t)ns);
        long long* sscan = (long long*)T.get(8 * (size_t)(ns + 1));
        if (!skeep || !sscan) return fail(c, -6, "device allocation failed (mask segments)");
        hipLaunchKernelGGL(mirp::mask_seg_keep_kernel, dim3(grid_for(ns, 256, 8192)), dim3(256), 0, c->stream, d_owner, ns, (const int*)keep, skeep);
        mirp::launch_excl_scan(c->stream, skeep, sscan, ns);
        long long skept = 0;
        HIPCHK(c, hipMemcpyAsync(&skept, sscan + ns, 8, hipMemcpyDeviceToHost, c->stream));
        hipLaunchKernelGGL(mirp::mask_compact_kernel, dim3(grid_for(ns, 256, 8192)), dim3(256), 0, c->stream, (const MirpAln*)d_segs, (const int*)skeep,
                           (const long long*)sscan, ns, d_segtmp);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpyAsync(d_segs, d_segtmp, sizeof(MirpAln) * (size_t)skept, hipMemcpyDeviceToDevice, c->stream));
        *nseg_io = skept;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpyAsync(d_alns, d_tmp, sizeof(MirpAln) * (size_t)kept, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    *n_io = kept;
    return 0;
}
