// Candidate-stage kernels: SAM-interval -> per-base strand-split weighted coverage scan, peak (run)
// extraction, gap-merge + precursor-window extension, window payload (sequence, candidate matures).
// Replaces `samtools depth plus.bam minus.bam | awk '$3+$4>CUT'` and the Python line loop
// (/root/reference/miR_PREFeR.py = MP :877-962), gen_candidate_region_typeA (MP:1246-1371) and the
// per-window `samtools faidx` / `samtools view` subprocesses of dump_piece (MP:1070-1198, 1374-1510).
//
// HBM layout: the genome is one byte array (contigs back to back); coverage lives in two int32
// difference arrays over a "guarded" coordinate space in which every contig owns len+1 slots, so a
// read's -w lands in its contig's guard slot at worst and depth returns to 0 between contigs.
// Algorithmic traffic: 16 B per alignment record + 16 B per base (one 4-B write and one 4-B read
// per base and strand: memset + single-pass decoupled-look-back scan), SURVEY.md section 8d.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <map>
#include <mutex>
#include "mirp_internal.h"

namespace mirp {

// ------------------------------------------------------------------------------------------
// a1: per-record weight min(N, CUT) scattered into the strand's difference array (MP:734-738, 870-873)
// ------------------------------------------------------------------------------------------
// The two slots of the strand's difference array that a record (or coverage segment) touches: its clamped interval [s, e) adds at s - 1 and takes back
// at e - 1 of the contig's guarded slice.  ONE definition for the pass that writes them and the pass that clears them again (the arrays must be
// all zero between passes: a clamp that differed between the two would leave stale counts behind).
struct CovSlots { int* d; long long i0, i1; bool ok; };
__device__ __forceinline__ CovSlots cov_slots(const MirpAln& r, const long long* __restrict__ goff, const long long* __restrict__ clen, int* diff_p, int* diff_m) {
    const long long L = clen[r.tid];
    long long s = r.pos, e = (long long)r.pos + r.len;
    if (s < 1) s = 1;
    if (e > L + 1) e = L + 1;
    CovSlots c;
    c.ok = s < e;
    c.d = (r.strand & 1) ? diff_m : diff_p;
    const long long base = goff[r.tid];
    c.i0 = base + s - 1; c.i1 = base + e - 1;
    return c;
}

__global__ void __launch_bounds__(256) cov_scatter_kernel(const MirpAln* __restrict__ alns, long long n, const long long* __restrict__ goff,
                                                          const long long* __restrict__ clen, int cutoff, int* __restrict__ diff_p,
                                                          int* __restrict__ diff_m) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const MirpAln r = alns[k];
        int w = (int)(r.depth > (unsigned)cutoff ? (unsigned)cutoff : r.depth);
        const CovSlots c = cov_slots(r, goff, clen, diff_p, diff_m);
        if (!c.ok || w == 0) continue;
        if (r.strand & 2) w = -w;          // coverage segment that takes a gapped alignment's own interval back out (mirp_load_coverage_segments)
        atomicAdd(&c.d[c.i0], w);
        atomicAdd(&c.d[c.i1], -w);
    }
}

// The difference arrays are zero outside the few positions the records touch: instead of clearing 8 bytes per genome base before every pass
// (hipMemset), the positions a pass wrote are cleared again after it (two stores per record).  Invariant: the arrays are all zero between passes.
__global__ void __launch_bounds__(256) cov_unscatter_kernel(const MirpAln* __restrict__ alns, long long n, const long long* __restrict__ goff,
                                                            const long long* __restrict__ clen, int* __restrict__ diff_p, int* __restrict__ diff_m) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const CovSlots c = cov_slots(alns[k], goff, clen, diff_p, diff_m);
        if (!c.ok) continue;
        c.d[c.i0] = 0;
        c.d[c.i1] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// a2 (first half): single-pass scan of both difference arrays -> depth, threshold, run starts.
// Decoupled look-back over 64-bit {flag, d+, d-} and {flag, n_starts, n_above} tile descriptors.
// ------------------------------------------------------------------------------------------
// 512 threads x 16 positions: the same 8192-position tile as 256 x 32, but 96 instead of 187 VGPRs, so twice the waves hide the two look-backs
// (measured on config[1] and on a cfg[4] shard: 0.50 / 0.26 of 8 TB/s against 0.46 / 0.256)
#ifndef SCAN_NT
#define SCAN_NT 512
#endif
#ifndef SCAN_IPT
#define SCAN_IPT 16
#endif
#define SCAN_TILE (SCAN_NT * SCAN_IPT)

__device__ __forceinline__ unsigned long long ld_status(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// pack two 31-bit non-negative values under a 2-bit flag
__device__ __forceinline__ unsigned long long pack2(unsigned flag, unsigned a, unsigned b) {
    return ((unsigned long long)flag << 62) | ((unsigned long long)(a & 0x7fffffffu) << 31) | (unsigned long long)(b & 0x7fffffffu);
}

// inclusive prefix sum over the 64 lanes of a wave in registers (DPP row shifts inside the rows of 16, then the row broadcasts)
__device__ __forceinline__ int wave_incl_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
    return x;
}

// Exclusive prefixes of a per-thread (a, b) pair sequence g[0..NV) in POSITION order for the wave-striped tile layout below: the order is
// (wave, v, lane).  On return ea[v], eb[v] = sum over everything before (wave, v, lane) inside the tile; tot_a / tot_b = tile totals.
template <int NV>
__device__ __forceinline__ void tile_excl_scan2(const int (&a)[NV], const int (&b)[NV], int (&ea)[NV], int (&eb)[NV], int& tot_a, int& tot_b,
                                                int* sh /* 2 * (SCAN_NT / 64) */) {
    const int wave = threadIdx.x >> 6;
    int ra = 0, rb = 0;                 // running row bases inside the wave
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int ia = wave_incl_scan(a[v]), ib = wave_incl_scan(b[v]);
        ea[v] = ra + ia - a[v]; eb[v] = rb + ib - b[v];
        ra += __builtin_amdgcn_readlane(ia, 63); rb += __builtin_amdgcn_readlane(ib, 63);
    }
    if ((threadIdx.x & 63) == 0) { sh[wave * 2] = ra; sh[wave * 2 + 1] = rb; }
    __syncthreads();
    int wa = 0, wb = 0, ta = 0, tb = 0;
#pragma unroll
    for (int w = 0; w < SCAN_NT / 64; w++) {
        const int xa = sh[w * 2], xb = sh[w * 2 + 1];
        if (w < wave) { wa += xa; wb += xb; }
        ta += xa; tb += xb;
    }
#pragma unroll
    for (int v = 0; v < NV; v++) { ea[v] += wa; eb[v] += wb; }
    tot_a = ta; tot_b = tb;
    __syncthreads();
}

struct RunStart { long long gx; int dp, dm; };

// Tile layout ("wave-striped"): a tile is SCAN_NT * SCAN_IPT consecutive positions; every wave owns a contiguous chunk of 64 * SCAN_IPT of
// them and reads it in SCAN_IPT / 4 rows of 256 positions, lane l taking the 16-byte vector l of the row: every load instruction of a wave is
// one fully coalesced kilobyte (a thread-contiguous layout makes each instruction touch 64 cache lines for 16 bytes apiece and thrashes the
// vector L1).  A thread therefore holds SCAN_NV groups of four consecutive positions, 256 positions apart.
#define SCAN_NV (SCAN_IPT / 4)
// first[t] = index of the first record (sorted by (tid, pos)) whose start slot lies in tile t or later; first[n_tiles] = n.  One thread per
// record boundary: where the tile index steps from tp to t, the tiles tp + 1 .. t begin at record k.
__device__ __forceinline__ long long cov_start_slot(const MirpAln& r, const long long* __restrict__ goff, const long long* __restrict__ clen) {
    const long long L = clen[r.tid];
    long long s = r.pos;
    if (s < 1) s = 1;
    if (s > L + 1) s = L + 1;          // (a record past the contig end contributes nothing; clamped so that the slots stay sorted across contigs)
    return goff[r.tid] + s - 1;
}
// The same pass leaves every tile's TOTAL of difference values in agg[strand][tile]: a record adds +w and -w, which cancel inside one tile, so only
// the few records whose two slots lie in different tiles (reads are tens of bases, a tile is 8192) touch it.  The exclusive prefix of those totals
// is the depth carried into a tile -- the fused scan reads it instead of looking back for it.
__global__ void __launch_bounds__(256) cov_tile_first_kernel(const MirpAln* __restrict__ alns, long long n, const long long* __restrict__ goff,
                                                             const long long* __restrict__ clen, long long n_tiles, int cutoff, int max_len,
                                                             long long* __restrict__ first, long long* __restrict__ back /* [n_tiles + 1] */,
                                                             int* __restrict__ agg /* [2][n_tiles], zero */) {
    // back[t] = first record that can reach tile t: its start slot lies at most max_len (the longest record) before the tile.  A tile reads the records
    // back[t] .. first[t + 1]: those of its own range plus the handful that hang over from the tile before (not that tile's whole record list).
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k <= n; k += (long long)gridDim.x * blockDim.x) {
        long long t = n_tiles, tb = n_tiles;
        if (k < n) {
            const MirpAln r = alns[k];
            const long long slot = cov_start_slot(r, goff, clen);
            t = slot / SCAN_TILE; tb = (slot + max_len) / SCAN_TILE;
            int w = (int)(r.depth > (unsigned)cutoff ? (unsigned)cutoff : r.depth);
            const CovSlots c = cov_slots(r, goff, clen, nullptr, nullptr);
            const long long ta = c.i0 / SCAN_TILE, tb = c.i1 / SCAN_TILE;
            if (c.ok && w != 0 && ta != tb) {
                if (r.strand & 2) w = -w;
                int* a = agg + ((r.strand & 1) ? n_tiles : 0);
                atomicAdd(&a[ta], w);
                atomicAdd(&a[tb], -w);
            }
        }
        const long long sp = k > 0 ? cov_start_slot(alns[k - 1], goff, clen) : 0;
        const long long tp = k > 0 ? sp / SCAN_TILE : -1, tbp = k > 0 ? (sp + max_len) / SCAN_TILE : -1;
        for (long long x = tp + 1; x <= t && x <= n_tiles; x++) first[x] = k;
        for (long long x = tbp + 1; x <= tb && x <= n_tiles; x++) back[x] = k;
    }
}
__global__ void __launch_bounds__(256) cov_maxlen_kernel(const MirpAln* __restrict__ alns, long long n, int* __restrict__ out) {
    int m = 0;
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) { const int l = alns[k].len; m = l > m ? l : m; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

// FUSED: the tile's difference values do not come from the dense arrays (one global atomic pair per record before, 8 bytes read per base here) but
// are built in LDS from the sorted records themselves: the records that start in this tile or close enough before it to reach it (back[] / first[];
// no record is longer than a tile -- checked on the host) add their +w / -w with LDS atomics.  A tile that holds a covered base, or continues a run, then writes its
// values to the dense arrays once, with plain coalesced stores: that is all the run walk ever reads.  No clearing pass, no global atomics.
#define COV_CT 256          // contigs whose {offset, length} fit the fused scan's LDS table (more: read from global memory)
template <bool FUSED>
__global__ void __launch_bounds__(SCAN_NT) cov_scan_kernel(int* __restrict__ diff_p, int* __restrict__ diff_m, long long gtot,
                                                           const MirpAln* __restrict__ alns, const long long* __restrict__ goff, const long long* __restrict__ clen,
                                                           int n_contigs, const long long* __restrict__ first, const long long* __restrict__ back,
                                                           const long long* __restrict__ carry_p,
                                                           const long long* __restrict__ carry_m,
                                                           int cutoff, unsigned long long* __restrict__ stat_d,
                                                           unsigned long long* __restrict__ stat_c, unsigned int* __restrict__ ticket,
                                                           RunStart* __restrict__ starts, long long starts_cap,
                                                           MirpDepthPos* __restrict__ depth_out /* nullable; global coords in .pos via hi/lo */,
                                                           long long depth_cap, long long* __restrict__ depth_gx /* nullable */,
                                                           unsigned long long* __restrict__ totals /* [0]=n_starts [1]=n_above */) {
    __shared__ int sh[2 * (SCAN_NT / 64) + 2];
    __shared__ unsigned int s_tile;
    __shared__ int s_carry[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    extern __shared__ __align__(16) int ld[];          // FUSED: [2][SCAN_TILE]: + strand, - strand; then [2][COV_CT] contig offsets and lengths
    // the contigs' {offset of the guarded slice, length} into LDS while the ticket is on its way: one dependent global round trip less per record batch
    long long* ct = reinterpret_cast<long long*>(ld + 2 * SCAN_TILE);
    const bool ct_lds = FUSED && n_contigs <= COV_CT;
    if constexpr (FUSED) { if (ct_lds) for (int x = tid; x < n_contigs; x += SCAN_NT) { ct[x] = goff[x]; ct[COV_CT + x] = clen[x]; } }
    __syncthreads();
    const unsigned int tile = s_tile;
    const long long base = (long long)tile * SCAN_TILE;
    const long long wbase = base + (long long)wave * (64 * SCAN_IPT);
    int vp[SCAN_IPT], vm[SCAN_IPT];
    if constexpr (FUSED) {
        const long long k0 = back[tile], k1 = first[tile + 1];
        if (k0 == k1) {          // no record starts in this tile or the one before it: nothing can change here (most tiles of a sparse sample)
#pragma unroll
            for (int v = 0; v < SCAN_IPT; v++) { vp[v] = 0; vm[v] = 0; }
        } else {
            for (int x = tid * 4; x < 2 * SCAN_TILE; x += SCAN_NT * 4) *reinterpret_cast<int4*>(ld + x) = make_int4(0, 0, 0, 0);
            __syncthreads();
            constexpr int RU = 4;          // records in flight per thread: the loop is a chain of dependent loads (record, then its contig's offsets) otherwise
            for (long long kb = k0 + tid; kb < k1; kb += (long long)SCAN_NT * RU) {
                MirpAln r[RU];
#pragma unroll
                for (int u = 0; u < RU; u++) { const long long k = kb + (long long)u * SCAN_NT; if (k < k1) r[u] = alns[k]; }
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    if (kb + (long long)u * SCAN_NT >= k1) continue;
                    int w = (int)(r[u].depth > (unsigned)cutoff ? (unsigned)cutoff : r[u].depth);
                    CovSlots c;
                    if (ct_lds) {          // cov_slots with the table in LDS
                        const long long L = ct[COV_CT + r[u].tid];
                        long long s = r[u].pos, e = (long long)r[u].pos + r[u].len;
                        if (s < 1) s = 1;
                        if (e > L + 1) e = L + 1;
                        c.ok = s < e;
                        const long long o = ct[r[u].tid];
                        c.i0 = o + s - 1; c.i1 = o + e - 1;
                    } else c = cov_slots(r[u], goff, clen, nullptr, nullptr);
                    if (!c.ok || w == 0) continue;
                    if (r[u].strand & 2) w = -w;
                    int* d = ld + ((r[u].strand & 1) ? SCAN_TILE : 0);
                    const long long a = c.i0 - base, b = c.i1 - base;
                    if (a >= 0 && a < SCAN_TILE) atomicAdd(&d[a], w);
                    if (b >= 0 && b < SCAN_TILE) atomicAdd(&d[b], -w);
                }
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < SCAN_NV; v++) {
                const int o = wave * (64 * SCAN_IPT) + v * 256 + lane * 4;
                const int4 a = *reinterpret_cast<const int4*>(ld + o);
                const int4 b = *reinterpret_cast<const int4*>(ld + SCAN_TILE + o);
                vp[v * 4] = a.x; vp[v * 4 + 1] = a.y; vp[v * 4 + 2] = a.z; vp[v * 4 + 3] = a.w;
                vm[v * 4] = b.x; vm[v * 4 + 1] = b.y; vm[v * 4 + 2] = b.z; vm[v * 4 + 3] = b.w;
            }
        }
    } else
#pragma unroll
    for (int v = 0; v < SCAN_NV; v++) {
        const long long x = wbase + v * 256 + lane * 4;
        if (x + 3 < gtot) {
            const int4 a = *reinterpret_cast<const int4*>(diff_p + x);
            const int4 b = *reinterpret_cast<const int4*>(diff_m + x);
            vp[v * 4] = a.x; vp[v * 4 + 1] = a.y; vp[v * 4 + 2] = a.z; vp[v * 4 + 3] = a.w;
            vm[v * 4] = b.x; vm[v * 4 + 1] = b.y; vm[v * 4 + 2] = b.z; vm[v * 4 + 3] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                vp[v * 4 + j] = (x + j < gtot) ? diff_p[x + j] : 0;
                vm[v * 4 + j] = (x + j < gtot) ? diff_m[x + j] : 0;
            }
        }
    }
    int gp[SCAN_NV], gm[SCAN_NV], ep[SCAN_NV], em[SCAN_NV];
#pragma unroll
    for (int v = 0; v < SCAN_NV; v++) { gp[v] = vp[v * 4] + vp[v * 4 + 1] + vp[v * 4 + 2] + vp[v * 4 + 3]; gm[v] = vm[v * 4] + vm[v * 4 + 1] + vm[v * 4 + 2] + vm[v * 4 + 3]; }
    int tp, tm;
    tile_excl_scan2<SCAN_NV>(gp, gm, ep, em, tp, tm, sh);
    // ---- look-back #1: depth carried into this tile (fused: known beforehand, see cov_tile_first_kernel)
    if constexpr (FUSED) {
        if (tid == 0) { s_carry[0] = (int)((unsigned long long)carry_p[tile] & 0x7fffffffull); s_carry[1] = (int)((unsigned long long)carry_m[tile] & 0x7fffffffull); }
    } else {
    if (tid == 0) st_status(&stat_d[tile], pack2(tile == 0 ? 2u : 1u, (unsigned)tp, (unsigned)tm));
    if (tid < 64) {
        unsigned cp = 0, cm = 0;
        if (tile > 0) {
            long long idx = (long long)tile - 1;
            while (true) {
                long long j = idx - lane;
                unsigned long long v = 0;
                unsigned flag = 3;
                if (j >= 0) { do { v = ld_status(&stat_d[j]); flag = (unsigned)(v >> 62); } while (flag == 0); }
                unsigned long long pm = __ballot(flag == 2);
                int stop = pm ? (__ffsll((long long)pm) - 1) : 64;
                unsigned a = (j >= 0 && lane <= stop) ? (unsigned)((v >> 31) & 0x7fffffffu) : 0u;
                unsigned b = (j >= 0 && lane <= stop) ? (unsigned)(v & 0x7fffffffu) : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                cp += a; cm += b;
                if (pm || idx - 64 < 0) break;
                idx -= 64;
            }
            if (lane == 0) st_status(&stat_d[tile], pack2(2u, cp + (unsigned)tp, cm + (unsigned)tm));
        }
        if (lane == 0) { s_carry[0] = (int)(cp & 0x7fffffffu); s_carry[1] = (int)(cm & 0x7fffffffu); }
    }
    }
    __syncthreads();
    // depths are >= 0 everywhere, but tile aggregates of a *difference* array can be negative: they are
    // carried modulo 2^31, which is exact as long as true depths stay below 2^31.
    // ---- per-position depth, threshold, run-start flags.  Bit v*4+j of the masks = position j of group v; the depth just before a group is its
    // exclusive prefix, so "was the previous position above the threshold" needs no neighbour.
    unsigned abovemask = 0, startmask = 0;
    int ns[SCAN_NV], na[SCAN_NV];
#pragma unroll
    for (int v = 0; v < SCAN_NV; v++) {
        int cp = (int)(((unsigned)s_carry[0] + (unsigned)ep[v]) & 0x7fffffffu), cm = (int)(((unsigned)s_carry[1] + (unsigned)em[v]) & 0x7fffffffu);
        const long long x0 = wbase + v * 256 + lane * 4;
        unsigned prev = (cp + cm > cutoff) ? 1u : 0u;
        unsigned am = 0, sm = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cp = (int)(((unsigned)cp + (unsigned)vp[v * 4 + j]) & 0x7fffffffu); cm = (int)(((unsigned)cm + (unsigned)vm[v * 4 + j]) & 0x7fffffffu);
            const unsigned ab = ((x0 + j < gtot) && (cp + cm > cutoff)) ? 1u : 0u;
            am |= ab << j; sm |= (ab & ~prev) << j;
            prev = ab;
        }
        abovemask |= am << (v * 4); startmask |= sm << (v * 4);
        ns[v] = __popc(sm); na[v] = __popc(am);
    }
    int es[SCAN_NV], ea[SCAN_NV], ts, ta;
    tile_excl_scan2<SCAN_NV>(ns, na, es, ea, ts, ta, sh);
    if constexpr (FUSED) {
        // ta = covered bases of the tile; a run can also reach in from the tile before (depth carried in above the threshold) and its walk then reads
        // this tile's first values.  Written per ROW of 256 positions (one coalesced kilobyte per array and wave): a run walk reads the positions of
        // its run and the one that ends it, so a row is needed if it holds a covered base or the depth just before it is above the threshold --
        // at a config[4] shard every tile holds a locus, but two rows in three hold none.
        const bool carried = (int)(((unsigned)s_carry[0] + (unsigned)s_carry[1]) & 0x7fffffffu) > cutoff;
        if (ta > 0 || carried) {
#pragma unroll
            for (int v = 0; v < SCAN_NV; v++) {
                const int before = (int)(((unsigned)s_carry[0] + (unsigned)ep[v]) & 0x7fffffffu) + (int)(((unsigned)s_carry[1] + (unsigned)em[v]) & 0x7fffffffu);
                if (__ballot(((abovemask >> (v * 4)) & 15u) != 0) == 0ull && __builtin_amdgcn_readfirstlane(before) <= cutoff) continue;
                const long long x = wbase + v * 256 + lane * 4;
                if (x + 3 < gtot) {
                    *reinterpret_cast<int4*>(diff_p + x) = make_int4(vp[v * 4], vp[v * 4 + 1], vp[v * 4 + 2], vp[v * 4 + 3]);
                    *reinterpret_cast<int4*>(diff_m + x) = make_int4(vm[v * 4], vm[v * 4 + 1], vm[v * 4 + 2], vm[v * 4 + 3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (x + j < gtot) { diff_p[x + j] = vp[v * 4 + j]; diff_m[x + j] = vm[v * 4 + j]; }
                }
            }
        }
    }
    // ---- look-back #2: ordered output offsets
    if (tid == 0) st_status(&stat_c[tile], pack2(tile == 0 ? 2u : 1u, (unsigned)ts, (unsigned)ta));
    if (tid < 64) {
        long long cs = 0, ca = 0;
        if (tile > 0) {
            long long idx = (long long)tile - 1;
            while (true) {
                long long j = idx - lane;
                unsigned long long v = 0;
                unsigned flag = 3;
                if (j >= 0) { do { v = ld_status(&stat_c[j]); flag = (unsigned)(v >> 62); } while (flag == 0); }
                unsigned long long pm = __ballot(flag == 2);
                int stop = pm ? (__ffsll((long long)pm) - 1) : 64;
                int a = (j >= 0 && lane <= stop) ? (int)((v >> 31) & 0x7fffffffu) : 0;
                int b = (j >= 0 && lane <= stop) ? (int)(v & 0x7fffffffu) : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                cs += a; ca += b;
                if (pm || idx - 64 < 0) break;
                idx -= 64;
            }
            if (lane == 0) st_status(&stat_c[tile], pack2(2u, (unsigned)(cs + ts), (unsigned)(ca + ta)));
        }
        if (lane == 0) { s_carry[2] = (int)cs; s_carry[3] = (int)ca; }
    }
    __syncthreads();
    if (abovemask) {     // the depths are recomputed here instead of being held in registers across the two look-backs (most threads own no covered base)
#pragma unroll
        for (int v = 0; v < SCAN_NV; v++) {
            if (!((abovemask >> (v * 4)) & 15u)) continue;
            int cp = (int)(((unsigned)s_carry[0] + (unsigned)ep[v]) & 0x7fffffffu), cm = (int)(((unsigned)s_carry[1] + (unsigned)em[v]) & 0x7fffffffu);
            long long os = (long long)s_carry[2] + es[v], oa = (long long)s_carry[3] + ea[v];
            const long long x0 = wbase + v * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cp = (int)(((unsigned)cp + (unsigned)vp[v * 4 + j]) & 0x7fffffffu); cm = (int)(((unsigned)cm + (unsigned)vm[v * 4 + j]) & 0x7fffffffu);
                if (startmask & (1u << (v * 4 + j))) {
                    if (os < starts_cap) { RunStart r; r.gx = x0 + j; r.dp = cp; r.dm = cm; starts[os] = r; }
                    os++;
                }
                if (abovemask & (1u << (v * 4 + j))) {
                    if (depth_out && oa < depth_cap) {
                        MirpDepthPos d; d.tid = 0; d.pos = 0; d.dp = cp; d.dm = cm;
                        depth_out[oa] = d; depth_gx[oa] = x0 + j;
                    }
                    oa++;
                }
            }
        }
    }
    if ((long long)(tile + 1) * SCAN_TILE >= gtot && tid == 0) {      // last tile: the totals are its inclusive prefix
        totals[0] = (unsigned long long)((long long)s_carry[2] + ts); totals[1] = (unsigned long long)((long long)s_carry[3] + ta);
    }
}

// ------------------------------------------------------------------------------------------
// a2 (second half): walk each run, strand vote (with the contig-change double count, MP:905-906 +
// 926-929), drop runs shorter than min_len (MP:956).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int contig_of(const long long* __restrict__ goff, int n_contigs, long long gx) {
    int lo = 0, hi = n_contigs - 1;   // last t with goff[t] <= gx
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (goff[mid] <= gx) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ void __launch_bounds__(256) run_walk_kernel(const RunStart* __restrict__ starts, long long n_runs, const int* __restrict__ diff_p,
                                                       const int* __restrict__ diff_m, long long gtot, int cutoff,
                                                       const long long* __restrict__ goff, int n_contigs, int min_len,
                                                       MirpPeak* __restrict__ runs, int* __restrict__ keep, int first_run_double) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n_runs; k += (long long)gridDim.x * blockDim.x) {
        RunStart r = starts[k];
        int dp = r.dp, dm = r.dm;
        long long sum_p = dp, sum_m = dm, x = r.gx + 1;
        // the walk is a chain of dependent steps, but the LOADS need not be: RW_B positions of both arrays are fetched at once, then stepped through
        // (one load pair per step made the kernel as long as its longest run times the memory latency: 0.24 ms at a config[4] shard)
#ifndef RW_B
#define RW_B 8
#endif
        for (bool open = true; open && x < gtot;) {
            int vp[RW_B], vm[RW_B];
#pragma unroll
            for (int q = 0; q < RW_B; q++) { const long long xx = x + q < gtot ? x + q : gtot - 1; vp[q] = diff_p[xx]; vm[q] = diff_m[xx]; }
#pragma unroll
            for (int q = 0; q < RW_B; q++) {
                if (!open || x >= gtot) break;
                dp += vp[q]; dm += vm[q];
                if (dp + dm > cutoff) { sum_p += dp; sum_m += dm; x++; } else open = false;
            }
        }
        int t = contig_of(goff, n_contigs, r.gx);
        if (k > 0) {
            int tprev = contig_of(goff, n_contigs, starts[k - 1].gx);
            if (tprev != t) { sum_p += r.dp; sum_m += r.dm; }
        } else if (first_run_double) {   // contig shard: a contig of another shard precedes this run in the depth file of the whole genome
            sum_p += r.dp; sum_m += r.dm;
        }
        MirpPeak p;
        p.tid = t; p.start = (int)(r.gx - goff[t] + 1); p.end = (int)(x - goff[t] + 1); p.strand = (sum_p > sum_m) ? 0 : 1;
        runs[k] = p;
        keep[k] = (p.end - p.start >= min_len) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) depth_fix_kernel(MirpDepthPos* __restrict__ d, const long long* __restrict__ gx, long long n,
                                                        const long long* __restrict__ goff, int n_contigs) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        int t = contig_of(goff, n_contigs, gx[k]);
        d[k].tid = t; d[k].pos = (int)(gx[k] - goff[t] + 1);
    }
}

// ------------------------------------------------------------------------------------------
// generic single-block exclusive scan of int32 (small arrays: runs, regions)
// ------------------------------------------------------------------------------------------
// A thread owns SCAN1_NT consecutive elements per round (serial scan in registers, one wave scan + one 16-entry combine per round): 16,384
// elements per round of the block instead of 1,024 -- at the 3 x 10^5 runs / regions of a config[4] rank shard 19 rounds instead of 300 (the
// stage calls this scan seven times; round 3: 0.2 ms per call, more than the coverage scan itself).
#define SCAN1_NT 16
__global__ void __launch_bounds__(1024) excl_scan_i32_kernel(const int* __restrict__ in, long long* __restrict__ out, long long n) {
    __shared__ long long sh[16];
    __shared__ long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (long long b = 0; b < n; b += 1024 * SCAN1_NT) {
        const long long k0 = b + (long long)tid * SCAN1_NT;
        int v[SCAN1_NT];
        if (k0 + SCAN1_NT <= n && (((size_t)(in + k0)) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < SCAN1_NT / 4; q++) {
                const int4 x = *reinterpret_cast<const int4*>(in + k0 + 4 * q);
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < SCAN1_NT; q++) v[q] = (k0 + q < n) ? in[k0 + q] : 0;
        }
        long long mine = 0;
#pragma unroll
        for (int q = 0; q < SCAN1_NT; q++) mine += v[q];
        long long iv = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { long long t = __shfl_up(iv, o); if (lane >= o) iv += t; }
        if (lane == 63) sh[wave] = iv;
        __syncthreads();
        long long wpre = 0, tot = 0;
        for (int w = 0; w < 16; w++) { if (w < wave) wpre += sh[w]; tot += sh[w]; }
        const long long carry = s_carry;
        long long run = carry + wpre + iv - mine;
#pragma unroll
        for (int q = 0; q < SCAN1_NT; q++) { if (k0 + q < n) out[k0 + q] = run; run += v[q]; }
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) out[n] = s_carry;
}

// The same scan over many workgroups for the arrays of a rank shard (10^5 .. 10^7 runs, regions, windows): single pass, decoupled look-back over
// one 64-bit descriptor {flag:2 | partial sum:62, two's complement} per 4,096-element block; the launcher clears the ticket counter and the
// descriptors of the blocks in use before every call (one fill of 8 bytes per block).
#define MSCAN_NT 256
#define MSCAN_IPT 16
#define MSCAN_TILE (MSCAN_NT * MSCAN_IPT)
#define MSCAN_MAX_BLOCKS 16384
__device__ __forceinline__ unsigned long long mscan_pack(unsigned flag, long long v) {
    return ((unsigned long long)flag << 62) | ((unsigned long long)v & 0x3fffffffffffffffull);
}
__global__ void __launch_bounds__(MSCAN_NT) excl_scan_i32_mb_kernel(const int* __restrict__ in, long long* __restrict__ out, long long n,
                                                                    unsigned long long* __restrict__ scratch) {
    __shared__ long long sh[MSCAN_NT / 64];
    __shared__ unsigned s_tile;
    __shared__ long long s_pre;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = (unsigned)atomicAdd(&scratch[0], 1ull);
    __syncthreads();
    const unsigned tile = s_tile;
    unsigned long long* status = scratch + 1;
    const long long k0 = (long long)tile * MSCAN_TILE + (long long)tid * MSCAN_IPT;
    int v[MSCAN_IPT];
    if (k0 + MSCAN_IPT <= n && (((size_t)(in + k0)) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < MSCAN_IPT / 4; q++) {
            const int4 x = *reinterpret_cast<const int4*>(in + k0 + 4 * q);
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < MSCAN_IPT; q++) v[q] = (k0 + q < n) ? in[k0 + q] : 0;
    }
    long long mine = 0;
#pragma unroll
    for (int q = 0; q < MSCAN_IPT; q++) mine += v[q];
    long long iv = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long t = __shfl_up(iv, o); if (lane >= o) iv += t; }
    if (lane == 63) sh[wave] = iv;
    __syncthreads();
    long long wpre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MSCAN_NT / 64; w++) { if (w < wave) wpre += sh[w]; tot += sh[w]; }
    if (tid == 0) st_status(&status[tile], mscan_pack(tile == 0 ? 2u : 1u, tot));
    if (tid < 64) {
        long long pre = 0;
        if (tile > 0) {
            long long idx = (long long)tile - 1;
            while (true) {
                const long long j = idx - lane;
                unsigned long long d = 0;
                unsigned flag = 3;
                if (j >= 0) { do { d = ld_status(&status[j]); flag = (unsigned)(d >> 62); } while (flag == 0); }
                const unsigned long long pm = __ballot(flag == 2);
                const int stop = pm ? (__ffsll((long long)pm) - 1) : 64;
                long long a = (j >= 0 && lane <= stop) ? ((long long)(d << 2) >> 2) : 0;          // sign-extend the 62-bit partial sum
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
                pre += a;
                if (pm || idx - 64 < 0) break;
                idx -= 64;
            }
            if (lane == 0) st_status(&status[tile], mscan_pack(2u, pre + tot));
        }
        if (lane == 0) s_pre = pre;
    }
    __syncthreads();
    long long run = s_pre + wpre + iv - mine;
#pragma unroll
    for (int q = 0; q < MSCAN_IPT; q++) { if (k0 + q < n) out[k0 + q] = run; run += v[q]; }
    if (tid == 0 && (long long)(tile + 1) * MSCAN_TILE >= n) out[n] = s_pre + tot;
}

// kept runs -> peaks in @SQ order (API output) and in sorted-contig order (pipeline order, MP:1309)
__global__ void __launch_bounds__(256) contig_peak_ranges_kernel(const MirpPeak* __restrict__ runs, long long n_runs, const long long* __restrict__ kscan,
                                                                 int n_contigs, long long* __restrict__ csq /* [n_contigs+1] kept-start per contig */) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= n_contigs; t += gridDim.x * blockDim.x) {
        long long lo = 0, hi = n_runs;   // first run with tid >= t
        while (lo < hi) { long long mid = (lo + hi) >> 1; if (runs[mid].tid < t) lo = mid + 1; else hi = mid; }
        csq[t] = kscan[lo];
    }
}
__global__ void contig_dest_kernel(const long long* __restrict__ csq, const int* __restrict__ order, int n_contigs, long long* __restrict__ cdest) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        long long acc = 0;
        for (int oi = 0; oi < n_contigs; oi++) { int t = order[oi]; cdest[t] = acc; acc += csq[t + 1] - csq[t]; }
    }
}
__global__ void __launch_bounds__(256) peak_compact_kernel(const MirpPeak* __restrict__ runs, const int* __restrict__ keep, const long long* __restrict__ kscan,
                                                           long long n_runs, const long long* __restrict__ csq, const long long* __restrict__ cdest,
                                                           MirpPeak* __restrict__ peaks_sq, MirpPeak* __restrict__ peaks_sorted) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n_runs; k += (long long)gridDim.x * blockDim.x) {
        if (!keep[k]) continue;
        MirpPeak p = runs[k];
        long long i = kscan[k];
        peaks_sq[i] = p;
        peaks_sorted[cdest[p.tid] + (i - csq[p.tid])] = p;
    }
}

// ------------------------------------------------------------------------------------------
// a3: gap merge (next_region_typeA MP:1256-1270) + window extension (extend_region MP:1272-1300)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) region_head_kernel(const MirpPeak* __restrict__ P, long long n, int max_gap, int* __restrict__ head) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x)
        head[k] = (k == 0 || P[k].tid != P[k - 1].tid || P[k].start - P[k - 1].end >= max_gap) ? 1 : 0;
}
__global__ void __launch_bounds__(256) region_first_kernel(const int* __restrict__ head, const long long* __restrict__ hscan, long long n,
                                                           long long* __restrict__ rfirst /* [n_regions+1] */) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k <= n; k += (long long)gridDim.x * blockDim.x) {
        if (k == n) rfirst[hscan[n]] = n;
        else if (head[k]) rfirst[hscan[k]] = k;
    }
}

__device__ __forceinline__ int extend_region(int s, int e, int L, long long seqlen, int out[2][2]) {
    int length = e - s;
    if (length > L + 50) return 0;
    if (length > L) { out[0][0] = s; out[0][1] = e; return 1; }
    if (length < 60) {
        long long ls = (long long)s - (L - length - 25) - 25, le = (long long)e + 25;
        long long rs = (long long)s - 25, re = (long long)e + (L - length - 25) + 25;
        if (ls < 0) ls = 0;
        if (le > seqlen) le = seqlen;
        if (rs < 0) rs = 0;
        if (re > seqlen) re = seqlen;
        out[0][0] = (int)ls; out[0][1] = (int)le; out[1][0] = (int)rs; out[1][1] = (int)re;
        return 2;
    }
    int ext = (L - length) / 2;
    long long left = (long long)s - ext, right = (long long)e + ext;
    if (left < 0) left = 1;
    if (right > seqlen) right = seqlen + 1;
    out[0][0] = (int)left; out[0][1] = (int)right;
    return 1;
}

// per region: number of FASTA entries (incl. the both-strand L/R duplication MP:1184-1192), locus flag, slot demand
__global__ void __launch_bounds__(256) region_count_kernel(const MirpPeak* __restrict__ P, const long long* __restrict__ rfirst, long long n_regions,
                                                           const long long* __restrict__ clen, int L, int* __restrict__ n_entries,
                                                           int* __restrict__ is_locus, int* __restrict__ n_slots) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n_regions; r += (long long)gridDim.x * blockDim.x) {
        long long f = rfirst[r], l = rfirst[r + 1];
        int ext[2][2];
        int nwin = extend_region(P[f].start, P[l - 1].end, L, clen[P[f].tid], ext);
        int has_p = 0, has_m = 0;
        for (long long k = f; k < l; k++) { if (P[k].strand == 0) has_p = 1; else has_m = 1; }
        int ne = 0;
        if (nwin > 0) ne = (has_p != has_m) ? nwin : (nwin == 1 ? 2 : 6);
        n_entries[r] = ne;
        is_locus[r] = nwin > 0 ? 1 : 0;
        n_slots[r] = ne * (int)(l - f);
    }
}

__global__ void __launch_bounds__(256) region_emit_kernel(const MirpPeak* __restrict__ P, const long long* __restrict__ rfirst, long long n_regions,
                                                          const long long* __restrict__ clen, int L, const long long* __restrict__ escan,
                                                          const long long* __restrict__ lscan, const long long* __restrict__ sscan,
                                                          MirpWindow* __restrict__ W, MirpLocus* __restrict__ loci, MirpPeak* __restrict__ wpeaks,
                                                          int* __restrict__ roles, int seq_stride) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n_regions; r += (long long)gridDim.x * blockDim.x) {
        long long f = rfirst[r], l = rfirst[r + 1];
        int npk = (int)(l - f);
        int ext[2][2];
        int rs = P[f].start, re = P[l - 1].end, tid = P[f].tid;
        int nwin = extend_region(rs, re, L, clen[tid], ext);
        if (nwin == 0) continue;
        MirpLocus lc;
        lc.tid = tid; lc.start = rs; lc.end = re; lc.n_windows = nwin;
        lc.w[0][0] = ext[0][0]; lc.w[0][1] = ext[0][1]; lc.w[1][0] = nwin > 1 ? ext[1][0] : 0; lc.w[1][1] = nwin > 1 ? ext[1][1] : 0;
        lc.peak_first = f; lc.n_peaks = npk; lc.pad = 0;
        loci[lscan[r]] = lc;
        int has_p = 0, has_m = 0;
        for (long long k = f; k < l; k++) { if (P[k].strand == 0) has_p = 1; else has_m = 1; }
        long long e = escan[r], slot = sscan[r];
        int emitted = 0;
        auto emit = [&](int idx, int strand, int only) {
            MirpWindow w;
            w.tid = tid; w.ws = ext[idx][0]; w.we = ext[idx][1]; w.strand = strand; w.loc_s = rs; w.loc_e = re;
            w.tag = nwin == 2 ? (idx == 0 ? 1 : 2) : 0;
            w.peak_off = slot; w.n_peaks = 0;
            for (long long k = f; k < l; k++) {
                if (only >= 0 && P[k].strand != only) continue;
                wpeaks[slot + w.n_peaks] = P[k]; w.n_peaks++;
            }
            w.n_matures = 0; w.pad0 = (int)f; w.mature_off = 2 * slot;   // pad0 carries the region's first peak until the payload kernel ran
            w.seq_off = e * (long long)seq_stride; w.seq_len = 0; w.pad1 = npk;
            W[e] = w;
            roles[e] = nwin == 2 ? ((emitted & 1) ? 2 : 1) : 0;   // filter_next_loci consumes (L,R) entries pairwise (MP:2394-2403)
            e++; slot += npk; emitted++;
        };
        if (has_p != has_m) {
            for (int idx = 0; idx < nwin; idx++) emit(idx, has_p ? 0 : 1, -1);
        } else {
            for (int idx = 0; idx < nwin; idx++)
                for (int s = 0; s < 2; s++)
                    for (int j = 0; j <= idx; j++) emit(j, s, s);
        }
    }
}

// ------------------------------------------------------------------------------------------
// a4-a6: window payload.  One wavefront per window: sequence gather (+ reverse complement),
// per-position most-abundant-read table in LDS, candidate matures.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ char rc_char(char c) {   // get_complement MP:232-235: upper-case ATGCU only
    switch (c) { case 'A': return 'U'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; case 'U': return 'A'; default: return c; }
}

// first_rec[w] = index of the first (tid, pos)-sorted record at or after the start of window w: one thread per window, a plain binary search --
// 2 x 10^5 independent searches in flight hide the chain of dependent loads that a search inside the payload kernel (one wave per window) paid in full.
__global__ void __launch_bounds__(256) window_first_record_kernel(const MirpWindow* __restrict__ W, long long n_windows, const MirpAln* __restrict__ alns,
                                                                  long long n_alns, long long* __restrict__ first_rec) {
    for (long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x; w < n_windows; w += (long long)gridDim.x * blockDim.x) {
        const int tid = W[w].tid, ws = W[w].ws;
        long long lo = 0, hi = n_alns;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            const int rt = alns[mid].tid, rp = alns[mid].pos;
            if (rt < tid || (rt == tid && rp < ws)) lo = mid + 1; else hi = mid;
        }
        first_rec[w] = lo;
    }
}

__device__ __forceinline__ int wave_max_i32(int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(x, o); x = t > x ? t : x; }
    return x;
}

__global__ void __launch_bounds__(64) window_payload_kernel(MirpWindow* __restrict__ W, long long n_windows, const MirpPeak* __restrict__ P,
                                                            const long long* __restrict__ first_rec,
                                                            const MirpAln* __restrict__ alns, long long n_alns, const unsigned char* __restrict__ genome,
                                                            const long long* __restrict__ gboff, const long long* __restrict__ clen,
                                                            double min_mature_depth, int wmax,
                                                            char* __restrict__ seqs, MirpMature* __restrict__ matures, int* __restrict__ rt_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long* best = (unsigned long long*)smem;        // [wmax]  (depth << 32 | ~index) of the most abundant read of the window strand that starts at pos
    int* dmax = (int*)(best + wmax);              // [wmax]  its depth, 0 = none
    int* tot = dmax + wmax;                       // [wmax]  total depth of the reads that start at pos (inspection copy only)
    unsigned short* lmax = (unsigned short*)(tot + wmax);   // [wmax]
    const int lane = threadIdx.x;
    // (fetching the next window's record, index and first 64 records while this one is worked on made the kernel slower: 0.47 -> 0.53 ms at a config[4] shard)
    for (long long w = blockIdx.x; w < n_windows; w += gridDim.x) {
        MirpWindow win = W[w];
        const int tid = win.tid, ws = win.ws, we = win.we, strand = win.strand;
        // ---- a4: samtools faidx chr:ws-(we-1): start 0 -> 1, end clamped to the contig (MP:1098-1105)
        long long s = ws < 1 ? 1 : ws, e = (long long)we - 1;
        if (e > clen[tid]) e = clen[tid];
        int len = e >= s ? (int)(e - s + 1) : 0;
        const unsigned char* g = genome + gboff[tid];
        char* dst = seqs + win.seq_off;
        if (strand == 0) for (int x = lane; x < len; x += 64) dst[x] = (char)g[s - 1 + x];
        else for (int x = lane; x < len; x += 64) dst[x] = rc_char((char)g[e - 1 - x]);
        // ---- a5: per start position, most abundant read of the window's strand; first seen wins ties (MP:1457).  The window's first record in
        // the (tid, pos)-sorted array comes from window_first_record_kernel; the records of [ws, we] are walked 64 at a time and a record votes for
        // its start position with the key (depth << 32 | ~index): the maximum is the deepest read and, among equals, the first in array order.
        // (Round 3 ran one binary search over ALL records per start position: 326 x 25 dependent loads per window, 3.6 ms at a config[4] shard.)
        const int width = we - ws + 1;
        for (int x = lane; x < width; x += 64) { best[x] = 0ull; tot[x] = 0; }
        const long long lo = first_rec[w];
        __syncthreads();
        for (long long k0 = lo;; k0 += 64) {
            const long long k = k0 + lane;
            bool in = false;
            if (k < n_alns) {
                const MirpAln r = alns[k];
                in = r.tid == tid && r.pos <= we;
                if (in && (int)r.strand == strand) {
                    const int x = r.pos - ws;
                    atomicMax(&best[x], ((unsigned long long)r.depth << 32) | (unsigned long long)(0xffffffffu - (unsigned)(k - lo)));
                    if (rt_out) atomicAdd(&tot[x], (int)r.depth);
                }
            }
            if (!(__ballot(in) >> 63)) break;               // the last lane is past the window (or the array): done
        }
        __syncthreads();
        for (int x = lane; x < width; x += 64) {
            const unsigned long long v = best[x];
            const int best_d = (int)(v >> 32);
            int best_l = 0;
            if (best_d > 0) best_l = alns[lo + (long long)(0xffffffffu - (unsigned)v)].len;
            dmax[x] = best_d; lmax[x] = (unsigned short)best_l;
            if (rt_out) {   // inspection copy of the a5 table (mirp_get_window_readtable): [len of max, depth of max, total depth] per start position
                int* o = rt_out + ((size_t)w * wmax + x) * 3;
                o[0] = best_l; o[1] = best_d; o[2] = tot[x];
            }
        }
        __syncthreads();
        // ---- a6: gen_matures_one_peak per same-strand peak of the region (MP:1472-1510).  The reference walks the positions of [peak start - 20,
        // peak end) in order: the first two with a read deeper than the bound are kept, the second is replaced by any later one that is strictly deeper
        // (so it ends as the FIRST deepest position behind the first kept one); with none above the bound the first deepest position of all stands
        // in.  Here a lane takes a position, 64 at a time, and the order-dependent picks are first-set-lane / first-lane-at-the-maximum.
        {
            int nm = 0;
            const long long f = win.pad0;
            const int npk = win.pad1;
            MirpMature* out = matures + win.mature_off;
            for (int k = 0; k < npk; k++) {
                const MirpPeak pk = P[f + k];
                if (pk.strand != strand) continue;
                int n = 0, c_pos[2] = {0, 0}, c_d[2] = {0, 0}, c_l[2] = {0, 0};
                int hi_pos = 0, hi_d = 0, hi_l = 0;
                const int p0 = pk.start - 20 > ws ? pk.start - 20 : ws, p1 = pk.end - 1 < we ? pk.end - 1 : we;      // inclusive
                for (int b = p0; b <= p1; b += 64) {
                    const int pos = b + lane;
                    const int d = pos <= p1 ? dmax[pos - ws] : 0;
                    const int l = pos <= p1 ? (int)lmax[pos - ws] : 0;
                    const int dm = wave_max_i32(d);
                    if (dm > hi_d) {
                        const int fl = __ffsll((long long)__ballot(d == dm)) - 1;
                        hi_pos = b + fl; hi_d = dm; hi_l = __shfl(l, fl);
                    }
                    unsigned long long q = __ballot(d > 0 && (double)d > min_mature_depth);
                    while (n < 2 && q) {
                        const int fl = __ffsll((long long)q) - 1;
                        c_pos[n] = b + fl; c_d[n] = __shfl(d, fl); c_l[n] = __shfl(l, fl);
                        n++;
                        q &= q - 1;
                    }
                    if (q) {          // n == 2: a later position replaces the second pick only if strictly deeper
                        const bool mine = (q >> lane) & 1ull;
                        const int m = wave_max_i32(mine ? d : 0);
                        if (m > c_d[1]) {
                            const int fl = __ffsll((long long)__ballot(mine && d == m)) - 1;
                            c_pos[1] = b + fl; c_d[1] = m; c_l[1] = __shfl(l, fl);
                        }
                    }
                }
                if (lane == 0) {
                    if (n == 0) {
                        MirpMature m; m.start = hi_d > 0 ? hi_pos : 0; m.end = hi_d > 0 ? hi_pos + hi_l : 0; m.strand = hi_d > 0 ? strand : -1; m.depth = hi_d;
                        out[nm] = m;
                    }
                    for (int x = 0; x < n; x++) { MirpMature m; m.start = c_pos[x]; m.end = c_pos[x] + c_l[x]; m.strand = strand; m.depth = c_d[x]; out[nm + x] = m; }
                }
                nm += n == 0 ? 1 : n;
            }
            if (lane == 0) {
                win.n_matures = nm; win.seq_len = len;
                if (!rt_out) { win.pad0 = 0; win.pad1 = 0; W[w] = win; }   // inspection mode re-runs on finished windows: leave them untouched
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------
static inline int grid_for(long long n, int block, int cap) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

void launch_cov_scatter(hipStream_t st, const MirpAln* alns, long long n, const long long* goff, const long long* clen, int cutoff, int* diff_p, int* diff_m) {
    if (n <= 0) return;
    hipLaunchKernelGGL(cov_scatter_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, alns, n, goff, clen, cutoff, diff_p, diff_m);
}
void launch_cov_unscatter(hipStream_t st, const MirpAln* alns, long long n, const long long* goff, const long long* clen, int* diff_p, int* diff_m) {
    if (n <= 0) return;
    hipLaunchKernelGGL(cov_unscatter_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, alns, n, goff, clen, diff_p, diff_m);
}
long long cov_scan_tiles(long long gtot) { return (gtot + SCAN_TILE - 1) / SCAN_TILE; }
void launch_cov_scan(hipStream_t st, const int* diff_p, const int* diff_m, long long gtot, int cutoff, unsigned long long* stat_d,
                     unsigned long long* stat_c, unsigned int* ticket, void* starts, long long starts_cap, MirpDepthPos* depth_out,
                     long long depth_cap, long long* depth_gx, unsigned long long* totals) {
    long long tiles = cov_scan_tiles(gtot);
    hipLaunchKernelGGL(cov_scan_kernel<false>, dim3((unsigned)tiles), dim3(SCAN_NT), 0, st, const_cast<int*>(diff_p), const_cast<int*>(diff_m), gtot, nullptr, nullptr,
                       nullptr, 0, nullptr, nullptr, nullptr, nullptr, cutoff, stat_d, stat_c, ticket, (RunStart*)starts, starts_cap, depth_out, depth_cap, depth_gx, totals);
}
int cov_scan_tile_positions() { return SCAN_TILE; }
void launch_cov_maxlen(hipStream_t st, const MirpAln* alns, long long n, int* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(cov_maxlen_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, alns, n, out);
}
size_t cov_fused_aux_bytes(long long gtot) { const size_t t = (size_t)cov_scan_tiles(gtot); return 8 * (t + 2) + 4 * 2 * t + 16 + 8 * 3 * (t + 2); }
hipError_t launch_cov_scan_fused(hipStream_t st, const MirpAln* alns, long long n, int max_len, const long long* goff, const long long* clen, int n_contigs, void* aux, int* diff_p,
                                 int* diff_m, long long gtot, int cutoff, unsigned long long* stat_d, unsigned long long* stat_c, unsigned int* ticket, void* starts,
                                 long long starts_cap, MirpDepthPos* depth_out, long long depth_cap, long long* depth_gx, unsigned long long* totals) {
    const long long tiles = cov_scan_tiles(gtot);
    const size_t lds = 2 * (size_t)SCAN_TILE * sizeof(int) + 2 * COV_CT * sizeof(long long);
    hipError_t e = hipFuncSetAttribute((const void*)cov_scan_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    // aux: first[tiles + 2] | agg[2][tiles] (+ pad) | carry_p[tiles + 2] | carry_m[tiles + 2] | back[tiles + 2]
    long long* first = (long long*)aux;
    int* agg = (int*)(first + tiles + 2);
    long long* carry_p = (long long*)(((uintptr_t)(agg + 2 * tiles) + 15) & ~(uintptr_t)15);
    long long* carry_m = carry_p + tiles + 2;
    long long* back = carry_m + tiles + 2;
    e = hipMemsetAsync(agg, 0, 4 * 2 * (size_t)tiles, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cov_tile_first_kernel, dim3(grid_for(n + 1, 256, 8192)), dim3(256), 0, st, alns, n, goff, clen, tiles, cutoff, max_len, first, back, agg);
    launch_excl_scan(st, agg, carry_p, tiles);
    launch_excl_scan(st, agg + tiles, carry_m, tiles);
    hipLaunchKernelGGL(cov_scan_kernel<true>, dim3((unsigned)tiles), dim3(SCAN_NT), lds, st, diff_p, diff_m, gtot, alns, goff, clen, n_contigs, first, back, carry_p, carry_m, cutoff, stat_d, stat_c, ticket,
                       (RunStart*)starts, starts_cap, depth_out, depth_cap, depth_gx, totals);
    return hipGetLastError();
}
size_t run_start_bytes() { return sizeof(RunStart); }
void launch_run_walk(hipStream_t st, const void* starts, long long n_runs, const int* diff_p, const int* diff_m, long long gtot, int cutoff,
                     const long long* goff, int n_contigs, int min_len, MirpPeak* runs, int* keep, int first_run_double) {
    if (n_runs <= 0) return;
    hipLaunchKernelGGL(run_walk_kernel, dim3(grid_for(n_runs, 256, 8192)), dim3(256), 0, st, (const RunStart*)starts, n_runs, diff_p, diff_m, gtot,
                       cutoff, goff, n_contigs, min_len, runs, keep, first_run_double);
}
void launch_depth_fix(hipStream_t st, MirpDepthPos* d, const long long* gx, long long n, const long long* goff, int n_contigs) {
    if (n <= 0) return;
    hipLaunchKernelGGL(depth_fix_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, d, gx, n, goff, n_contigs);
}
// Scratch of the many-workgroup scan, one per stream (calls on a stream are ordered, so its descriptors and ticket counter have one user at a time);
// released by release_scan_scratch when the stream's owner goes away.
namespace {
struct ScanScratch { unsigned long long* dev = nullptr; };
std::mutex g_scan_mu;
std::map<hipStream_t, ScanScratch> g_scan;
}  // namespace
void release_scan_scratch(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_scan_mu);
    auto it = g_scan.find(st);
    if (it == g_scan.end()) return;
    if (it->second.dev) (void)hipFree(it->second.dev);
    g_scan.erase(it);
}
void launch_excl_scan(hipStream_t st, const int* in, long long* out, long long n) {
    const long long blocks = (n + MSCAN_TILE - 1) / MSCAN_TILE;
    if (n > 1024 * SCAN1_NT && blocks <= MSCAN_MAX_BLOCKS) {
        std::lock_guard<std::mutex> lk(g_scan_mu);
        ScanScratch& s = g_scan[st];
        const size_t bytes = 8 * (size_t)(MSCAN_MAX_BLOCKS + 1);
        if (!s.dev && hipMalloc((void**)&s.dev, bytes) != hipSuccess) { s.dev = nullptr; (void)hipGetLastError(); }
        if (s.dev && hipMemsetAsync(s.dev, 0, 8 * (size_t)(blocks + 1), st) == hipSuccess) {
            hipLaunchKernelGGL(excl_scan_i32_mb_kernel, dim3((unsigned)blocks), dim3(MSCAN_NT), 0, st, in, out, n, s.dev);
            return;
        }
    }
    hipLaunchKernelGGL(excl_scan_i32_kernel, dim3(1), dim3(1024), 0, st, in, out, n);
}
void launch_peak_compact(hipStream_t st, const MirpPeak* runs, const int* keep, const long long* kscan, long long n_runs, int n_contigs,
                         const int* order, long long* csq, long long* cdest, MirpPeak* peaks_sq, MirpPeak* peaks_sorted) {
    hipLaunchKernelGGL(contig_peak_ranges_kernel, dim3(grid_for(n_contigs + 1, 256, 1024)), dim3(256), 0, st, runs, n_runs, kscan, n_contigs, csq);
    hipLaunchKernelGGL(contig_dest_kernel, dim3(1), dim3(1), 0, st, csq, order, n_contigs, cdest);
    if (n_runs > 0)
        hipLaunchKernelGGL(peak_compact_kernel, dim3(grid_for(n_runs, 256, 8192)), dim3(256), 0, st, runs, keep, kscan, n_runs, csq, cdest, peaks_sq, peaks_sorted);
}
void launch_region_head(hipStream_t st, const MirpPeak* P, long long n, int max_gap, int* head) {
    if (n <= 0) return;
    hipLaunchKernelGGL(region_head_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, P, n, max_gap, head);
}
void launch_region_first(hipStream_t st, const int* head, const long long* hscan, long long n, long long* rfirst) {
    hipLaunchKernelGGL(region_first_kernel, dim3(grid_for(n + 1, 256, 8192)), dim3(256), 0, st, head, hscan, n, rfirst);
}
void launch_region_count(hipStream_t st, const MirpPeak* P, const long long* rfirst, long long n_regions, const long long* clen, int L,
                         int* n_entries, int* is_locus, int* n_slots) {
    if (n_regions <= 0) return;
    hipLaunchKernelGGL(region_count_kernel, dim3(grid_for(n_regions, 256, 8192)), dim3(256), 0, st, P, rfirst, n_regions, clen, L, n_entries, is_locus, n_slots);
}
void launch_region_emit(hipStream_t st, const MirpPeak* P, const long long* rfirst, long long n_regions, const long long* clen, int L,
                        const long long* escan, const long long* lscan, const long long* sscan, MirpWindow* W, MirpLocus* loci, MirpPeak* wpeaks,
                        int* roles, int seq_stride) {
    if (n_regions <= 0) return;
    hipLaunchKernelGGL(region_emit_kernel, dim3(grid_for(n_regions, 256, 8192)), dim3(256), 0, st, P, rfirst, n_regions, clen, L, escan, lscan, sscan, W,
                       loci, wpeaks, roles, seq_stride);
}
void launch_window_payload(hipStream_t st, MirpWindow* W, long long n_windows, const MirpPeak* P, const MirpAln* alns, long long n_alns,
                           const unsigned char* genome, const long long* gboff, const long long* clen, double min_mature_depth, int wmax, char* seqs,
                           MirpMature* matures, long long* first_rec, int* rt_out) {
    if (n_windows <= 0) return;
    size_t lds = (size_t)wmax * 18 + 32;
    hipLaunchKernelGGL(window_first_record_kernel, dim3(grid_for(n_windows, 256, 8192)), dim3(256), 0, st, W, n_windows, alns, n_alns, first_rec);
    hipLaunchKernelGGL(window_payload_kernel, dim3(grid_for(n_windows, 1, 16384)), dim3(64), lds, st, W, n_windows, P, first_rec, alns, n_alns, genome, gboff, clen,
                       min_mature_depth, wmax, seqs, matures, rt_out);
}

}  // namespace mirp
