// Device-resident candidate -> fold -> predict pipeline behind the C-ABI (include/mirprefer.h).
// Host orchestration only; every stage leaves its results in HBM for the next one.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "mirp_ctx.h"

namespace mirp {
// small kernels local to the pipeline (window index for the folder, result selection / gather)
__global__ void window_index_kernel(const MirpWindow* __restrict__ W, long long n, long long* __restrict__ offs, int* __restrict__ lens) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        offs[k] = W[k].seq_off;
        lens[k] = W[k].seq_len;
    }
}
// filter_next_loci pairing (MP:2373-2432): role 0 = single ('0' entry), 1 = first of an (L,R) pair, 2 = second (evaluated only if the first failed)
__global__ void result_keep_kernel(const int* __restrict__ roles, const int* __restrict__ n_out, long long n, int* __restrict__ keep) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        int r = roles[k];
        keep[k] = (n_out[k] > 0 && (r != 2 || n_out[k - 1] == 0)) ? 1 : 0;
    }
}
__global__ void result_gather_kernel(const int* __restrict__ keep, const long long* __restrict__ kscan, long long n, const MirpMirna* __restrict__ out,
                                     const char* __restrict__ ss, int ss_stride, int max_lines, const int* __restrict__ side_idx,
                                     const char* __restrict__ ss2, int max_lines2, MirpMirna* __restrict__ res, char* __restrict__ text) {
    // one wavefront per window
    const int lane = threadIdx.x & 63;
    for (long long w = blockIdx.x * (long long)(blockDim.x / 64) + (threadIdx.x >> 6); w < n; w += (long long)gridDim.x * (blockDim.x / 64)) {
        if (!keep[w]) continue;
        MirpMirna m = out[w * MIRP_MAX_MIRNA_PER_WINDOW];
        long long i = kscan[w];
        if (lane == 0) res[i] = m;
        const int sk = side_idx ? side_idx[w] : -1;   // re-folded at full capacity: its lines live in the side buffer
        const char* src = sk >= 0 ? ss2 + ((size_t)sk * max_lines2 + m.line) * ss_stride + m.ss_off
                                  : ss + ((size_t)w * max_lines + m.line) * ss_stride + m.ss_off;
        char* dst = text + (size_t)i * ss_stride;
        for (int x = lane; x < ss_stride; x += 64) dst[x] = (x < m.ss_len) ? src[x] : (char)0;
    }
}
// Windows whose structure lines did not fit the default capacity (status 1): compact list + per-window slot (unordered; the slot map is what counts)
__global__ void side_compact_kernel(const int* __restrict__ status, long long n, int* __restrict__ side_idx, int* __restrict__ side_list,
                                    unsigned int* __restrict__ count) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        int slot = -1;
        if (status[k] == 1) { slot = (int)atomicAdd(count, 1u); side_list[slot] = (int)k; }
        side_idx[k] = slot;
    }
}
__global__ void side_gather_kernel(const int* __restrict__ side_list, int n_side, const long long* __restrict__ offs, const int* __restrict__ lens,
                                   long long* __restrict__ soffs, int* __restrict__ slens) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_side; k += gridDim.x * blockDim.x) {
        soffs[k] = offs[side_list[k]];
        slens[k] = lens[side_list[k]];
    }
}
// the per-window summary of a re-folded window is the one of its full-capacity run
__global__ void side_scatter_kernel(const int* __restrict__ side_list, int n_side, const int* __restrict__ nl2, const int* __restrict__ mfe2,
                                    const int* __restrict__ st2, int* __restrict__ nl, int* __restrict__ mfe, int* __restrict__ st) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_side; k += gridDim.x * blockDim.x) {
        const int w = side_list[k];
        nl[w] = nl2[k]; mfe[w] = mfe2[k]; st[w] = st2[k];
    }
}
}  // namespace mirp

template <class T>
static T* host_copy(mirp_ctx* c, const void* dev, size_t n) {
    T* h = (T*)std::calloc(std::max<size_t>(n, 1), sizeof(T));
    if (!h) return nullptr;
    if (n && hipMemcpy(h, dev, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) { std::free(h); return nullptr; }
    return h;
}

static int read_ll(mirp_ctx* c, const void* dev, long long* out) {
    if (hipMemcpyAsync(out, dev, sizeof(long long), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    return 0;
}

extern "C" int mirp_load_genome(mirp_ctx* c, int32_t n_contigs, const int64_t* contig_len, const uint8_t* seq_concat) {
    if (!c) return -1;
    if (n_contigs < 1 || !contig_len || !seq_concat) return fail(c, -1, "mirp_load_genome: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    c->n_contigs = n_contigs;
    c->h_clen.assign(contig_len, contig_len + n_contigs);
    c->h_goff.resize(n_contigs + 1); c->h_gboff.resize(n_contigs + 1);
    long long g = 0, b = 0;
    for (int t = 0; t < n_contigs; t++) {
        if (contig_len[t] < 0 || contig_len[t] > 0x7ffffff0LL) return fail(c, -1, "mirp_load_genome: contig length out of range");
        c->h_goff[t] = g; c->h_gboff[t] = b; g += contig_len[t] + 1; b += contig_len[t];
    }
    c->h_goff[n_contigs] = g; c->h_gboff[n_contigs] = b;
    c->gtot = g; c->gbytes = b;
    if (c->genome.ensure((size_t)b + 16) || c->clen.ensure(8 * (size_t)n_contigs) || c->goff.ensure(8 * (size_t)(n_contigs + 1)) ||
        c->gboff.ensure(8 * (size_t)(n_contigs + 1)))
        return fail(c, -6, "device allocation failed (genome)");
    HIPCHK(c, hipMemcpy(c->genome.p, seq_concat, (size_t)b, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->clen.p, c->h_clen.data(), 8 * (size_t)n_contigs, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->goff.p, c->h_goff.data(), 8 * (size_t)(n_contigs + 1), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->gboff.p, c->h_gboff.data(), 8 * (size_t)(n_contigs + 1), hipMemcpyHostToDevice));
    c->have_candidate = c->have_fold = c->have_result = false;
    return 0;
}

extern "C" int mirp_load_alignments(mirp_ctx* c, const MirpAln* alns, int64_t n) {
    if (!c) return -1;
    if (n < 0 || (n > 0 && !alns)) return fail(c, -1, "mirp_load_alignments: bad argument");
    if (c->n_contigs == 0) return fail(c, -1, "mirp_load_alignments: load the genome first");
    HIPCHK(c, hipSetDevice(c->device));
    for (int64_t k = 0; k < n; k++) {
        if (alns[k].tid < 0 || alns[k].tid >= c->n_contigs) return fail(c, -1, "mirp_load_alignments: tid out of range");
        if (k > 0 && (alns[k].tid < alns[k - 1].tid || (alns[k].tid == alns[k - 1].tid && alns[k].pos < alns[k - 1].pos)))
            return fail(c, -1, "mirp_load_alignments: records are not sorted by (tid, pos)");
        if (alns[k].sample >= MIRP_MAX_SAMPLES) return fail(c, -1, "mirp_load_alignments: sample index out of range");
    }
    if (c->alns.ensure(sizeof(MirpAln) * (size_t)std::max<int64_t>(n, 1))) return fail(c, -6, "device allocation failed (alignments)");
    if (n) HIPCHK(c, hipMemcpy(c->alns.p, alns, sizeof(MirpAln) * (size_t)n, hipMemcpyHostToDevice));
    c->n_alns = n; c->n_segs = 0; c->ingest_resident = false; c->max_aln_len = -1;
    c->have_candidate = c->have_fold = c->have_result = false;
    return 0;
}

extern "C" int mirp_load_coverage_segments(mirp_ctx* c, const MirpAln* segs, int64_t n) {
    if (!c) return -1;
    if (n < 0 || (n > 0 && !segs)) return fail(c, -1, "mirp_load_coverage_segments: bad argument");
    if (c->n_contigs == 0) return fail(c, -1, "mirp_load_coverage_segments: load the genome first");
    HIPCHK(c, hipSetDevice(c->device));
    for (int64_t k = 0; k < n; k++)
        if (segs[k].tid < 0 || segs[k].tid >= c->n_contigs) return fail(c, -1, "mirp_load_coverage_segments: tid out of range");
    if (c->segs.ensure(sizeof(MirpAln) * (size_t)std::max<int64_t>(n, 1))) return fail(c, -6, "device allocation failed (segments)");
    if (n) HIPCHK(c, hipMemcpy(c->segs.p, segs, sizeof(MirpAln) * (size_t)n, hipMemcpyHostToDevice));
    c->n_segs = n;
    c->have_candidate = c->have_fold = c->have_result = false;
    return 0;
}

// coverage: memset + scatter + single-pass scan. depth_out optional (second run for mirp_get_depth).
static int run_coverage(mirp_ctx* c, MirpDepthPos* depth_out, long long depth_cap, long long* depth_gx) {
    const long long gtot = c->gtot;
    const long long tiles = mirp::cov_scan_tiles(gtot);
    if (c->diff.ensure(8 * (size_t)(gtot + 8)) || c->stat.ensure(16 * (size_t)tiles + 64) ||
        c->starts.ensure(mirp::run_start_bytes() * (size_t)std::max<long long>(c->n_alns + c->n_segs, 1)) || c->totals.ensure(64))
        return fail(c, -6, "device allocation failed (coverage)");
    int* diff_p = (int*)c->diff.p;
    int* diff_m = diff_p + ((gtot + 3) / 4) * 4;   // keep 16-B alignment of both arrays
    unsigned long long* stat_d = (unsigned long long*)c->stat.p;
    unsigned long long* stat_c = stat_d + tiles;
    unsigned int* ticket = (unsigned int*)(stat_c + tiles);
    HIPCHK(c, hipMemsetAsync(c->stat.p, 0, 16 * (size_t)tiles + 64, c->stream));
    HIPCHK(c, hipMemsetAsync(c->totals.p, 0, 64, c->stream));
    // Fused path (no coverage segments, no record longer than a scan tile): the scan builds every tile's difference values from the sorted records
    // in LDS and writes the dense arrays only where a run walk will read them -- no global atomics, no clearing pass.
    // Measured (profiles/tools/cov_time.py): cfg[4] shard, 0.1 records per base: 1.27 ms fused against 2.11 ms + the clearing pass; config[1],
    // 0.002 records per base: 0.137 against 0.123 ms (two more launches, nothing to gain from 138 k atomics) -- so the density picks the path.
    // mirp_set_coverage_path forces it (tests run both on the same input).
    c->cov_fused = false;
    const bool want = c->cov_mode >= 0 ? c->cov_mode == 1 : c->n_alns >= c->gtot / 32;
    if (c->n_segs == 0 && c->n_alns > 0 && want) {
        if (c->max_aln_len < 0) {
            int* d_max = (int*)((char*)c->totals.p + 32);          // (cleared above)
            mirp::launch_cov_maxlen(c->stream, (const MirpAln*)c->alns.p, c->n_alns, d_max);
            int m = 0;
            HIPCHK(c, hipMemcpyAsync(&m, d_max, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->max_aln_len = m;
        }
        c->cov_fused = c->max_aln_len <= mirp::cov_scan_tile_positions();
    }
    if (c->cov_fused) {
        if (c->tile_first.ensure(mirp::cov_fused_aux_bytes(gtot))) return fail(c, -6, "device allocation failed (coverage)");
        c->diff_clean_ptr = nullptr;          // the dense arrays hold the written tiles' values from now on: the atomic path clears them before it runs again
        HIPCHK(c, mirp::launch_cov_scan_fused(c->stream, (const MirpAln*)c->alns.p, c->n_alns, c->max_aln_len, (const long long*)c->goff.p, (const long long*)c->clen.p,
                                              c->n_contigs, c->tile_first.p, diff_p, diff_m, gtot, c->cand.cutoff, stat_d, stat_c, ticket, c->starts.p,
                                              std::max<long long>(c->n_alns + c->n_segs, 1), depth_out, depth_cap, depth_gx, (unsigned long long*)c->totals.p));
        return 0;
    }
    // the difference arrays are all zero between passes (clean_coverage clears what a pass wrote); a full clear only for a new buffer or after
    // a pass that did not get to clean up
    if (c->diff_clean_ptr != c->diff.p || c->diff_clean_bytes < 8 * (size_t)(gtot + 8)) {
        HIPCHK(c, hipMemsetAsync(c->diff.p, 0, c->diff.cap, c->stream));
        c->diff_clean_bytes = c->diff.cap;
    }
    c->diff_clean_ptr = nullptr;          // dirty from here until clean_coverage
    mirp::launch_cov_scatter(c->stream, (const MirpAln*)c->alns.p, c->n_alns, (const long long*)c->goff.p, (const long long*)c->clen.p, c->cand.cutoff,
                             diff_p, diff_m);
    if (c->n_segs > 0)       // gapped alignments: their own [pos, pos + len(SEQ)) taken back out, their M / = / X blocks added
        mirp::launch_cov_scatter(c->stream, (const MirpAln*)c->segs.p, c->n_segs, (const long long*)c->goff.p, (const long long*)c->clen.p, c->cand.cutoff,
                                 diff_p, diff_m);
    mirp::launch_cov_scan(c->stream, diff_p, diff_m, gtot, c->cand.cutoff, stat_d, stat_c, ticket, c->starts.p, std::max<long long>(c->n_alns + c->n_segs, 1),
                          depth_out, depth_cap, depth_gx, (unsigned long long*)c->totals.p);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// clears the positions the last run_coverage wrote (two per record / segment): the arrays are all zero again
static int clean_coverage(mirp_ctx* c) {
    if (c->cov_fused) return 0;          // the fused scan wrote whole tiles with plain stores and relies on no invariant of the arrays
    int* diff_p = (int*)c->diff.p;
    int* diff_m = diff_p + ((c->gtot + 3) / 4) * 4;
    mirp::launch_cov_unscatter(c->stream, (const MirpAln*)c->alns.p, c->n_alns, (const long long*)c->goff.p, (const long long*)c->clen.p, diff_p, diff_m);
    if (c->n_segs > 0)
        mirp::launch_cov_unscatter(c->stream, (const MirpAln*)c->segs.p, c->n_segs, (const long long*)c->goff.p, (const long long*)c->clen.p, diff_p, diff_m);
    HIPCHK(c, hipGetLastError());
    c->diff_clean_ptr = c->diff.p;
    return 0;
}

extern "C" int mirp_set_contig_shard(mirp_ctx* c, int32_t preceded_by_coverage_elsewhere) {
    if (!c) return -1;
    c->shard_first_run_double = preceded_by_coverage_elsewhere ? 1 : 0;
    c->have_candidate = false; c->have_fold = false; c->have_result = false;
    return 0;
}

extern "C" int mirp_select_windows(mirp_ctx* c, int64_t first, int64_t count) {
    if (!c) return -1;
    if (!c->have_candidate) return fail(c, -1, "mirp_select_windows: run mirp_candidate first");
    const long long total = c->sel_total >= 0 ? c->sel_total : c->n_windows;
    if (count < 0) {          // back to the whole list
        c->win_first = 0; c->n_windows = total; c->sel_total = -1;
    } else {
        if (first < 0 || first + count > total) return fail(c, -1, "mirp_select_windows: range outside the window list");
        c->sel_total = total; c->win_first = first; c->n_windows = count;
    }
    c->have_fold = false; c->have_result = false;
    return 0;
}

extern "C" int mirp_limit_windows(mirp_ctx* c, int64_t n_keep) {
    if (!c) return -1;
    if (!c->have_candidate) return fail(c, -1, "mirp_limit_windows: run mirp_candidate first");
    if (n_keep < 0 || n_keep > c->n_windows) return fail(c, -1, "mirp_limit_windows: n_keep out of range");
    // every later stage reads the window arrays as prefixes of length n_windows (offsets into wpeaks / matures / sequences stay valid)
    if (c->sel_total >= 0) return fail(c, -1, "mirp_limit_windows: a window view is active (mirp_select_windows)");
    c->n_windows = n_keep;
    c->have_fold = false; c->have_result = false;
    return 0;
}

extern "C" int mirp_excl_scan_i32(mirp_ctx* c, const int32_t* in, int64_t n, int64_t* out) {
    if (!c) return -1;
    if (n < 0 || !out || (n > 0 && !in)) return fail(c, -1, "mirp_excl_scan_i32: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    TmpDevice T;
    int* d_in = (int*)T.get(4 * (size_t)std::max<int64_t>(n, 1));
    long long* d_out = (long long*)T.get(8 * (size_t)(n + 1));
    if (!d_in || !d_out) return fail(c, -6, "device allocation failed");
    if (n > 0) HIPCHK(c, hipMemcpyAsync(d_in, in, 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    mirp::launch_excl_scan(c->stream, d_in, d_out, n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, d_out, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mirp_candidate(mirp_ctx* c, const MirpCandidateParams* params, const int32_t* contig_order, int64_t* n_peaks_out,
                              int64_t* n_loci_out, int64_t* n_windows_out) {
    if (!c) return -1;
    if (!params || !contig_order) return fail(c, -1, "mirp_candidate: null argument");
    if (c->n_contigs == 0) return fail(c, -1, "mirp_candidate: no genome loaded");
    if (c->ingest_resident && c->ingest_n_contigs != c->n_contigs) return fail(c, -1, "mirp_candidate: the SAM header and the genome do not have the same contigs");
    if (params->precursor_len < 60 || params->precursor_len > 3000) return fail(c, -1, "Error: allowed precursor range: 60-3000");
    if (params->cutoff < 2) return fail(c, -1, "Error: READS_DEPTH_CUTOFF should >=2.");
    HIPCHK(c, hipSetDevice(c->device));
    c->cand = *params;
    c->have_candidate = c->have_fold = c->have_result = false;
    const int nc = c->n_contigs;
    if (c->order.ensure(4 * (size_t)nc)) return fail(c, -6, "device allocation failed");
    HIPCHK(c, hipMemcpyAsync(c->order.p, contig_order, 4 * (size_t)nc, hipMemcpyHostToDevice, c->stream));
    hipStream_t st = c->stream;
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    int rc = run_coverage(c, nullptr, 0, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    long long tot[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(tot, c->totals.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->n_runs = tot[0]; c->n_above = tot[1];
    if (c->n_runs > std::max<long long>(c->n_alns + c->n_segs, 1)) return fail(c, -7, "internal: run-start capacity exceeded");
    const long long nr = c->n_runs;
    const int* diff_p = (const int*)c->diff.p;
    const int* diff_m = diff_p + ((c->gtot + 3) / 4) * 4;
    if (c->runs.ensure(sizeof(MirpPeak) * (size_t)std::max<long long>(nr, 1)) || c->keep.ensure(4 * (size_t)std::max<long long>(nr, 1)) ||
        c->kscan.ensure(8 * (size_t)(nr + 1)) || c->csq.ensure(8 * (size_t)(nc + 2)) || c->cdest.ensure(8 * (size_t)(nc + 2)))
        return fail(c, -6, "device allocation failed (runs)");
    mirp::launch_run_walk(st, c->starts.p, nr, diff_p, diff_m, c->gtot, c->cand.cutoff, (const long long*)c->goff.p, nc, c->cand.min_peak_len,
                          (MirpPeak*)c->runs.p, (int*)c->keep.p, c->shard_first_run_double);
    HIPCHK(c, hipEventRecord(c->ev[4], st));
    if (int rc2 = clean_coverage(c)) return rc2;          // the run walk was the last reader of the difference arrays
    HIPCHK(c, hipEventRecord(c->ev[5], st));
    mirp::launch_excl_scan(st, (const int*)c->keep.p, (long long*)c->kscan.p, nr);
    long long np = 0;
    if (read_ll(c, (const long long*)c->kscan.p + nr, &np)) return fail(c, -2, "D2H failed");
    c->n_peaks = np;
    if (c->peaks_sq.ensure(sizeof(MirpPeak) * (size_t)std::max<long long>(np, 1)) || c->peaks_sorted.ensure(sizeof(MirpPeak) * (size_t)std::max<long long>(np, 1)))
        return fail(c, -6, "device allocation failed (peaks)");
    mirp::launch_peak_compact(st, (const MirpPeak*)c->runs.p, (const int*)c->keep.p, (const long long*)c->kscan.p, nr, nc, (const int*)c->order.p,
                              (long long*)c->csq.p, (long long*)c->cdest.p, (MirpPeak*)c->peaks_sq.p, (MirpPeak*)c->peaks_sorted.p);
    // ---- regions
    if (c->head.ensure(4 * (size_t)std::max<long long>(np, 1)) || c->hscan.ensure(8 * (size_t)(np + 1))) return fail(c, -6, "device allocation failed (regions)");
    const MirpPeak* P = (const MirpPeak*)c->peaks_sorted.p;
    mirp::launch_region_head(st, P, np, c->cand.max_gap, (int*)c->head.p);
    mirp::launch_excl_scan(st, (const int*)c->head.p, (long long*)c->hscan.p, np);
    long long nreg = 0;
    if (read_ll(c, (const long long*)c->hscan.p + np, &nreg)) return fail(c, -2, "D2H failed");
    c->n_regions = nreg;
    const size_t r1 = (size_t)std::max<long long>(nreg, 1);
    if (c->rfirst.ensure(8 * (r1 + 1)) || c->nent.ensure(4 * r1) || c->isloc.ensure(4 * r1) || c->nslots.ensure(4 * r1) || c->escan.ensure(8 * (r1 + 1)) ||
        c->lscan.ensure(8 * (r1 + 1)) || c->sscan.ensure(8 * (r1 + 1)))
        return fail(c, -6, "device allocation failed (regions)");
    mirp::launch_region_first(st, (const int*)c->head.p, (const long long*)c->hscan.p, np, (long long*)c->rfirst.p);
    mirp::launch_region_count(st, P, (const long long*)c->rfirst.p, nreg, (const long long*)c->clen.p, c->cand.precursor_len, (int*)c->nent.p,
                              (int*)c->isloc.p, (int*)c->nslots.p);
    mirp::launch_excl_scan(st, (const int*)c->nent.p, (long long*)c->escan.p, nreg);
    mirp::launch_excl_scan(st, (const int*)c->isloc.p, (long long*)c->lscan.p, nreg);
    mirp::launch_excl_scan(st, (const int*)c->nslots.p, (long long*)c->sscan.p, nreg);
    long long nw = 0, nl = 0, ns = 0;
    if (read_ll(c, (const long long*)c->escan.p + nreg, &nw) || read_ll(c, (const long long*)c->lscan.p + nreg, &nl) ||
        read_ll(c, (const long long*)c->sscan.p + nreg, &ns))
        return fail(c, -2, "D2H failed");
    c->n_windows = nw; c->n_loci = nl; c->n_slots = ns;
    c->seq_stride = ((c->cand.precursor_len + 50 + 2 + 15) / 16) * 16;
    const size_t w1 = (size_t)std::max<long long>(nw, 1);
    if (c->windows.ensure(sizeof(MirpWindow) * w1) || c->loci.ensure(sizeof(MirpLocus) * (size_t)std::max<long long>(nl, 1)) ||
        c->wpeaks.ensure(sizeof(MirpPeak) * (size_t)std::max<long long>(ns, 1)) || c->matures.ensure(sizeof(MirpMature) * 2 * (size_t)std::max<long long>(ns, 1)) ||
        c->wseqs.ensure(w1 * c->seq_stride + 16) || c->woffs.ensure(8 * (w1 + 1)) || c->wlens.ensure(4 * w1) || c->roles.ensure(4 * w1))
        return fail(c, -6, "device allocation failed (windows)");
    mirp::launch_region_emit(st, P, (const long long*)c->rfirst.p, nreg, (const long long*)c->clen.p, c->cand.precursor_len, (const long long*)c->escan.p,
                             (const long long*)c->lscan.p, (const long long*)c->sscan.p, (MirpWindow*)c->windows.p, (MirpLocus*)c->loci.p,
                             (MirpPeak*)c->wpeaks.p, (int*)c->roles.p, c->seq_stride);
    const int wmax = c->cand.precursor_len + 52;
    mirp::launch_window_payload(st, (MirpWindow*)c->windows.p, nw, P, (const MirpAln*)c->alns.p, c->n_alns, (const unsigned char*)c->genome.p,
                                (const long long*)c->gboff.p, (const long long*)c->clen.p, c->cand.cutoff * 0.5 /* MP:3404 */, wmax, (char*)c->wseqs.p,
                                (MirpMature*)c->matures.p, (long long*)c->woffs.p /* scratch here: window_index_kernel fills it below */);
    if (nw > 0)
        hipLaunchKernelGGL(mirp::window_index_kernel, dim3((unsigned)std::min<long long>((nw + 255) / 256, 4096)), dim3(256), 0, st,
                           (const MirpWindow*)c->windows.p, nw, (long long*)c->woffs.p, (int*)c->wlens.p);
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    float a = 0, b = 0;
    float cl = 0;
    (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&cl, c->ev[4], c->ev[5]);
    c->ms[0] = a + cl; c->ms[1] = b - cl;      // coverage = scatter + scan + clearing what the pass wrote
    c->have_candidate = true;
    if (n_peaks_out) *n_peaks_out = np;
    if (n_loci_out) *n_loci_out = nl;
    if (n_windows_out) *n_windows_out = nw;
    return 0;
}

extern "C" int mirp_get_depth(mirp_ctx* c, MirpDepthPos** depth, int64_t* n_depth) {
    if (!c) return -1;
    if (!depth || !n_depth) return fail(c, -1, "mirp_get_depth: null argument");
    if (!c->have_candidate) return fail(c, -1, "mirp_get_depth: run mirp_candidate first");
    HIPCHK(c, hipSetDevice(c->device));
    const long long na = c->n_above;
    TmpDevice T;
    MirpDepthPos* d = (MirpDepthPos*)T.get(sizeof(MirpDepthPos) * (size_t)std::max<long long>(na, 1));
    long long* gx = (long long*)T.get(8 * (size_t)std::max<long long>(na, 1));
    if (!d || !gx) return fail(c, -6, "device allocation failed (depth)");
    int rc = run_coverage(c, d, na, gx);
    if (rc) return rc;
    if (int rc2 = clean_coverage(c)) return rc2;
    mirp::launch_depth_fix(c->stream, d, gx, na, (const long long*)c->goff.p, c->n_contigs);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    MirpDepthPos* h = host_copy<MirpDepthPos>(c, d, (size_t)na);
    if (!h) return fail(c, -2, "D2H failed");
    *depth = h; *n_depth = na;
    return 0;
}

extern "C" int mirp_get_peaks(mirp_ctx* c, MirpPeak** peaks, int64_t* n) {
    if (!c) return -1;
    if (!peaks || !n) return fail(c, -1, "mirp_get_peaks: null argument");
    if (!c->have_candidate) return fail(c, -1, "mirp_get_peaks: run mirp_candidate first");
    HIPCHK(c, hipSetDevice(c->device));
    MirpPeak* h = host_copy<MirpPeak>(c, c->peaks_sq.p, (size_t)c->n_peaks);
    if (!h) return fail(c, -2, "D2H failed");
    *peaks = h; *n = c->n_peaks;
    return 0;
}

extern "C" int mirp_get_loci(mirp_ctx* c, MirpLocus** loci, int64_t* n_loci, MirpPeak** peaks_sorted, int64_t* n_peaks) {
    if (!c) return -1;
    if (!loci || !n_loci || !peaks_sorted || !n_peaks) return fail(c, -1, "mirp_get_loci: null argument");
    if (!c->have_candidate) return fail(c, -1, "mirp_get_loci: run mirp_candidate first");
    HIPCHK(c, hipSetDevice(c->device));
    MirpLocus* h = host_copy<MirpLocus>(c, c->loci.p, (size_t)c->n_loci);
    MirpPeak* hp = host_copy<MirpPeak>(c, c->peaks_sorted.p, (size_t)c->n_peaks);
    if (!h || !hp) { std::free(h); std::free(hp); return fail(c, -2, "D2H failed"); }
    *loci = h; *n_loci = c->n_loci; *peaks_sorted = hp; *n_peaks = c->n_peaks;
    return 0;
}

extern "C" int mirp_get_windows(mirp_ctx* c, MirpWindow** windows, int64_t* n_windows, MirpPeak** wpeaks, int64_t* n_wpeaks, MirpMature** matures,
                                int64_t* n_matures, char** seqs, int64_t* n_seq_bytes) {
    if (!c) return -1;
    if (!windows || !n_windows || !wpeaks || !n_wpeaks || !matures || !n_matures || !seqs || !n_seq_bytes) return fail(c, -1, "mirp_get_windows: null argument");
    if (!c->have_candidate) return fail(c, -1, "mirp_get_windows: run mirp_candidate first");
    HIPCHK(c, hipSetDevice(c->device));
    MirpWindow* hw = host_copy<MirpWindow>(c, c->windows.p, (size_t)c->n_windows);
    MirpPeak* hp = host_copy<MirpPeak>(c, c->wpeaks.p, (size_t)c->n_slots);
    MirpMature* hm = host_copy<MirpMature>(c, c->matures.p, 2 * (size_t)c->n_slots);
    char* hs = host_copy<char>(c, c->wseqs.p, (size_t)c->n_windows * c->seq_stride);
    if (!hw || !hp || !hm || !hs) { std::free(hw); std::free(hp); std::free(hm); std::free(hs); return fail(c, -2, "D2H failed"); }
    *windows = hw; *n_windows = c->n_windows; *wpeaks = hp; *n_wpeaks = c->n_slots; *matures = hm; *n_matures = 2 * c->n_slots;
    *seqs = hs; *n_seq_bytes = c->n_windows * c->seq_stride;
    return 0;
}

extern "C" int mirp_get_window_readtable(mirp_ctx* c, int32_t** table, int32_t* width, int64_t* n_windows) {
    if (!c) return -1;
    if (!table || !width || !n_windows) return fail(c, -1, "mirp_get_window_readtable: null argument");
    if (!c->have_candidate) return fail(c, -1, "mirp_get_window_readtable: run mirp_candidate first");
    HIPCHK(c, hipSetDevice(c->device));
    const int wmax = c->cand.precursor_len + 52;
    const long long nw = c->n_windows;
    TmpDevice T;
    int* d = (int*)T.get(sizeof(int) * 3 * (size_t)wmax * (size_t)std::max<long long>(nw, 1));
    if (!d) return fail(c, -6, "device allocation failed (read table)");
    HIPCHK(c, hipMemsetAsync(d, 0, sizeof(int) * 3 * (size_t)wmax * (size_t)std::max<long long>(nw, 1), c->stream));
    long long* first_rec = (long long*)T.get(8 * (size_t)std::max<long long>(nw, 1));
    if (!first_rec) return fail(c, -6, "device allocation failed (read table)");
    mirp::launch_window_payload(c->stream, (MirpWindow*)c->windows.p, nw, (const MirpPeak*)c->peaks_sorted.p, (const MirpAln*)c->alns.p, c->n_alns,
                                (const unsigned char*)c->genome.p, (const long long*)c->gboff.p, (const long long*)c->clen.p, c->cand.cutoff * 0.5, wmax,
                                (char*)c->wseqs.p, (MirpMature*)c->matures.p, first_rec, d);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    int32_t* h = host_copy<int32_t>(c, d, 3 * (size_t)wmax * (size_t)nw);
    if (!h) return fail(c, -2, "D2H failed");
    *table = h; *width = wmax; *n_windows = nw;
    return 0;
}

extern "C" int mirp_fold(mirp_ctx* c, int32_t span, int32_t max_lines) {
    if (!c) return -1;
    if (!c->have_candidate) return fail(c, -1, "mirp_fold: run mirp_candidate first");
    if (span < 1 || max_lines < 1) return fail(c, -1, "mirp_fold: bad span/max_lines");
    HIPCHK(c, hipSetDevice(c->device));
    const long long nw = c->n_windows;
    const int n_cap = c->seq_stride;   // >= longest window
    const int stride = ((n_cap + 3 + 7) / 8) * 8;
    const size_t w1 = (size_t)std::max<long long>(nw, 1);
    const size_t per_win = (size_t)max_lines * stride;
    if (c->lines.ensure(sizeof(MirpFoldLine) * w1 * max_lines) || c->ss.ensure(w1 * per_win) || c->nlines.ensure(4 * w1) || c->mfe.ensure(4 * w1) ||
        c->status.ensure(4 * w1))
        return fail(c, -6, "device allocation failed (fold)");
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    c->n_side = 0; c->side_max_lines = 0;
    long long fallbacks = 0;
    {
        int rc = mirp_run_fold(c, (const unsigned char*)c->wseqs.p, c->v_woffs(), c->v_wlens(), (int)nw, n_cap, span, max_lines,
                               stride, (MirpFoldLine*)c->lines.p, (char*)c->ss.p, (int*)c->nlines.p, (int*)c->mfe.p, (int*)c->status.p);
        if (rc) return rc;
        fallbacks = c->last_fallback;
    }
    const double main_kernel_ms[2] = {c->fold_kernel_ms[0], c->fold_kernel_ms[1]};
    // RNALfold has no limit on the number of structure lines (MP:3053); a window that produced more than max_lines (tandem repeats: up to
    // one line per start position) was flagged, not truncated.  Only those windows are folded again, at the capacity no window can
    // exceed, into side buffers that the predict stage and the text writers read instead of the window's slot in the main buffers.
    int big = n_cap + 2;
    if (nw > 0 && max_lines < big) {
        if (c->side_idx.ensure(4 * w1) || c->side_list.ensure(4 * w1) || c->side_cnt.ensure(64)) return fail(c, -6, "device allocation failed (fold overflow list)");
        unsigned int* cnt = (unsigned int*)c->side_cnt.p;
        HIPCHK(c, hipMemsetAsync(cnt, 0, 4, c->stream));
        hipLaunchKernelGGL(mirp::side_compact_kernel, dim3((unsigned)std::min<long long>((nw + 255) / 256, 4096)), dim3(256), 0, c->stream,
                           (const int*)c->status.p, nw, (int*)c->side_idx.p, (int*)c->side_list.p, cnt);
        unsigned int ns = 0;
        HIPCHK(c, hipMemcpyAsync(&ns, cnt, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (ns > 0) {
            // a flagged window reports the number of lines it needs: the side buffers get the largest of them (<= n + 2), not the worst case
            std::vector<int> hnl((size_t)nw), hst((size_t)nw);
            HIPCHK(c, hipMemcpy(hnl.data(), c->nlines.p, 4 * (size_t)nw, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(hst.data(), c->status.p, 4 * (size_t)nw, hipMemcpyDeviceToHost));
            int need = max_lines + 1;
            for (long long w = 0; w < nw; w++) if (hst[(size_t)w] == 1) need = std::max(need, hnl[(size_t)w]);
            big = std::min(big, ((need + 7) / 8) * 8);
            const size_t per2 = (size_t)big * stride;
            if (c->side_offs.ensure(8 * (size_t)ns + 8) || c->side_lens.ensure(4 * (size_t)ns) || c->lines2.ensure(sizeof(MirpFoldLine) * (size_t)ns * big) ||
                c->ss2.ensure((size_t)ns * per2) || c->nlines2.ensure(4 * (size_t)ns) || c->mfe2.ensure(4 * (size_t)ns) || c->status2.ensure(4 * (size_t)ns))
                return fail(c, -6, "device allocation failed (fold overflow buffers)");
            const dim3 g((ns + 255) / 256), b(256);
            hipLaunchKernelGGL(mirp::side_gather_kernel, g, b, 0, c->stream, (const int*)c->side_list.p, (int)ns, c->v_woffs(),
                               c->v_wlens(), (long long*)c->side_offs.p, (int*)c->side_lens.p);
            int rc = mirp_run_fold(c, (const unsigned char*)c->wseqs.p, (const long long*)c->side_offs.p, (const int*)c->side_lens.p, (int)ns, n_cap, span, big,
                                   stride, (MirpFoldLine*)c->lines2.p, (char*)c->ss2.p, (int*)c->nlines2.p, (int*)c->mfe2.p, (int*)c->status2.p);
            if (rc) return rc;
            fallbacks += c->last_fallback;
            hipLaunchKernelGGL(mirp::side_scatter_kernel, g, b, 0, c->stream, (const int*)c->side_list.p, (int)ns, (const int*)c->nlines2.p,
                               (const int*)c->mfe2.p, (const int*)c->status2.p, (int*)c->nlines.p, (int*)c->mfe.p, (int*)c->status.p);
            c->n_side = ns; c->side_max_lines = big;
        }
    }
    c->last_fallback = fallbacks;
    c->fold_kernel_ms[0] = main_kernel_ms[0]; c->fold_kernel_ms[1] = main_kernel_ms[1];
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    float a = 0;
    (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
    c->ms[2] = a;
    c->fold_stride = stride; c->fold_max_lines = max_lines; c->fold_span = span;
    c->have_fold = true;
    return 0;
}

extern "C" int mirp_get_fold(mirp_ctx* c, MirpFoldLine** lines, char** ss, int32_t* ss_stride, int32_t* max_lines, int32_t** n_lines, int32_t** mfe,
                             int32_t** status) {
    if (!c) return -1;
    if (!lines || !ss || !ss_stride || !max_lines || !n_lines || !mfe || !status) return fail(c, -1, "mirp_get_fold: null argument");
    if (!c->have_fold) return fail(c, -1, "mirp_get_fold: run mirp_fold first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nw = (size_t)c->n_windows, ml = (size_t)c->fold_max_lines;
    MirpFoldLine* hl = host_copy<MirpFoldLine>(c, c->lines.p, nw * ml);
    char* hs = host_copy<char>(c, c->ss.p, nw * ml * c->fold_stride);
    int32_t* hn = host_copy<int32_t>(c, c->nlines.p, nw);
    int32_t* hm = host_copy<int32_t>(c, c->mfe.p, nw);
    int32_t* ht = host_copy<int32_t>(c, c->status.p, nw);
    if (!hl || !hs || !hn || !hm || !ht) { std::free(hl); std::free(hs); std::free(hn); std::free(hm); std::free(ht); return fail(c, -2, "D2H failed"); }
    *lines = hl; *ss = hs; *ss_stride = c->fold_stride; *max_lines = c->fold_max_lines; *n_lines = hn; *mfe = hm; *status = ht;
    return 0;
}

extern "C" int mirp_get_fold_summary(mirp_ctx* c, int32_t** n_lines, int32_t** mfe, int32_t** status, int64_t* n_windows) {
    if (!c) return -1;
    if (!n_lines || !mfe || !status || !n_windows) return fail(c, -1, "mirp_get_fold_summary: null argument");
    if (!c->have_fold) return fail(c, -1, "mirp_get_fold_summary: run mirp_fold first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nw = (size_t)c->n_windows;
    int32_t* hn = host_copy<int32_t>(c, c->nlines.p, nw);
    int32_t* hm = host_copy<int32_t>(c, c->mfe.p, nw);
    int32_t* ht = host_copy<int32_t>(c, c->status.p, nw);
    if (!hn || !hm || !ht) { std::free(hn); std::free(hm); std::free(ht); return fail(c, -2, "D2H failed"); }
    *n_lines = hn; *mfe = hm; *status = ht; *n_windows = c->n_windows;
    return 0;
}

extern "C" int mirp_get_fold_overflow(mirp_ctx* c, int32_t** windows, int64_t* n, MirpFoldLine** lines, char** ss, int32_t* ss_stride, int32_t* max_lines,
                                      int32_t** n_lines) {
    if (!c) return -1;
    if (!windows || !n || !lines || !ss || !ss_stride || !max_lines || !n_lines) return fail(c, -1, "mirp_get_fold_overflow: null argument");
    if (!c->have_fold) return fail(c, -1, "mirp_get_fold_overflow: run mirp_fold first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t ns = (size_t)c->n_side, ml = (size_t)c->side_max_lines;
    int32_t* hw = host_copy<int32_t>(c, c->side_list.p, ns);
    MirpFoldLine* hl = host_copy<MirpFoldLine>(c, c->lines2.p, ns * ml);
    char* hs = host_copy<char>(c, c->ss2.p, ns * ml * c->fold_stride);
    int32_t* hn = host_copy<int32_t>(c, c->nlines2.p, ns);
    if (!hw || !hl || !hs || !hn) { std::free(hw); std::free(hl); std::free(hs); std::free(hn); return fail(c, -2, "D2H failed"); }
    *windows = hw; *n = (int64_t)ns; *lines = hl; *ss = hs; *ss_stride = c->fold_stride; *max_lines = c->side_max_lines; *n_lines = hn;
    return 0;
}

// the filter kernel over the resident windows: the main launch (fold output at the default line capacity) and, when windows were folded
// again at full capacity, a second launch over those with the side buffers
static int launch_predict_resident(mirp_ctx* c, const MirpPredictParams& pp, unsigned int* rcount, int* rpool, unsigned int rcap, int rstride) {
    const long long nw = c->n_windows;
    const int* skip = c->n_side > 0 ? (const int*)c->side_idx.p : nullptr;
    std::string err;
    if (c->p_need.ensure(12 * (size_t)std::max<long long>(nw, 1))) return fail(c, -6, "device allocation failed (predict)");
    // every launch re-runs the windows that exceeded a capacity of the kernel (structures, pieces per line, candidate matures) at capacities
    // sized for them (run_predict_launch): the reference has no such limits
    if (int rc = mirp::run_predict_launch(c->stream, c->n_cu, c->v_windows(), (int)nw, (const MirpMature*)c->matures.p, (const MirpAln*)c->alns.p, c->n_alns,
                                          (const MirpFoldLine*)c->lines.p, (const char*)c->ss.p, c->fold_stride, c->fold_max_lines, (const int*)c->nlines.p, pp,
                                          (MirpMirna*)c->p_out.p, (int*)c->p_nout.p, (int*)c->p_status.p, rcount, rpool, rcap, rstride, nullptr, 0, skip, &err, (int*)c->p_need.p))
        return fail(c, rc, "mirp_predict: " + err);
    if (c->n_side > 0) {
        if (mirp::predict_lds_bytes_min(c->side_max_lines, c->fold_stride) > 160 * 1024)
            return fail(c, -5, "mirp_predict: a window with more structure lines than the default capacity exceeds the LDS budget of the predict kernel at this PRECURSOR_LEN");
        if (int rc = mirp::run_predict_launch(c->stream, c->n_cu, c->v_windows(), (int)nw, (const MirpMature*)c->matures.p, (const MirpAln*)c->alns.p, c->n_alns,
                                              (const MirpFoldLine*)c->lines2.p, (const char*)c->ss2.p, c->fold_stride, c->side_max_lines, (const int*)c->nlines2.p, pp,
                                              (MirpMirna*)c->p_out.p, (int*)c->p_nout.p, (int*)c->p_status.p, rcount, rpool, rcap, rstride, (const int*)c->side_list.p,
                                              (int)c->n_side, nullptr, &err, (int*)c->p_need.p))
            return fail(c, rc, "mirp_predict (windows over the default line capacity): " + err);
    }
    return 0;
}

extern "C" int mirp_predict(mirp_ctx* c, const MirpPredictParams* pp, MirpMirna** result, int64_t* n_result, char** ss_text, int32_t* ss_stride,
                            int32_t** n_passed, int32_t** status, int64_t* n_windows) {
    if (!c) return -1;
    if (!pp || !result || !n_result || !ss_text || !ss_stride || !n_passed || !status || !n_windows) return fail(c, -1, "mirp_predict: null argument");
    if (!c->have_fold) return fail(c, -1, "mirp_predict: run mirp_fold first");
    if (pp->n_samples < 1 || pp->n_samples > MIRP_MAX_SAMPLES) return fail(c, -1, "mirp_predict: n_samples out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (mirp::predict_lds_bytes_min(c->fold_max_lines, c->fold_stride) > 160 * 1024)
        return fail(c, -5, "mirp_predict: max_lines*ss_stride exceeds the LDS budget of the predict kernel");
    const long long nw = c->n_windows;
    const size_t w1 = (size_t)std::max<long long>(nw, 1);
    if (c->p_out.ensure(sizeof(MirpMirna) * w1 * MIRP_MAX_MIRNA_PER_WINDOW) || c->p_nout.ensure(4 * w1) || c->p_status.ensure(4 * w1) ||
        c->p_keep.ensure(4 * w1) || c->p_kscan.ensure(8 * (w1 + 1)))
        return fail(c, -6, "device allocation failed (predict)");
    hipStream_t st = c->stream;
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    long long nres = 0;
    if (nw > 0) {
        if (int rc = launch_predict_resident(c, *pp, nullptr, nullptr, 0u, 0)) return rc;
        HIPCHK(c, hipEventRecord(c->ev[1], st));
        hipLaunchKernelGGL(mirp::result_keep_kernel, dim3((unsigned)std::min<long long>((nw + 255) / 256, 4096)), dim3(256), 0, st, c->v_roles(),
                           (const int*)c->p_nout.p, nw, (int*)c->p_keep.p);
        mirp::launch_excl_scan(st, (const int*)c->p_keep.p, (long long*)c->p_kscan.p, nw);
        if (read_ll(c, (const long long*)c->p_kscan.p + nw, &nres)) return fail(c, -2, "D2H failed");
        if (c->p_res.ensure(sizeof(MirpMirna) * (size_t)std::max<long long>(nres, 1)) || c->p_text.ensure((size_t)std::max<long long>(nres, 1) * c->fold_stride))
            return fail(c, -6, "device allocation failed (result)");
        hipLaunchKernelGGL(mirp::result_gather_kernel, dim3((unsigned)std::min<long long>((nw + 3) / 4, 8192)), dim3(256), 0, st, (const int*)c->p_keep.p,
                           (const long long*)c->p_kscan.p, nw, (const MirpMirna*)c->p_out.p, (const char*)c->ss.p, c->fold_stride, c->fold_max_lines,
                           c->n_side > 0 ? (const int*)c->side_idx.p : (const int*)nullptr, (const char*)c->ss2.p, c->side_max_lines,
                           (MirpMirna*)c->p_res.p, (char*)c->p_text.p);
    } else {
        HIPCHK(c, hipEventRecord(c->ev[1], st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    float a = 0;
    (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
    c->ms[3] = a;
    MirpMirna* hr = host_copy<MirpMirna>(c, c->p_res.p, (size_t)nres);
    char* ht = host_copy<char>(c, c->p_text.p, (size_t)nres * c->fold_stride);
    int32_t* hn = host_copy<int32_t>(c, c->p_nout.p, (size_t)nw);
    int32_t* hst = host_copy<int32_t>(c, c->p_status.p, (size_t)nw);
    if (!hr || !ht || !hn || !hst) { std::free(hr); std::free(ht); std::free(hn); std::free(hst); return fail(c, -2, "D2H failed"); }
    *result = hr; *n_result = nres; *ss_text = ht; *ss_stride = c->fold_stride; *n_passed = hn; *status = hst; *n_windows = nw;
    c->n_result = nres; c->have_result = true;
    return 0;
}

// -d mode of the reference (dict_why_not_miRNA_reasons of check_loci, MP:2206-2347): re-runs the filter kernel in its reasons mode and returns
// one int32 record per window (field 1 == -1: {window, -1, n_structures, any mature in range, n_passed}) and per evaluated (mature, structure)
// pair: {window, mature index, structure index, line, off, len, get_maturestar_info code, expression flags, fold_s, fold_e, star_s, star_e,
// depth this strand, antisense, mature, isoform, star, imperfect star x3, mature-star distance, mature depth per sample...}; stride ints each.
// Flags: 1 too close, 2 star but too few reads on the duplex, 4 no star and not allowed, 8 too many start positions, 16 ratio too small,
// 32 mature depth <= 100, 64 not in all samples, 128 passed, 256 no read on the precursor (the reference divides by zero there).
extern "C" int mirp_predict_reasons(mirp_ctx* c, const MirpPredictParams* pp, int32_t** records, int64_t* n_records, int32_t* stride) {
    if (!c) return -1;
    if (!pp || !records || !n_records || !stride) return fail(c, -1, "mirp_predict_reasons: null argument");
    if (!c->have_fold) return fail(c, -1, "mirp_predict_reasons: run mirp_fold first");
    if (pp->n_samples < 1 || pp->n_samples > MIRP_MAX_SAMPLES) return fail(c, -1, "mirp_predict_reasons: n_samples out of range");
    HIPCHK(c, hipSetDevice(c->device));
    const long long nw = c->n_windows;
    const int rstride = 21 + pp->n_samples;
    *records = nullptr; *n_records = 0; *stride = rstride;
    if (nw <= 0) return 0;
    const size_t w1 = (size_t)nw;
    if (c->p_out.ensure(sizeof(MirpMirna) * w1 * MIRP_MAX_MIRNA_PER_WINDOW) || c->p_nout.ensure(4 * w1) || c->p_status.ensure(4 * w1))
        return fail(c, -6, "device allocation failed (predict)");
    TmpDevice T;
    unsigned int* d_cnt = (unsigned int*)T.get(16);
    if (!d_cnt) return fail(c, -6, "device allocation failed (reasons)");
    unsigned int cap = (unsigned int)std::min<unsigned long long>((unsigned long long)nw * 96ull + 1024ull, 0x7fffffffull / (unsigned)rstride);
    for (int attempt = 0; attempt < 2; attempt++) {
        int* d_pool = (int*)T.get((size_t)cap * rstride * 4);
        if (!d_pool) return fail(c, -6, "device allocation failed (reasons pool)");
        HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, c->stream));
        if (int rc = launch_predict_resident(c, *pp, d_cnt, d_pool, cap, rstride)) return rc;
        unsigned int n = 0;
        HIPCHK(c, hipMemcpyAsync(&n, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (n <= cap) {
            int32_t* h = host_copy<int32_t>(c, d_pool, (size_t)n * rstride);
            if (!h) return fail(c, -2, "D2H failed");
            size_t k2 = 0;          // records of a first pass whose window was run again at larger capacities carry window -1
            for (size_t k = 0; k < n; k++)
                if (h[k * rstride] >= 0) { if (k2 != k) std::memcpy(h + k2 * rstride, h + k * rstride, 4 * (size_t)rstride); k2++; }
            *records = h; *n_records = (int64_t)k2;
            return 0;
        }
        if ((unsigned long long)n * rstride > 0x7fffffffull) return fail(c, -6, "mirp_predict_reasons: record pool too large");
        cap = n;
    }
    return fail(c, -2, "mirp_predict_reasons: record pool overflow");
}

extern "C" int mirp_last_timings(mirp_ctx* c, double ms[4]) {
    if (!c || !ms) return -1;
    for (int i = 0; i < 4; i++) ms[i] = c->ms[i];
    return 0;
}

extern "C" int mirp_last_fold_kernel_ms(mirp_ctx* c, double ms[2]) {
    if (!c || !ms) return -1;
    ms[0] = c->fold_kernel_ms[0]; ms[1] = c->fold_kernel_ms[1];
    return 0;
}

extern "C" int64_t mirp_last_fold_fallbacks(mirp_ctx* c) { return c ? (int64_t)c->last_fallback : -1; }
extern "C" int64_t mirp_last_fold_overflow(mirp_ctx* c) { return c ? (int64_t)c->n_side : -1; }
extern "C" int mirp_last_coverage_fused(mirp_ctx* c) { return c ? (c->cov_fused ? 1 : 0) : -1; }
extern "C" int mirp_set_fold_split_path(mirp_ctx* c, int32_t mode) {
    if (!c) return -1;
    if (mode < 0 || mode > 1) return fail(c, -1, "mirp_set_fold_split_path: mode is 0 (split candidates) or 1 (dense splits)");
    c->fold_dense = mode;
    return 0;
}
extern "C" int64_t mirp_last_fold_dense(mirp_ctx* c) { return c ? (int64_t)c->last_dense : -1; }
extern "C" int mirp_set_coverage_path(mirp_ctx* c, int32_t mode) {
    if (!c) return -1;
    if (mode < -1 || mode > 1) return fail(c, -1, "mirp_set_coverage_path: mode is -1 (by record density), 0 (atomic scatter) or 1 (fused scan)");
    c->cov_mode = mode;
    return 0;
}
