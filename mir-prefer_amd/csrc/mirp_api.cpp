// C-ABI layer of libmirprefer.so (see include/mirprefer.h).  Host orchestration only: device
// buffers, streams, kernel launches.  There is NO CPU fallback: every entry point fails loudly
// if no gfx950 device / HIP runtime is usable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "mirp_internal.h"

#define MIRP_ABI_VERSION 7   // 7: mirp_write_result_reports, mirp_fold_predict_report_stream, mirp_select_windows, mirp_dist_comm_info, MIRP_MAX_SAMPLES 255; 6: mirp_last_coverage_fused, mirp_fold_batch_summary, mirp_predict_batch_reasons, text writers; 5: mirp_dist_*, mirp_gather_loci / mirp_gather_records, mirp_read_fasta, mirp_ingest_sams_shard; 4: MirpSamData.segs, mirp_ingest_sams_gpu, mirp_load_coverage_segments; 2: mirp_set_fold_model, mirp_ingest_sams; 3: mirp_predict returns the per-window capacity status, mirp_get_fold_overflow
#define MIRP_NMAX 3096

#include "mirp_ctx.h"

extern "C" int mirp_abi_version(void) { return MIRP_ABI_VERSION; }

extern "C" void mirp_destroy(mirp_ctx* c);

extern "C" int mirp_create(int device, mirp_ctx** out) {
    if (!out) return -1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -3;  // no GPU: fail loudly, no fallback
    if (device < 0 || device >= ndev) return -4;
    mirp_ctx* c = new mirp_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return -2; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return -2; }
    FoldParams* hp = new FoldParams();
    mirp_fill_fold_params(hp);
    if (hp->gen_wing_d != mirp::fold_lds_gen_wing_d()) {      // the fill kernel compiles this property of the table in (a1_gen_row_w)
        std::fprintf(stderr, "mirp_create: energy model and fill kernel disagree on the saturation of the asymmetry term (%d vs %d)\n", hp->gen_wing_d, mirp::fold_lds_gen_wing_d());
        delete hp; delete c; return -5;
    }
    if (hp->xb_bias < 0) { std::fprintf(stderr, "mirp_create: inner-pair terms outside the fill kernel's byte-table range\n"); delete hp; delete c; return -5; }
    for (int t = 1; t <= 6; t++)          // the fill kernel's list entries carry these two terms of a cell as 10-bit signed fields (fold_lds_kernel.hip, ENT_OUTER)
        for (int a = 0; a < 5; a++)
            for (int b = 0; b < 5; b++)
                if (hp->mismatchI[t][a][b] < -512 || hp->mismatchI[t][a][b] > 511 || hp->mismatch1nI[t][a][b] < -512 || hp->mismatch1nI[t][a][b] > 511) {
                    std::fprintf(stderr, "mirp_create: interior-loop mismatch energies outside the fill kernel's list-entry range\n");
                    delete hp; delete c; return -5;
                }
    if (hipMalloc((void**)&c->d_params, sizeof(FoldParams)) != hipSuccess ||
        hipMemcpy(c->d_params, hp, sizeof(FoldParams), hipMemcpyHostToDevice) != hipSuccess) {
        delete hp; delete c; return -2;
    }
    delete hp;
    for (int i = 0; i < 6; i++)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) { mirp_destroy(c); return -2; }
    *out = c;
    return 0;
}

extern "C" int mirp_set_fold_model(mirp_ctx* c, int32_t model) {
    if (!c) return -1;
    if (model != MIRP_FOLD_MODEL_VIENNA_212 && model != MIRP_FOLD_MODEL_VIENNA_185) return fail(c, -1, "mirp_set_fold_model: unknown model");
    (void)hipSetDevice(c->device);
    if (model == MIRP_FOLD_MODEL_VIENNA_185 && !c->d_params185l) {
        FoldParams* hl = new FoldParams();
        mirp_fill_fold_params_t1999(hl);
        if (hipMalloc((void**)&c->d_params185l, sizeof(FoldParams)) != hipSuccess ||
            hipMemcpy(c->d_params185l, hl, sizeof(FoldParams), hipMemcpyHostToDevice) != hipSuccess) {
            delete hl;
            if (c->d_params185l) { (void)hipFree(c->d_params185l); c->d_params185l = nullptr; }
            return fail(c, -6, "mirp_set_fold_model: device allocation failed");
        }
        delete hl;
    }
    if (model == MIRP_FOLD_MODEL_VIENNA_185 && !c->d_params185) {
        FoldParams185* hp = new FoldParams185();
        mirp_fill_fold_params185(hp);
        if (hipMalloc((void**)&c->d_params185, sizeof(FoldParams185)) != hipSuccess ||
            hipMemcpy(c->d_params185, hp, sizeof(FoldParams185), hipMemcpyHostToDevice) != hipSuccess) {
            delete hp;
            if (c->d_params185) { (void)hipFree(c->d_params185); c->d_params185 = nullptr; }
            return fail(c, -6, "mirp_set_fold_model: device allocation failed");
        }
        delete hp;
    }
    c->fold_model = model;
    c->have_fold = false; c->have_result = false;
    return 0;
}

extern "C" void mirp_destroy(mirp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)mirp_wait_text(c);
    (void)mirp_dist_finalize(c);
    c->dist_tmp.release();
    c->seqs.release(); c->offs.release(); c->ws.release(); c->lines.release(); c->ss.release();
    c->nlines.release(); c->mfe.release(); c->status.release(); c->carch.release(); c->fctl.release(); c->flist.release(); c->wstate.release(); c->dlist.release();
    c->blines.release(); c->bss.release(); c->bnlines.release(); c->bmfe.release(); c->bstatus.release();
    for (DevBuf* b : {&c->genome, &c->clen, &c->goff, &c->gboff, &c->alns, &c->order, &c->diff, &c->stat, &c->starts, &c->totals, &c->runs,
                      &c->keep, &c->kscan, &c->csq, &c->cdest, &c->peaks_sq, &c->peaks_sorted, &c->head, &c->hscan, &c->rfirst, &c->nent,
                      &c->isloc, &c->nslots, &c->escan, &c->lscan, &c->sscan, &c->windows, &c->roles, &c->loci, &c->wpeaks, &c->matures,
                      &c->wseqs, &c->woffs, &c->wlens, &c->segs, &c->sort_tmp, &c->sort_counts, &c->side_cnt, &c->side_idx, &c->side_list, &c->side_offs, &c->side_lens, &c->lines2, &c->ss2,
                      &c->nlines2, &c->mfe2, &c->status2, &c->p_out, &c->p_nout, &c->p_status, &c->p_keep, &c->p_kscan, &c->p_res, &c->p_text})
        b->release();
    for (int i = 0; i < 6; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (hipEvent_t ev : c->fold_ev) (void)hipEventDestroy(ev);
    if (c->d_params) (void)hipFree(c->d_params);
    if (c->d_params185) (void)hipFree(c->d_params185);
    if (c->d_params185l) (void)hipFree(c->d_params185l);
    if (c->stream) { mirp::release_scan_scratch(c->stream); (void)hipStreamDestroy(c->stream); }
    delete c;
}

extern "C" const char* mirp_last_error(const mirp_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" void mirp_free(void* p) { std::free(p); }

static int fold_batch_impl(mirp_ctx* c, const char* seqs, const int64_t* offsets, int32_t n_seqs, int32_t span,
                           int32_t max_lines, MirpFoldLine** lines, char** ss, int32_t* ss_stride_out,
                           int32_t** n_lines, int32_t** mfe, int32_t** status, bool want_text) {
    if (!c) return -1;
    if (!seqs || !offsets || n_seqs < 0 || (want_text && (!lines || !ss || !ss_stride_out)) || !n_lines || !mfe || !status)
        return fail(c, -1, "mirp_fold_batch: null argument");
    int32_t stride_dummy = 0;
    if (!ss_stride_out) ss_stride_out = &stride_dummy;
    if (span < 1 || max_lines < 1) return fail(c, -1, "mirp_fold_batch: bad span/max_lines");
    HIPCHK(c, hipSetDevice(c->device));
    int n_max = 1;
    for (int i = 0; i < n_seqs; i++) {
        int64_t n = offsets[i + 1] - offsets[i];
        if (n < 0) return fail(c, -1, "mirp_fold_batch: offsets not monotone");
        if (n > MIRP_NMAX) return fail(c, -5, "mirp_fold_batch: sequence longer than MIRP_NMAX");
        n_max = std::max<int>(n_max, (int)n);
    }
    const int stride = ((n_max + 3 + 7) / 8) * 8;
    *ss_stride_out = stride;
    const size_t nl = (size_t)n_seqs * max_lines;
    MirpFoldLine* h_lines = (MirpFoldLine*)std::calloc(want_text ? std::max<size_t>(nl, 1) : 1, sizeof(MirpFoldLine));
    char* h_ss = (char*)std::calloc(want_text ? std::max<size_t>(nl * stride, 1) : 1, 1);
    int32_t* h_nl = (int32_t*)std::calloc(std::max(n_seqs, 1), sizeof(int32_t));
    int32_t* h_mfe = (int32_t*)std::calloc(std::max(n_seqs, 1), sizeof(int32_t));
    int32_t* h_st = (int32_t*)std::calloc(std::max(n_seqs, 1), sizeof(int32_t));
    auto bail = [&](int code, const std::string& m) {
        std::free(h_lines); std::free(h_ss); std::free(h_nl); std::free(h_mfe); std::free(h_st);
        return fail(c, code, m);
    };
    if (!h_lines || !h_ss || !h_nl || !h_mfe || !h_st) return bail(-6, "mirp_fold_batch: host allocation failed");
    if (n_seqs > 0) {
        const size_t total = (size_t)offsets[n_seqs] - (size_t)offsets[0];
        if (c->seqs.ensure(total + 16) || c->offs.ensure(sizeof(long long) * (n_seqs + 1)))
            return bail(-6, "device allocation failed (inputs)");
        std::vector<long long> rel(n_seqs + 1);
        for (int i = 0; i <= n_seqs; i++) rel[i] = offsets[i] - offsets[0];
        if (hipMemcpyAsync(c->seqs.p, seqs + offsets[0], total, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(c->offs.p, rel.data(), sizeof(long long) * (n_seqs + 1), hipMemcpyHostToDevice, c->stream) != hipSuccess)
            return bail(-2, "H2D copy failed");
        // batches bound the structure-text buffer
        const size_t per_win = (size_t)max_lines * stride;
        int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_seqs, ((size_t)1 << 30) / per_win));
        // the batch path has its own output buffers: a batch call between mirp_fold and mirp_predict (the imported windows of a re-balanced run,
        // balance.py) must not touch the resident fold output
        if (c->blines.ensure(sizeof(MirpFoldLine) * (size_t)batch * max_lines) || c->bss.ensure((size_t)batch * per_win) ||
            c->bnlines.ensure(4 * (size_t)n_seqs) || c->bmfe.ensure(4 * (size_t)n_seqs) || c->bstatus.ensure(4 * (size_t)n_seqs))
            return bail(-6, "device allocation failed (outputs)");
        double kms[2] = {0, 0};
        long long fallbacks = 0;
        for (int b0 = 0; b0 < n_seqs; b0 += batch) {
            const int nb = std::min(batch, n_seqs - b0);
            // windows of this batch are addressed relative to b0: shift the pointers
            int rc = mirp_run_fold(c, (const unsigned char*)c->seqs.p, (const long long*)c->offs.p + b0, nullptr, nb, n_max, span, max_lines, stride,
                                   (MirpFoldLine*)c->blines.p, (char*)c->bss.p, (int*)c->bnlines.p + b0, (int*)c->bmfe.p + b0, (int*)c->bstatus.p + b0);
            if (rc) return bail(rc, c->err);
            kms[0] += c->fold_kernel_ms[0]; kms[1] += c->fold_kernel_ms[1]; fallbacks += c->last_fallback;
            if (want_text &&
                (hipMemcpyAsync(h_lines + (size_t)b0 * max_lines, c->blines.p, sizeof(MirpFoldLine) * (size_t)nb * max_lines,
                                hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                 hipMemcpyAsync(h_ss + (size_t)b0 * per_win, c->bss.p, (size_t)nb * per_win, hipMemcpyDeviceToHost, c->stream) != hipSuccess))
                return bail(-2, "D2H copy failed");
            if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(-2, "fold kernel execution failed");
        }
        c->fold_kernel_ms[0] = kms[0]; c->fold_kernel_ms[1] = kms[1]; c->last_fallback = fallbacks;      // over all batches of this call
        if (hipMemcpy(h_nl, c->bnlines.p, 4 * (size_t)n_seqs, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(h_mfe, c->bmfe.p, 4 * (size_t)n_seqs, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(h_st, c->bstatus.p, 4 * (size_t)n_seqs, hipMemcpyDeviceToHost) != hipSuccess)
            return bail(-2, "D2H copy failed");
    }
    if (want_text) { *lines = h_lines; *ss = h_ss; } else { std::free(h_lines); std::free(h_ss); }
    *n_lines = h_nl; *mfe = h_mfe; *status = h_st;
    return 0;
}

extern "C" int mirp_fold_batch(mirp_ctx* c, const char* seqs, const int64_t* offsets, int32_t n_seqs, int32_t span,
                               int32_t max_lines, MirpFoldLine** lines, char** ss, int32_t* ss_stride_out,
                               int32_t** n_lines, int32_t** mfe, int32_t** status) {
    return fold_batch_impl(c, seqs, offsets, n_seqs, span, max_lines, lines, ss, ss_stride_out, n_lines, mfe, status, true);
}

extern "C" int mirp_fold_batch_summary(mirp_ctx* c, const char* seqs, const int64_t* offsets, int32_t n_seqs, int32_t span, int32_t max_lines,
                                       int32_t** n_lines, int32_t** mfe, int32_t** status) {
    return fold_batch_impl(c, seqs, offsets, n_seqs, span, max_lines, nullptr, nullptr, nullptr, n_lines, mfe, status, false);
}

// ------------------------------------------------------------------------------------------
template <class T>
static T* host_alloc(size_t n) { return (T*)std::calloc(std::max<size_t>(n, 1), sizeof(T)); }

static int predict_batch_impl(mirp_ctx* c, const MirpWindow* windows, int32_t n_windows, const MirpMature* matures, int64_t n_matures,
                              const MirpAln* alns, int64_t n_alns, const MirpFoldLine* lines, const char* ss, int32_t ss_stride,
                              int32_t max_lines, const int32_t* n_lines, const MirpPredictParams* pp, MirpMirna** mirnas,
                              int32_t** n_mirnas, int32_t** status, int32_t** reasons, int64_t* n_reasons, int32_t* reasons_stride) {
    if (!c) return -1;
    if (!windows || !matures || !alns || !lines || !ss || !n_lines || !pp || !mirnas || !n_mirnas || !status || n_windows < 0)
        return fail(c, -1, "mirp_predict_batch: null argument");
    if (pp->n_samples < 1 || pp->n_samples > MIRP_MAX_SAMPLES) return fail(c, -1, "mirp_predict_batch: n_samples out of range");
    if (ss_stride % 8 != 0) return fail(c, -1, "mirp_predict_batch: ss_stride must be a multiple of 8");
    HIPCHK(c, hipSetDevice(c->device));
    if (mirp::predict_lds_bytes_min(max_lines, ss_stride) > 160 * 1024)
        return fail(c, -5, "mirp_predict_batch: max_lines*ss_stride exceeds the LDS budget of the predict kernel");
    TmpDevice T;
    const size_t nl = (size_t)n_windows * max_lines;
    void* d_w = T.get(sizeof(MirpWindow) * n_windows);
    void* d_m = T.get(sizeof(MirpMature) * n_matures);
    void* d_a = T.get(sizeof(MirpAln) * n_alns);
    void* d_l = T.get(sizeof(MirpFoldLine) * nl);
    void* d_s = T.get(nl * ss_stride);
    void* d_n = T.get(4 * (size_t)n_windows);
    void* d_o = T.get(sizeof(MirpMirna) * (size_t)n_windows * MIRP_MAX_MIRNA_PER_WINDOW);
    void* d_no = T.get(4 * (size_t)n_windows);
    void* d_st = T.get(4 * (size_t)n_windows);
    if (!d_w || !d_m || !d_a || !d_l || !d_s || !d_n || !d_o || !d_no || !d_st) return fail(c, -6, "device allocation failed");
    MirpMirna* h_o = host_alloc<MirpMirna>((size_t)n_windows * MIRP_MAX_MIRNA_PER_WINDOW);
    int32_t* h_no = host_alloc<int32_t>(n_windows);
    int32_t* h_st = host_alloc<int32_t>(n_windows);
    auto bail = [&](int code, const std::string& m) { std::free(h_o); std::free(h_no); std::free(h_st); return fail(c, code, m); };
    if (n_windows > 0) {
        if (hipMemcpyAsync(d_w, windows, sizeof(MirpWindow) * n_windows, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_m, matures, sizeof(MirpMature) * n_matures, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_a, alns, sizeof(MirpAln) * n_alns, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_l, lines, sizeof(MirpFoldLine) * nl, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_s, ss, nl * ss_stride, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_n, n_lines, 4 * (size_t)n_windows, hipMemcpyHostToDevice, c->stream) != hipSuccess)
            return bail(-2, "H2D copy failed");
        int grid = std::min(n_windows, c->n_cu * 16);
        // reasons mode (-d): one int32 record per window and per evaluated (mature, structure) pair, layout as mirp_predict_reasons
        const int rstride = 21 + pp->n_samples;
        unsigned int rcap = reasons ? (unsigned int)std::min<unsigned long long>((unsigned long long)n_windows * 96ull + 1024ull, 0x7fffffffull / (unsigned)rstride) : 0u;
        unsigned int* d_cnt = reasons ? (unsigned int*)T.get(16) : nullptr;
        int* d_pool = reasons ? (int*)T.get((size_t)rcap * rstride * 4) : nullptr;
        if (reasons && (!d_cnt || !d_pool)) return bail(-6, "device allocation failed (reasons pool)");
        if (reasons && hipMemsetAsync(d_cnt, 0, 16, c->stream) != hipSuccess) return bail(-2, "memset failed");
        (void)grid;
        {
            std::string err;
            if (int rc = mirp::run_predict_launch(c->stream, c->n_cu, (const MirpWindow*)d_w, n_windows, (const MirpMature*)d_m, (const MirpAln*)d_a, n_alns,
                                                  (const MirpFoldLine*)d_l, (const char*)d_s, ss_stride, max_lines, (const int*)d_n, *pp, (MirpMirna*)d_o,
                                                  (int*)d_no, (int*)d_st, d_cnt, d_pool, rcap, rstride, nullptr, 0, nullptr, &err))
                return bail(rc, "mirp_predict_batch: " + err);
        }
        if (reasons) {
            unsigned int n = 0;
            if (hipMemcpyAsync(&n, d_cnt, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
                return bail(-2, "predict kernel execution failed");
            if (n > rcap) return bail(-6, "mirp_predict_batch_reasons: record pool overflow");
            int32_t* h = host_alloc<int32_t>((size_t)n * rstride);
            if (!h || (n && hipMemcpy(h, d_pool, (size_t)n * rstride * 4, hipMemcpyDeviceToHost) != hipSuccess)) { std::free(h); return bail(-2, "D2H copy failed"); }
            size_t k2 = 0;          // records of a first pass whose window was run again at larger capacities carry window -1
            for (size_t k = 0; k < n; k++)
                if (h[k * rstride] >= 0) { if (k2 != k) std::memcpy(h + k2 * rstride, h + k * rstride, 4 * (size_t)rstride); k2++; }
            *reasons = h; *n_reasons = (int64_t)k2; *reasons_stride = rstride;
        }
        if (hipMemcpyAsync(h_o, d_o, sizeof(MirpMirna) * (size_t)n_windows * MIRP_MAX_MIRNA_PER_WINDOW, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(h_no, d_no, 4 * (size_t)n_windows, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(h_st, d_st, 4 * (size_t)n_windows, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            return bail(-2, "predict kernel execution failed");
    }
    *mirnas = h_o; *n_mirnas = h_no; *status = h_st;
    return 0;
}

extern "C" int mirp_predict_batch(mirp_ctx* c, const MirpWindow* windows, int32_t n_windows, const MirpMature* matures, int64_t n_matures,
                                  const MirpAln* alns, int64_t n_alns, const MirpFoldLine* lines, const char* ss, int32_t ss_stride,
                                  int32_t max_lines, const int32_t* n_lines, const MirpPredictParams* pp, MirpMirna** mirnas,
                                  int32_t** n_mirnas, int32_t** status) {
    return predict_batch_impl(c, windows, n_windows, matures, n_matures, alns, n_alns, lines, ss, ss_stride, max_lines, n_lines, pp, mirnas, n_mirnas, status,
                              nullptr, nullptr, nullptr);
}

extern "C" int mirp_predict_batch_reasons(mirp_ctx* c, const MirpWindow* windows, int32_t n_windows, const MirpMature* matures, int64_t n_matures,
                                          const MirpAln* alns, int64_t n_alns, const MirpFoldLine* lines, const char* ss, int32_t ss_stride,
                                          int32_t max_lines, const int32_t* n_lines, const MirpPredictParams* pp, MirpMirna** mirnas,
                                          int32_t** n_mirnas, int32_t** status, int32_t** reasons, int64_t* n_reasons, int32_t* reasons_stride) {
    if (!reasons || !n_reasons || !reasons_stride) return fail(c, -1, "mirp_predict_batch_reasons: null argument");
    *reasons = nullptr; *n_reasons = 0; *reasons_stride = 0;
    return predict_batch_impl(c, windows, n_windows, matures, n_matures, alns, n_alns, lines, ss, ss_stride, max_lines, n_lines, pp, mirnas, n_mirnas, status,
                              reasons, n_reasons, reasons_stride);
}

int mirp_run_fold(mirp_ctx* c, const unsigned char* d_seqs, const long long* d_offs, const int* d_lens, int n_work, int n_cap, int span,
                  int max_lines, int stride, MirpFoldLine* d_lines, char* d_ss, int* d_nlines, int* d_mfe, int* d_status) {
    if (n_work <= 0) return 0;
    c->last_fallback = 0;
    const bool m185 = c->fold_model == MIRP_FOLD_MODEL_VIENNA_185;
    // generic kernels (tables in a global workspace): every window when the LDS-resident path does not apply, else its flagged windows
    auto run_generic = [&](const int* work_list, int n_generic) -> int {
        if (m185) {
            if (mirp::fold185_lds_bytes(n_cap, max_lines) > 160 * 1024) return fail(c, -5, "LDS budget exceeded (vienna-1.8.5 kernel: window or max_lines too large)");
            const size_t slot = mirp::fold185_ws_slot_ints(n_cap, span);
            int slots = (int)std::max<size_t>(1, std::min<size_t>((size_t)c->n_cu * 96, ((size_t)128 << 30) / (slot * 4)));      // windows per batch (fill kernel, then epilogue kernel)
            slots = std::min(slots, n_generic);
            while (c->ws.ensure((size_t)slots * slot * 4)) {          // (the device may be shared: take fewer windows per batch before giving up)
                if (slots <= 1) return fail(c, -6, "device allocation failed (fold workspace)");
                slots = (slots + 1) / 2; (void)hipGetLastError();
            }
            hipError_t e = mirp::launch_fold185(c->stream, slots, c->d_params185, d_seqs, d_offs, d_lens, work_list, n_generic, span, n_cap, (int*)c->ws.p, slot,
                                                max_lines, stride, d_lines, d_ss, d_nlines, d_mfe, d_status);
            if (e != hipSuccess) return fail(c, -2, std::string("fold (vienna-1.8.5) kernel launch failed: ") + hipGetErrorString(e));
            return 0;
        }
        const size_t slot_ints = mirp::fold_generic_ws_slot_ints(n_cap, span);          // c, fML, DML ring, split-candidate pool of one window
        int wg_per_cu = 96;         // windows per CU and batch (the hardware keeps as many resident as registers and LDS allow: 6 of the fill, 8 of the epilogue): one batch for
                                    // 20,000 windows -- every batch ends with a tail of idle CUs (three batches of 8,192: 0.075 s at L = 301, one: 0.069), and 288 GB hold the 45 - 75 GB
        if (const char* e = std::getenv("MIRP_GENERIC_WG_PER_CU")) wg_per_cu = std::max(1, std::atoi(e));      // dev: occupancy experiments (profiles/tools/l400_time.py)
        int slots = (int)std::max<size_t>(1, std::min<size_t>((size_t)c->n_cu * wg_per_cu, ((size_t)128 << 30) / (slot_ints * 4)));      // PRECURSOR_LEN = 3000: 160 MB a slot
        slots = std::min(slots, n_generic);
        while (c->ws.ensure((size_t)slots * slot_ints * 4)) {          // (the device may be shared: take fewer windows per batch before giving up)
            if (slots <= 1) return fail(c, -6, "device allocation failed (fold workspace)");
            slots = (slots + 1) / 2; (void)hipGetLastError();
        }
        mirp::launch_fold_generic(c->stream, slots, c->d_params, d_seqs, d_offs, d_lens, work_list, n_generic, span, n_cap, (int*)c->ws.p, slot_ints,
                                  max_lines, stride, d_lines, d_ss, d_nlines, d_mfe, d_status);
        HIPCHK(c, hipGetLastError());
        return 0;
    };
    if (!m185 && mirp::fold_generic_lds_bytes(n_cap, max_lines) > 160 * 1024) return fail(c, -5, "LDS budget exceeded (window or max_lines too large)");
    // the generic kernel ranks interior-loop candidates by energy * 1024 + shape in 32 bits (fold_kernel.hip, GEN_EMAX): energies below 10^6 in magnitude
    if (!m185 && n_cap > 5000) return fail(c, -5, "window longer than 5,000 nt");
    const int* work_list = nullptr;
    int n_generic = n_work;
    if (span <= mirp::fold_lds_max_span() && mirp::fold_lds_bytes(max_lines) <= 160 * 1024) {
        // fill kernel (one 1024-thread workgroup per CU, tables in LDS) + epilogue kernel (many small workgroups) per sub-batch;
        // the two exchange the c / fML triangles of every window through per-window slabs in HBM
        const size_t slab = mirp::fold_lds_slab_shorts(std::min(n_cap, mirp::fold_lds_max_n() + 2));
        const int sub = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_work, ((size_t)8 << 30) / (slab * 6)));   // three 16-bit triangles per window: c, fML, trace-back codes
        if (c->carch.ensure((size_t)sub * slab * 6) || c->fctl.ensure(1024) || c->flist.ensure(4 * (size_t)n_work) || c->wstate.ensure(4 * (size_t)sub) ||
            c->dlist.ensure(4 * (size_t)sub))
            return fail(c, -6, "device allocation failed (fold LDS kernel)");
        HIPCHK(c, hipMemsetAsync(c->fctl.p, 0, 1024, c->stream));
        unsigned int* ctl = (unsigned int*)c->fctl.p;
        // diagnostics exist only in a -DMIRP_DIAG build (`make DIAG=1`, profiles/tools/): MIRP_FOLD_DEBUG=<flags> ablates phases (results then
        // wrong), MIRP_FOLD_CLOCKS=1 prints phase clocks, MIRP_FOLD_DUMP=<path> dumps slabs.  The shipped library reads no environment.
#if defined(MIRP_DIAG) || defined(MIRP_LITE_CLOCKS)
        const char* dbg_env = std::getenv("MIRP_FOLD_DEBUG");
        int dbg_flags = dbg_env ? std::atoi(dbg_env) : 0;
        const char* clk_env = std::getenv("MIRP_FOLD_CLOCKS");
        long long* dbg_cycles = clk_env ? (long long*)(ctl + 8) : nullptr;
        if (clk_env && std::atoi(clk_env) == 2) dbg_flags |= 1 << 20;      // light mode: per wave only busy (reported as splits) and barrier wait
#elif defined(MIRP_ABLATE) || defined(MIRP_TIMING_ONLY)
        const int dbg_flags = 16;          // timing experiments whose fill results are wrong by construction: no epilogue (it could walk garbage forever)
        long long* dbg_cycles = nullptr;
#else
        const int dbg_flags = 0;
        long long* dbg_cycles = nullptr;
#endif
        int n_sub = 0;
        for (int b0 = 0; b0 < n_work; b0 += sub) {
            const int nb = std::min(sub, n_work - b0);
            if (b0 > 0) HIPCHK(c, hipMemsetAsync(c->fctl.p, 0, 16, c->stream));   // the work counters and the dense pass's list length; the fallback count (ctl[4]) and the number of windows handed to the dense pass (ctl[5]) keep accumulating
            const int grid = std::min(nb, c->n_cu);
            const int grid_epi = std::min(nb, c->n_cu * 8);
            while ((int)c->fold_ev.size() < 3 * (n_sub + 1)) { hipEvent_t ev; HIPCHK(c, hipEventCreate(&ev)); c->fold_ev.push_back(ev); }
            hipEvent_t* ev3 = &c->fold_ev[3 * n_sub];
            HIPCHK(c, hipEventRecord(ev3[0], c->stream));
            hipError_t e = mirp::launch_fold_lds(c->stream, m185 ? 1 : 0, grid, grid_epi, m185 ? c->d_params185l : c->d_params, d_seqs, d_offs + b0, d_lens ? d_lens + b0 : nullptr, nb, b0, span,
                                                 (short*)c->carch.p, slab, (int*)c->wstate.p, ctl, (int*)c->flist.p, ctl + 4, max_lines, stride,
                                                 d_lines + (size_t)b0 * max_lines, d_ss + (size_t)b0 * max_lines * stride, d_nlines + b0, d_mfe + b0,
                                                 d_status + b0, dbg_flags, dbg_cycles, ev3[1], (int*)c->dlist.p, c->fold_dense);
            if (e != hipSuccess) return fail(c, -2, std::string("fold LDS kernel launch failed: ") + hipGetErrorString(e));
            HIPCHK(c, hipEventRecord(ev3[2], c->stream));
            n_sub++;
        }
        unsigned int nfb2[2] = {0, 0};
        HIPCHK(c, hipMemcpyAsync(nfb2, ctl + 4, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const unsigned int nfb = nfb2[0];
        c->last_fallback = nfb;
        c->last_dense = nfb2[1];
        c->fold_kernel_ms[0] = c->fold_kernel_ms[1] = 0;
        for (int k = 0; k < n_sub; k++) {
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, c->fold_ev[3 * k], c->fold_ev[3 * k + 1]);
            (void)hipEventElapsedTime(&b, c->fold_ev[3 * k + 1], c->fold_ev[3 * k + 2]);
            c->fold_kernel_ms[0] += a; c->fold_kernel_ms[1] += b;
        }
#if defined(MIRP_DIAG) || defined(MIRP_LITE_CLOCKS)
        if (const char* dump = std::getenv("MIRP_FOLD_DUMP")) {   // diagnostics: c / fML slabs of the first window of the last sub-batch
            std::vector<short> h(3 * slab);
            HIPCHK(c, hipMemcpy(h.data(), c->carch.p, 6 * slab, hipMemcpyDeviceToHost));
            if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(h.data(), 2, h.size(), f); std::fclose(f); }
        }
        if (dbg_cycles) {
            long long cyc[4 + 64 + 8];
            HIPCHK(c, hipMemcpy(cyc, dbg_cycles, sizeof(cyc), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[mirp fold clocks] windows=%d setup=%lld fillA=%lld fillB=%lld writeout=%lld (sum over workgroups, s_memtime ticks)\n", n_work,
                         cyc[0], cyc[1], cyc[2], cyc[3]);
            for (int w = 0; w < 16; w++)
                std::fprintf(stderr, "[mirp fold clocks] wave %2d: phaseB=%lld interior=%lld splits=%lld barrier=%lld\n", w, cyc[4 + 4 * w], cyc[5 + 4 * w], cyc[6 + 4 * w],
                             cyc[7 + 4 * w]);
            for (int b = 0; b < 4; b++)
                std::fprintf(stderr, "[mirp fold clocks] wave 9, diagonals with %d%s blocks: %lld, interior ticks %lld\n", b, b == 3 ? "+" : "", cyc[72 + b], cyc[68 + b]);
        }
#endif
#ifdef MIRP_EPI_CLOCKS
        mirp::fold_lds_epi_clocks_print();
#endif
#ifdef MIRP_L2_CLOCKS
        if (!m185) mirp::fold_lds2_clocks_print();
#endif
        if (nfb == 0) return 0;
        work_list = (const int*)c->flist.p;
        n_generic = (int)nfb;
    }
    return run_generic(work_list, n_generic);
}
