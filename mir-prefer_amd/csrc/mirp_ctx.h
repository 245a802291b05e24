// Context and helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "mirp_internal.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct TmpDevice {   // scoped device allocations
    std::vector<void*> ptrs;
    ~TmpDevice() { for (void* p : ptrs) (void)hipFree(p); }
    void* get(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return p;
    }
};

struct mirp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int n_cu = 256;
    FoldParams* d_params = nullptr;
    FoldParams185* d_params185 = nullptr;   // created on the first use of the vienna-1.8.5 model (generic kernel)
    FoldParams* d_params185l = nullptr;     // Turner-1999 values in the layout of the LDS-resident kernels
    int fold_model = MIRP_FOLD_MODEL_VIENNA_212;
    DevBuf seqs, offs, ws, lines, ss, nlines, mfe, status, carch, fctl, flist, wstate, dlist;
    DevBuf blines, bss, bnlines, bmfe, bstatus;      // outputs of mirp_fold_batch (kept apart from the resident fold output of mirp_fold)
    long long last_fallback = 0;
    // ---- device-resident pipeline state (mirp_pipeline.cpp)
    int n_contigs = 0;
    long long gtot = 0, gbytes = 0, n_alns = 0;
    std::vector<long long> h_clen, h_goff, h_gboff;
    DevBuf genome, clen, goff, gboff, alns, order, segs, sort_tmp, sort_counts;
    long long n_segs = 0;
    DevBuf tile_first;               // fused coverage scan: first record of every tile (candidate_kernels.hip)
    int max_aln_len = -1;             // longest resident record (reference span), -1 = not known: the fused scan needs it <= one tile
    int fold_dense = 0;               // mirp_set_fold_split_path: 1 = dense multiloop splits for every window
    unsigned int last_dense = 0;      // windows of the last fold the candidate-pool pass handed to the dense fill kernel
    int cov_mode = -1;                // mirp_set_coverage_path: -1 = by record density, 0 = atomic scatter, 1 = fused scan (where the input allows it)
    bool cov_fused = false;           // the last run_coverage took the fused path (nothing to clear afterwards)
    void* diff_clean_ptr = nullptr;   // the difference arrays at this address are all zero (run_coverage / clean_coverage)
    size_t diff_clean_bytes = 0;
    bool ingest_resident = false;     // the alignments came from mirp_ingest_sams_gpu (already validated and sorted on the device)
    int ingest_n_contigs = 0;
    DevBuf diff, stat, starts, totals, runs, keep, kscan, csq, cdest, peaks_sq, peaks_sorted;
    DevBuf head, hscan, rfirst, nent, isloc, nslots, escan, lscan, sscan, windows, roles, loci, wpeaks, matures, wseqs, woffs, wlens;
    DevBuf p_out, p_nout, p_status, p_keep, p_kscan, p_res, p_text, p_need;
    // windows whose structure lines exceed the default capacity: re-folded alone at full capacity into these side buffers (mirp_fold)
    DevBuf side_cnt, side_idx, side_list, side_offs, side_lens, lines2, ss2, nlines2, mfe2, status2;
    long long n_side = 0;
    int side_max_lines = 0;
    MirpCandidateParams cand = {0, 0, 0, 0};
    long long n_runs = 0, n_above = 0, n_peaks = 0, n_regions = 0, n_loci = 0, n_windows = 0, n_slots = 0;
    int seq_stride = 0, fold_stride = 0, fold_max_lines = 0, fold_span = 0;
    bool have_candidate = false, have_fold = false;
    int shard_first_run_double = 0;   // contig shard whose first covered contig is not the first covered contig of the whole genome
    double ms[4] = {0, 0, 0, 0};
    double fold_kernel_ms[2] = {0, 0};   // fill / epilogue kernels of the last mirp_run_fold (LDS-resident path)
    std::vector<hipEvent_t> fold_ev;     // 3 events per sub-batch, created on demand
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // ---- multi-GPU (mirp_dist.cpp): RCCL communicator of this context's device, one process per GPU
    void* comm = nullptr;
    int dist_rank = 0, dist_world = 1;
    std::string dist_dir;             // local transport (mirp_dist_init_local): ranks that share a GPU exchange through files in this directory
    long long dist_seq = 0;
    bool dist_broken = false;         // a wait on a peer expired (or RCCL reported an asynchronous error): the communicator was aborted, every later exchange fails at once
    DevBuf dist_tmp;
    struct TextJob { std::thread th; int rc = 0; std::string err; };
    std::vector<std::shared_ptr<TextJob>> text_jobs;      // text artefacts still being formatted / written behind the caller (mirp_wait_text)
    // window view (mirp_select_windows): the fold and the filter run on windows [win_first, win_first + n_windows) of the candidate stage's list;
    // sel_total = the list's length while a view is active, -1 otherwise
    long long win_first = 0, sel_total = -1;
    const MirpWindow* v_windows() const { return (const MirpWindow*)windows.p + win_first; }
    const int* v_roles() const { return (const int*)roles.p + win_first; }
    const long long* v_woffs() const { return (const long long*)woffs.p + win_first; }
    const int* v_wlens() const { return (const int*)wlens.p + win_first; }
    long long n_result = 0;           // records of the last mirp_predict (p_res / p_text), what mirp_gather_loci sends
    bool have_result = false;
};

static inline int fail(mirp_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail((c), -2, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)


// Folds n_work device-resident windows (seqs/offs/lens as the kernels expect) into the context's fold output buffers.
// Uses the LDS-resident kernel when span allows and re-runs flagged windows (length / int16 range) with the generic kernel.
// sort_kernels.hip: stable device sort by (tid, pos) / keep-region filter of the resident record array
int mirp_device_sort_alns(mirp_ctx* c, MirpAln* d_alns, MirpAln* d_tmp, long long n, int posbits, int tidbits);
int mirp_device_mask_alns(mirp_ctx* c, MirpAln* d_alns, MirpAln* d_tmp, long long* n_io, MirpAln* d_segs, MirpAln* d_segtmp, const int* d_owner, const int* d_seg_span,
                          long long* nseg_io,
                          const long long* d_rfirst, const int* d_rstart, const int* d_remax);

namespace mirp {
// mirp_dist.cpp: collectives on the context's communicator (no-ops / local copies without one)
int dist_all_counts(mirp_ctx* c, long long mine, std::vector<long long>& counts);
int dist_gatherv_bytes(mirp_ctx* c, const void* d_src, long long mine, int dst, void* d_dst, const std::vector<long long>& counts);
int dist_alltoallv_bytes(mirp_ctx* c, const void* d_send, const std::vector<long long>& send_off, const std::vector<long long>& send_cnt, void* d_recv,
                         const std::vector<long long>& recv_off, const std::vector<long long>& recv_cnt);
int dist_allgather_ll(mirp_ctx* c, const long long* mine, int n, std::vector<long long>& out);
int dist_agree(mirp_ctx* c, int local_rc, const char* what);
}  // namespace mirp

int mirp_run_fold(mirp_ctx* c, const unsigned char* d_seqs, const long long* d_offs, const int* d_lens, int n_work, int n_cap, int span,
                  int max_lines, int stride, MirpFoldLine* d_lines, char* d_ss, int* d_nlines, int* d_mfe, int* d_status);
