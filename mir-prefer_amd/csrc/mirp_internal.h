// Internal declarations shared by the HIP kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>
#include "../../include/mirprefer.h"
#include "fold_params.h"

namespace mirp {

size_t fold_generic_lds_bytes(int n_cap, int max_lines);
size_t fold_generic_ws_slot_ints(int n_cap, int span);
void launch_fold_generic(hipStream_t stream, int grid, const FoldParams* P, const unsigned char* seqs, const long long* offs,
                         const int* lens, const int* work_list, int n_work, int span, int n_cap, int* ws, size_t ws_slot_ints, int max_lines,
                         int ss_stride, MirpFoldLine* out_lines, char* out_ss, int* out_nlines, int* out_mfe, int* out_status);

// fold185_kernel.hip: vienna-1.8.5 compatibility mode
size_t fold185_lds_bytes(int n_cap, int max_lines);
size_t fold185_ws_slot_ints(int n_cap, int span);
hipError_t launch_fold185(hipStream_t stream, int grid, const FoldParams185* P, const unsigned char* seqs, const long long* offs, const int* lens,
                          const int* work_list, int n_work,
                          int span, int n_cap, int* ws, size_t ws_slot_ints, int max_lines, int ss_stride, MirpFoldLine* out_lines, char* out_ss,
                          int* out_nlines, int* out_mfe, int* out_status);

struct PredictCaps { int p_cap, s_cap, m_cap, l_cap; };   // pieces per line, structures per window, candidate matures per window, staged lines per window (predict_kernel.hip)
PredictCaps predict_default_caps(int max_lines, int ss_stride);
size_t predict_lds_bytes(int max_lines, int ss_stride);
size_t predict_lds_bytes(int max_lines, int ss_stride, PredictCaps caps);
size_t predict_lds_bytes_min(int max_lines, int ss_stride);          // without the staged text (which moves to global memory when it does not fit)
size_t predict_lds_bytes_min(int max_lines, int ss_stride, PredictCaps caps);
hipError_t launch_predict(hipStream_t stream, int grid, const MirpWindow* windows, int n_windows, const MirpMature* matures,
                          const MirpAln* alns, long long n_alns, const MirpFoldLine* lines, const char* ss, int ss_stride, int max_lines,
                          const int* n_lines, MirpPredictParams pp, MirpMirna* out, int* n_out, int* status, unsigned int* rcount = nullptr,
                          int* rpool = nullptr, unsigned int rcap = 0, int rstride = 0, const int* wsel = nullptr, int n_sel = 0,
                          const int* skip = nullptr, const int* wslot = nullptr, const PredictCaps* caps = nullptr, int* need = nullptr);
// launch + re-run of the windows that exceeded a capacity, with capacities sized for them (predict_kernel.hip)
int run_predict_launch(hipStream_t stream, int n_cu, const MirpWindow* windows, int n_windows, const MirpMature* matures, const MirpAln* alns, long long n_alns,
                       const MirpFoldLine* lines, const char* ss, int ss_stride, int max_lines, const int* n_lines, MirpPredictParams pp, MirpMirna* out, int* n_out,
                       int* status, unsigned int* rcount, int* rpool, unsigned int rcap, int rstride, const int* wsel, int n_sel, const int* skip, std::string* err,
                       int* need_buf = nullptr);

// fold_lds_kernel.hip
size_t fold_lds_bytes(int max_lines);
size_t fold_lds_epilogue_bytes(int max_lines);
size_t fold_lds_slab_shorts(int n_cap);
#ifdef MIRP_EPI_CLOCKS
void fold_lds_epi_clocks_print();
#endif
int fold_lds_max_n();
int fold_lds_gen_wing_d();
int fold_lds_max_span();
hipError_t launch_fold_lds(hipStream_t stream, int model, int grid, int grid_epi, const FoldParams* P, const unsigned char* seqs, const long long* offs, const int* lens,
                           int n_work, int win_base, int span, short* slabs, size_t slab_shorts, int* win_state, unsigned int* work_counter, int* fallback_list,
                           unsigned int* fallback_count, int max_lines, int ss_stride, MirpFoldLine* out_lines, char* out_ss, int* out_nlines, int* out_mfe,
                           int* out_status, int dbg_flags, long long* dbg_cycles, hipEvent_t ev_between, int* dense_list, int force_dense);

// fold_lds2_kernel.hip: fill kernel of the default model, two diagonals per barrier interval
size_t fold_lds2_bytes();
#ifdef MIRP_L2_CLOCKS
void fold_lds2_clocks_print();
#endif
hipError_t launch_fold_lds2_fill(hipStream_t stream, int grid, const FoldParams* P, const unsigned char* seqs, const long long* offs, const int* lens, int n_work,
                                 int win_base, int span, short* slabs, size_t slab_shorts, int* win_state, unsigned int* work_counter, int* fallback_list,
                                 unsigned int* fallback_count, int* out_nlines, int* out_mfe, int* out_status);

// candidate_kernels.hip
void launch_cov_scatter(hipStream_t st, const MirpAln* alns, long long n, const long long* goff, const long long* clen, int cutoff, int* diff_p, int* diff_m);
void launch_cov_unscatter(hipStream_t st, const MirpAln* alns, long long n, const long long* goff, const long long* clen, int* diff_p, int* diff_m);
long long cov_scan_tiles(long long gtot);
size_t run_start_bytes();
int cov_scan_tile_positions();
void launch_cov_maxlen(hipStream_t st, const MirpAln* alns, long long n, int* out);
size_t cov_fused_aux_bytes(long long gtot);
hipError_t launch_cov_scan_fused(hipStream_t st, const MirpAln* alns, long long n, int max_len /* longest record */, const long long* goff, const long long* clen, int n_contigs, void* aux, int* diff_p,
                                 int* diff_m, long long gtot, int cutoff, unsigned long long* stat_d, unsigned long long* stat_c, unsigned int* ticket, void* starts,
                                 long long starts_cap, MirpDepthPos* depth_out, long long depth_cap, long long* depth_gx, unsigned long long* totals);
void launch_cov_scan(hipStream_t st, const int* diff_p, const int* diff_m, long long gtot, int cutoff, unsigned long long* stat_d,
                     unsigned long long* stat_c, unsigned int* ticket, void* starts, long long starts_cap, MirpDepthPos* depth_out,
                     long long depth_cap, long long* depth_gx, unsigned long long* totals);
void launch_run_walk(hipStream_t st, const void* starts, long long n_runs, const int* diff_p, const int* diff_m, long long gtot, int cutoff,
                     const long long* goff, int n_contigs, int min_len, MirpPeak* runs, int* keep, int first_run_double);
void launch_depth_fix(hipStream_t st, MirpDepthPos* d, const long long* gx, long long n, const long long* goff, int n_contigs);
void launch_excl_scan(hipStream_t st, const int* in, long long* out, long long n);      // out[n + 1]; arrays beyond 16,384 elements: many-workgroup look-back scan
void release_scan_scratch(hipStream_t st);                                              // frees that scan's per-stream descriptors (before the stream is destroyed)
void launch_peak_compact(hipStream_t st, const MirpPeak* runs, const int* keep, const long long* kscan, long long n_runs, int n_contigs,
                         const int* order, long long* csq, long long* cdest, MirpPeak* peaks_sq, MirpPeak* peaks_sorted);
void launch_region_head(hipStream_t st, const MirpPeak* P, long long n, int max_gap, int* head);
void launch_region_first(hipStream_t st, const int* head, const long long* hscan, long long n, long long* rfirst);
void launch_region_count(hipStream_t st, const MirpPeak* P, const long long* rfirst, long long n_regions, const long long* clen, int L,
                         int* n_entries, int* is_locus, int* n_slots);
void launch_region_emit(hipStream_t st, const MirpPeak* P, const long long* rfirst, long long n_regions, const long long* clen, int L,
                        const long long* escan, const long long* lscan, const long long* sscan, MirpWindow* W, MirpLocus* loci, MirpPeak* wpeaks,
                        int* roles, int seq_stride);
void launch_window_payload(hipStream_t st, MirpWindow* W, long long n_windows, const MirpPeak* P, const MirpAln* alns, long long n_alns,
                           const unsigned char* genome, const long long* gboff, const long long* clen, double min_mature_depth, int wmax, char* seqs,
                           MirpMature* matures, long long* first_rec /* scratch, n_windows entries */, int* rt_out = nullptr);

}  // namespace mirp
