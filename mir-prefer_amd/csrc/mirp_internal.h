// Internal declarations shared by the HIP kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/mirprefer.h"
#include "fold_params.h"

namespace mirp {

size_t fold_generic_lds_bytes(int n_cap, int max_lines);
size_t fold_generic_ws_slot_ints(int n_cap, int span);
void launch_fold_generic(hipStream_t stream, int grid, const FoldParams* P, const unsigned char* seqs, const long long* offs,
                         const int* work_list, int n_work, int span, int n_cap, int* ws, size_t ws_slot_ints, int max_lines,
                         int ss_stride, MirpFoldLine* out_lines, char* out_ss, int* out_nlines, int* out_mfe, int* out_status);

size_t predict_lds_bytes(int max_lines, int ss_stride);
hipError_t launch_predict(hipStream_t stream, int grid, const MirpWindow* windows, int n_windows, const MirpMature* matures,
                          const MirpAln* alns, long long n_alns, const MirpFoldLine* lines, const char* ss, int ss_stride, int max_lines,
                          const int* n_lines, MirpPredictParams pp, MirpMirna* out, int* n_out, int* status);

}  // namespace mirp
