/*
 * mirprefer.h -- C-ABI of libmirprefer.so: the MI355X (gfx950) replacement for the subprocess
 * boundary of miR-PREFeR's candidate -> fold -> predict hot path.
 *
 * The reference has no FFI; the boundary it crosses is `samtools depth | awk`, `samtools faidx`,
 * `samtools view` and `RNALfold -L` subprocesses.  Each entry point below
 * names the reference interface (file:line in /root/reference/miR_PREFeR.py, "MP") it replaces.
 *
 * Conventions: plain C types only; inputs are caller-owned and borrowed for the call; outputs are
 * library-owned host buffers released with mirp_free(); every function returns 0 on success and a
 * negative code on error, with text available from mirp_last_error(); one context per device,
 * calls on a context are not re-entrant, different contexts are independent.
 */
#ifndef MIRPREFER_H
#define MIRPREFER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirp_ctx mirp_ctx;

/* One structure line of `RNALfold -L` output: "<ss> (<energy/100 %6.2f>) <start %4d>". */
typedef struct {
    int32_t start;   /* 1-based start column as RNALfold prints it */
    int32_t len;     /* strlen of the dot-bracket text (incl. dangle dots) */
    int32_t energy;  /* 0.01 kcal/mol */
    int32_t printed; /* 0 if RNALfold's containment rule suppresses the line */
} MirpFoldLine;

int mirp_create(int device, mirp_ctx** out);
void mirp_destroy(mirp_ctx* ctx);
const char* mirp_last_error(const mirp_ctx* ctx);
void mirp_free(void* p);
/* ABI version of this header; bumped on any signature change. */
int mirp_abi_version(void);

/*
 * Replaces: `RNALfold -L <span>` on a FASTA chunk (MP:3047-3119, command MP:3053, consumer MP:1541-1599).
 * seqs: concatenated sequence bytes (any case, T or U); offsets[n_seqs+1]: byte offsets into seqs.
 * Out (library-owned): lines[n_seqs*max_lines], ss[n_seqs*max_lines*ss_stride] (NUL-terminated texts),
 * n_lines[n_seqs], mfe[n_seqs] (0.01 kcal/mol, RNALfold's final " (%6.2f)" line), status[n_seqs]
 * (0 ok, 1 = more than max_lines structures, <0 = error for that sequence).
 */
int mirp_fold_batch(mirp_ctx* ctx, const char* seqs, const int64_t* offsets, int32_t n_seqs, int32_t span,
                    int32_t max_lines, MirpFoldLine** lines, char** ss, int32_t* ss_stride, int32_t** n_lines,
                    int32_t** mfe, int32_t** status);

#ifdef __cplusplus
}
#endif
#endif
