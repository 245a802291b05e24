/* oracle/oracle.h -- TEST INFRASTRUCTURE (CPU restatement of the reference hot path). Not product code. */
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_LINES 384
#define ORACLE_MAX_SS 3200   /* lines of PRECURSOR_LEN up to 3000 (+50 +dangles), the upper limit of the reference (MP:167-184) */

typedef struct {
    char ss[ORACLE_MAX_SS]; /* dot-bracket text exactly as RNALfold prints it (incl. dangle dots) */
    int len;
    int energy;             /* 0.01 kcal/mol; RNALfold prints energy/100 with %6.2f */
    int start;              /* 1-based start column as printed */
} OracleFoldLine;

typedef struct {
    int n_lines;
    int overflow;
    int mfe;                /* f3[1], 0.01 kcal/mol */
    OracleFoldLine lines[ORACLE_MAX_LINES];
} OracleFoldResult;

/* Any number of lines of any length (PRECURSOR_LEN up to 3000): the same folds with the lines as text, "structure energy start\n" each, in a malloc'd
   buffer the caller frees with oracle_free_text; model 0 = vienna-2.1.2, 1 = vienna-1.8.5. 0 = ok. */
typedef struct { char *buf; size_t len, cap; int n_lines; } OracleTextSink;
int oracle_lfold_text(const char *seq, int n, int span, int model, char **text, int *n_lines, int *mfe);
void oracle_free_text(char *text);
static inline void oracle_sink_add(OracleTextSink *k, const char *lead, const char *body, int energy, int start) {
    const size_t need = strlen(lead) + strlen(body) + 40;
    if (k->len + need > k->cap) { k->cap = (k->cap + need) * 2; k->buf = (char *)realloc(k->buf, k->cap); }
    k->len += (size_t)sprintf(k->buf + k->len, "%s%s %d %d\n", lead, body, energy, start);
    k->n_lines++;
}

/* RNALfold -L span on one sequence (vienna-2.1.2 flavour: Turner-2004, dangles=2). 0 = ok. */
int oracle_lfold(const char *seq, int n, int span, OracleFoldResult *out);
/* the same for the vienna-1.8.5 flavour (Turner-1999, dangles=1, multi-component structure strings). 0 = ok. */
int oracle_lfold185(const char *seq, int n, int span, OracleFoldResult *out);


/* ---- candidate stage (oracle/candidate.c) ---- */
typedef struct { int32_t tid, pos; uint32_t depth; uint16_t len; uint8_t strand, sample; } OracleAln; /* pos 1-based */
typedef struct { int32_t tid, pos, dp, dm; } OracleDepthPos;          /* one line of bam.depth.cut<CUT> */
typedef struct { int32_t tid, start, end, strand; } OraclePeak;         /* [start,end) 1-based, strand 0 '+', 1 '-' */
typedef struct { int32_t start, end, strand, depth; } OracleMature;     /* strand -1 = the (0,0,0,0) fallback */
typedef struct {
    int32_t tid, ws, we, strand, loc_s, loc_e, tag; /* tag 0 / 1='L' / 2='R' */
    int32_t n_peaks; int64_t peak_off;
    int32_t n_matures; int64_t mature_off;
    int64_t seq_off; int32_t seq_len;
} OracleWindow;
typedef struct { int32_t tid, start, end, n_windows; int32_t w[2][2]; int64_t peak_first; int32_t n_peaks; } OracleLocus;

int oracle_coverage_peaks(const OracleAln *a, size_t n, const int64_t *contig_len, int n_contigs, int cutoff, int min_len,
                          OracleDepthPos **depth_out, size_t *n_depth, OraclePeak **peaks_out, size_t *n_peaks);
int oracle_make_windows(const OraclePeak *peaks, size_t n_peaks, const OracleAln *a, size_t na, const char *const *genome,
                        const int64_t *contig_len, int n_contigs, const int *contig_order, int max_gap, int precursor_len,
                        double min_mature_depth, OracleWindow **w_out, size_t *nw_out, OraclePeak **wpeaks_out, size_t *nwpeaks,
                        OracleMature **mat_out, size_t *nmat, char **seq_out, size_t *nseq, OracleLocus **loci_out, size_t *nloci_out);
void oracle_free(void *p);

/* ---- predict stage (oracle/predict.c) ---- */
#define ORACLE_MAX_STRUCTS 512
typedef struct { double norm_energy; int32_t fold_start; int32_t sstype; int32_t len; char ss[ORACLE_MAX_SS]; } OracleStruct;
/* a8: structures of one window from its RNALfold lines */
int oracle_structures(const OracleFoldLine *lines, int n_lines, int minlen, OracleStruct *out, int max_out);

/* fail codes of get_maturestar_info (a9); 0 = ok */
enum {
    MS_OK = 0, MS_FAIL_MATCHED_BASES, MS_FAIL_NOT_IN_FOLD_REGION, MS_FAIL_NOT_IN_ONE_ARM, MS_FAIL_MATCH_LT_14,
    MS_FAIL_OVERLAP, MS_FAIL_STAR_OUT_OF_FOLD, MS_FAIL_STAR_NOT_IN_ONE_ARM, MS_FAIL_TOO_MANY_BULGE_OR_LOOP,
    MS_FAIL_MAX_BULGE_GT_2, MS_FAIL_TOTAL_LOOP_GT_5, MS_FAIL_NUM_BULGE_GT_2, MS_REFERENCE_EXCEPTION
};
typedef struct { int32_t code; int32_t star_s, star_e, fold_s, fold_e; int32_t prime5, total_dots, total_bps; int32_t star_l0, star_l1, mat_l0, mat_l1; } OracleMatureStar;
int oracle_duplex_code(const char *mature, int ml, const char *star, int sl);   /* stat_duplex + pass_stat_duplex alone: MS_* code */
int oracle_maturestar(const char *ss, int len, int m0, int m1, int foldstart, int regionstart, int regionend, int strand, OracleMatureStar *out);

#define ORACLE_MAX_SAMPLES 256
typedef struct {
    int32_t reads_pre[ORACLE_MAX_SAMPLES], reads_mature[ORACLE_MAX_SAMPLES], reads_star[ORACLE_MAX_SAMPLES], reads_antisense[ORACLE_MAX_SAMPLES];
    int32_t reads_isoform[ORACLE_MAX_SAMPLES], reads_inside[ORACLE_MAX_SAMPLES], bases_with_reads_start[ORACLE_MAX_SAMPLES];
    int32_t imperfect[ORACLE_MAX_SAMPLES][3];
    int64_t total_this_strand, total_anti, total_mature, total_isoform, total_star /* after max with imperfect */, total_star_perfect;
    int64_t total_imperfect[3];
    int32_t mature_star_distance;
    int32_t has_imperfect_key;   /* 'max_imperfect_star' in dict */
    int32_t imperfect_which, imperfect_start, imperfect_end; int64_t max_imperfect;
    double ratio_total, ratio_both, ratio_iso;
    double ratio_start[ORACLE_MAX_SAMPLES];
    int32_t exception;           /* reference would raise (ZeroDivisionError) */
} OracleExpr;
int oracle_expression(const OracleAln *a, size_t na, int n_samples, int tid, int ws, int we, int fold_s, int fold_e, int m0, int m1,
                      int star_s, int star_e, int strand, int allow_3nt, OracleExpr *out);

typedef struct {
    int32_t window;              /* index of the FASTA entry that produced the record */
    int32_t tid, fold_s, fold_e, mat_s, mat_e, star_s, star_e, strand, has_star;
    int32_t ss_len; char ss[ORACLE_MAX_SS];
    int64_t total_depth_mature, total_depth_star;
} OracleMirna;
typedef struct { int32_t n_samples, min_mature_len, max_mature_len, allow_3nt, allow_no_star, minlen; } OraclePredictParams;
/* a11 for one window: returns number of miRNA records written (first = the one the reference keeps), 0 = failed */
int oracle_check_loci(const OracleStruct *st, int n_st, const OracleMature *mat, int n_mat, const OracleWindow *w, const OracleAln *a,
                      size_t na, const OraclePredictParams *pp, OracleMirna *out, int max_out);

#ifdef __cplusplus
}
#endif
