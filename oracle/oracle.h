/* oracle/oracle.h -- TEST INFRASTRUCTURE (CPU restatement of the reference hot path). Not product code. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_LINES 256
#define ORACLE_MAX_SS 512

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

/* RNALfold -L span on one sequence (vienna-2.1.2 flavour: Turner-2004, dangles=2). 0 = ok. */
int oracle_lfold(const char *seq, int n, int span, OracleFoldResult *out);

#ifdef __cplusplus
}
#endif
