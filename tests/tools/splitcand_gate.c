/* tests/tools/splitcand_gate.c -- DEV TOOL / TEST INFRASTRUCTURE (uses the CPU oracle's tables; never part of the product).
 *
 * Host-side gate for an exact candidate-list sparsification of the multiloop split of the local fold
 * (RNALfold fill, SURVEY.md App. B "Fill": DML(a,b) = min_k fML[a][k] + fML[k+1][b]; reference call site
 * /root/reference/miR_PREFeR.py:3053).  With ML_BASE = 0:
 *     DML(i,j) = min( DML(i,j-1), min_{s in Cand(j), i+TURN+2 <= s <= j-TURN-1} fML[i][s-1] + fML[s][j] )
 * where Cand(j) = { s : fML[s][j] is realised STRICTLY by its pair term } (strictly below fML[s+1][j], fML[s][j-1], DML(s,j)).
 * The tool (1) checks that identity cell by cell against the dense minimum, and (2) counts what a GPU kernel would pay:
 * list sizes, candidates visited per cell, and the per-wave maximum (a 64-lane wave runs as long as its longest list).
 *
 *   gcc -O2 -o /tmp/splitcand_gate tests/tools/splitcand_gate.c -lm && /tmp/splitcand_gate < windows.txt
 */
#include "../../oracle/lfold.c"

double g_colbp, g_rowbp, g_both; long g_maxcolbp;
int main(int argc, char **argv) {
    int span = argc > 1 ? atoi(argv[1]) : 300;
    char line[4096];
    double tot_dense = 0, tot_sparse = 0, tot_cells = 0, tot_cand = 0, tot_paired = 0, tot_wave64 = 0, tot_wave_dense = 0, tot_list_at = 0;
    double tot_cols = 0; long maxlist = 0, maxtotal = 0; long nwin = 0, mism = 0;
    double hist_d[400] = {0}, hist_dn[400] = {0};
    long hmax[64] = {0}, htot[64] = {0};
    while (fgets(line, sizeof line, stdin)) {
        int n = (int)strlen(line);
        while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        if (n < 10) continue;
        Fold F; F.n = n; F.M = span;
        F.seq = (char *)calloc(n + 16, 1); F.S = (int *)calloc(n + 2, sizeof(int));
        for (int i = 1; i <= n; i++) {
            char ch = (char)toupper((unsigned char)line[i - 1]); if (ch == 'T') ch = 'U';
            F.seq[i] = ch; F.S[i] = ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'U' ? 4 : 0;
        }
        F.S[0] = F.S[n]; F.S[n + 1] = F.S[1];
        size_t cells = (size_t)(n + 2) * (span + 2);
        F.c = (int *)malloc(cells * sizeof(int)); F.fML = (int *)malloc(cells * sizeof(int));
        F.pt = (unsigned char *)calloc(cells, 1); F.f3 = (int *)calloc(n + span + 8, sizeof(int));
        for (size_t x = 0; x < cells; x++) F.c[x] = F.fML[x] = INF;
        for (int i = 1; i <= n; i++)
            for (int j = i + TURN + 1; j <= n && j - i <= span - 1; j++) F.pt[IDX(&F, i, j)] = (unsigned char)PAIR[F.S[i]][F.S[j]];
        fill(&F);
        /* candidates */
        unsigned char *cand = (unsigned char *)calloc(cells, 1);
        int *dml = (int *)malloc(cells * sizeof(int));
        long total = 0; static double tot_colbp = 0, tot_rowbp = 0, tot_both = 0; static long maxcolbp = 0;
        long colbp = 0, rowbp = 0, both = 0;
        for (int i = 1; i <= n; i++)
            for (int j = i + TURN + 1; j <= n && j <= i + span; j++) {
                int v = mget(&F, i, j), d = DML(&F, i, j);
                dml[IDX(&F, i, j)] = d;
                int other = imin(imin(mget(&F, i + 1, j), mget(&F, i, j - 1)), d);
                if (v < other && v < INF / 2) { cand[IDX(&F, i, j)] = 1; total++; }
                { int up = mget(&F, i + 1, j), lf = mget(&F, i, j - 1); if (v < INF / 2) { if (v < up) colbp++; if (v < lf) rowbp++; if (v < up && v < lf) both++; } }
                tot_cells++;
                if (ptype(&F, i, j)) tot_paired++;
            }
        tot_colbp += colbp; tot_rowbp += rowbp; tot_both += both; if (colbp > maxcolbp) maxcolbp = colbp;
        if (feof(stdin) || 1) { extern double g_colbp, g_rowbp, g_both; extern long g_maxcolbp; g_colbp = tot_colbp; g_rowbp = tot_rowbp; g_both = tot_both; g_maxcolbp = maxcolbp; }
        tot_cand += total; if (total > maxtotal) maxtotal = total;
        long wmaxcol = 0;
        for (int j = 1; j <= n; j++) { long c = 0; for (int s = 1; s < j; s++) if (j - s <= span && j - s > TURN && cand[IDX(&F, s, j)]) c++; if (c > maxlist) maxlist = c; if (c > wmaxcol) wmaxcol = c; tot_cols++; }
        hmax[wmaxcol > 63 ? 63 : wmaxcol]++; htot[total / 256 > 63 ? 63 : total / 256]++;
        /* per-cell sparse work + check of the identity, diagonal by diagonal */
        for (int d = TURN + 1; d <= span && d < n; d++) {
            int ncell = n - d;
            long wmax = 0; int lanes = 0;
            for (int i = 1; i <= ncell; i++) {
                int j = i + d;
                long cnt = 0; int best = (d - 1 > TURN) ? dml[IDX(&F, i, j - 1)] : INF;
                if (best > INF) best = INF;
                for (int s = i + TURN + 2; s <= j - TURN - 1; s++)
                    if (cand[IDX(&F, s, j)]) { cnt++; best = imin(best, mget(&F, i, s - 1) + mget(&F, s, j)); }
                int dense = dml[IDX(&F, i, j)];
                /* equality where either is finite */
                if ((dense < INF / 2 || best < INF / 2) && dense != best) { if (mism < 5) fprintf(stderr, "MISMATCH n=%d i=%d j=%d dense=%d sparse=%d\n", n, i, j, dense, best); mism++; }
                tot_sparse += cnt; hist_d[d] += cnt; hist_dn[d] += 1;
                if (d > 8) tot_dense += d - 8;
                if (cnt > wmax) wmax = cnt;
                lanes++;
                if (lanes == 64 || i == ncell) { tot_wave64 += (double)wmax * 64; tot_wave_dense += (d > 8 ? (d - 8) : 0) * 64.0; wmax = 0; lanes = 0; }
            }
        }
        nwin++;
        free(cand); free(dml); free(F.seq); free(F.S); free(F.c); free(F.fML); free(F.pt); free(F.f3);
    }
    printf("windows %ld  mismatches %ld\n", nwin, mism);
    printf("cells/window %.0f  paired %.0f (%.3f)  candidates %.0f (%.4f of cells, %.3f of paired)  max per window %ld\n", tot_cells / nwin, tot_paired / nwin,
           tot_paired / tot_cells, tot_cand / nwin, tot_cand / tot_cells, tot_cand / tot_paired, maxtotal);
    printf("column breakpoints (fML(i,j) < fML(i+1,j)) %.0f per window (max %ld), row breakpoints (< fML(i,j-1)) %.0f, both %.0f\n", g_colbp / nwin, g_maxcolbp, g_rowbp / nwin, g_both / nwin);
    printf("mean |Cand(j)| %.2f  max |Cand(j)| %ld\n", tot_cand / tot_cols, maxlist);
    printf("dense splits/window %.0f   sparse visits/window %.0f   ratio %.4f\n", tot_dense / nwin, tot_sparse / nwin, tot_sparse / tot_dense);
    printf("wave64 (max over lanes x 64): sparse %.0f  dense %.0f  ratio %.4f\n", tot_wave64 / nwin, tot_wave_dense / nwin, tot_wave64 / tot_wave_dense);
    printf("per-window max column count: "); for (int k = 0; k < 64; k++) if (hmax[k]) printf("%d:%ld ", k, hmax[k]); printf("\n");
    printf("per-window total candidates / 256: "); for (int k = 0; k < 64; k++) if (htot[k]) printf("%d:%ld ", k, htot[k]); printf("\n");
    for (int d = 20; d <= span; d += 40) printf("  d=%d: mean visits per cell %.2f (dense %d)\n", d, hist_dn[d] ? hist_d[d] / hist_dn[d] : 0, d - 8);
    return mism != 0;
}
