// Host-side construction of the device energy model from the generated Turner-2004 tables.
#include "fold_params.h"
#include "energy_params_t2004.h"
#include "energy_params_t1999.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

static inline int clamp0(int v) { return v > 0 ? 0 : v; }

template <class PP>
static void fill_gen_key(PP* p) {
    for (int u = 6; u <= MIRP_MAXLOOP; u++)
        for (int n1 = 0; n1 < 32; n1++) {          // n1 runs up to u - 2 = 28
            unsigned v = 65535u << 10;
            if (n1 >= 2 && n1 <= u - 2) {
                int y = std::abs(2 * n1 - u) * p->ninio;
                v = ((unsigned)(p->internal_loop[u] + (y < p->MAX_NINIO ? y : p->MAX_NINIO)) << 10) | (unsigned)(n1 << 5 | (u - n1));
            }
            p->gen_key[u - 6][n1] = v;
        }
    for (int u = 6; u <= MIRP_MAXLOOP; u++) {
        for (int m = 0; m < 32; m++) p->gen_key2[u - 6][m] = m + 2 < 32 ? p->gen_key[u - 6][m + 2] : 65535u << 10;
        for (int x = 0; x < 4; x++) p->gen_keyt[u - 6][x] = u - 5 + x >= 0 ? p->gen_key[u - 6][u - 5 + x] : 65535u << 10;
    }
}

static void fill_derived(FoldParams* p) {
    fill_gen_key(p);
    p->gen_wing_d = p->ninio > 0 ? (p->MAX_NINIO + p->ninio - 1) / p->ninio : 1 << 20;
    for (int u = 6; u <= MIRP_MAXLOOP; u++) p->gen_wing_key[u - 6] = ((unsigned)(p->internal_loop[u] + p->MAX_NINIO) << 10) | 63u;
    // inner-pair terms of the bulge and 1 x n candidates relative to the ring entry (fold_lds_common.h: XB = TerminalAU - mismatchI, X1 = mismatch1nI -
    // mismatchI of the inner pair seen from inside): stored as bytes, so their ranges become a bias that the candidates' size-term keys take back
    {
        static const int ptype[5][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 5}, {0, 0, 0, 1, 0}, {0, 0, 2, 0, 3}, {0, 6, 0, 4, 0}};      // [S[p]][S[q]], A C G U = 1 2 3 4
        static const int rt[8] = {0, 2, 1, 4, 3, 6, 5, 7};
        int lo_b = 0, hi_b = 0, lo_1 = 0, hi_1 = 0;
        for (int sp = 1; sp <= 4; sp++)
            for (int sq = 1; sq <= 4; sq++) {
                const int t2 = rt[ptype[sp][sq]];
                if (!t2) continue;
                for (int a = 0; a < 5; a++)
                    for (int b = 0; b < 5; b++) {
                        const int mi = p->mismatchI[t2][a][b];
                        const int xb = (t2 > 2 ? p->TerminalAU : 0) - mi, x1 = p->mismatch1nI[t2][a][b] - mi;
                        lo_b = xb < lo_b ? xb : lo_b; hi_b = xb > hi_b ? xb : hi_b; lo_1 = x1 < lo_1 ? x1 : lo_1; hi_1 = x1 > hi_1 ? x1 : hi_1;
                    }
            }
        p->xb_bias = -lo_b; p->x1_bias = -lo_1;
        if (hi_b - lo_b > 255 || hi_1 - lo_1 > 255 || p->xb_bias > 2047 || p->x1_bias > 2047) p->xb_bias = p->x1_bias = -1;      // (mirp_create refuses the set)
    }
    for (int u = 0; u <= MIRP_MAXLOOP; u++) {
        const unsigned kb = (unsigned)(p->bulge[u] + 2048 - (p->xb_bias > 0 ? p->xb_bias : 0)) << 10;
        p->kb0_key[u] = kb | (unsigned)u;
        p->kb1_key[u] = kb | (unsigned)(u << 5);
        int y = (u - 1) * p->ninio;
        const unsigned k1 = (unsigned)((u >= 1 && u + 1 <= MIRP_MAXLOOP ? p->internal_loop[u + 1] : 0) + (y < p->MAX_NINIO ? y : p->MAX_NINIO) + 2048 - (p->x1_bias > 0 ? p->x1_bias : 0)) << 10;
        p->k1n0_key[u] = k1 | (unsigned)(1 << 5 | u);
        p->k1n1_key[u] = k1 | (unsigned)(u << 5 | 1);
    }
    for (int rr = 0; rr < 32; rr++)
        for (int u = 0; u < 32; u++) p->ring_rowoff[rr][u] = ((rr - u) & 31) * MIRP_RING_CSTR;
}

void mirp_fill_fold_params(FoldParams* p) {
    std::memset(p, 0, sizeof(*p));
    std::memcpy(p->stack, T04_stack, sizeof(p->stack));
    std::memcpy(p->bulge, T04_bulge, sizeof(p->bulge));
    std::memcpy(p->internal_loop, T04_internal_loop, sizeof(p->internal_loop));
    std::memcpy(p->mismatchI, T04_mismatchI, sizeof(p->mismatchI));
    std::memcpy(p->mismatchH, T04_mismatchH, sizeof(p->mismatchH));
    std::memcpy(p->mismatch1nI, T04_mismatch1nI, sizeof(p->mismatch1nI));
    std::memcpy(p->mismatch23I, T04_mismatch23I, sizeof(p->mismatch23I));
    for (int t = 0; t < 8; t++)
        for (int a = 0; a < 5; a++) {
            p->dangle5[t][a] = clamp0(T04_dangle5[t][a]);
            p->dangle3[t][a] = clamp0(T04_dangle3[t][a]);
            for (int b = 0; b < 5; b++) {
                p->mismatchM[t][a][b] = clamp0(T04_mismatchM[t][a][b]);
                p->mismatchExt[t][a][b] = clamp0(T04_mismatchExt[t][a][b]);
            }
        }
    std::memcpy(p->int11, T04_int11, sizeof(p->int11));
    std::memcpy(p->int21, T04_int21, sizeof(p->int21));
    std::memcpy(p->int22, T04_int22, sizeof(p->int22));
    for (int u = 0; u < MIRP_HP_MAX; u++)
        p->hairpinE[u] = (u <= 30) ? T04_hairpin[u] : T04_hairpin[30] + (int)(T04_LXC * std::log((double)u / 30.));
    for (int k = 0; k < T04_N_TETRALOOPS; k++) { std::strncpy(p->tetra[k], T04_Tetraloops[k], 7); p->tetraE[k] = T04_Tetraloop_E[k]; }
    for (int k = 0; k < T04_N_TRILOOPS; k++) { std::strncpy(p->tri[k], T04_Triloops[k], 7); p->triE[k] = T04_Triloop_E[k]; }
    for (int k = 0; k < T04_N_HEXALOOPS; k++) { std::strncpy(p->hexa[k], T04_Hexaloops[k], 11); p->hexaE[k] = T04_Hexaloop_E[k]; }
    p->ML_closing = T04_ML_closing;
    p->ML_intern = T04_ML_intern;
    p->TerminalAU = T04_TerminalAU;
    p->ninio = T04_ninio;
    p->MAX_NINIO = T04_MAX_NINIO;
    p->n_tri = T04_N_TRILOOPS; p->n_tetra = T04_N_TETRALOOPS; p->n_hexa = T04_N_HEXALOOPS;
    fill_derived(p);
}

void mirp_fill_fold_params185(FoldParams185* p) {
    std::memset(p, 0, sizeof(*p));
    std::memcpy(p->stack, T99_stack, sizeof(p->stack));
    std::memcpy(p->bulge, T99_bulge, sizeof(p->bulge));
    std::memcpy(p->internal_loop, T99_internal_loop, sizeof(p->internal_loop));
    std::memcpy(p->mismatchI, T99_mismatchI, sizeof(p->mismatchI));
    std::memcpy(p->mismatchH, T99_mismatchH, sizeof(p->mismatchH));
    for (int t = 0; t < 8; t++)
        for (int a = 0; a < 5; a++) { p->dangle5[t][a] = clamp0(T99_dangle5[t][a]); p->dangle3[t][a] = clamp0(T99_dangle3[t][a]); }
    std::memcpy(p->int11, T99_int11, sizeof(p->int11));
    std::memcpy(p->int21, T99_int21, sizeof(p->int21));
    std::memcpy(p->int22, T99_int22, sizeof(p->int22));
    for (int u = 0; u < MIRP_HP_MAX; u++)
        p->hairpinE[u] = (u <= 30) ? T99_hairpin[u] : T99_hairpin[30] + (int)(T99_LXC * std::log((double)u / 30.));
    p->n_tetra = T99_N_TETRALOOPS;
    for (int k = 0; k < T99_N_TETRALOOPS; k++) { std::strncpy(p->tetra[k], T99_Tetraloops[k], 7); p->tetraE[k] = T99_Tetraloop_E[k]; }
    p->ML_closing = T99_ML_closing;
    p->ML_intern = T99_ML_intern;
    p->TerminalAU = T99_TerminalAU;
    p->ninio = T99_ninio;
    p->MAX_NINIO = T99_MAX_NINIO;
    for (int u = 0; u <= MIRP_MAXLOOP; u++)
        for (int n1 = 0; n1 < 32; n1++) {
            const int y = std::abs(2 * n1 - u) * p->ninio;
            p->gen_e[u][n1] = ((p->internal_loop[u] + (y < p->MAX_NINIO ? y : p->MAX_NINIO)) << 10) | ((n1 & 31) << 5) | ((u - n1) & 31);
        }
    for (int u = 0; u <= MIRP_MAXLOOP; u++) {
        for (int m = 0; m < 32; m++) p->gen_e1[u][m] = p->gen_e[u][m + 1 < 32 ? m + 1 : 31];
        for (int x = 0; x < 4; x++) p->gen_et[u][x] = p->gen_e[u][u - 4 + x >= 0 ? u - 4 + x : 0];
    }
}

void mirp_fill_fold_params_t1999(FoldParams* p) {
    std::memset(p, 0, sizeof(*p));
    std::memcpy(p->stack, T99_stack, sizeof(p->stack));
    std::memcpy(p->bulge, T99_bulge, sizeof(p->bulge));
    std::memcpy(p->internal_loop, T99_internal_loop, sizeof(p->internal_loop));
    std::memcpy(p->mismatchI, T99_mismatchI, sizeof(p->mismatchI));
    std::memcpy(p->mismatch1nI, T99_mismatchI, sizeof(p->mismatch1nI));
    std::memcpy(p->mismatch23I, T99_mismatchI, sizeof(p->mismatch23I));
    std::memcpy(p->mismatchH, T99_mismatchH, sizeof(p->mismatchH));
    for (int t = 0; t < 8; t++)
        for (int a = 0; a < 5; a++) { p->dangle5[t][a] = clamp0(T99_dangle5[t][a]); p->dangle3[t][a] = clamp0(T99_dangle3[t][a]); }
    std::memcpy(p->int11, T99_int11, sizeof(p->int11));
    std::memcpy(p->int21, T99_int21, sizeof(p->int21));
    std::memcpy(p->int22, T99_int22, sizeof(p->int22));
    for (int u = 0; u < MIRP_HP_MAX; u++)
        p->hairpinE[u] = (u <= 30) ? T99_hairpin[u] : T99_hairpin[30] + (int)(T99_LXC * std::log((double)u / 30.));
    p->n_tri = 0; p->n_hexa = 0; p->n_tetra = T99_N_TETRALOOPS;
    for (int k = 0; k < T99_N_TETRALOOPS; k++) { std::strncpy(p->tetra[k], T99_Tetraloops[k], 7); p->tetraE[k] = T99_Tetraloop_E[k]; }
    p->ML_closing = T99_ML_closing;
    p->ML_intern = T99_ML_intern;
    p->TerminalAU = T99_TerminalAU;
    p->ninio = T99_ninio;
    p->MAX_NINIO = T99_MAX_NINIO;
    fill_derived(p);
}
