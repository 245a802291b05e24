// Device-side energy model for the L-bounded local fold (replaces the RNALfold subprocess,
// reference call site /root/reference/miR_PREFeR.py:3053-3064).  Turner-2004 nearest-neighbour
// parameters, "vienna-2.1.2" flavour (dangles = 2).  Integers in 0.01 kcal/mol.
#pragma once
#include <stdint.h>

#define MIRP_TURN 3
#define MIRP_MAXLOOP 30
#define MIRP_INF 10000000
#define MIRP_RING_CSTR 354        // row stride (shorts) of the fill kernel's c ring: CSTR of fold_lds_common.h (static_assert there)
#define MIRP_HP_MAX 3104          // hairpin size table (log-extrapolated above 30 on the host)

struct alignas(16) FoldParams {
    // derived, read with scalar loads by the LDS fill kernel (wave-uniform loop shapes).  An interior-loop candidate is ranked by the key
    // (energy term << 10) | (n1 << 5 | n2): the minimum key is the minimum energy and, among equal energies, the first shape in the
    // backtrack's search order (p ascending, q descending), which is what the trace-back code of the cell must name.
    // (also read by the generic kernel, fold_kernel.hip: one v_mad_i32_i24 per generic candidate)
    unsigned gen_key[25][32];     // generic loops [u-6][n1], 2 <= n1 <= u-2: (internal_loop[u] + min(MAX_NINIO, |2 n1 - u| ninio)) << 10 | n1 << 5 | (u - n1)
    unsigned kb0_key[31];         // bulge n1 = 0, n2 = u:   (bulge[u] + 2048) << 10 | u
    unsigned kb1_key[31];         // bulge n1 = u, n2 = 0:   (bulge[u] + 2048) << 10 | u << 5
    unsigned k1n0_key[31];        // 1 x k loop (n1 = 1, n2 = k): (internal_loop[k+1] + min(MAX_NINIO, (k-1) ninio) + 2048) << 10 | 1 << 5 | k
    unsigned k1n1_key[31];        // k x 1 loop (n1 = k, n2 = 1): ... | k << 5 | 1
    // generic loops whose asymmetry term is saturated (|n1 - n2| >= gen_wing_d: ninio |n1 - n2| >= MAX_NINIO): all of a row's share one
    // size / asymmetry term, so the fill kernel takes the minimum over their ring entries first and adds the term once.  That minimum names
    // no shape: code 63 = "some loop with n1 >= 2", which sorts after every shape with n1 <= 1 and before every other one (TB_GENERIC, fold_epilogue.h)
    unsigned gen_wing_key[25];    // [u-6]: (internal_loop[u] + MAX_NINIO) << 10 | 63
    int gen_wing_d;               // smallest |n1 - n2| with a saturated term: 5 with Turner-2004 (ninio 60, MAX_NINIO 300)
    // bias of the fill kernel's byte tables of inner-pair terms (LdsTables::XB / X1, fold_lds_common.h): table entry = term + bias in 0..255; the
    // kb*_key / k1n*_key tables above carry the matching -bias
    int xb_bias, x1_bias;
    // ring-row offsets of the fill kernel's c ring (fold_lds_common.h: diagonal dd lives in row dd & 31, CSTR shorts per row): [r0 & 31][u] = ((r0 - u) & 31) * CSTR.
    // The bulge / 1 x n jobs read a different ring row per candidate; with the row of the interval's r0 in SGPRs (one s_load burst per block) the
    // address is one vector add instead of s_add + s_and + s_mul + v_add per candidate.
    int ring_rowoff[32][32];
    // (the key tables come first so that the fill kernel's scalar loads reach them with immediate offsets from the one base pointer)
    int stack[8][8];
    int bulge[31];
    int internal_loop[31];
    int mismatchI[8][5][5];
    int mismatchH[8][5][5];
    int mismatchM[8][5][5];       // clamped <= 0
    int mismatch1nI[8][5][5];
    int mismatch23I[8][5][5];
    int mismatchExt[8][5][5];     // clamped <= 0
    int dangle5[8][5];            // clamped <= 0
    int dangle3[8][5];            // clamped <= 0
    int hairpinE[MIRP_HP_MAX];
    int tetraE[32], triE[2], hexaE[4];
    char tetra[32][8];
    char tri[2][8];
    char hexa[4][12];
    int ML_closing, ML_intern, TerminalAU, ninio, MAX_NINIO;
    int n_tri, n_tetra, n_hexa;   // motifs in use
    // (behind the tables of the LDS-resident fill kernel, whose layout -- and scalar-cache footprint -- they must not move)
    // the same terms as the generic kernel (fold_kernel.hip) reads them, four per scalar load: its chunks of four candidates start at n1 = 2, 6, ...
    alignas(16) unsigned gen_key2[25][32];    // [u-6][m] = gen_key[u-6][m + 2]
    alignas(16) unsigned gen_keyt[25][4];     // [u-6][x] = gen_key[u-6][u - 5 + x]: the last four generic shapes of a size, a chunk of their own
    // the three big tables last: everything above stays within the immediate-offset range of scalar loads
    int int11[8][8][5][5];
    int int21[8][8][5][5][5];
    int int22[8][8][5][5][5][5];
};

// Energy model of the "vienna-1.8.5" compatibility mode (Turner-1999 parameters as shipped in ViennaRNA 1.8.5, dangles = 1).
struct alignas(16) FoldParams185 {
    // derived, read with scalar loads by fold185_kernel's interior-loop interval (wave-uniform loop shapes): size + asymmetry term of the generic loops --
    // in this model every loop but stack, bulge, 1 x 1, 1 x 2, 2 x 2 --: internal_loop[u] + min(MAX_NINIO, |2 n1 - u| ninio)
    // -- as keys, term << 10 | n1 << 5 | n2 (the minimum key is the minimum energy and, among equal energies, the loop the backtrack's search finds first:
    // the trace-back code of the cell), and in the order the kernel's chunks of four candidates read them (one 16-byte scalar load each)
    int gen_e[31][32];            // the keys of size u, [u][n1]
    int gen_e1[31][32];           // [u][m] = gen_e[u][m + 1]: chunks start at n1 = 1
    int gen_et[31][4];            // [u][x] = gen_e[u][u - 4 + x]: the last four generic shapes of a size (n1 = u - 4 .. u - 1)
    int stack[8][8];
    int bulge[31];
    int internal_loop[31];
    int mismatchI[8][5][5];
    int mismatchH[8][5][5];
    int dangle5[8][5];            // clamped <= 0
    int dangle3[8][5];            // clamped <= 0
    int int11[8][8][5][5];
    int int21[8][8][5][5][5];
    int int22[8][8][5][5][5][5];
    int hairpinE[MIRP_HP_MAX];    // size table, log-extrapolated above 30 on the host
    int n_tetra;
    int tetraE[32];               // bonus added to the hairpin energy (not a total, unlike Turner-2004)
    char tetra[32][8];
    int ML_closing, ML_intern, TerminalAU, ninio, MAX_NINIO;
};
void mirp_fill_fold_params185(FoldParams185* p);

// Fills *p from the generated Turner-2004 tables (host side, mirp_params.cpp).
void mirp_fill_fold_params(FoldParams* p);
// Turner-1999 (ViennaRNA 1.8.5) values in the same layout for the LDS-resident kernels of the vienna-1.8.5 model: the 1 x n and 2 x 3
// loops of that model use the generic formula, i.e. mismatch1nI = mismatch23I = mismatchI; tetraE holds bonuses; mismatchM / mismatchExt unused.
void mirp_fill_fold_params_t1999(FoldParams* p);
