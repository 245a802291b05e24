// Device-side energy functions shared by the fill and backtrack phases of the local fold.
// Semantics: SURVEY.md Appendix B/B2 (observable behaviour of RNALfold 2.1.2, Turner-2004, d2);
// replaces the subprocess at /root/reference/miR_PREFeR.py:3053-3064.
#pragma once
#include <hip/hip_runtime.h>
#include "fold_params.h"

namespace mirp {

// pair type of bases (a,b) in 0..4 (N A C G U): CG=1 GC=2 GU=3 UG=4 AU=5 UA=6.
// 25 entries x 3 bits packed into two words (index = a*5+b).
__device__ __forceinline__ int pair_type(int a, int b) {
    // idx: A-U(1,4)=9 ->5 ; C-G(2,3)=13 ->1 ; G-C(3,2)=17 ->2 ; G-U(3,4)=19 ->3 ; U-A(4,1)=21 ->6 ; U-G(4,3)=23 ->4
    const unsigned long long lo = (5ull << 27) | (1ull << 39) | (2ull << 51) | (3ull << 57); // idx 0..20 (3 bits each)
    const unsigned int hi = (6u << 0) | (4u << 6);                                          // idx 21..24
    int idx = a * 5 + b;
    return idx < 21 ? (int)((lo >> (3 * idx)) & 7ull) : (int)((hi >> (3 * (idx - 21))) & 7u);
}

__device__ __forceinline__ int rtype_of(int t) {
    // {0,2,1,4,3,6,5,7}
    return (int)((0x75634120u >> (4 * t)) & 15u);
}

__device__ __forceinline__ int e_mlstem(const FoldParams* __restrict__ P, int type, int a, int b) {
    int e = P->ML_intern + (type > 2 ? P->TerminalAU : 0);
    if (a >= 0 && b >= 0) e += P->mismatchM[type][a][b];
    else if (a >= 0) e += P->dangle5[type][a];
    else if (b >= 0) e += P->dangle3[type][b];
    return e;
}

__device__ __forceinline__ int e_extloop(const FoldParams* __restrict__ P, int type, int a, int b) {
    int e = (type > 2 ? P->TerminalAU : 0);
    if (a >= 0 && b >= 0) e += P->mismatchExt[type][a][b];
    else if (a >= 0) e += P->dangle5[type][a];
    else if (b >= 0) e += P->dangle3[type][b];
    return e;
}

// type2 already rtype'd
__device__ __forceinline__ int e_intloop(const FoldParams* __restrict__ P, int n1, int n2, int type, int type2,
                                         int si1, int sj1, int sp1, int sq1) {
    int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1;
    if (nl == 0) return P->stack[type][type2];
    if (ns == 0) {
        int e = P->bulge[nl];
        if (nl == 1) e += P->stack[type][type2];
        else e += (type > 2 ? P->TerminalAU : 0) + (type2 > 2 ? P->TerminalAU : 0);
        return e;
    }
    if (ns == 1) {
        if (nl == 1) return P->int11[type][type2][si1][sj1];
        if (nl == 2) return (n1 == 1) ? P->int21[type][type2][si1][sq1][sj1] : P->int21[type2][type][sq1][si1][sp1];
        int x = (nl - 1) * P->ninio;
        return P->internal_loop[nl + 1] + (x < P->MAX_NINIO ? x : P->MAX_NINIO) + P->mismatch1nI[type][si1][sj1] +
               P->mismatch1nI[type2][sq1][sp1];
    }
    if (ns == 2) {
        if (nl == 2) return P->int22[type][type2][si1][sp1][sq1][sj1];
        if (nl == 3) return P->internal_loop[5] + P->ninio + P->mismatch23I[type][si1][sj1] + P->mismatch23I[type2][sq1][sp1];
    }
    int x = (nl - ns) * P->ninio;
    return P->internal_loop[nl + ns] + (x < P->MAX_NINIO ? x : P->MAX_NINIO) + P->mismatchI[type][si1][sj1] +
           P->mismatchI[type2][sq1][sp1];
}

}  // namespace mirp
