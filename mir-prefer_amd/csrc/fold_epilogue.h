// Shared device code of the local-fold kernels: window context, hairpin / exterior / multiloop
// terms, the wave-cooperative backtrack and the f3 + enumeration + output epilogue.
// Semantics: SURVEY.md Appendix B/B2 (RNALfold 2.1.2, Turner-2004, d2); replaces the subprocess at
// /root/reference/miR_PREFeR.py:3053-3064.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "fold_device.h"
#include "mirp_internal.h"

namespace mirp {

#define TURN MIRP_TURN
#define MAXLOOP MIRP_MAXLOOP
#define INF MIRP_INF

// Optional LDS-resident int16 copies of the small parameter tables for the backtracks / exterior sweep (a pointer chase
// through global memory costs ~1 us per traced pair otherwise).  int11/int21/int22 and the hairpin size table stay global.
struct EpiTables {
    short stack[64], bulge[32], internal_loop[32];
    short mismatchI[200], mismatchH[200], mismatchM[200], mismatch1nI[200], mismatch23I[200], mismatchExt[200];
    short dangle5[40], dangle3[40];
    short ML_closing, ML_intern, TerminalAU, ninio, MAX_NINIO, pad[3];
    short hairpin[304];     // size term for loops of 3..303 unpaired bases (pair distances up to 304)
};

__device__ inline void fill_epi_tables(EpiTables* E, const FoldParams* __restrict__ P, int tid, int nt) {
    for (int x = tid; x < 64; x += nt) E->stack[x] = (short)min(P->stack[x >> 3][x & 7], 32767);
    for (int x = tid; x < 31; x += nt) { E->bulge[x] = (short)min(P->bulge[x], 32767); E->internal_loop[x] = (short)min(P->internal_loop[x], 32767); }
    for (int x = tid; x < 200; x += nt) {
        int t = x / 25, a = (x % 25) / 5, b = x % 5;
        E->mismatchI[x] = (short)min(P->mismatchI[t][a][b], 32767); E->mismatchH[x] = (short)min(P->mismatchH[t][a][b], 32767);
        E->mismatchM[x] = (short)P->mismatchM[t][a][b]; E->mismatch1nI[x] = (short)min(P->mismatch1nI[t][a][b], 32767);
        E->mismatch23I[x] = (short)min(P->mismatch23I[t][a][b], 32767); E->mismatchExt[x] = (short)P->mismatchExt[t][a][b];
    }
    for (int x = tid; x < 304; x += nt) E->hairpin[x] = (short)min(P->hairpinE[x], 32767);
    for (int x = tid; x < 40; x += nt) { E->dangle5[x] = (short)P->dangle5[x / 5][x % 5]; E->dangle3[x] = (short)P->dangle3[x / 5][x % 5]; }
    if (tid == 0) { E->ML_closing = (short)P->ML_closing; E->ML_intern = (short)P->ML_intern; E->TerminalAU = (short)P->TerminalAU; E->ninio = (short)P->ninio; E->MAX_NINIO = (short)P->MAX_NINIO; E->pad[0] = 0; }
}

#define XTAB_N 900
// exterior-loop term by bases: xtab[((si * 6 + a) * 5 + sj) * 6 + b] = e_extloop(pair_type(si, sj), a, b) with a / b = 5 for "no neighbour" (window end)
__device__ inline void fill_ext_table(short* xtab, const FoldParams* __restrict__ P, int tid, int nt) {
    for (int x = tid; x < XTAB_N; x += nt) {
        const int b = x % 6, sj = x / 6 % 5, a = x / 30 % 6, si = x / 180;
        const int type = pair_type(si, sj);
        int e = 0;
        if (type) {
            e = type > 2 ? P->TerminalAU : 0;
            if (a < 5 && b < 5) e += P->mismatchExt[type][a][b];
            else if (a < 5) e += P->dangle5[type][a];
            else if (b < 5) e += P->dangle3[type][b];
        }
        xtab[x] = (short)e;
    }
}

struct WinCtx {
    const EpiTables* E = nullptr;   // LDS tables (optional)
    const FoldParams* __restrict__ P;
    const unsigned char* S;   // LDS, 0..n+1
    const unsigned char* seq; // LDS, upper-case RNA chars, 1-based
    const int* f3;            // LDS, 0..n+2
    const short* spec;        // LDS, special hairpin energy per i for u==3,4,6 at [k*ldspec + i], SHRT_MIN = none
    int ldspec;
    int n, D;                 // D = max pair distance (min(span-1, n-1))
    // tiled exterior sweep only (f3_sweep_tiled): the exterior-loop term of a pair as ONE table read
    const short* xtab = nullptr;          // LDS [5 (S[i])][6 (5' neighbour, 5 = none)][5 (S[j])][6 (3' neighbour, 5 = none)], TerminalAU included
    const unsigned char* pq2 = nullptr;   // LDS per position j: byte offset (S[j] * 6 + 3' neighbour code) * 2 into a row of xtab
    short* pp = nullptr;                  // LDS per row i (written by the sweep): the first partner j that realises f3[i], where f3[i] != f3[i+1]
};

// special hairpin motifs (tri/tetra/hexa loops) per closing position i: spec[k*ld + i], -32768 = none
__device__ inline void special_hairpins(const FoldParams* __restrict__ P, const unsigned char* seq, int n, short* spec, int ld, int tid, int nt) {
    for (int x = tid; x <= n; x += nt) {
        short s3 = -32768, s4 = -32768, s6 = -32768;
        if (x >= 1) {
            if (x + 4 <= n)
                for (int k = 0; k < P->n_tri; k++) { bool m = true; for (int t = 0; t < 5; t++) m = m && (seq[x + t] == (unsigned char)P->tri[k][t]); if (m && s3 == -32768) s3 = (short)P->triE[k]; }
            if (x + 5 <= n)
                for (int k = 0; k < P->n_tetra; k++) { bool m = true; for (int t = 0; t < 6; t++) m = m && (seq[x + t] == (unsigned char)P->tetra[k][t]); if (m && s4 == -32768) s4 = (short)P->tetraE[k]; }
            if (x + 7 <= n)
                for (int k = 0; k < P->n_hexa; k++) { bool m = true; for (int t = 0; t < 8; t++) m = m && (seq[x + t] == (unsigned char)P->hexa[k][t]); if (m && s6 == -32768) s6 = (short)P->hexaE[k]; }
        }
        spec[x] = s3; spec[ld + x] = s4; spec[2 * ld + x] = s6;
    }
}

__device__ __forceinline__ int ptype_at(const WinCtx& X, int i, int j) {
    int d = j - i;
    if (d <= TURN || d > X.D) return 0;
    return pair_type(X.S[i], X.S[j]);
}

__device__ __forceinline__ int e_hairpin(const WinCtx& X, int i, int j, int type) {
    int u = j - i - 1;
    if (u == 4) { int s = X.spec[X.ldspec + i]; if (s != -32768) return s; }
    else if (u == 6) { int s = X.spec[2 * X.ldspec + i]; if (s != -32768) return s; }
    else if (u == 3) {
        int s = X.spec[i];
        if (s != -32768) return s;
        if (X.E) return X.E->hairpin[3] + (type > 2 ? X.E->TerminalAU : 0);
        return X.P->hairpinE[3] + (type > 2 ? X.P->TerminalAU : 0);
    }
    if (X.E && u < 304) return X.E->hairpin[u] + X.E->mismatchH[type * 25 + X.S[i + 1] * 5 + X.S[j - 1]];
    return X.P->hairpinE[u] + X.P->mismatchH[type][X.S[i + 1]][X.S[j - 1]];
}

__device__ __forceinline__ int mlstem_x(const WinCtx& X, int type, int a, int b) {
    if (X.E) {
        int e = X.E->ML_intern + (type > 2 ? X.E->TerminalAU : 0);
        if (a >= 0 && b >= 0) e += X.E->mismatchM[type * 25 + a * 5 + b];
        else if (a >= 0) e += X.E->dangle5[type * 5 + a];
        else if (b >= 0) e += X.E->dangle3[type * 5 + b];
        return e;
    }
    return e_mlstem(X.P, type, a, b);
}

__device__ __forceinline__ int ext_term(const WinCtx& X, int i, int k, int type) {
    const int a = i > 1 ? (int)X.S[i - 1] : -1, b = k < X.n ? (int)X.S[k + 1] : -1;
    if (X.E) {
        int e = (type > 2 ? X.E->TerminalAU : 0);
        if (a >= 0 && b >= 0) e += X.E->mismatchExt[type * 25 + a * 5 + b];
        else if (a >= 0) e += X.E->dangle5[type * 5 + a];
        else if (b >= 0) e += X.E->dangle3[type * 5 + b];
        return e;
    }
    return e_extloop(X.P, type, a, b);
}
__device__ __forceinline__ int ml_term(const WinCtx& X, int i, int j, int type) {
    return mlstem_x(X, type, i > 1 ? (int)X.S[i - 1] : -1, j < X.n ? (int)X.S[j + 1] : -1);
}

// interior loop via the LDS tables when present
__device__ __forceinline__ int intloop_x(const WinCtx& X, int n1, int n2, int type, int type2, int si1, int sj1, int sp1, int sq1) {
    if (!X.E) return e_intloop(X.P, n1, n2, type, type2, si1, sj1, sp1, sq1);
    const EpiTables& T = *X.E;
    const FoldParams* __restrict__ P = X.P;
    int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1;
    if (nl == 0) return T.stack[type * 8 + type2];
    if (ns == 0) {
        int e = T.bulge[nl];
        if (nl == 1) e += T.stack[type * 8 + type2];
        else e += (type > 2 ? T.TerminalAU : 0) + (type2 > 2 ? T.TerminalAU : 0);
        return e;
    }
    if (ns == 1) {
        if (nl == 1) return P->int11[type][type2][si1][sj1];
        if (nl == 2) return (n1 == 1) ? P->int21[type][type2][si1][sq1][sj1] : P->int21[type2][type][sq1][si1][sp1];
        int x = (nl - 1) * T.ninio;
        return T.internal_loop[nl + 1] + (x < T.MAX_NINIO ? x : T.MAX_NINIO) + T.mismatch1nI[type * 25 + si1 * 5 + sj1] + T.mismatch1nI[type2 * 25 + sq1 * 5 + sp1];
    }
    if (ns == 2) {
        if (nl == 2) return P->int22[type][type2][si1][sp1][sq1][sj1];
        if (nl == 3) return T.internal_loop[5] + T.ninio + T.mismatch23I[type * 25 + si1 * 5 + sj1] + T.mismatch23I[type2 * 25 + sq1 * 5 + sp1];
    }
    int x = (nl - ns) * T.ninio;
    return T.internal_loop[nl + ns] + (x < T.MAX_NINIO ? x : T.MAX_NINIO) + T.mismatchI[type * 25 + si1 * 5 + sj1] + T.mismatchI[type2 * 25 + sq1 * 5 + sp1];
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

// first set lane of a 64-bit ballot, or -1
__device__ __forceinline__ int first_lane(unsigned long long mask) { return mask ? (__ffsll((long long)mask) - 1) : -1; }

#ifdef MIRP_EPI_CLOCKS     // diagnostics build: phase clocks of wave 0 (lane 0) of every epilogue workgroup, summed in LDS, flushed once per workgroup
static __device__ unsigned long long g_epi_clk[32];
__device__ inline long long* epi_acc() { __shared__ long long acc[32]; return acc; }
#define EPI_CNT(k) do { if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)&epi_acc()[k], 1ull); } while (0)
#define EPI_T0() long long _t = clock64()
#define EPI_T(k) do { if (threadIdx.x == 0) { const long long _n = clock64(); epi_acc()[k] += _n - _t; _t = clock64(); } else { _t = clock64(); } } while (0)
#define EPI_INIT() do { if (threadIdx.x < 32) epi_acc()[threadIdx.x] = 0; __syncthreads(); } while (0)
#define EPI_FLUSH() do { __syncthreads(); if (threadIdx.x < 32) atomicAdd(&g_epi_clk[threadIdx.x], (unsigned long long)epi_acc()[threadIdx.x]); } while (0)
#else
#define EPI_T0() do {} while (0)
#define EPI_T(k) do {} while (0)
#define EPI_INIT() do {} while (0)
#define EPI_FLUSH() do {} while (0)
#define EPI_CNT(k) do {} while (0)
#endif
#define BT_STACK 96
#define MIRP_EPI_DMAX 300   // largest pair distance a tiled archive holds (fold_lds_kernel.hip: LDMAX)
#define TB_GENERIC 64   // trace-back code of a cell whose interior loop is one with n1 >= 2 that the fill kernel did not name (1 + (1 << 5 | 31): no real shape)
#define BT_LINE 16      // cells of a helix line fetched per round trip (each is its own cache line of the trace-back triangle)

// Wave-cooperative backtrack of one locally optimal structure (all 64 lanes call it with
// wave-uniform arguments).  buf: per-wave LDS char buffer; stk: per-wave LDS sector stack.
// Returns string length (>0) or a negative error code.  First-match-wins search orders follow
// SURVEY.md App. B "Backtrack" + B2 (exterior partner scan descending).
template <class Tab>
__device__ int backtrack_wave(const WinCtx& X, const Tab& T, int start, int jend, int span, char* buf, int* stk) {
    const int lane = threadIdx.x & 63;
    const int n = X.n;
    int len0 = (n - start < span + 1 ? n - start : span + 1) + 2;
    for (int x = lane; x < len0; x += 64) buf[x] = '-';
    int sp = 0;
    if (lane == 0) { stk[0] = start; stk[1] = jend; stk[2] = 0; }
    sp = 1;
    __builtin_amdgcn_wave_barrier();
    while (sp > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sp--;
        int i = stk[3 * sp], j = stk[3 * sp + 1], ml = stk[3 * sp + 2];
        i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j); ml = __builtin_amdgcn_readfirstlane(ml);
        if (j < i + TURN + 1) continue;
        if (sp + 3 >= BT_STACK) return -20;
        if (ml == 0) {
            int fij = X.f3[i];
            if (fij == X.f3[i + 1]) {
                if (lane == 0) { stk[3 * sp] = i + 1; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 0; }
                sp++;
                continue;
            }
            int found = -1;
            // descending scan; the caller's j is its own hit + 2 for the first segment, so the top four candidates are probed alone first
            // (every lane of a round pulls its own cache line of the archive: a full round costs 64 lines)
            for (int kb = j, width = 4; kb >= i + TURN + 1 && found < 0; kb -= width, width = 64) {
                EPI_CNT(16);
                int k = kb - lane;
                bool ok = false;
                if (lane < width && k >= i + TURN + 1) {
                    int type = ptype_at(X, i, k);
                    if (type) ok = (fij == T.C(k - i, i) + ext_term(X, i, k, type) + X.f3[k + 1]);
                }
                int fl = first_lane(__ballot(ok));
                if (fl >= 0) found = kb - fl;
            }
            if (found < 0) return -21;
            int k = found;
            if (j == n) {
                if (lane == 0) { stk[3 * sp] = k + 1; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 0; }
                sp++;
            }
            j = k;
            if (lane == 0) {
                buf[i - start] = '(';
                buf[j - start] = ')';
                if (j < n) buf[j + 1 - start] = '.';
            }
        } else {
            int d = j - i;
            EPI_CNT(17);
            int fij = T.M(d, i);
            if (T.M(d - 1, i) == fij) {
                if (lane == 0) { stk[3 * sp] = i; stk[3 * sp + 1] = j - 1; stk[3 * sp + 2] = 1; }
                sp++;
                continue;
            }
            if (T.M(d - 1, i + 1) == fij) {
                if (lane == 0) { stk[3 * sp] = i + 1; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 1; }
                sp++;
                continue;
            }
            int type = ptype_at(X, i, j);
            bool ok = false;
            EPI_CNT(18);
            if (type) ok = (T.C(d, i) + ml_term(X, i, j, type) == fij);
            if (!ok) {
                int found = -1;
                for (int kb = i + 1 + TURN; kb <= j - 2 - TURN && found < 0; kb += 64) {
                    EPI_CNT(19);
                    int k = kb + lane;
                    bool hit = false;
                    if (k <= j - 2 - TURN) hit = (fij == T.M(k - i, i) + T.M(j - k - 1, k + 1));
                    int fl = first_lane(__ballot(hit));
                    if (fl >= 0) found = kb + fl;
                }
                if (found < 0) return -22;
                if (lane == 0) {
                    stk[3 * sp] = i; stk[3 * sp + 1] = found; stk[3 * sp + 2] = 1;
                    stk[3 * sp + 3] = found + 1; stk[3 * sp + 4] = j; stk[3 * sp + 5] = 1;
                }
                sp += 2;
                continue;
            }
            if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
        }
        // (i,j) is a traced pair: follow stacks / interior loops until a hairpin or a multiloop
        for (;;) {
            // With trace-back codes the chain is followed a whole helix per memory round trip: lane l fetches the code of (i+l, j-l); stacked
            // pairs and symmetric loops stay on that line, so the walk below only goes back to memory after an asymmetric loop.  A code > 0
            // also says that the hairpin does not realise c, so no energy is read along the way.
            {
                const int il = i + lane, jl = j - lane;
                EPI_CNT(20);
                int cl = (lane < BT_LINE && jl - il >= TURN + 1) ? T.TB(jl - il, il) : 0;
                if (__builtin_amdgcn_readfirstlane(cl) > 0) {
                    int pos = 0;
                    for (;;) {
                        const int c = __builtin_amdgcn_readlane(cl, pos);
                        if (c <= 0) break;                                  // (i,j) = line position pos: hairpin or multiloop
                        const int n1 = (c - 1) >> 5, n2 = (c - 1) & 31;
                        i += 1 + n1; j -= 1 + n2;
                        if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
                        if (n1 != n2 || pos + 1 + n1 >= BT_LINE) { pos = -1; break; }   // off the line (or past it): fetch again
                        pos += 1 + n1;
                    }
                    if (pos < 0) continue;
                }
            }
            int type = ptype_at(X, i, j);
            EPI_CNT(21);
            int cij = T.C(j - i, i);
            if (cij == e_hairpin(X, i, j, type)) break;
            int pmax = (j - 2 - TURN < i + MAXLOOP + 1) ? j - 2 - TURN : i + MAXLOOP + 1;
            int fp = -1, fq = -1;
            // Trace-back code of the fill kernel (T.TB >= 0): 1 + (n1 << 5 | n2) names the interior loop this search would find first,
            // 0 says that none realises c(i,j) (multiloop).  Tables without codes (TB < 0) are searched: p ascending, q descending.
            EPI_CNT(22);
            const int code = __builtin_amdgcn_readfirstlane(T.TB(j - i, i));
            if (code > 0) { fp = i + 1 + ((code - 1) >> 5); fq = j - 1 - ((code - 1) & 31); }
            if (code < 0)
            for (int pb = i + 1; pb <= pmax && fp < 0; pb += 2) {
                int p = pb + (lane >> 5), q = j - 1 - (lane & 31);
                int minq = j - i + p - MAXLOOP - 2;
                if (minq < p + 1 + TURN) minq = p + 1 + TURN;
                bool hit = false;
                if (p <= pmax && q >= minq) {
                    int t2 = pair_type(X.S[p], X.S[q]);
                    if (t2) {
                        t2 = rtype_of(t2);
                        int e = intloop_x(X, p - i - 1, j - q - 1, type, t2, X.S[i + 1], X.S[j - 1], X.S[p - 1], X.S[q + 1]);
                        hit = (cij == e + T.C(q - p, p));
                    }
                }
                int fl = first_lane(__ballot(hit));
                if (fl >= 0) { fp = pb + (fl >> 5); fq = j - 1 - (fl & 31); }
            }
            if (fp >= 0) {
                i = fp; j = fq;
                if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
                continue;
            }
            int mm = X.P->ML_closing + mlstem_x(X, rtype_of(type), X.S[j - 1], X.S[i + 1]);
            int found = -1;
            for (int kb = i + 2 + TURN; kb <= j - 3 - TURN && found < 0; kb += 64) {
                EPI_CNT(23);
                int k = kb + lane;
                bool hit = false;
                if (k <= j - 3 - TURN) hit = (cij == T.M(k - i - 1, i + 1) + T.M(j - k - 2, k + 1) + mm);
                int fl = first_lane(__ballot(hit));
                if (fl >= 0) found = kb + fl;
            }
            if (found < 0) return -23;
            if (lane == 0) {
                stk[3 * sp] = i + 1; stk[3 * sp + 1] = found; stk[3 * sp + 2] = 1;
                stk[3 * sp + 3] = found + 1; stk[3 * sp + 4] = j - 1; stk[3 * sp + 5] = 1;
            }
            sp += 2;
            break;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int last = 0;
    for (int x = lane; x < len0; x += 64)
        if (buf[x] != '-') last = x;
    {
        int v = last;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
        last = v;
    }
    int L = last + 1;
    for (int x = lane; x < L; x += 64)
        if (buf[x] == '-') buf[x] = '.';
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return L;
}

// Interior loop of a pair (i, j) whose trace-back code is TB_GENERIC: some loop with n1 >= 2 realises c(i,j) = cij and no loop with n1 <= 1 does
// (fill kernel: the packed generic rows carry no shape code).  The reference's search from p = i + 3 on -- p ascending, q descending, first match;
// lane = (p parity, n2), NB rounds of candidates fetched together.  Returns p << 16 | q, or -1.  Kept out of line: inlined into the backtrack
// its energy function's table pointers pushed the whole kernel over its register budget (35 -> 93 spilled VGPRs), for a path few pairs take.
template <class Tab>
__device__ __forceinline__ int bt_search_unnamed(const WinCtx& X, const Tab& T, int i, int j, int cij) {
    const int lane = threadIdx.x & 63;
    const int type = ptype_at(X, i, j);
    const int hp = lane >> 5, n2 = lane & 31, q = j - 1 - n2;
    constexpr int NR = (MAXLOOP - 2) / 2 + 1;         // pairs of n1 = 2 .. MAXLOOP (+ 1)
    constexpr int NB = 3;
    int res = -1;
#pragma unroll 1
    for (int kb = 0; kb < NR && res < 0; kb += NB) {
        int cc[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int n1 = 2 + 2 * (kb + k) + hp, p = i + 1 + n1;
            cc[k] = INF;
            if (n1 + n2 <= MAXLOOP && q - p >= TURN + 1) cc[k] = T.C(q - p, p);
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            if (res < 0) {
                const int n1 = 2 + 2 * (kb + k) + hp, p = i + 1 + n1;
                bool hit = false;
                if (cc[k] < INF) {
                    const int t2 = rtype_of(pair_type(X.S[p], X.S[q]));
                    hit = (cij == cc[k] + intloop_x(X, n1, n2, type, t2, X.S[i + 1], X.S[j - 1], X.S[p - 1], X.S[q + 1]));
                }
                const int fl = first_lane(__ballot(hit));
                if (fl >= 0) res = (i + 3 + 2 * (kb + k) + (fl >> 5)) << 16 | (j - 1 - (fl & 31));
            }
        }
    }
    return res;
}

// Backtrack over a TILED archive with trace-back codes (Tab::kTiled).  Same search orders and results as backtrack_wave; what changes is how often
// it goes to memory: the walk is a chain of dependent round trips (~1 us each under load), so every fetch brings an 8 x 8 PATCH of the archive
// anchored at the current pair or segment (i, j) -- lane (a, b) holds cell (i + a, j - b): trace-back code, c and (for a multiloop segment) fML,
// a handful of 128-byte tiles per table -- and the walk continues in registers (v_readlane) until it leaves the patch:
//   * exterior segment: the top four partner candidates (the caller's j is its own hit + 2) are row 0 of the patch, so the partner scan and the
//     first stretch of the helix are one fetch;
//   * helix: stacked pairs, bulges and interior loops are followed inside the patch (an asymmetric loop does not cost a fetch), the closing
//     pair's c is already there for the hairpin test;
//   * multiloop segment: the run of unpaired bases that the reference trims one pop at a time (j first, then i) is walked inside the patch, the
//     pair test of the segment's final (i, j) needs no further fetch, and the helix starts in the same patch;
//   * multiloop splits: all split points of the segment are fetched at once (two fML reads per lane and 64 points), first match ascending.
template <class Tab>
__device__ int backtrack_wave_tiled(const WinCtx& X, const Tab& T, int start, int jend, int span, char* buf, int* stk) {
    const int lane = threadIdx.x & 63;
    const int n = X.n;
    const int pa = lane >> 3, pb = lane & 7;
    constexpr int NSR = (MIRP_EPI_DMAX + 63) / 64;   // rounds of 64 split points that cover any segment
    int len0 = (n - start < span + 1 ? n - start : span + 1) + 2;
    for (int x = lane; x < len0; x += 64) buf[x] = '-';
    int sp = 0;
    if (lane == 0) { stk[0] = start; stk[1] = jend; stk[2] = 0; }
    sp = 1;
    __builtin_amdgcn_wave_barrier();
    while (sp > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sp--;
        int i = stk[3 * sp], j = stk[3 * sp + 1], ml = stk[3 * sp + 2];
        i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j); ml = __builtin_amdgcn_readfirstlane(ml);
        if (j < i + TURN + 1) continue;
        if (sp + 3 >= BT_STACK) return -20;
        int tbv = 0, cv = INF;          // the current patch: trace-back codes and c of the cells (i + pa, j - pb) at fetch time
        int r = 0, c = 0;               // the walk's position inside it
        bool defer = false;
        if (ml == 2) {
            // a traced pair whose loop the fill kernel did not name (TB_GENERIC, pushed by the walk below)
            const int hit = bt_search_unnamed(X, T, i, j, T.C(j - i, i));
            if (hit < 0) return -25;
            i = hit >> 16; j = hit & 0xffff;
            if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
            r = 8; c = 8;
        } else if (ml == 0) {
            // unpaired 5' bases: the reference pops (i + 1, j) while f3[i] == f3[i + 1]; here 64 positions per step
            for (;;) {
                const int x = i + lane;
                const bool diff = x <= n ? (X.f3[x] != X.f3[x + 1]) : true;
                const int fl = first_lane(__ballot(diff));
                if (fl < 0) { i += 64; continue; }
                i += fl;
                break;
            }
            if (j < i + TURN + 1) continue;
            const int fij = X.f3[i];
            EPI_CNT(16);
            {
                const int ia = i + pa, jb = j - pb, dd = jb - ia;
                if (dd >= TURN + 1) { const int o = T.at(dd, ia); tbv = T.TBat(o); cv = T.Cat(o); }
            }
            // partners descending.  The caller's j is its own hit + 2 for a structure's first segment: the top four candidates are row 0 of the patch
            int found = -1;
            {
                const int k = j - pb;
                bool ok = false;
                if (pa == 0 && pb < 4 && k >= i + TURN + 1) {
                    const int type = ptype_at(X, i, k);
                    if (type) ok = (fij == cv + ext_term(X, i, k, type) + X.f3[k + 1]);
                }
                const int fl = first_lane(__ballot(ok));
                if (fl >= 0) { found = j - fl; c = fl; }
            }
            for (int kb = j - 4; kb >= i + TURN + 1 && found < 0; kb -= 64) {      // rare: a full round of 64 is a row of the archive = 8 tiles
                const int k = kb - lane;
                bool ok = false;
                if (k >= i + TURN + 1) {
                    const int type = ptype_at(X, i, k);
                    if (type) ok = (fij == T.C(k - i, i) + ext_term(X, i, k, type) + X.f3[k + 1]);
                }
                const int fl = first_lane(__ballot(ok));
                if (fl >= 0) { found = kb - fl; c = 8; }       // outside the patch: the helix fetches its own
            }
            if (found < 0) return -21;
            const int k = found;
            if (j == n) {
                if (lane == 0) { stk[3 * sp] = k + 1; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 0; }
                sp++;
            }
            j = k;
            if (lane == 0) {
                buf[i - start] = '(';
                buf[j - start] = ')';
                if (j < n) buf[j + 1 - start] = '.';
            }
        } else {
            int fij, cij;
            for (;;) {      // trim unpaired bases inside patches of fML: j first, then i (the reference's order), until neither matches
                EPI_CNT(17);
                const int ia = i + pa, jb = j - pb, dd = jb - ia;
                int mv = INF;
                tbv = 0; cv = INF;
                if (dd >= TURN + 1) { const int o = T.at(dd, ia); mv = T.Mat(o); cv = T.Cat(o); tbv = T.TBat(o); }
                fij = __builtin_amdgcn_readlane(mv, 0);
                r = 0; c = 0;
                bool edge = false;
                for (;;) {
                    if (r == 7 || c == 7) { edge = true; break; }
                    if (__builtin_amdgcn_readlane(mv, r * 8 + c + 1) == fij) { c++; continue; }
                    if (__builtin_amdgcn_readlane(mv, (r + 1) * 8 + c) == fij) { r++; continue; }
                    break;
                }
                i += r; j -= c;
                if (edge && (r | c)) continue;      // left the patch: fetch again at the new segment
                if (edge) return -24;               // (0, 0) is never on the edge
                cij = __builtin_amdgcn_readlane(cv, r * 8 + c);
                break;
            }
            const int type = ptype_at(X, i, j);
            bool ok = false;
            if (type) ok = (cij + ml_term(X, i, j, type) == fij);
            if (!ok) {
                // all split points at once, first match ascending
                EPI_CNT(19);
                int m1[NSR], m2[NSR];
                const int k0 = i + 1 + TURN, k1 = j - 2 - TURN;
#pragma unroll
                for (int q = 0; q < NSR; q++) {
                    m1[q] = INF; m2[q] = INF;
                    if (k0 + 64 * q <= k1) {                    // wave-uniform: most segments need one round
                        const int k = k0 + 64 * q + lane;
                        if (k <= k1) { m1[q] = T.M(k - i, i); m2[q] = T.M(j - k - 1, k + 1); }
                    }
                }
                int found = -1;
#pragma unroll
                for (int q = 0; q < NSR; q++) {
                    if (k0 + 64 * q <= k1 && found < 0) {
                        const int fl = first_lane(__ballot(fij == m1[q] + m2[q]));
                        if (fl >= 0) found = k0 + 64 * q + fl;
                    }
                }
                if (found < 0) return -22;
                if (lane == 0) {
                    stk[3 * sp] = i; stk[3 * sp + 1] = found; stk[3 * sp + 2] = 1;
                    stk[3 * sp + 3] = found + 1; stk[3 * sp + 4] = j; stk[3 * sp + 5] = 1;
                }
                sp += 2;
                continue;
            }
            if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
        }
        // (i,j) is a traced pair at position (r, c) of the current patch (c > 7: not in it): follow the trace-back codes until a hairpin or a multiloop
        for (;;) {
            if (r > 7 || c > 7) {       // left the patch: fetch the one anchored at the current pair
                EPI_CNT(20);
                const int ia = i + pa, jb = j - pb, dd = jb - ia;
                tbv = 0; cv = INF;
                if (dd >= TURN + 1) { const int o = T.at(dd, ia); tbv = T.TBat(o); cv = T.Cat(o); }
                r = 0; c = 0;
            }
            for (;;) {
                const int code = __builtin_amdgcn_readlane(tbv, r * 8 + c);
                if (code <= 0) break;                               // hairpin or multiloop closes here
                if (code == TB_GENERIC) {       // unnamed loop: searched at the top of the sector loop, where the patch and the walk's position are dead
                    if (lane == 0) { stk[3 * sp] = i; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 2; }
                    sp++;
                    defer = true;
                    break;
                }
                const int n1 = (code - 1) >> 5, n2 = (code - 1) & 31;
                i += 1 + n1; j -= 1 + n2;
                if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
                r += 1 + n1; c += 1 + n2;
                if (r > 7 || c > 7) break;
            }
            if (defer) break;
            if (r > 7 || c > 7) continue;
            const int cij = __builtin_amdgcn_readlane(cv, r * 8 + c);
            const int type = ptype_at(X, i, j);
            if (cij == e_hairpin(X, i, j, type)) break;
            const int mm = X.P->ML_closing + mlstem_x(X, rtype_of(type), X.S[j - 1], X.S[i + 1]);
            EPI_CNT(23);
            int m1[NSR], m2[NSR];
            const int k0 = i + 2 + TURN, k1 = j - 3 - TURN;
#pragma unroll
            for (int q = 0; q < NSR; q++) {
                m1[q] = INF; m2[q] = INF;
                if (k0 + 64 * q <= k1) {                        // wave-uniform
                    const int k = k0 + 64 * q + lane;
                    if (k <= k1) { m1[q] = T.M(k - i - 1, i + 1); m2[q] = T.M(j - k - 2, k + 1); }
                }
            }
            int found = -1;
#pragma unroll
            for (int q = 0; q < NSR; q++) {
                if (k0 + 64 * q <= k1 && found < 0) {
                    const int fl = first_lane(__ballot(cij == m1[q] + m2[q] + mm));
                    if (fl >= 0) found = k0 + 64 * q + fl;
                }
            }
            if (found < 0) return -23;
            if (lane == 0) {
                stk[3 * sp] = i + 1; stk[3 * sp + 1] = found; stk[3 * sp + 2] = 1;
                stk[3 * sp + 3] = found + 1; stk[3 * sp + 4] = j - 1; stk[3 * sp + 5] = 1;
            }
            sp += 2;
            break;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int last = 0;
    for (int x = lane; x < len0; x += 64)
        if (buf[x] != '-') last = x;
    {
        int v = last;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
        last = v;
    }
    int L = last + 1;
    for (int x = lane; x < L; x += 64)
        if (buf[x] == '-') buf[x] = '.';
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return L;
}

// Exterior sweep over a TILED archive (Tab::kTiled: 8 x 8 tiles over (row, diagonal), see fold_lds_kernel.hip).
// f3[i] = min(f3[i+1], min_j c(i,j) + ext(i,j) + f3[j+1]) is sequential in i, but only through f3.  Rows are taken in blocks of 8 x (waves) rows,
// top block first.  Step 1: every wave takes ONE row block of the archive (8 rows) and has ALL of its tiles in flight at once: lane = diagonal
// (d = 4 + lane, + 64 per group), a lane's 16-byte load is the 8 rows of its diagonal, a wave-wide load is 1 KB of consecutive tiles.  A cell then
// costs three LDS reads (position code of j, the exterior term from xtab, f3[j+1]) and eight VALU instructions: c is finite only where (i, j)
// pairs, so no pair type is computed, and the term table is indexed by bases and neighbour codes, so nothing is selected.  Partners j whose
// f3[j+1] is final (j+1 above the block) go into the lane's eight running minima, which meet in LDS (ds_min); the few partners inside the block
// (first group only) are parked in LDS, transposed: innerT[j+1][row].
// Step 2 (wave 0, lane = row): the sequential chain through the block runs in registers -- one v_readlane + add + min per row, the innerT reads do
// not depend on the chain.  The backtrack stacks are idle here and serve as scratch.
template <class Tab, int NT>
__device__ void f3_sweep_tiled(const WinCtx& X, const Tab& T, int* f3, int* scratch) {
    constexpr int NW = NT / 64, RBK = 8 * NW;
    constexpr int NG = (MIRP_EPI_DMAX - TURN - 1) / 64 + 1;   // groups of 64 diagonals
    constexpr int SW_BIAS = 1 << 17, KEY_INF = 0x7fffffff;     // energies of an exterior decomposition stay far inside +-2^17
    static_assert(RBK <= 64, "step 2 maps the rows of a block onto the lanes of one wave");
    static_assert(RBK <= 64 + TURN, "only the first group of diagonals has partners inside the block");
    static_assert((RBK + RBK * RBK / 2) * 4 <= NW * 3 * BT_STACK * 4, "f3 scratch must fit the backtrack stacks");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = X.n, D = X.D;
    int* part = scratch;                                       // [RBK] per-row minimum over the partners above the block
    short* innerT = reinterpret_cast<short*>(scratch + RBK);   // [RBK (j+1 - blk_lo)][RBK (row - blk_lo)] c + ext of the partners inside the block
    const char* xtab = reinterpret_cast<const char*>(X.xtab);
    const int top = n - TURN - 1;                              // highest row that can pair
    EPI_T0();
    for (int g = top >= 1 ? (top - 1) / RBK : -1; g >= 0; g--) {
        const int blk_lo = g * RBK + 1, blk_hi = blk_lo + RBK - 1;
        const int tb = g * NW + wave;                          // this wave's row block: rows i0 .. i0 + 7
        const int i0 = 8 * tb + 1;
        const int dmax_b = D < n - i0 ? D : n - i0;            // wave-uniform: the block's first row reaches furthest
        // all tiles of the row block in flight: 16 bytes = the 8 rows of diagonal d (cells past a row's end are never looked at)
        uint4 cg[NG];
        const uint4* base = reinterpret_cast<const uint4*>(T.carch + T.off[tb]) + lane;
#pragma unroll
        for (int u = 0; u < NG; u++) {
            cg[u] = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
            if (TURN + 1 + 64 * u + lane <= dmax_b) cg[u] = base[64 * u];
        }
        for (int x = tid; x < RBK; x += NT) part[x] = KEY_INF;
        for (int x = tid; x < RBK * RBK / 2; x += NT) reinterpret_cast<int*>(innerT)[x] = 0x7fff7fff;
        __syncthreads();
        EPI_T(8);
#ifdef MIRP_EPI_CLOCKS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        EPI_T(9);
#endif
        {
            int best[8];      // keys (c + f3[j+1] + SW_BIAS) << 9 | j: the minimum is the minimum energy and, among equal energies, the first partner
#pragma unroll
            for (int rr = 0; rr < 8; rr++) best[rr] = KEY_INF;
            // byte offset of row i's part of xtab (lanes 0-7 hold the block's rows): wave-uniform per row, fetched with v_readlane
            int vrow = 0;
            {
                const int i = i0 + (lane & 7);
                if (i <= n) vrow = (X.S[i] * 6 + (i > 1 ? (int)X.S[i - 1] : 5)) * 60;
            }
#pragma unroll
            for (int u = 0; u < NG; u++) {
                if (TURN + 1 + 64 * u <= dmax_b) {              // wave-uniform
                    const int d = TURN + 1 + 64 * u + lane;
                    const int j0 = i0 + d;                      // partner of row i0; row i0 + rr pairs with j0 + rr
                    const int nrow_ok = d <= D ? n - j0 : -1;   // rows rr <= nrow_ok have j <= n
                    const int kb0 = (SW_BIAS << 9) + j0;
                    const unsigned cw[4] = {cg[u].x, cg[u].y, cg[u].z, cg[u].w};
#pragma unroll
                    for (int h = 0; h < 8; h += 4) {            // four rows at a time (registers)
                        int pq[4], fj[4], tv[4];
                        // lanes past the window's end read whatever LDS holds there (out of range: zero): their cells are not valid
#pragma unroll
                        for (int q = 0; q < 4; q++) pq[q] = X.pq2[j0 + h + q];
#pragma unroll
                        for (int q = 0; q < 4; q++) fj[q] = f3[j0 + h + q + 1];
#pragma unroll
                        for (int q = 0; q < 4; q++) tv[q] = *reinterpret_cast<const short*>(xtab + __builtin_amdgcn_readlane(vrow, h + q) + pq[q]);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int rr = h + q;
                            const int cv = (int)(short)((rr & 1) ? cw[rr >> 1] >> 16 : cw[rr >> 1] & 0xffffu);
                            bool valid = cv != 0x7fff && rr <= nrow_ok;     // c is finite only where (i, j) pairs
                            if (u == 0) {
                                const bool above = j0 + rr >= blk_hi;       // f3[j+1] final: j+1 above the block
                                if (valid && !above) innerT[(j0 + rr + 1 - blk_lo) * RBK + 8 * wave + rr] = (short)(cv + tv[q]);   // partner inside the block
                                valid = valid && above;
                            }
                            const int c = cv + tv[q] + fj[q];
                            const int cm = valid ? (c << 9) + kb0 + rr : KEY_INF;
                            best[rr] = cm < best[rr] ? cm : best[rr];
                        }
                    }
                }
            }
            // hand-issued ds_min: the compiler's atomic optimizer would turn each atomicMin into a loop over the active lanes
            {
                const unsigned pa = (unsigned)(size_t)(__attribute__((address_space(3))) int*)(part + 8 * wave);
#pragma unroll
                for (int rr = 0; rr < 8; rr++) asm volatile("ds_min_i32 %0, %1 offset:%2" : : "v"(pa), "v"(best[rr]), "n"(4 * rr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        EPI_T(10);
        __syncthreads();
        EPI_T(11);
        if (wave == 0) {
            const int r = lane & (RBK - 1);
            int best = part[r];
            int fx = blk_hi + 1 <= n + 2 ? f3[blk_hi + 1] : 0, fo = 0, po = 0;
#pragma unroll 4
            for (int xr = RBK - 1; xr >= 0; xr--) {
                const int bx = __builtin_amdgcn_readlane(best, xr);
                const int bv = (bx >> 9) - SW_BIAS;                // KEY_INF decodes to a large positive value
                fx = bv < fx ? bv : fx;                            // f3 of row blk_lo + xr
                fo = lane == xr ? fx : fo;
                po = lane == xr ? (bx & 511) : po;                 // its first partner (meaningful where f3 drops at this row)
                const int e = innerT[xr * RBK + r];                // rows that have row xr + blk_lo - 1 as a partner
                const int c = ((e + fx) << 9) + (SW_BIAS << 9) + (blk_lo + xr - 1);
                best = (e != 0x7fff && c < best) ? c : best;
            }
            if (lane < RBK && blk_lo + lane <= n) { f3[blk_lo + lane] = fo; X.pp[blk_lo + lane] = (short)po; }
        }
        EPI_T(12);
        __syncthreads();
        EPI_T(13);
    }
}

// Shared epilogue: f3 sweep, enumeration of structure starts, parallel backtracks, RNALfold's
// "print prev unless contained in new" rule, output records.  Called by every thread of the
// workgroup after the tables are complete (and visible).
template <class Tab, int NT>
__device__ void fold_epilogue(const WinCtx& X, const Tab& T, int span, int* f3 /*LDS, writable*/, int* starts /*LDS [max_lines]*/,
                              int* lens /*LDS [max_lines]*/, char* btbuf /*LDS NT/64 * bufstride*/, int bufstride,
                              int* btstk /*LDS NT/64 * 3*BT_STACK*/, int* sh_misc /*LDS >= 4 ints*/,
                              int win, int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines,
                              char* __restrict__ out_ss, int* __restrict__ out_nlines, int* __restrict__ out_mfe,
                              int* __restrict__ out_status) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = X.n, D = X.D;
    // ---- f3 (exterior) sweep.  f3[i] = min(f3[i+1], min_j c(i,j) + ext(i,j) + f3[j+1]) is sequential in i, but only through f3:
    // rows are processed in blocks of RB.  Step 1 (waves take the rows of the block round-robin, all in parallel): the partners j whose f3[j+1] is already
    // final (j+1 above the block) are reduced to one partial minimum per row, and the few partners inside the block (span < NW)
    // are fetched to LDS.  Step 2 (wave 0): the short sequential chain through the block touches LDS only.  The backtrack stacks
    // are idle here and serve as scratch: part[NW], inner[NW][NW].
    for (int x = tid; x < n + 3; x += NT) f3[x] = 0;
    __syncthreads();
    EPI_T0();
#ifdef MIRP_X_EPI_NOSWEEP          // timing experiment: no exterior sweep (f3 stays 0: no structures follow)
    if constexpr (Tab::kTiled) {}
    else {
#else
    if constexpr (Tab::kTiled) f3_sweep_tiled<Tab, NT>(X, T, f3, btstk);
    else {
#endif
    constexpr int NW = NT / 64;
    constexpr int RB = 32;   // rows per block: 32 consecutive i share their cache lines of every archived diagonal
    static_assert(RB == 32, "step 1 maps a half-wave onto the rows of a block");
    static_assert((RB + RB * RB) * 4 <= NW * 3 * BT_STACK * 4, "f3 scratch must fit the backtrack stacks");
    int* part = btstk;
    int* inner = btstk + RB;
    for (int i_hi = n - TURN - 1; i_hi >= 1; i_hi -= RB) {
        // Step 1 walks the block diagonal by diagonal: a half-wave holds the RB rows (lane = row), so its 32 cells of one archived diagonal
        // are one contiguous 64-byte read that is consumed at once (row-major order re-fetched every line once per row from HBM when
        // hundreds of windows share an L2).  Waves and half-waves stride over the diagonals; the per-row minima meet in LDS.
        for (int x = tid; x < RB + RB * RB; x += NT) part[x] = INF;      // part[RB] and inner[RB][RB] are contiguous
        __syncthreads();
        {
            const int r = lane & (RB - 1);
            const int i = i_hi - r;
            int best = INF;
            if (i >= 1) {
                const int si = X.S[i];
                constexpr int UNR = 8;      // archive reads in flight per lane: the sweep is a chain of memory round trips otherwise
                for (int d0 = TURN + 1 + 2 * wave + (lane >> 5); d0 <= D && i + d0 <= n; d0 += 2 * NW * UNR) {
                    int cv[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; u++) {
                        const int d = d0 + u * 2 * NW;
                        cv[u] = (d <= D && i + d <= n) ? T.C(d, i) : INF;
                    }
#pragma unroll
                    for (int u = 0; u < UNR; u++) {
                        const int d = d0 + u * 2 * NW, j = i + d;
                        if (cv[u] >= INF) continue;
                        const int type = pair_type(si, X.S[j]);
                        if (type) {
                            const int e = cv[u] + ext_term(X, i, j, type);
                            if (j >= i_hi) { const int c = e + f3[j + 1]; best = c < best ? c : best; }   // f3[j+1] final: j+1 above the block
                            else inner[r * RB + d - TURN - 1] = e;                                         // partner inside the block
                        }
                    }
                }
            }
            if (best < INF) atomicMin(&part[r], best);
        }
        __syncthreads();
        if (wave == 0) {
            const int i_lo = (i_hi - RB + 1 > 1) ? i_hi - RB + 1 : 1;
            for (int r = i_hi; r >= i_lo; r--) {
                const int w = i_hi - r;
                int best = f3[r + 1];
                const int p = part[w];
                best = p < best ? p : best;
                if (lane < RB) {
                    const int e = inner[w * RB + lane];
                    if (e < INF) { const int c = e + f3[r + TURN + 1 + lane + 1]; best = c < best ? c : best; }
                }
                best = wave_min(best);
                if (lane == 0) f3[r] = best;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
    }
    }
    EPI_T(0);
    if (wave == 0) {
        // ---- structure starts, descending: l>=2 with f3[l]!=f3[l+1] && f3[l-1]==f3[l]; l==1 with f3[1]!=f3[2]
        int cnt = 0;
        for (int base = n - TURN - 1; base >= 1; base -= 64) {
            int l = base - lane;
            bool is = false;
            if (l >= 1) is = (f3[l] != f3[l + 1]) && (l == 1 || f3[l - 1] == f3[l]);
            unsigned long long mask = __ballot(is);
            int rank = __popcll(mask & ((1ull << lane) - 1ull));
            if (is && cnt + rank < max_lines) starts[cnt + rank] = l;
            cnt += __popcll(mask);
        }
        if (lane == 0) { sh_misc[0] = cnt < max_lines ? cnt : max_lines; sh_misc[1] = cnt > max_lines ? 1 : 0; sh_misc[2] = 0; sh_misc[3] = cnt; sh_misc[4] = 0; }
    }
    __syncthreads();
    const int nst = sh_misc[0];
    EPI_T(1);
    // ---- backtracks: one wave per start
    char* mybuf = btbuf + wave * bufstride;
    int* mystk = btstk + wave * 3 * BT_STACK;
    // (the trip count is bounded on purpose: with an endless for (;;) around the counter draw this loop was miscompiled -- the kernel hung on a
    // one-window batch -- and any bound made it go away)
    for (int it = 0; it <= nst; it++) {
        int k = wave + it * (NT / 64);
        if constexpr (Tab::kTiled) {      // structures differ a lot in length: the waves draw them from a counter instead of striding
            int t = 0;
            if (lane == 0) t = atomicAdd(&sh_misc[4], 1);
            k = __builtin_amdgcn_readfirstlane(t);
        }
        if (k >= nst) break;
        int lind = starts[k];
        EPI_CNT(25);
        int fij = f3[lind];
        // "short backtrack": first partner (ascending) that realises f3[lind]; the tiled sweep has recorded it
        int pp = -1;
        if constexpr (Tab::kTiled) pp = X.pp[lind];
        else
        for (int pb = lind + TURN; pb <= lind + span && pp < 0; pb += 64) {
            EPI_CNT(24);
            int q = pb + lane;
            bool hit = false;
            if (q <= lind + span && q <= n) {
                int type = ptype_at(X, lind, q);
                if (type) hit = (fij == T.C(q - lind, lind) + ext_term(X, lind, q, type) + f3[q + 1]);
            }
            int fl = first_lane(__ballot(hit));
            if (fl >= 0) pp = pb + fl;
        }
        int L = -10;
        EPI_T(2);
        if (pp >= 0) {
#ifdef MIRP_X_EPI_NOBT             // timing experiment: sweep and enumeration only
            if constexpr (Tab::kTiled) L = -10;
#else
            if constexpr (Tab::kTiled) L = backtrack_wave_tiled(X, T, lind, (pp + 2 < n ? pp + 2 : n), span, mybuf, mystk);
#endif
#ifdef MIRP_X_GEN_NOBT              // timing experiment (generic path): sweep and enumeration only
            else L = -10;
#else
            else L = backtrack_wave(X, T, lind, (pp + 2 < n ? pp + 2 : n), span, mybuf, mystk);
#endif
        }
        EPI_T(3);
        if (L < 0) {
            if (lane == 0) { sh_misc[2] = L; lens[k] = 0; }
            continue;
        }
        // write printed text: leading '.' for starts >= 2 (5' dangle base), none for the start-1 structure
        int lead = lind >= 2 ? 1 : 0;
        char* dst = out_ss + ((size_t)win * max_lines + k) * ss_stride;
        if (L + lead + 1 > ss_stride) { if (lane == 0) { sh_misc[2] = -30; lens[k] = 0; } continue; }
        if (lane == 0 && lead) dst[0] = '.';
        for (int x = lane; x < L; x += 64) dst[x + lead] = mybuf[x];
        if (lane == 0) {
            dst[L + lead] = 0;
            lens[k] = L;
            MirpFoldLine ln;
            ln.start = lead ? lind - 1 : 1;
            ln.len = L + lead;
            ln.energy = lead ? (f3[lind] - f3[lind + L - 1]) : (f3[1] - f3[L]);
            ln.printed = 1;
            out_lines[(size_t)win * max_lines + k] = ln;
        }
        EPI_T(4);
    }
    EPI_T(5);
    __syncthreads();
    EPI_T(6);
    // ---- RNALfold prints `prev` unless it is contained in `new` (the next start); the start-1
    // structure never takes part as `new`, and the last start>=2 structure is always printed.
    for (int k = wave; k + 1 < nst; k += NT / 64) {
        int prev_i = starts[k], new_i = starts[k + 1];
        if (new_i < 2) continue;
        int lp = lens[k], Ln = lens[k + 1];
        if (lp <= 0 || Ln <= 0) continue;
        int i = new_i - 1;
        int off = prev_i - i;
        const char* prev = out_ss + ((size_t)win * max_lines + k) * ss_stride + 1;
        const char* nw = out_ss + ((size_t)win * max_lines + k + 1) * ss_stride + 1;
        bool differ = false;
        for (int t = lane; t < lp; t += 64) {
            char a = (off + t < Ln) ? nw[off + t] : (char)0;
            if (a != prev[t]) differ = true;
        }
        bool anyd = __ballot(differ) != 0ull;
        bool print = (i + Ln < prev_i + lp) || anyd;
        if (lane == 0 && !print) out_lines[(size_t)win * max_lines + k].printed = 0;
    }
    EPI_T(7);
    if (tid == 0) {
        out_nlines[win] = sh_misc[1] ? sh_misc[3] : nst;      // over capacity (status 1): the number of lines the window needs
        out_mfe[win] = f3[1];
        out_status[win] = sh_misc[2] ? sh_misc[2] : (sh_misc[1] ? 1 : 0);
    }
    __syncthreads();
}


}  // namespace mirp
