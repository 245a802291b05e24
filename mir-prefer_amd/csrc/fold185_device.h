// Device code shared by the kernels of the "vienna-1.8.5" fold model (RNALfold 1.8.5 as the reference bundles it for Linux: Turner-1999
// parameters, dangles = 1, full multi-component backtracks; call site /root/reference/miR_PREFeR.py:3053-3064): energy terms, the
// wave-cooperative first-match backtrack and the exterior sweep + enumeration + output epilogue, templated on the table accessor
// (int32 tables of the generic kernel, or the 16-bit slabs written by the LDS-resident fill kernel).  Behaviour: SURVEY.md Appendix B (d1).
#pragma once
#include <hip/hip_runtime.h>
#include "fold_device.h"
#include "mirp_internal.h"

namespace mirp {
namespace v185 {

#if defined(MIRP_EPI_CLOCKS) && defined(EPI_T)      // phase clocks of the LDS-path epilogue (dev build; fold_epilogue.h defines the macros)
#define V185_T0() EPI_T0()
#define V185_T(k) EPI_T(k)
#else
#define V185_T0() do {} while (0)
#define V185_T(k) do {} while (0)
#endif

#define V_TURN 3
#define V_MAXLOOP 30
#define V_INF 1000000
#define V_NT 256
#ifndef V_G
#define V_G 2          // lanes per cell of the generic kernel's interval B (L = 400, four candidates a lane and turn: 1 -> 0.304 s, 2 -> 0.278, 4 -> 0.288, 8 -> 0.345)
#endif
#define V_BT_STACK 192
#ifndef MIRP_EPI_DMAX
#define MIRP_EPI_DMAX 300   // largest pair distance a tiled archive holds (fold_epilogue.h)
#endif
#define V_BT_LINE 16     // cells of a helix line fetched per round trip (every further cell is one more tile of the archive)

struct GTab185 {   // int32 tables in a global workspace (fold185_kernel)
    static constexpr bool kTiled = false;
    int* __restrict__ c;
    int* __restrict__ m;
    int* __restrict__ dm;
    const unsigned short* __restrict__ tb = nullptr;   // trace-back codes of the fill (round 5): 1 + (n1 << 5 | n2) = the interior loop the backtrack's search would find first, 0 = none
    int ld, n, M;
    __device__ __forceinline__ int C(int i, int j) const { const int d = j - i; return (d <= V_TURN || d > M || i < 1 || j > n) ? V_INF : c[(size_t)d * ld + i]; }
    __device__ __forceinline__ int Mm(int i, int j) const { const int d = j - i; return (d <= V_TURN || d > M || i < 1 || j > n) ? V_INF : m[(size_t)d * ld + i]; }
    __device__ __forceinline__ int DM(int i, int j) const { const int d = j - i; return (d <= V_TURN || d > M || i < 1 || j > n) ? V_INF : dm[(size_t)d * ld + i]; }
    __device__ __forceinline__ int TB(int i, int j) const { const int d = j - i; return !tb ? -1 : (d <= V_TURN || d > M || i < 1 || j > n) ? 0 : (int)tb[(size_t)d * ld + i]; }
};

template <class PT>
struct Ctx {
    const PT* __restrict__ P;
    const unsigned char* S;     // LDS 0..n+1
    const short* tetra;         // LDS: tetraloop bonus of the hairpin closed at i (0 = none)
    const int* f3;              // LDS
    int n, M;
    const short* dg = nullptr;  // optional LDS copy: dangle5[t * 5 + base] at [0, 40), dangle3 at [40, 80) (the global tables cost a round trip per read)
};
template <class PT>
__device__ __forceinline__ int D5(const Ctx<PT>& X, int t, int b) { return X.dg ? (int)X.dg[t * 5 + b] : X.P->dangle5[t][b]; }
template <class PT>
__device__ __forceinline__ int D3(const Ctx<PT>& X, int t, int b) { return X.dg ? (int)X.dg[40 + t * 5 + b] : X.P->dangle3[t][b]; }

template <class PT>
__device__ __forceinline__ int ptype(const Ctx<PT>& X, int i, int j) {
    const int d = j - i;
    if (d <= V_TURN || d > X.M - 1 || i < 1 || j > X.n) return 0;
    return pair_type(X.S[i], X.S[j]);
}
template <class PT>
__device__ __forceinline__ int AU(const Ctx<PT>& X, int t) { return t > 2 ? X.P->TerminalAU : 0; }
template <class PT>
__device__ __forceinline__ int MLi(const Ctx<PT>& X, int t) { return X.P->ML_intern + AU(X, t); }

template <class PT>
__device__ __forceinline__ int hairpin(const Ctx<PT>& X, int i, int j, int type) {
    const int u = j - i - 1;
    int e = X.P->hairpinE[u < MIRP_HP_MAX ? u : MIRP_HP_MAX - 1];
    if (u == 4) e += X.tetra[i];
    if (u == 3) e += AU(X, type);
    else e += X.P->mismatchH[type][X.S[i + 1]][X.S[j - 1]];
    return e;
}

template <class PT>
__device__ __forceinline__ int loopE(const Ctx<PT>& X, int n1, int n2, int type, int type2, int si1, int sj1, int sp1, int sq1) {
    const PT* __restrict__ P = X.P;
    const int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1;
    if (nl == 0) return P->stack[type][type2];
    if (ns == 0) {
        int e = P->bulge[nl];
        if (nl == 1) e += P->stack[type][type2];
        else e += AU(X, type) + AU(X, type2);
        return e;
    }
    if (ns == 1 && nl == 1) return P->int11[type][type2][si1][sj1];
    if (ns == 1 && nl == 2) return n1 == 1 ? P->int21[type][type2][si1][sq1][sj1] : P->int21[type2][type][sq1][si1][sp1];
    if (ns == 2 && nl == 2) return P->int22[type][type2][si1][sp1][sq1][sj1];
    int x = (nl - ns) * P->ninio;
    return P->internal_loop[n1 + n2] + (x < P->MAX_NINIO ? x : P->MAX_NINIO) + P->mismatchI[type][si1][sj1] + P->mismatchI[type2][sq1][sp1];
}

__device__ __forceinline__ int first_lane(unsigned long long mask) { return mask ? (__ffsll((long long)mask) - 1) : -1; }

// Wave-cooperative backtrack (all 64 lanes, wave-uniform arguments).  Returns the string length or a negative error code.
template <class PT, class TabT>
__device__ int backtrack(const Ctx<PT>& X, const TabT& T, int start, int maxdist, char* buf, int bufcap, int* stk) {
    const int lane = threadIdx.x & 63;
    const int n = X.n;
    const PT* __restrict__ P = X.P;
    const int len0 = (n - start < maxdist ? n - start : maxdist) + 1;
    if (len0 + 3 > bufcap) return -9;
    for (int x = lane; x < len0 + 3; x += 64) buf[x] = x < len0 ? '-' : (char)0;
    int sp = 0;
    if (lane == 0) { stk[0] = start; stk[1] = (n < start + maxdist + 1 ? n : start + maxdist + 1); stk[2] = 0; }
    sp = 1;
    __builtin_amdgcn_wave_barrier();
    while (sp > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sp--;
        int i = __builtin_amdgcn_readfirstlane(stk[3 * sp]), j = __builtin_amdgcn_readfirstlane(stk[3 * sp + 1]), ml = __builtin_amdgcn_readfirstlane(stk[3 * sp + 2]);
        if (j < i + V_TURN + 1) continue;
        if (sp + 3 >= V_BT_STACK) return -20;
        if (ml == 0) {
            const int fij = X.f3[i];
            if (fij == X.f3[i + 1]) {
                if (lane == 0) { stk[3 * sp] = i + 1; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 0; }
                sp++;
                continue;
            }
            // ascending scan over the partner k; inside one k the four tests run in the original order and later hits overwrite earlier ones
            int fk = -1, ftr = 0, fjj = 0;
            for (int kb = i + V_TURN + 1; kb <= j && fk < 0; kb += 64) {
                const int k = kb + lane;
                int traced = 0, jj = k + 1;
                if (k <= j) {
                    int t = ptype(X, i + 1, k);
                    if (t) {
                        const int cc = T.C(i + 1, k) + D5(X, t, X.S[i]) + AU(X, t);
                        if (fij == cc + X.f3[k + 1]) traced = i + 1;
                        if (k < n && fij == cc + X.f3[k + 2] + D3(X, t, X.S[k + 1])) { traced = i + 1; jj = k + 2; }
                    }
                    t = ptype(X, i, k);
                    if (t) {
                        const int cc = T.C(i, k) + AU(X, t);
                        if (fij == cc + X.f3[k + 1]) traced = i;
                        if (k < n && fij == cc + X.f3[k + 2] + D3(X, t, X.S[k + 1])) { traced = i; jj = k + 2; }
                    }
                }
                const int fl = first_lane(__ballot(traced != 0));
                if (fl >= 0) { fk = kb + fl; ftr = __shfl(traced, fl); fjj = __shfl(jj, fl); }
            }
            if (fk < 0) return -21;
            if (j == n) {
                if (lane == 0) { stk[3 * sp] = fjj; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 0; }
                sp++;
            }
            i = ftr; j = fk;
            if (lane == 0) {
                buf[i - start] = '(';
                buf[j - start] = ')';
                if (fjj == fk + 2) buf[fk + 1 - start] = '.';
            }
        } else {
            const int fij = T.Mm(i, j);
            if (T.Mm(i, j - 1) + 0 == fij) {
                if (lane == 0) { stk[3 * sp] = i; stk[3 * sp + 1] = j - 1; stk[3 * sp + 2] = 1; }
                sp++;
                continue;
            }
            if (T.Mm(i + 1, j) + 0 == fij) {
                if (lane == 0) { stk[3 * sp] = i + 1; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 1; }
                sp++;
                continue;
            }
            int t = ptype(X, i, j);
            const int cij = T.C(i, j) + MLi(X, t);
            t = ptype(X, i + 1, j);
            const int ci1j = T.C(i + 1, j) + D5(X, t, X.S[i]) + MLi(X, t);
            t = ptype(X, i, j - 1);
            const int cij1 = T.C(i, j - 1) + D3(X, t, X.S[j]) + MLi(X, t);
            t = ptype(X, i + 1, j - 1);
            const int ci1j1 = T.C(i + 1, j - 1) + D5(X, t, X.S[i]) + D3(X, t, X.S[j]) + MLi(X, t);
            if (fij == cij || fij == ci1j || fij == cij1 || fij == ci1j1) {
                if (fij == ci1j) i++;
                else if (fij == cij1) j--;
                else if (fij == ci1j1) { i++; j--; }
                if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
            } else {
                int found = -1;
                for (int kb = i + 1 + V_TURN; kb <= j - 2 - V_TURN && found < 0; kb += 64) {
                    const int k = kb + lane;
                    bool hit = false;
                    if (k <= j - 2 - V_TURN) hit = (fij == T.Mm(i, k) + T.Mm(k + 1, j));
                    const int fl = first_lane(__ballot(hit));
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
        }
        // (i,j) is a traced pair: follow stacks / interior loops until a hairpin or a multiloop
        for (;;) {
            // with trace-back codes the chain is followed a whole helix per memory round trip (see backtrack_wave in fold_epilogue.h)
            {
                const int il = i + lane, jl = j - lane;
                int cl = (lane < V_BT_LINE && jl - il >= V_TURN + 1) ? T.TB(il, jl) : 0;
                if (__builtin_amdgcn_readfirstlane(cl) > 0) {
                    int pos = 0;
                    for (;;) {
                        const int c = __builtin_amdgcn_readlane(cl, pos);
                        if (c <= 0) break;
                        const int n1 = (c - 1) >> 5, n2 = (c - 1) & 31;
                        i += 1 + n1; j -= 1 + n2;
                        if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
                        if (n1 != n2 || pos + 1 + n1 >= V_BT_LINE) { pos = -1; break; }
                        pos += 1 + n1;
                    }
                    if (pos < 0) continue;
                }
            }
            const int type = ptype(X, i, j);
            const int cij = T.C(i, j);
            if (cij == hairpin(X, i, j, type)) break;
            const int pmax = (j - 2 - V_TURN < i + V_MAXLOOP + 1) ? j - 2 - V_TURN : i + V_MAXLOOP + 1;
            int fp = -1, fq = -1;
            // trace-back code of the LDS fill kernel (>= 0): 1 + (n1 << 5 | n2) names the interior loop this search would find first, 0 = none
            const int code = __builtin_amdgcn_readfirstlane(T.TB(i, j));
            if (code > 0) { fp = i + 1 + ((code - 1) >> 5); fq = j - 1 - ((code - 1) & 31); }
            if (code < 0)
            for (int pb = i + 1; pb <= pmax && fp < 0; pb += 2) {
                const int p = pb + (lane >> 5), q = j - 1 - (lane & 31);
                int minq = j - i + p - V_MAXLOOP - 2;
                if (minq < p + 1 + V_TURN) minq = p + 1 + V_TURN;
                bool hit = false;
                if (p <= pmax && q >= minq) {
                    int t2 = ptype(X, p, q);
                    if (t2) {
                        t2 = rtype_of(t2);
                        hit = (cij == loopE(X, p - i - 1, j - q - 1, type, t2, X.S[i + 1], X.S[j - 1], X.S[p - 1], X.S[q + 1]) + T.C(p, q));
                    }
                }
                const int fl = first_lane(__ballot(hit));
                if (fl >= 0) { fp = pb + (fl >> 5); fq = j - 1 - (fl & 31); }
            }
            if (fp >= 0) {
                i = fp; j = fq;
                if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
                continue;
            }
            const int tt = rtype_of(type);
            const int mm = P->ML_closing + MLi(X, tt);
            const int e5 = D5(X, tt, X.S[j - 1]), e3 = D3(X, tt, X.S[i + 1]);
            int fk = -1, fv = 0;
            for (int kb = i + 2 + V_TURN; kb <= j - 3 - V_TURN && fk < 0; kb += 64) {
                const int k = kb + lane;
                int v = 0;      // 1: plain, 2: i1 = i+2, 3: j1 = j-2, 4: both (first in this order)
                if (k <= j - 3 - V_TURN) {
                    if (cij == T.Mm(i + 1, k) + T.Mm(k + 1, j - 1) + mm) v = 1;
                    else if (cij == T.Mm(i + 2, k) + T.Mm(k + 1, j - 1) + mm + e3) v = 2;
                    else if (cij == T.Mm(i + 1, k) + T.Mm(k + 1, j - 2) + mm + e5) v = 3;
                    else if (cij == T.Mm(i + 2, k) + T.Mm(k + 1, j - 2) + mm + e3 + e5) v = 4;
                }
                const int fl = first_lane(__ballot(v != 0));
                if (fl >= 0) { fk = kb + fl; fv = __shfl(v, fl); }
            }
            if (fk < 0) return -23;
            const int i1 = (fv == 2 || fv == 4) ? i + 2 : i + 1, j1 = (fv == 3 || fv == 4) ? j - 2 : j - 1;
            if (lane == 0) {
                stk[3 * sp] = i1; stk[3 * sp + 1] = fk; stk[3 * sp + 2] = 1;
                stk[3 * sp + 3] = fk + 1; stk[3 * sp + 4] = j1; stk[3 * sp + 5] = 1;
            }
            sp += 2;
            break;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // length = C string length (the 3' dangle dot of the last helix may sit one cell behind the nominal buffer), trailing '-' stripped
    int L = len0;
    while (L < len0 + 2 && buf[L] != 0) L++;
    while (L > 1 && buf[L - 1] == '-') L--;
    for (int x = lane; x < L; x += 64)
        if (buf[x] == '-') buf[x] = '.';
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return L;
}

// The same backtrack over the TILED archive of the LDS-resident fill kernel (TabT::kTiled, trace-back codes present): identical search orders
// and results, fewer trips to memory -- see backtrack_wave_tiled in fold_epilogue.h.  Every fetch is an 8 x 8 patch anchored at the current pair or
// segment (lane (a, b) holds cell (i + a, j - b): trace-back code, c and, for a multiloop segment, fML).  The run of unpaired bases of a
// multiloop segment (j first, then i, one pop each in the reference) is walked inside the fML patch, the four dangle variants of the segment's
// closing pair are the patch cells (r, c), (r+1, c), (r, c+1), (r+1, c+1), the helix follows the codes inside the patch, and all split points of a
// multiloop are fetched at once.
template <class PT, class TabT>
__device__ int backtrack_tiled(const Ctx<PT>& X, const TabT& T, int start, int maxdist, char* buf, int bufcap, int* stk) {
    const int lane = threadIdx.x & 63;
    const int n = X.n;
    const PT* __restrict__ P = X.P;
    const int pa = lane >> 3, pb = lane & 7;
    constexpr int NSR = 5;      // rounds of 64 split points that cover any segment (pair distances <= 300)
    const int len0 = (n - start < maxdist ? n - start : maxdist) + 1;
    if (len0 + 3 > bufcap) return -9;
    for (int x = lane; x < len0 + 3; x += 64) buf[x] = x < len0 ? '-' : (char)0;
    int sp = 0;
    if (lane == 0) { stk[0] = start; stk[1] = (n < start + maxdist + 1 ? n : start + maxdist + 1); stk[2] = 0; }
    sp = 1;
    __builtin_amdgcn_wave_barrier();
    while (sp > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sp--;
        int i = __builtin_amdgcn_readfirstlane(stk[3 * sp]), j = __builtin_amdgcn_readfirstlane(stk[3 * sp + 1]), ml = __builtin_amdgcn_readfirstlane(stk[3 * sp + 2]);
        if (j < i + V_TURN + 1) continue;
        if (sp + 3 >= V_BT_STACK) return -20;
        int tbv = 0, cv = V_INF;        // the current patch
        int r = 8, c = 8;               // the walk's position inside it (8: none fetched)
        if (ml == 0) {
            for (;;) {                  // unpaired 5' bases, 64 positions per step (the reference pops (i + 1, j) while f3[i] == f3[i + 1])
                const int x = i + lane;
                const bool diff = x <= n ? (X.f3[x] != X.f3[x + 1]) : true;
                const int fl = first_lane(__ballot(diff));
                if (fl < 0) { i += 64; continue; }
                i += fl;
                break;
            }
            if (j < i + V_TURN + 1) continue;
            const int fij = X.f3[i];
            // ascending scan over the partner k; inside one k the four tests run in the original order and later hits overwrite earlier ones
            int fk = -1, ftr = 0, fjj = 0;
            for (int kb = i + V_TURN + 1; kb <= j && fk < 0; kb += 64) {
                const int k = kb + lane;
                int traced = 0, jj = k + 1;
                if (k <= j) {
                    const int c1 = T.C(i + 1, k), c0 = T.C(i, k);
                    int t = ptype(X, i + 1, k);
                    if (t) {
                        const int cc = c1 + D5(X, t, X.S[i]) + AU(X, t);
                        if (fij == cc + X.f3[k + 1]) traced = i + 1;
                        if (k < n && fij == cc + X.f3[k + 2] + D3(X, t, X.S[k + 1])) { traced = i + 1; jj = k + 2; }
                    }
                    t = ptype(X, i, k);
                    if (t) {
                        const int cc = c0 + AU(X, t);
                        if (fij == cc + X.f3[k + 1]) traced = i;
                        if (k < n && fij == cc + X.f3[k + 2] + D3(X, t, X.S[k + 1])) { traced = i; jj = k + 2; }
                    }
                }
                const int fl = first_lane(__ballot(traced != 0));
                if (fl >= 0) { fk = kb + fl; ftr = __shfl(traced, fl); fjj = __shfl(jj, fl); }
            }
            if (fk < 0) return -21;
            if (j == n) {
                if (lane == 0) { stk[3 * sp] = fjj; stk[3 * sp + 1] = j; stk[3 * sp + 2] = 0; }
                sp++;
            }
            i = ftr; j = fk;
            if (lane == 0) {
                buf[i - start] = '(';
                buf[j - start] = ')';
                if (fjj == fk + 2) buf[fk + 1 - start] = '.';
            }
        } else {
            int fij;
            for (;;) {      // trim unpaired bases inside patches of fML: j first, then i, until neither matches
                const int ia = i + pa, jb = j - pb;
                const int mv = T.Mm(ia, jb);
                cv = T.C(ia, jb);
                tbv = (jb - ia >= V_TURN + 1 && jb - ia <= T.D) ? T.TB(ia, jb) : 0;
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
                if (edge) return -24;
                break;
            }
            // the segment's closing pair with its four dangle variants: patch cells (r, c), (r+1, c), (r, c+1), (r+1, c+1)   (r, c <= 6 here)
            int t = ptype(X, i, j);
            const int cij = __builtin_amdgcn_readlane(cv, r * 8 + c) + MLi(X, t);
            t = ptype(X, i + 1, j);
            const int ci1j = __builtin_amdgcn_readlane(cv, (r + 1) * 8 + c) + D5(X, t, X.S[i]) + MLi(X, t);
            t = ptype(X, i, j - 1);
            const int cij1 = __builtin_amdgcn_readlane(cv, r * 8 + c + 1) + D3(X, t, X.S[j]) + MLi(X, t);
            t = ptype(X, i + 1, j - 1);
            const int ci1j1 = __builtin_amdgcn_readlane(cv, (r + 1) * 8 + c + 1) + D5(X, t, X.S[i]) + D3(X, t, X.S[j]) + MLi(X, t);
            if (fij == cij || fij == ci1j || fij == cij1 || fij == ci1j1) {
                if (fij == ci1j) { i++; r++; }                 // the reference's order of preference
                else if (fij == cij1) { j--; c++; }
                else if (fij == ci1j1) { i++; j--; r++; c++; }
                if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
            } else {
                // all split points at once, first match ascending
                int m1[NSR], m2[NSR];
                const int k0 = i + 1 + V_TURN, k1 = j - 2 - V_TURN;
#pragma unroll
                for (int q = 0; q < NSR; q++) {
                    m1[q] = V_INF; m2[q] = V_INF;
                    if (k0 + 64 * q <= k1) {
                        const int k = k0 + 64 * q + lane;
                        if (k <= k1) { m1[q] = T.Mm(i, k); m2[q] = T.Mm(k + 1, j); }
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
        }
        // (i,j) is a traced pair at position (r, c) of the current patch (> 7: not in it): follow the trace-back codes until a hairpin or a multiloop
        for (;;) {
            if (r > 7 || c > 7) {
                const int ia = i + pa, jb = j - pb;
                cv = T.C(ia, jb);
                tbv = (jb - ia >= V_TURN + 1 && jb - ia <= T.D) ? T.TB(ia, jb) : 0;
                r = 0; c = 0;
            }
            for (;;) {
                const int code = __builtin_amdgcn_readlane(tbv, r * 8 + c);
                if (code <= 0) break;
                const int n1 = (code - 1) >> 5, n2 = (code - 1) & 31;
                i += 1 + n1; j -= 1 + n2;
                if (lane == 0) { buf[i - start] = '('; buf[j - start] = ')'; }
                r += 1 + n1; c += 1 + n2;
                if (r > 7 || c > 7) break;
            }
            if (r > 7 || c > 7) continue;
            const int cij = __builtin_amdgcn_readlane(cv, r * 8 + c);
            const int type = ptype(X, i, j);
            if (cij == hairpin(X, i, j, type)) break;
            const int tt = rtype_of(type);
            const int mm = P->ML_closing + MLi(X, tt);
            const int e5 = D5(X, tt, X.S[j - 1]), e3 = D3(X, tt, X.S[i + 1]);
            // all split points at once; per point the four variants in the original order, first point ascending
            int fk = -1, fv = 0;
            {
                int a1[NSR], a2[NSR], b1[NSR], b2[NSR];
                const int k0 = i + 2 + V_TURN, k1 = j - 3 - V_TURN;
#pragma unroll
                for (int q = 0; q < NSR; q++) {
                    a1[q] = V_INF; a2[q] = V_INF; b1[q] = V_INF; b2[q] = V_INF;
                    if (k0 + 64 * q <= k1) {
                        const int k = k0 + 64 * q + lane;
                        if (k <= k1) { a1[q] = T.Mm(i + 1, k); a2[q] = T.Mm(i + 2, k); b1[q] = T.Mm(k + 1, j - 1); b2[q] = T.Mm(k + 1, j - 2); }
                    }
                }
#pragma unroll
                for (int q = 0; q < NSR; q++) {
                    if (k0 + 64 * q <= k1 && fk < 0) {
                        int v = 0;      // 1: plain, 2: i1 = i+2, 3: j1 = j-2, 4: both (first in this order)
                        if (cij == a1[q] + b1[q] + mm) v = 1;
                        else if (cij == a2[q] + b1[q] + mm + e3) v = 2;
                        else if (cij == a1[q] + b2[q] + mm + e5) v = 3;
                        else if (cij == a2[q] + b2[q] + mm + e3 + e5) v = 4;
                        const int fl = first_lane(__ballot(v != 0));
                        if (fl >= 0) { fk = k0 + 64 * q + fl; fv = __shfl(v, fl); }
                    }
                }
            }
            if (fk < 0) return -23;
            const int i1 = (fv == 2 || fv == 4) ? i + 2 : i + 1, j1 = (fv == 3 || fv == 4) ? j - 2 : j - 1;
            if (lane == 0) {
                stk[3 * sp] = i1; stk[3 * sp + 1] = fk; stk[3 * sp + 2] = 1;
                stk[3 * sp + 3] = fk + 1; stk[3 * sp + 4] = j1; stk[3 * sp + 5] = 1;
            }
            sp += 2;
            break;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int L = len0;
    while (L < len0 + 2 && buf[L] != 0) L++;
    while (L > 1 && buf[L - 1] == '-') L--;
    for (int x = lane; x < L; x += 64)
        if (buf[x] == '-') buf[x] = '.';
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return L;
}

// Exterior sweep over the TILED archive (TabT::kTiled: 8 x 8 tiles over (row, diagonal), 16 bytes = the 8 rows of one diagonal), the layout of
// f3_sweep_tiled of the default model: a wave takes one row block of 8 rows and has all of its tiles in flight at once, lane = diagonal, a wave-wide
// load is 1 KB of consecutive tiles.  What dangles 1 adds is the c(i+1, j) term: cell (i+1, j) is row + 1 on diagonal - 1, i.e. the NEXT row of the
// PREVIOUS lane's vector -- one lane shift of the loaded vectors (lane 0 takes lane 63 of the group before), and for the block's last row the first
// row of the next row block on the previous lane's diagonal (a ninth, 2-byte value per lane).  (The row-per-lane sweep this replaces read 16 bytes
// of a 128-byte tile per access: 4.7 of the epilogue's 10.1 ms.)  Partners whose f3 is final go into the lane's eight running minima, which meet
// in LDS; partners inside the block of 8 x (waves) rows park their two terms in LDS by the row whose f3 they wait for; wave 0 then runs the chain
// through the block in registers.
template <class PT, class TabT, int NT>
__device__ void sweep_tiled(const Ctx<PT>& X, const TabT& T, int* f3, int* scratch) {
    constexpr int NW = NT / 64, RBK = 8 * NW;
    constexpr int NG = (MIRP_EPI_DMAX + 1 - V_TURN - 1) / 64 + 1;      // groups of 64 partner distances 4 .. span (c reaches span - 1, the c(i+1, j) term one further)
    static_assert(RBK <= 64, "the chain maps the rows of a block onto the lanes of one wave");
    static_assert(RBK <= 64 + V_TURN, "only the first group of distances has partners inside the block");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = X.n, D = T.D;
    const unsigned char* S = X.S;
    int* part = scratch;                      // [RBK] per-row minimum over the partners whose f3 is final
    int* innerA = scratch + RBK;              // [RBK (row waited for)][RBK (row)] term continued by f3[j+1]
    int* innerB = innerA + RBK * RBK;         // the same for the 3' dangle term, continued by f3[j+2]
    const int tau = X.P->TerminalAU;
    const int top = n - V_TURN - 1;           // highest row that can pair
    for (int g = top >= 1 ? (top - 1) / RBK : -1; g >= 0; g--) {
        const int blk_lo = g * RBK + 1, blk_hi = blk_lo + RBK - 1;
        const int tb = g * NW + wave, i0 = 8 * tb + 1;
        const int dmax_b = D < n - i0 ? D : n - i0;                    // wave-uniform: the block's first row reaches furthest
        const uint4* base = reinterpret_cast<const uint4*>(T.carch + T.off[tb]) + lane;
        const unsigned short* nbase = reinterpret_cast<const unsigned short*>(T.carch) + (i0 + 8 <= n ? T.off[tb + 1] : 0) + 8 * lane;
        // two groups of 64 distances in flight (the one at hand and the next): all five at once cost 25 registers under this kernel's budget of 64
        auto load = [&](int u, uint4& c4, unsigned& x1) {
            const int d = V_TURN + 1 + 64 * u + lane;
            c4 = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
            x1 = 0x7fffu;
            if (d <= dmax_b) c4 = base[64 * u];
            if (d <= D && i0 + 8 + d <= n) x1 = nbase[8 * 64 * u];
        };
        uint4 cnext; unsigned xnext;
        load(0, cnext, xnext);
        for (int x = tid; x < RBK + 2 * RBK * RBK; x += NT) part[x] = V_INF;
        __syncthreads();
        {
            int best[8];
#pragma unroll
            for (int rr = 0; rr < 8; rr++) best[rr] = V_INF;
            const int vS = S[i0 + lane <= n + 1 ? i0 + lane : n + 1];      // lanes 0 .. 8: the bases of rows i0 .. i0 + 8 (wave-uniform per row: v_readlane)
            unsigned last[5] = {0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fffu};      // lane 63 of the group before (scalars); distance 3 does not pair
#pragma unroll 1
            for (int u = 0; u < NG; u++) {
                const uint4 cgu = cnext; const unsigned nxu = xnext;
                if (u + 1 < NG && V_TURN + 1 + 64 * (u + 1) <= dmax_b + 1) load(u + 1, cnext, xnext);
                if (V_TURN + 1 + 64 * u <= dmax_b + 1) {              // wave-uniform (distance dmax_b + 1 still has its c(i+1, j) term)
                    const int d = V_TURN + 1 + 64 * u + lane;
                    const int j0 = i0 + d;                            // partner of row i0; row i0 + rr pairs with j0 + rr
                    // the previous lane's vector: rows i0 .. i0 + 8 of diagonal d - 1
                    unsigned pw[5];
                    {
                        const unsigned own[5] = {cgu.x, cgu.y, cgu.z, cgu.w, nxu};
#pragma unroll
                        for (int q = 0; q < 5; q++) {
                            const unsigned v = (unsigned)__shfl_up((int)own[q], 1);
                            pw[q] = lane == 0 ? last[q] : v;
                            last[q] = (unsigned)__builtin_amdgcn_readlane((int)own[q], 63);
                        }
                    }
                    const unsigned cw[4] = {cgu.x, cgu.y, cgu.z, cgu.w};
                    int sjn = S[j0 <= n + 1 ? j0 : n + 1];           // base of j (carried from row to row), then of j + 1 (out of range: never used)
#pragma unroll
                    for (int rr = 0; rr < 8; rr++) {
                        const int i = i0 + rr, j = j0 + rr;
                        const int sj0 = sjn;
                        sjn = S[j + 1 <= n + 1 ? j + 1 : n + 1];
                        const int ca = (int)(short)((rr & 1) ? cw[rr >> 1] >> 16 : cw[rr >> 1] & 0xffffu);
                        const int r1 = rr + 1;
                        const int cb = r1 < 8 ? (int)(short)((r1 & 1) ? pw[r1 >> 1] >> 16 : pw[r1 >> 1] & 0xffffu) : (int)(short)(pw[4] & 0xffffu);
                        const bool inw = j <= n;
                        const int si = __builtin_amdgcn_readlane(vS, rr), si1 = __builtin_amdgcn_readlane(vS, rr + 1);
                        int a = V_INF, b = V_INF;
                        if (inw && ca != 0x7fff && d <= D) {          // c is finite only where the two bases pair
                            const int t = pair_type(si, sj0);
                            const int e = ca + (t > 2 ? tau : 0);
                            a = e;
                            if (j < n) b = e + D3(X, t, sjn);
                        }
                        if (inw && cb != 0x7fff && d - 1 > V_TURN && d - 1 <= D) {
                            const int t = pair_type(si1, sj0);
                            const int e = cb + D5(X, t, si) + (t > 2 ? tau : 0);
                            a = e < a ? e : a;
                            if (j < n) { const int v = e + D3(X, t, sjn); b = v < b ? v : b; }
                        }
                        if (a < V_INF) {
                            if (j >= blk_hi) {                        // f3[j+1], f3[j+2] final (f3[n+1] = f3[n+2] = 0)
                                int v = f3[j + 1] + a; best[rr] = v < best[rr] ? v : best[rr];
                                if (j < n) { v = f3[j + 2] + b; best[rr] = v < best[rr] ? v : best[rr]; }
                            } else {                                  // continued by rows of this block (first group of distances only)
                                innerA[(j + 1 - blk_lo) * RBK + (i - blk_lo)] = a;
                                if (j + 2 > blk_hi) { const int v = f3[j + 2] + b; best[rr] = v < best[rr] ? v : best[rr]; }
                                else innerB[(j + 2 - blk_lo) * RBK + (i - blk_lo)] = b;
                            }
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
        __syncthreads();
        if (wave == 0) {
            const int r = lane & (RBK - 1);
            int best = part[r];
            int fx = f3[blk_hi + 1 <= n + 2 ? blk_hi + 1 : n + 2], fo = 0;
#pragma unroll 4
            for (int xr = RBK - 1; xr >= 0; xr--) {
                const int bx = __builtin_amdgcn_readlane(best, xr);
                fx = bx < fx ? bx : fx;                                // f3 of row blk_lo + xr
                fo = lane == xr ? fx : fo;
                const int ea = innerA[xr * RBK + r], eb = innerB[xr * RBK + r];      // terms of row r that wait for this row
                int v = ea + fx; v = ea < V_INF ? v : V_INF; best = v < best ? v : best;
                v = eb + fx; v = eb < V_INF ? v : V_INF; best = v < best ? v : best;
            }
            if (lane < RBK && blk_lo + lane <= n) f3[blk_lo + lane] = fo;
        }
        __syncthreads();
    }
}

// Exterior sweep (sequential in i, partners reduced in parallel), enumeration of the structure starts, one backtrack per start and wave,
// RNALfold's "print prev unless contained in new" rule, output records.  Called by all NT threads of the workgroup; f3 must be zero-filled.
template <class PT, class TabT, int NT>
__device__ void epilogue(const Ctx<PT>& X, const TabT& T, int* f3, int* starts, int* lens, int* btstk, int* red, char* btbuf, int nc, int win, int max_lines,
                         int ss_stride, MirpFoldLine* __restrict__ out_lines, char* __restrict__ out_ss, int* __restrict__ out_nlines,
                         int* __restrict__ out_mfe, int* __restrict__ out_status) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = X.n, M = X.M;
    const PT* __restrict__ P = X.P;
    const unsigned char* S = X.S;
    V185_T0();
        // ---- exterior sweep: f3[i] = min(f3[i+1], min_j { c(i,j), c(i+1,j) + dangle5 } + AU, continued by f3[j+1] or dangle3 + f3[j+2]) is sequential in i only
        // through f3, so rows go in blocks of RB.  Step 1 (all waves): a half-wave holds the rows of the block (lane = row) and walks the partners
        // diagonal by diagonal -- its 32 cells of one archived diagonal are one contiguous read.  Partners at or above the block top have
        // final f3 values and are reduced to one minimum per row; the few partners inside the block leave their two terms in LDS.  Step 2
        // (wave 0): the short sequential chain through the block touches LDS only.  The backtrack stacks are idle here and serve as scratch.
        if constexpr (TabT::kTiled) {
            static_assert((NT / 64) * 8 + 2 * (NT / 64) * 8 * (NT / 64) * 8 <= (NT / 64) * 3 * V_BT_STACK, "exterior-sweep scratch must fit the backtrack stacks");
            sweep_tiled<PT, TabT, NT>(X, T, f3, btstk);
        } else {
            constexpr int NW = NT / 64, RB = 32;
            static_assert(RB + 2 * RB * RB <= NW * 3 * V_BT_STACK, "exterior-sweep scratch must fit the backtrack stacks");
            int* part = btstk;                  // [RB] per-row minimum over the partners with final f3
            int* innerA = btstk + RB;           // [RB][RB] term continued by f3[j+1]
            int* innerB = innerA + RB * RB;     // [RB][RB] term continued by f3[j+2] (3' dangle)
            for (int i_hi = n - V_TURN - 1; i_hi >= 1; i_hi -= RB) {
                for (int x = tid; x < RB + 2 * RB * RB; x += NT) part[x] = V_INF;
                __syncthreads();
                {
                    const int r = lane & (RB - 1), i = i_hi - r;
                    int best = V_INF;
                    {
                        const int jmax = i >= 1 ? ((i + M < n) ? i + M : n) : 0;
                        constexpr int UNR = 4;      // archive reads in flight per lane (two rows each): the sweep is a chain of memory round trips otherwise
                        for (int d0 = V_TURN + 1 + 2 * wave + (lane >> 5); d0 <= M; d0 += 2 * NW * UNR) {
                            int ca[UNR], cb[UNR];
#pragma unroll
                            for (int u = 0; u < UNR; u++) {
                                const int j = i + d0 + u * 2 * NW;
                                const bool in = i >= 1 && j <= jmax;
                                ca[u] = in ? T.C(i, j) : V_INF;
                                cb[u] = in ? T.C(i + 1, j) : V_INF;
                            }
#pragma unroll
                            for (int u = 0; u < UNR; u++) {
                                const int d = d0 + u * 2 * NW, j = i + d;
                                if (!(i >= 1 && j <= jmax)) continue;
                                int a = V_INF, b = V_INF;
                                // c is finite only where the two bases pair: the pair type (two sequence reads) is looked up for those cells only
                                if (ca[u] < V_INF) {
                                    const int t = ptype(X, i, j);
                                    const int e = ca[u] + AU(X, t);
                                    a = e;
                                    if (j < n) b = e + D3(X, t, S[j + 1]);
                                }
                                if (cb[u] < V_INF) {
                                    const int t = ptype(X, i + 1, j);
                                    const int e = cb[u] + D5(X, t, S[i]) + AU(X, t);
                                    a = e < a ? e : a;
                                    if (j < n) { const int v = e + D3(X, t, S[j + 1]); b = v < b ? v : b; }
                                }
                                if (a < V_INF) {
                                    if (j >= i_hi) {        // f3[j+1], f3[j+2] final (f3[n+1] = f3[n+2] = 0)
                                        int v = f3[j + 1] + a; best = v < best ? v : best;
                                        if (j < n) { v = f3[j + 2] + b; best = v < best ? v : best; }
                                    } else {                // continued by rows of this block: parked transposed, by the row whose f3 they wait for
                                        innerA[(i_hi - j - 1) * RB + r] = a;
                                        if (j + 2 > i_hi) { const int v = f3[j + 2] + b; best = v < best ? v : best; }
                                        else innerB[(i_hi - j - 2) * RB + r] = b;
                                    }
                                }
                            }
                        }
                    }
                    if (best < V_INF) atomicMin(&part[r], best);
                }
                __syncthreads();
                if (wave == 0) {
                    // the sequential chain through the block in registers (lane = row r, i = i_hi - r): when f3 of row x becomes final, every row that
                    // parked a term waiting for it takes it (one v_readlane + two adds + two mins per row; the LDS reads do not depend on the chain)
                    const int r = lane & (RB - 1);
                    int best = part[r];
                    int fx = f3[i_hi + 1], fo = 0;
#pragma unroll 4
                    for (int xr = 0; xr < RB; xr++) {
                        const int bx = __builtin_amdgcn_readlane(best, xr);
                        fx = bx < fx ? bx : fx;                            // f3 of row i_hi - xr
                        fo = lane == xr ? fx : fo;
                        const int ea = innerA[xr * RB + r], eb = innerB[xr * RB + r];
                        int v = ea + fx; v = ea < V_INF ? v : V_INF; best = v < best ? v : best;
                        v = eb + fx; v = eb < V_INF ? v : V_INF; best = v < best ? v : best;
                    }
                    if (lane < RB && i_hi - lane >= 1) f3[i_hi - lane] = fo;
                }
                __syncthreads();
            }
        }

        V185_T(0);
        // ---- structure starts, descending: l >= 2 with f3[l] != f3[l+1] && f3[l-1] == f3[l]; l == 1 with f3[1] != f3[2], or when there is
        // no other start at all (probed on the binary: a structure-free window still prints the start-1 backtrack ".")
        if (wave == 0) {
            int cnt = 0;
            for (int base = n - V_TURN - 1; base >= 2; base -= 64) {
                const int l = base - lane;
                bool is = false;
                if (l >= 2) is = (f3[l] != f3[l + 1]) && (f3[l - 1] == f3[l]);
                const unsigned long long mask = __ballot(is);
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                if (is && cnt + rank < max_lines) starts[cnt + rank] = l;
                cnt += __popcll(mask);
            }
            if (n >= V_TURN + 2 && (f3[1] != f3[2] || cnt == 0)) {
                if (lane == 0 && cnt < max_lines) starts[cnt] = 1;
                cnt++;
            }
            if (lane == 0) { red[8] = cnt < max_lines ? cnt : max_lines; red[9] = cnt > max_lines ? 1 : 0; red[10] = 0; red[11] = cnt; red[7] = 0; }
        }
        __syncthreads();
        const int nst = red[8];
        V185_T(1);
        char* mybuf = btbuf + wave * (nc + 8);
        int* mystk = btstk + wave * 3 * V_BT_STACK;
        // structures differ a lot in length: the waves draw them from a counter (red[7]) instead of striding (bounded trip count, as in fold_epilogue.h)
        for (int it = 0; it <= nst; it++) {
            int t = 0;
            if (lane == 0) t = atomicAdd(&red[7], 1);
            const int k = __builtin_amdgcn_readfirstlane(t);
            if (k >= nst) break;
            const int lind = starts[k];
            int L;
            if constexpr (TabT::kTiled) L = backtrack_tiled(X, T, lind, lind == 1 ? M : M + 1, mybuf, nc + 8, mystk);
            else L = backtrack(X, T, lind, lind == 1 ? M : M + 1, mybuf, nc + 8, mystk);
            V185_T(3);
            if (L < 0) { if (lane == 0) { red[10] = L; lens[k] = 0; } continue; }
            if (L + 1 > ss_stride) { if (lane == 0) { red[10] = -30; lens[k] = 0; } continue; }
            char* dst = out_ss + ((size_t)win * max_lines + k) * ss_stride;
            for (int x = lane; x < L; x += 64) dst[x] = mybuf[x];
            if (lane == 0) {
                dst[L] = 0;
                lens[k] = L;
                MirpFoldLine ln;
                ln.start = lind; ln.len = L; ln.energy = f3[lind] - f3[lind + L]; ln.printed = 1;
                out_lines[(size_t)win * max_lines + k] = ln;
            }
            V185_T(4);
        }
        V185_T(5);
        __syncthreads();
        V185_T(6);
        // ---- RNALfold prints `prev` unless it is contained in `new` (the next start); the start-1 structure never takes part as `new`
        for (int k = wave; k + 1 < nst; k += NT / 64) {
            const int prev_i = starts[k], new_i = starts[k + 1];
            if (new_i < 2) continue;
            const int lp = lens[k], Ln = lens[k + 1];
            if (lp <= 0 || Ln <= 0) continue;
            const int i = new_i - 1;
            const int off = prev_i - i;
            const char* prev = out_ss + ((size_t)win * max_lines + k) * ss_stride;
            const char* nw = out_ss + ((size_t)win * max_lines + k + 1) * ss_stride;
            bool differ = false;
            for (int t = lane; t < lp; t += 64) {
                const char a = (off + t < Ln) ? nw[off + t] : (char)0;
                if (a != prev[t]) differ = true;
            }
            const bool anyd = __ballot(differ) != 0ull;
            const bool print = (i + Ln < prev_i + lp) || anyd;
            if (lane == 0 && !print) out_lines[(size_t)win * max_lines + k].printed = 0;
        }
        if (tid == 0) {
            out_nlines[win] = red[9] ? red[11] : nst;      // over capacity (status 1): the number of lines the window needs
            out_mfe[win] = f3[1];
            out_status[win] = red[10] ? red[10] : (red[9] ? 1 : 0);
        }
        __syncthreads();
}

}  // namespace v185
}  // namespace mirp
