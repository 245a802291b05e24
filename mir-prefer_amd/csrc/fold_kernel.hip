// Batched L-bounded Zuker MFE local fold with enumeration of locally optimal structures.
// One precursor window per workgroup.  Replaces `RNALfold -L <PRECURSOR_LEN>`
// (/root/reference/miR_PREFeR.py:3047-3119, command line :3053) for gfx950.
//
// This file holds the GENERIC kernel: DP tables (c, fML) live in a per-workgroup global
// workspace laid out diagonal-major ((d, i) -> d*ld + i) so that every table read of the
// anti-diagonal wavefront fill is coalesced across the lanes that own consecutive cells.
// It supports any window length up to MIRP_NMAX and any span, and is the fallback for
// windows the LDS-resident fast kernel (fold_lds_kernel.hip) flags as out of its range.
#include <hip/hip_runtime.h>
#include "fold_device.h"
#include "mirp_internal.h"

namespace mirp {

#define TURN MIRP_TURN
#define MAXLOOP MIRP_MAXLOOP
#define INF MIRP_INF

struct GTab {
    int* __restrict__ c;
    int* __restrict__ m;
    int ld;
    __device__ __forceinline__ int C(int d, int i) const { return c[(size_t)d * ld + i]; }
    __device__ __forceinline__ int M(int d, int i) const { return m[(size_t)d * ld + i]; }
};

struct WinCtx {
    const FoldParams* __restrict__ P;
    const unsigned char* S;   // LDS, 0..n+1
    const unsigned char* seq; // LDS, upper-case RNA chars, 1-based
    const int* f3;            // LDS, 0..n+2
    const short* spec;        // LDS, special hairpin energy per i for u==3,4,6 at [k*ldspec + i], SHRT_MIN = none
    int ldspec;
    int n, D;                 // D = max pair distance (min(span-1, n-1))
};

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
        return X.P->hairpinE[3] + (type > 2 ? X.P->TerminalAU : 0);
    }
    return X.P->hairpinE[u] + X.P->mismatchH[type][X.S[i + 1]][X.S[j - 1]];
}

__device__ __forceinline__ int ext_term(const WinCtx& X, int i, int k, int type) {
    return e_extloop(X.P, type, i > 1 ? (int)X.S[i - 1] : -1, k < X.n ? (int)X.S[k + 1] : -1);
}
__device__ __forceinline__ int ml_term(const WinCtx& X, int i, int j, int type) {
    return e_mlstem(X.P, type, i > 1 ? (int)X.S[i - 1] : -1, j < X.n ? (int)X.S[j + 1] : -1);
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

// first set lane of a 64-bit ballot, or -1
__device__ __forceinline__ int first_lane(unsigned long long mask) { return mask ? (__ffsll((long long)mask) - 1) : -1; }

#define BT_STACK 96

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
            for (int kb = j; kb >= i + TURN + 1 && found < 0; kb -= 64) {
                int k = kb - lane;
                bool ok = false;
                if (k >= i + TURN + 1) {
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
            if (type) ok = (T.C(d, i) + ml_term(X, i, j, type) == fij);
            if (!ok) {
                int found = -1;
                for (int kb = i + 1 + TURN; kb <= j - 2 - TURN && found < 0; kb += 64) {
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
            int type = ptype_at(X, i, j);
            int cij = T.C(j - i, i);
            if (cij == e_hairpin(X, i, j, type)) break;
            int pmax = (j - 2 - TURN < i + MAXLOOP + 1) ? j - 2 - TURN : i + MAXLOOP + 1;
            int fp = -1, fq = -1;
            for (int pb = i + 1; pb <= pmax && fp < 0; pb += 2) {
                int p = pb + (lane >> 5), q = j - 1 - (lane & 31);
                int minq = j - i + p - MAXLOOP - 2;
                if (minq < p + 1 + TURN) minq = p + 1 + TURN;
                bool hit = false;
                if (p <= pmax && q >= minq) {
                    int t2 = pair_type(X.S[p], X.S[q]);
                    if (t2) {
                        t2 = rtype_of(t2);
                        int e = e_intloop(X.P, p - i - 1, j - q - 1, type, t2, X.S[i + 1], X.S[j - 1], X.S[p - 1], X.S[q + 1]);
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
            int mm = X.P->ML_closing + e_mlstem(X.P, rtype_of(type), X.S[j - 1], X.S[i + 1]);
            int found = -1;
            for (int kb = i + 2 + TURN; kb <= j - 3 - TURN && found < 0; kb += 64) {
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
    // ---- f3 (exterior) sweep: sequential in i, lane-parallel over the partner j (wave 0)
    for (int x = tid; x < n + 3; x += NT) f3[x] = 0;
    __syncthreads();
    if (wave == 0) {
        for (int i = n - TURN - 1; i >= 1; i--) {
            int best = f3[i + 1];
            int jmax = (i + D < n) ? i + D : n;
            for (int j = i + TURN + 1 + lane; j <= jmax; j += 64) {
                int type = pair_type(X.S[i], X.S[j]);
                if (type) {
                    int e = f3[j + 1] + T.C(j - i, i) + ext_term(X, i, j, type);
                    best = e < best ? e : best;
                }
            }
            best = wave_min(best);
            if (lane == 0) f3[i] = best;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
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
        if (lane == 0) { sh_misc[0] = cnt < max_lines ? cnt : max_lines; sh_misc[1] = cnt > max_lines ? 1 : 0; sh_misc[2] = 0; }
    }
    __syncthreads();
    const int nst = sh_misc[0];
    // ---- backtracks: one wave per start
    char* mybuf = btbuf + wave * bufstride;
    int* mystk = btstk + wave * 3 * BT_STACK;
    for (int k = wave; k < nst; k += NT / 64) {
        int lind = starts[k];
        int fij = f3[lind];
        // "short backtrack": first partner (ascending) that realises f3[lind]
        int pp = -1;
        for (int pb = lind + TURN; pb <= lind + span && pp < 0; pb += 64) {
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
        if (pp >= 0) L = backtrack_wave(X, T, lind, (pp + 2 < n ? pp + 2 : n), span, mybuf, mystk);
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
    }
    __syncthreads();
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
    if (tid == 0) {
        out_nlines[win] = nst;
        out_mfe[win] = f3[1];
        out_status[win] = sh_misc[2] ? sh_misc[2] : (sh_misc[1] ? 1 : 0);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Generic kernel: tables in global workspace.
// ------------------------------------------------------------------------------------------
#define GEN_NT 256
#define GEN_G 8   // lanes cooperating on one cell

__global__ void __launch_bounds__(GEN_NT) fold_generic_kernel(
    const FoldParams* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs,
    const int* __restrict__ win_lens, const int* __restrict__ work_list, int n_work, int span, int n_cap, int* __restrict__ ws, size_t ws_slot_ints,
    int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines, char* __restrict__ out_ss,
    int* __restrict__ out_nlines, int* __restrict__ out_mfe, int* __restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS carve-up (n_cap = max window length this launch supports)
    const int nc = n_cap + 8;
    int* f3 = (int*)smem;                                  // nc ints
    int* starts = f3 + nc;                                 // max_lines
    int* lens = starts + max_lines;                        // max_lines
    int* btstk = lens + max_lines;                         // (NT/64)*3*BT_STACK
    int* sh_misc = btstk + (GEN_NT / 64) * 3 * BT_STACK;   // 8
    short* spec = (short*)(sh_misc + 8);                   // 3*nc
    unsigned char* S = (unsigned char*)(spec + 3 * nc);    // nc
    unsigned char* seq = S + nc;                           // nc
    char* btbuf = (char*)(seq + nc);                       // (NT/64)*nc

    const int tid = threadIdx.x;
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int win = work_list ? work_list[w] : w;
        const long long o0 = offs[win];
        const int n = win_lens ? win_lens[win] : (int)(offs[win + 1] - o0);
        if (n < 1 || n > n_cap) {
            if (tid == 0) { out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = n < 1 ? 0 : -40; }
            continue;
        }
        const int D = (span - 1 < n - 1) ? span - 1 : n - 1;
        // ---- stage the sequence: upper-case, T->U, numeric code
        for (int x = tid; x <= n + 1; x += GEN_NT) {
            unsigned char ch = 0;
            if (x >= 1 && x <= n) {
                ch = seqs[o0 + x - 1];
                if (ch >= 'a' && ch <= 'z') ch -= 32;
                if (ch == 'T') ch = 'U';
            }
            seq[x] = ch;
            S[x] = ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'U' ? 4 : 0;
        }
        __syncthreads();
        if (tid == 0) { S[0] = S[n]; S[n + 1] = S[1]; }
        // special hairpin motifs (tri/tetra/hexa loops) per closing position i
        for (int x = tid; x <= n; x += GEN_NT) {
            short s3 = -32768, s4 = -32768, s6 = -32768;
            if (x >= 1) {
                if (x + 4 <= n)
                    for (int k = 0; k < 2; k++) {
                        bool m = true;
                        for (int t = 0; t < 5; t++) m = m && (seq[x + t] == (unsigned char)P->tri[k][t]);
                        if (m && s3 == -32768) s3 = (short)P->triE[k];
                    }
                if (x + 5 <= n)
                    for (int k = 0; k < 16; k++) {
                        bool m = true;
                        for (int t = 0; t < 6; t++) m = m && (seq[x + t] == (unsigned char)P->tetra[k][t]);
                        if (m && s4 == -32768) s4 = (short)P->tetraE[k];
                    }
                if (x + 7 <= n)
                    for (int k = 0; k < 4; k++) {
                        bool m = true;
                        for (int t = 0; t < 8; t++) m = m && (seq[x + t] == (unsigned char)P->hexa[k][t]);
                        if (m && s6 == -32768) s6 = (short)P->hexaE[k];
                    }
            }
            spec[x] = s3; spec[nc + x] = s4; spec[2 * nc + x] = s6;
        }
        GTab T;
        T.ld = n_cap + 2;
        T.c = ws + (size_t)blockIdx.x * ws_slot_ints;
        T.m = T.c + ws_slot_ints / 2;
        // diagonal TURN of fML must read as INF
        for (int x = tid; x <= n; x += GEN_NT) T.m[(size_t)TURN * T.ld + x] = INF;
        __syncthreads();
        WinCtx X;
        X.P = P; X.S = S; X.seq = seq; X.f3 = f3; X.spec = spec; X.ldspec = nc; X.n = n; X.D = D;

        // ---- anti-diagonal wavefront fill: all cells with the same d = j - i are independent
        const int sub = tid % GEN_G;
        for (int d = TURN + 1; d <= D; d++) {
            const int ncell = n - d;
            for (int cell = tid / GEN_G; cell < ((ncell + GEN_NT / GEN_G - 1) / (GEN_NT / GEN_G)) * (GEN_NT / GEN_G); cell += GEN_NT / GEN_G) {
                const bool live = cell < ncell;
                const int i = cell + 1, j = i + d;
                int type = 0;
                int best = INF, mdec = INF;
                if (live) {
                    type = pair_type(S[i], S[j]);
                    if (type) {
                        if (sub == 0) best = e_hairpin(X, i, j, type);
                        const int si1 = S[i + 1], sj1 = S[j - 1];
                        const int pmax = (j - 2 - TURN < i + MAXLOOP + 1) ? j - 2 - TURN : i + MAXLOOP + 1;
                        for (int p = i + 1 + sub; p <= pmax; p += GEN_G) {
                            int minq = j - i + p - MAXLOOP - 2;
                            if (minq < p + 1 + TURN) minq = p + 1 + TURN;
                            const int sp1 = S[p - 1], Sp = S[p];
                            for (int q = minq; q < j; q++) {
                                int t2 = pair_type(Sp, S[q]);
                                if (!t2) continue;
                                t2 = rtype_of(t2);
                                int e = e_intloop(P, p - i - 1, j - q - 1, type, t2, si1, sj1, sp1, S[q + 1]) + T.C(q - p, p);
                                best = e < best ? e : best;
                            }
                        }
                        // multiloop closed by (i,j): DML(i+1, j-1)
                        int dec = INF;
                        for (int k = i + 2 + TURN + sub; k <= j - 3 - TURN; k += GEN_G) {
                            int e = T.M(k - i - 1, i + 1) + T.M(j - k - 2, k + 1);
                            dec = e < dec ? e : dec;
                        }
                        dec += P->ML_closing + e_mlstem(P, rtype_of(type), sj1, si1);
                        best = dec < best ? dec : best;
                    }
                    for (int k = i + 1 + TURN + sub; k <= j - 2 - TURN; k += GEN_G) {
                        int e = T.M(k - i, i) + T.M(j - k - 1, k + 1);
                        mdec = e < mdec ? e : mdec;
                    }
                }
#pragma unroll
                for (int o = GEN_G / 2; o > 0; o >>= 1) {
                    int t = __shfl_xor(best, o); best = t < best ? t : best;
                    int u = __shfl_xor(mdec, o); mdec = u < mdec ? u : mdec;
                }
                if (live && sub == 0) {
                    if (best > INF) best = INF;
                    int mm = T.M(d - 1, i + 1);
                    int m2 = T.M(d - 1, i);
                    mm = m2 < mm ? m2 : mm;
                    if (type) { int e = best + ml_term(X, i, j, type); mm = e < mm ? e : mm; }
                    mm = mdec < mm ? mdec : mm;
                    if (mm > INF) mm = INF;
                    T.c[(size_t)d * T.ld + i] = type ? best : INF;
                    T.m[(size_t)d * T.ld + i] = mm;
                }
            }
            __syncthreads();
        }
        fold_epilogue<GTab, GEN_NT>(X, T, span, f3, starts, lens, btbuf, nc, btstk, sh_misc, win, max_lines, ss_stride,
                                   out_lines, out_ss, out_nlines, out_mfe, out_status);
    }
}

size_t fold_generic_lds_bytes(int n_cap, int max_lines) {
    const int nc = n_cap + 8;
    size_t b = 0;
    b += sizeof(int) * nc;                          // f3
    b += sizeof(int) * 2 * max_lines;               // starts, lens
    b += sizeof(int) * (GEN_NT / 64) * 3 * BT_STACK;
    b += sizeof(int) * 8;
    b += sizeof(short) * 3 * nc;
    b += 2 * nc;
    b += (GEN_NT / 64) * nc;
    return (b + 15) & ~(size_t)15;
}

size_t fold_generic_ws_slot_ints(int n_cap, int span) {
    size_t D = (size_t)(span < n_cap ? span : n_cap) + 1;
    size_t per = D * (size_t)(n_cap + 2);
    return 2 * ((per + 63) & ~(size_t)63);
}

void launch_fold_generic(hipStream_t stream, int grid, const FoldParams* P, const unsigned char* seqs, const long long* offs,
                         const int* lens, const int* work_list, int n_work, int span, int n_cap, int* ws, size_t ws_slot_ints, int max_lines,
                         int ss_stride, MirpFoldLine* out_lines, char* out_ss, int* out_nlines, int* out_mfe, int* out_status) {
    size_t lds = fold_generic_lds_bytes(n_cap, max_lines);
    hipLaunchKernelGGL(fold_generic_kernel, dim3(grid), dim3(GEN_NT), lds, stream, P, seqs, offs, lens, work_list, n_work, span, n_cap, ws,
                       ws_slot_ints, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status);
}

}  // namespace mirp
