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
#include "fold_epilogue.h"

namespace mirp {

#define TURN MIRP_TURN
#define MAXLOOP MIRP_MAXLOOP
#define INF MIRP_INF

struct GTab {
    static constexpr bool kTiled = false;
    int* __restrict__ c;
    int* __restrict__ m;
    int ld;
    __device__ __forceinline__ int C(int d, int i) const { return c[(size_t)d * ld + i]; }
    __device__ __forceinline__ int M(int d, int i) const { return m[(size_t)d * ld + i]; }
    __device__ __forceinline__ int TB(int, int) const { return -1; }   // no trace-back codes: the backtrack searches
};

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
                    for (int k = 0; k < P->n_tri; k++) {
                        bool m = true;
                        for (int t = 0; t < 5; t++) m = m && (seq[x + t] == (unsigned char)P->tri[k][t]);
                        if (m && s3 == -32768) s3 = (short)P->triE[k];
                    }
                if (x + 5 <= n)
                    for (int k = 0; k < P->n_tetra; k++) {
                        bool m = true;
                        for (int t = 0; t < 6; t++) m = m && (seq[x + t] == (unsigned char)P->tetra[k][t]);
                        if (m && s4 == -32768) s4 = (short)P->tetraE[k];
                    }
                if (x + 7 <= n)
                    for (int k = 0; k < P->n_hexa; k++) {
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
