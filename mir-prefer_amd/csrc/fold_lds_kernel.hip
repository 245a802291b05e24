// LDS-resident batched L-bounded Zuker local fold for precursor windows (n <= 352, span <= 300):
// the production case PRECURSOR_LEN = 300 (/root/reference/miR_PREFeR.py:90, RNALfold -L at :3053).
//
// One window per workgroup (1024 threads = 16 wavefronts), one workgroup per CU:
//   * fML lives entirely in LDS as a triangular int16 table, diagonal-major: (d,i) -> off(d)+i, so
//     the two operands of a multiloop split are read at consecutive addresses by consecutive lanes;
//   * c keeps its last 32 anti-diagonals in an LDS ring (interior loops reach back MAXLOOP+2) and is
//     archived once, coalesced, as int16 to a per-workgroup global slab for the exterior (f3) sweep
//     and the backtracks;
//   * per anti-diagonal: phase A = interior-loop candidates (work items = (paired cell, p); the q
//     partners come from per-base partner bitmasks, so unpairable (p,q) are never visited) and
//     multiloop splits (lane = cell, sub-ranges of the split point across wave groups), both
//     reduced with LDS atomic min; phase B = one thread per cell finalises c, fML, DML;
//   * INF is never read inside the split loop: fML is monotone (ML_BASE = 0), so each row/column
//     has a first-finite distance and the split range is clipped to it;
//   * windows whose energies leave the int16 range are flagged and re-run by the generic kernel.
// No MFMA: integer min-plus DP with irregular table lookups.
#include <hip/hip_runtime.h>
#include "fold_epilogue.h"

namespace mirp {

#define LNT 1024
#define LNW (LNT / 64)
#define LCAP 352            // window length capacity
#define LDMAX 299           // max pair distance (span 300)
#define I16_INF 0x7fff
#define FIN_LIMIT 32000

struct LdsTables {          // int16 copies of the hot parameter tables
    short stack[64];
    short bulge[32];
    short internal_loop[32];
    short mismatchI[200], mismatchH[200], mismatchM[200], mismatch1nI[200], mismatch23I[200];
    short hairpinE[LCAP];
    short ML_closing, ML_intern, TerminalAU, ninio, MAX_NINIO, pad[3];
};

struct LTab {               // table accessors for the shared epilogue/backtrack
    const short* fml;       // LDS
    const int* off;         // LDS: triangular offset of diagonal d (valid for d >= 4)
    const short* carch;     // global archive of c, (d,i) -> d*LCAP + i
    __device__ __forceinline__ int C(int d, int i) const { int v = carch[(size_t)d * LCAP + i]; return v == I16_INF ? INF : v; }
    __device__ __forceinline__ int M(int d, int i) const {
        if (d < 4) return INF;
        int v = fml[off[d] + i];
        return v == I16_INF ? INF : v;
    }
};

__device__ __forceinline__ int lds_mlstem(const LdsTables& T, const FoldParams* __restrict__ P, int type, int a, int b) {
    int e = T.ML_intern + (type > 2 ? T.TerminalAU : 0);
    if (a >= 0 && b >= 0) e += T.mismatchM[type * 25 + a * 5 + b];
    else if (a >= 0) e += P->dangle5[type][a];
    else if (b >= 0) e += P->dangle3[type][b];
    return e;
}

// interior-loop energy with LDS tables for the common classes; type2 already rtype'd
__device__ __forceinline__ int lds_intloop(const LdsTables& T, const FoldParams* __restrict__ P, int n1, int n2, int type, int type2,
                                           int si1, int sj1, int sp1, int sq1) {
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
        return T.internal_loop[nl + 1] + (x < T.MAX_NINIO ? x : T.MAX_NINIO) + T.mismatch1nI[type * 25 + si1 * 5 + sj1] +
               T.mismatch1nI[type2 * 25 + sq1 * 5 + sp1];
    }
    if (ns == 2) {
        if (nl == 2) return P->int22[type][type2][si1][sp1][sq1][sj1];
        if (nl == 3) return T.internal_loop[5] + T.ninio + T.mismatch23I[type * 25 + si1 * 5 + sj1] + T.mismatch23I[type2 * 25 + sq1 * 5 + sp1];
    }
    int x = (nl - ns) * T.ninio;
    return T.internal_loop[nl + ns] + (x < T.MAX_NINIO ? x : T.MAX_NINIO) + T.mismatchI[type * 25 + si1 * 5 + sj1] +
           T.mismatchI[type2 * 25 + sq1 * 5 + sp1];
}

struct LdsLayout {
    size_t fml, aux, f3, rowfin, colfin, S, seq, spec, pmask, list, off, tabs, misc, starts, lens, total;
};
__host__ __device__ inline LdsLayout lds_layout(int max_lines) {
    LdsLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
    // fML triangle for d = 4..LDMAX at n = LCAP
    size_t tri = 0;
    for (int d = 4; d <= LDMAX; d++) tri += (size_t)(LCAP - d);
    L.fml = take(tri * 2);
    // fill-phase scratch (c ring 32 diagonals, DML ring 3, accumulators), re-used by the epilogue for backtrack buffers/stacks
    size_t fill_aux = (size_t)32 * LCAP * 2 + (size_t)3 * LCAP * 4 + (size_t)2 * LCAP * 4;
    size_t bt_aux = (size_t)LNW * (LCAP + 8) + (size_t)LNW * 3 * BT_STACK * 4;
    L.aux = take(fill_aux > bt_aux ? fill_aux : bt_aux);
    L.f3 = take((LCAP + 8) * 4);
    L.rowfin = take((LCAP + 8) * 2);
    L.colfin = take((LCAP + 8) * 2);
    L.S = take(LCAP + 8);
    L.seq = take(LCAP + 8);
    L.spec = take((size_t)3 * (LCAP + 8) * 2);
    L.pmask = take(5 * 12 * 4);
    L.list = take((size_t)2 * LCAP * 2 + 16);
    L.off = take((LDMAX + 2) * 4);
    L.tabs = take(sizeof(LdsTables));
    L.misc = take(16 * 4);
    L.starts = take((size_t)max_lines * 4);
    L.lens = take((size_t)max_lines * 4);
    L.total = o;
    return L;
}

__global__ void __launch_bounds__(LNT) fold_lds_kernel(
    const FoldParams* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs, const int* __restrict__ win_lens,
    int n_work, int span, short* __restrict__ carch_all, unsigned int* __restrict__ work_counter, int* __restrict__ fallback_list,
    unsigned int* __restrict__ fallback_count, int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines, char* __restrict__ out_ss,
    int* __restrict__ out_nlines, int* __restrict__ out_mfe, int* __restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsLayout LY = lds_layout(max_lines);
    short* fml = (short*)(smem + LY.fml);
    short* cring = (short*)(smem + LY.aux);                         // [32][LCAP]
    int* dmlring = (int*)(cring + 32 * LCAP);                       // [3][LCAP]
    int* cpart = dmlring + 3 * LCAP;                                // [LCAP]
    int* mdec = cpart + LCAP;                                       // [LCAP]
    char* btbuf = (char*)(smem + LY.aux);                           // epilogue alias
    int* btstk = (int*)(smem + LY.aux + (((size_t)LNW * (LCAP + 8) + 15) & ~(size_t)15));
    int* f3 = (int*)(smem + LY.f3);
    short* rowfin = (short*)(smem + LY.rowfin);
    short* colfin = (short*)(smem + LY.colfin);
    unsigned char* S = smem + LY.S;
    unsigned char* seq = smem + LY.seq;
    short* spec = (short*)(smem + LY.spec);
    unsigned int* pmask = (unsigned int*)(smem + LY.pmask);         // [5][12]
    unsigned short* list = (unsigned short*)(smem + LY.list);       // [2][LCAP]
    int* off = (int*)(smem + LY.off);
    LdsTables& T = *(LdsTables*)(smem + LY.tabs);
    int* misc = (int*)(smem + LY.misc);                             // 0: next window, 1: overflow flag, 2,3: list counts, 4..: epilogue sh_misc
    int* starts = (int*)(smem + LY.starts);
    int* lens = (int*)(smem + LY.lens);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = LCAP + 8;
    short* carch = carch_all + (size_t)blockIdx.x * (size_t)(LDMAX + 1) * LCAP;

    // ---- one-time: hot parameter tables into LDS
    for (int x = tid; x < 64; x += LNT) T.stack[x] = (short)min(P->stack[x >> 3][x & 7], (int)I16_INF);
    for (int x = tid; x < 31; x += LNT) { T.bulge[x] = (short)min(P->bulge[x], (int)I16_INF); T.internal_loop[x] = (short)min(P->internal_loop[x], (int)I16_INF); }
    for (int x = tid; x < 200; x += LNT) {
        int t = x / 25, a = (x % 25) / 5, b = x % 5;
        T.mismatchI[x] = (short)min(P->mismatchI[t][a][b], (int)I16_INF); T.mismatchH[x] = (short)min(P->mismatchH[t][a][b], (int)I16_INF);
        T.mismatchM[x] = (short)P->mismatchM[t][a][b]; T.mismatch1nI[x] = (short)min(P->mismatch1nI[t][a][b], (int)I16_INF);
        T.mismatch23I[x] = (short)min(P->mismatch23I[t][a][b], (int)I16_INF);
    }
    for (int x = tid; x < LCAP; x += LNT) T.hairpinE[x] = (short)min(P->hairpinE[x], (int)I16_INF);
    if (tid == 0) { T.ML_closing = (short)P->ML_closing; T.ML_intern = (short)P->ML_intern; T.TerminalAU = (short)P->TerminalAU; T.ninio = (short)P->ninio; T.MAX_NINIO = (short)P->MAX_NINIO; }
    __syncthreads();

    for (;;) {
        if (tid == 0) misc[0] = (int)atomicAdd(work_counter, 1u);
        __syncthreads();
        const int win = misc[0];
        __syncthreads();
        if (win >= n_work) break;
        const long long o0 = offs[win];
        const int n = win_lens ? win_lens[win] : (int)(offs[win + 1] - o0);
        if (n < 1 || n > LCAP - 2) {   // wave-uniform: empty window, or too long for this kernel (-> generic kernel)
            if (tid == 0) {
                out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0;
                if (n >= 1) { unsigned int k = atomicAdd(fallback_count, 1u); fallback_list[k] = win; }
            }
        } else {
        const int D = (span - 1 < n - 1) ? span - 1 : n - 1;
        // ---- stage sequence, codes, special hairpins, partner masks, triangular offsets
        for (int x = tid; x <= n + 1; x += LNT) {
            unsigned char ch = 0;
            if (x >= 1 && x <= n) {
                ch = seqs[o0 + x - 1];
                if (ch >= 'a' && ch <= 'z') ch -= 32;
                if (ch == 'T') ch = 'U';
            }
            seq[x] = ch;
            S[x] = ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'U' ? 4 : 0;
        }
        for (int x = tid; x < 60; x += LNT) pmask[x] = 0u;
        for (int x = tid; x < LCAP + 8; x += LNT) { rowfin[x] = 20000; colfin[x] = 20000; }
        for (int x = tid; x < 3 * LCAP; x += LNT) dmlring[x] = INF;
        for (int x = tid; x < 2 * LCAP; x += LNT) cpart[x] = INF;   // cpart + mdec
        if (tid == 0) {
            int o = 0;
            for (int d = 4; d <= LDMAX + 1; d++) { off[d] = o; o += (n - d > 0 ? n - d : 0); }
            misc[1] = 0; misc[2] = 0; misc[3] = 0;
        }
        __syncthreads();
        if (tid == 0) { S[0] = S[n]; S[n + 1] = S[1]; }
        for (int x = tid; x <= n; x += LNT) {
            short s3 = -32768, s4 = -32768, s6 = -32768;
            if (x >= 1) {
                if (x + 4 <= n)
                    for (int k = 0; k < 2; k++) { bool m = true; for (int t = 0; t < 5; t++) m = m && (seq[x + t] == (unsigned char)P->tri[k][t]); if (m && s3 == -32768) s3 = (short)P->triE[k]; }
                if (x + 5 <= n)
                    for (int k = 0; k < 16; k++) { bool m = true; for (int t = 0; t < 6; t++) m = m && (seq[x + t] == (unsigned char)P->tetra[k][t]); if (m && s4 == -32768) s4 = (short)P->tetraE[k]; }
                if (x + 7 <= n)
                    for (int k = 0; k < 4; k++) { bool m = true; for (int t = 0; t < 8; t++) m = m && (seq[x + t] == (unsigned char)P->hexa[k][t]); if (m && s6 == -32768) s6 = (short)P->hexaE[k]; }
                // partner masks: bit q of pmask[b] set iff base b can pair with S[q]
                int b = S[x];
                if (b == 1) atomicOr(&pmask[4 * 12 + (x >> 5)], 1u << (x & 31));                                        // A pairs with U
                else if (b == 2) atomicOr(&pmask[3 * 12 + (x >> 5)], 1u << (x & 31));                                   // C pairs with G
                else if (b == 3) { atomicOr(&pmask[2 * 12 + (x >> 5)], 1u << (x & 31)); atomicOr(&pmask[4 * 12 + (x >> 5)], 1u << (x & 31)); }  // G with C,U
                else if (b == 4) { atomicOr(&pmask[1 * 12 + (x >> 5)], 1u << (x & 31)); atomicOr(&pmask[3 * 12 + (x >> 5)], 1u << (x & 31)); }  // U with A,G
            }
            spec[x] = s3; spec[nc + x] = s4; spec[2 * nc + x] = s6;
        }
        // paired-cell list of the first diagonal
        if (D >= 4)
            for (int x = tid; x < n - 4; x += LNT) {
                int i = x + 1;
                if (pair_type(S[i], S[i + 4])) { int k = atomicAdd(&misc[2], 1); list[k] = (unsigned short)i; }
            }
        __syncthreads();
        WinCtx X;
        X.P = P; X.S = S; X.seq = seq; X.f3 = f3; X.spec = spec; X.ldspec = nc; X.n = n; X.D = D;

        // ---- anti-diagonal wavefront
        for (int d = 4; d <= D; d++) {
            const int ncell = n - d;
            const int cur = d & 1;
            const unsigned short* clist = list + cur * LCAP;
            const int ncp = misc[2 + cur];
            // phase A1: interior loops (incl. stacks and bulges) of paired cells; items = (cell, p)
            for (int it = tid; it < ncp * 32; it += LNT) {
                const int i = clist[it >> 5], j = i + d;
                const int p = i + 1 + (it & 31);
                int pmax = j - 2 - TURN; if (pmax > i + MAXLOOP + 1) pmax = i + MAXLOOP + 1;
                if (p > pmax) continue;
                int qlo = p + d - MAXLOOP - 2; if (qlo < p + 1 + TURN) qlo = p + 1 + TURN;
                const int width = j - qlo;              // q in [qlo, j-1]
                if (width <= 0) continue;
                const int Sp = S[p];
                const unsigned int* pm = pmask + Sp * 12;
                unsigned int w0 = pm[qlo >> 5], w1 = pm[(qlo >> 5) + 1];
                int sh = qlo & 31;
                unsigned int bits = sh ? ((w0 >> sh) | (w1 << (32 - sh))) : w0;
                bits &= (width >= 32) ? 0xffffffffu : ((1u << width) - 1u);
                if (!bits) continue;
                const int type = pair_type(S[i], S[j]);
                const int si1 = S[i + 1], sj1 = S[j - 1], sp1 = S[p - 1], n1 = p - i - 1;
                const short* crow = cring + p;
                int best = INF;
                while (bits) {
                    int b = __ffs(bits) - 1;
                    bits &= bits - 1;
                    int q = qlo + b;
                    int t2 = rtype_of(pair_type(Sp, S[q]));
                    int e = lds_intloop(T, P, n1, j - q - 1, type, t2, si1, sj1, sp1, S[q + 1]) + (int)crow[((q - p) & 31) * LCAP];
                    best = e < best ? e : best;
                }
                if (best < INF) atomicMin(&cpart[i], best);
            }
            // phase A2: multiloop splits DML(i,j) over the finite range of row i / column j.
            // The split point t is wave-uniform (scalar address arithmetic); lanes = consecutive cells.
            {
                const int ncpad = (ncell + 63) & ~63;
                const int nsub = LNT / ncpad;            // >= 2 for ncell <= 512
                const int cell = tid % ncpad;
                const int sub = __builtin_amdgcn_readfirstlane(tid / ncpad);
                if (sub < nsub) {
                    const int i = cell + 1, j = i + d;
                    int tlo = 30000, rng = 0;
                    if (cell < ncell) {
                        int a = rowfin[i]; if (a < 4) a = 4;
                        int b = d - 1 - colfin[j]; if (b > d - 5) b = d - 5;
                        if (b >= a) { tlo = a; rng = b - a; }
                    }
                    int best = INF;
                    for (int t = 4 + sub; t <= d - 5; t += nsub) {
                        const int u = d - t - 1;
                        const int o1 = (t - 4) * n - ((t * (t - 1)) / 2 - 6);
                        const int o2 = (u - 4) * n - ((u * (u - 1)) / 2 - 6) + t + 1;
                        if ((unsigned)(t - tlo) <= (unsigned)rng) {
                            int e = (int)fml[o1 + i] + (int)fml[o2 + i];
                            best = e < best ? e : best;
                        }
                    }
                    if (best < INF) atomicMin(&mdec[i], best);
                }
            }
            __syncthreads();
            // phase B: finalise the cells of this diagonal; build the paired list of the next one
            if (tid == 0) misc[2 + (cur ^ 1)] = 0;
            __syncthreads();
            for (int x = tid; x < ncell; x += LNT) {
                const int i = x + 1, j = i + d;
                const int type = pair_type(S[i], S[j]);
                int cv = INF;
                const int md = mdec[i];
                if (type) {
                    cv = cpart[i];
                    int h = e_hairpin(X, i, j, type);
                    cv = h < cv ? h : cv;
                    int dml = dmlring[((d + 1) % 3) * LCAP + i + 1];     // (d-2) mod 3
                    if (dml < INF) {
                        int e = dml + T.ML_closing + lds_mlstem(T, P, rtype_of(type), S[j - 1], S[i + 1]);
                        cv = e < cv ? e : cv;
                    }
                }
                int m = INF;
                if (d > 4) {
                    int a = fml[off[d - 1] + i], b = fml[off[d - 1] + i + 1];
                    a = a == I16_INF ? INF : a; b = b == I16_INF ? INF : b;
                    m = a < b ? a : b;
                }
                if (type) { int e = cv + lds_mlstem(T, P, type, i > 1 ? (int)S[i - 1] : -1, j < n ? (int)S[j + 1] : -1); m = e < m ? e : m; }
                m = md < m ? md : m;
                if ((cv < INF && (cv > FIN_LIMIT || cv < -FIN_LIMIT)) || (m < INF && (m > FIN_LIMIT || m < -FIN_LIMIT)) ||
                    (md < INF && (md > FIN_LIMIT || md < -FIN_LIMIT))) misc[1] = 1;
                const short c16 = cv >= INF ? (short)I16_INF : (short)cv;
                const short m16 = m >= INF ? (short)I16_INF : (short)m;
                cring[(d & 31) * LCAP + i] = c16;
                carch[(size_t)d * LCAP + i] = c16;
                fml[off[d] + i] = m16;
                dmlring[(d % 3) * LCAP + i] = md;
                if (m < INF) { if (rowfin[i] > d) rowfin[i] = (short)d; if (colfin[j] > d) colfin[j] = (short)d; }
                cpart[i] = INF; mdec[i] = INF;
                if (d + 1 <= D && i + d + 1 <= n && pair_type(S[i], S[i + d + 1])) {
                    int k = atomicAdd(&misc[2 + (cur ^ 1)], 1);
                    list[(cur ^ 1) * LCAP + k] = (unsigned short)i;
                }
            }
            __syncthreads();
        }
        const int overflow = misc[1];
        __syncthreads();
        if (overflow) {   // int16 range exceeded: hand the window to the generic kernel
            if (tid == 0) { unsigned int k = atomicAdd(fallback_count, 1u); fallback_list[k] = win; out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; }
        } else {
            __threadfence_block();
            LTab TB;
            TB.fml = fml; TB.off = off; TB.carch = carch;
            fold_epilogue<LTab, LNT>(X, TB, span, f3, starts, lens, btbuf, LCAP + 8, btstk, misc + 4, win, max_lines, ss_stride, out_lines, out_ss,
                                     out_nlines, out_mfe, out_status);
        }
        }   // window fits this kernel
        __syncthreads();
    }
}

size_t fold_lds_bytes(int max_lines) { return lds_layout(max_lines).total; }
size_t fold_lds_carch_shorts_per_wg() { return (size_t)(LDMAX + 1) * LCAP; }
int fold_lds_max_n() { return LCAP - 2; }
int fold_lds_max_span() { return LDMAX + 1; }

hipError_t launch_fold_lds(hipStream_t stream, int grid, const FoldParams* P, const unsigned char* seqs, const long long* offs, const int* lens,
                           int n_work, int span, short* carch, unsigned int* work_counter, int* fallback_list, unsigned int* fallback_count,
                           int max_lines, int ss_stride, MirpFoldLine* out_lines, char* out_ss, int* out_nlines, int* out_mfe, int* out_status) {
    size_t lds = fold_lds_bytes(max_lines);
    hipError_t e = hipFuncSetAttribute((const void*)fold_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fold_lds_kernel, dim3(grid), dim3(LNT), lds, stream, P, seqs, offs, lens, n_work, span, carch, work_counter, fallback_list,
                       fallback_count, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status);
    return hipGetLastError();
}

}  // namespace mirp
