// LDS-resident batched L-bounded Zuker local fold for precursor windows (n <= 352, span <= 300):
// the production case PRECURSOR_LEN = 300 (/root/reference/miR_PREFeR.py:90, RNALfold -L at :3053).
//
// One window per workgroup (1024 threads = 16 wavefronts), one workgroup per CU:
//   * fML lives entirely in LDS as a triangular int16 table, diagonal-major: (d,i) -> off(d)+i, so
//     the two operands of a multiloop split are read at consecutive addresses by consecutive lanes;
//   * c keeps its last 32 anti-diagonals in an LDS ring (interior loops reach back MAXLOOP+2) and is
//     archived once, coalesced, as int16 to a per-workgroup global slab for the exterior (f3) sweep
//     and the backtracks;
//   * per anti-diagonal: phase A = interior-loop candidates (32 lanes per paired cell; the ring stores
//     G0 = c + inner mismatch, so a generic candidate is one LDS read; stack/bulge/1xn/2x3 in four
//     class-homogeneous rounds; the four small loops that need the big int11/21/22 tables are issued
//     first and consumed last) and multiloop splits (lane = cell, wave-uniform split point, scalar
//     offsets), reduced with shuffles + LDS atomic min; phase B = one thread per cell finalises
//     c, fML, DML and runs in the same barrier interval as phase A of the next diagonal;
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
#define FML_BIAS 32000       // fML is kept in LDS as uint16 (value + FML_BIAS), 65535 = INF; finite fML must stay in [-32000, 767] so that the sum of
#define FML_MAX 767          // two finite entries (<= 65534) can never be mistaken for a sum that involves INF (>= 65535)

struct LdsTables {          // int16 copies of the hot parameter tables
    short stack[64];
    short bulge[32];
    short internal_loop[32];
    short mismatchI[200], mismatchH[200], mismatchM[200], mismatch1nI[200], mismatch23I[200];
    unsigned short penK[25 * 34];       // generic interior loops: [u-6][n1] = il[u] + min(MAX_NINIO, |2 n1 - u| ninio) for 2 <= n1 <= u-2, else 65535
    unsigned short ocombo[120];         // the other classes: n1 | n2 << 5 | class << 10
    short n_gcombo, n_ocombo;
    unsigned char rt2[28];              // rtype(pair_type(a, b)) at [a*5+b]
    short ML_closing, ML_intern, TerminalAU, ninio, MAX_NINIO, pad[3];
};
#define CSTR 354            // c-ring row stride in shorts (177 dwords: odd, spreads LDS banks)

struct LTab {               // table accessors for the shared epilogue/backtrack
    const short* fml;       // triangular fML (per-window global slab in the epilogue kernel)
    const int* off;         // LDS: triangular offset of diagonal d (valid for d >= 4)
    const short* carch;     // global archive of c, triangular like fML: (d,i) -> off[d] + i (keeps a workgroup's slab ~88 KB, L2-friendly)
    __device__ __forceinline__ int C(int d, int i) const { int v = carch[off[d] + i]; return v == I16_INF ? INF : v; }
    __device__ __forceinline__ int M(int d, int i) const {
        if (d < 4) return INF;
        const int v = (unsigned short)fml[off[d] + i];
        return v == 65535 ? INF : v - FML_BIAS;
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
    size_t fml, aux, f3, S, seq, spec, list, off, tabs, misc, starts, lens, total;
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
    size_t fill_aux = (size_t)32 * CSTR * 2 + (size_t)3 * LCAP * 2 + (size_t)4 * LCAP * 4;   // c ring, DML ring (int16), 2 x {cpart, mdec}
    size_t bt_aux = (size_t)LNW * (LCAP + 8) + (size_t)LNW * 3 * BT_STACK * 4;
    L.aux = take(fill_aux > bt_aux ? fill_aux : bt_aux);
    L.f3 = take((LCAP + 8) * 4);
    L.S = take(LCAP + 8);
    L.seq = L.f3;   // staged characters are only needed while the special-hairpin table is built; f3 is epilogue-only
    L.spec = take((size_t)3 * (LCAP + 8) * 2);
    L.list = take((size_t)3 * LCAP * 2 + 16);
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
    int n_work, int win_base, int span, short* __restrict__ slabs, size_t slab_shorts, int* __restrict__ win_state,
    unsigned int* __restrict__ work_counter, int* __restrict__ fallback_list,
    unsigned int* __restrict__ fallback_count, int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines, char* __restrict__ out_ss,
    int* __restrict__ out_nlines, int* __restrict__ out_mfe, int* __restrict__ out_status, int dbg_flags, long long* __restrict__ dbg_cycles) {
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsLayout LY = lds_layout(max_lines);
    long long tA = 0, tB = 0, tS = 0, tE = 0, t0 = 0;   // diagnostic phase clocks (thread 0 only, dbg_cycles != nullptr)
    unsigned short* fml = (unsigned short*)(smem + LY.fml);   // biased uint16 (see FML_BIAS)
    unsigned short* cring = (unsigned short*)(smem + LY.aux);       // [32][CSTR] G0 + 32768 as uint16, 65535 = INF
    short* dmlring = (short*)(cring + 32 * CSTR);                             // [3][LCAP] int16
    int* acc = (int*)(dmlring + 3 * LCAP);                          // [2 (diagonal parity)][2 (cpart, mdec)][LCAP]
    char* btbuf = (char*)(smem + LY.aux);                           // epilogue alias
    int* btstk = (int*)(smem + LY.aux + (((size_t)LNW * (LCAP + 8) + 15) & ~(size_t)15));
    int* f3 = (int*)(smem + LY.f3);
    unsigned char* S = smem + LY.S;
    unsigned char* seq = smem + LY.seq;
    short* spec = (short*)(smem + LY.spec);
    unsigned short* list = (unsigned short*)(smem + LY.list);       // [3][LCAP], list of diagonal d in buffer d % 3
    int* off = (int*)(smem + LY.off);
    LdsTables& T = *(LdsTables*)(smem + LY.tabs);
    int* misc = (int*)(smem + LY.misc);                             // 0: next window, 1: overflow flag, 2..4: list counts, 8..: epilogue sh_misc
    int* starts = (int*)(smem + LY.starts);
    int* lens = (int*)(smem + LY.lens);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = LCAP + 8;

    // ---- one-time: hot parameter tables into LDS
    for (int x = tid; x < 64; x += LNT) T.stack[x] = (short)min(P->stack[x >> 3][x & 7], (int)I16_INF);
    for (int x = tid; x < 31; x += LNT) { T.bulge[x] = (short)min(P->bulge[x], (int)I16_INF); T.internal_loop[x] = (short)min(P->internal_loop[x], (int)I16_INF); }
    for (int x = tid; x < 200; x += LNT) {
        int t = x / 25, a = (x % 25) / 5, b = x % 5;
        T.mismatchI[x] = (short)min(P->mismatchI[t][a][b], (int)I16_INF); T.mismatchH[x] = (short)min(P->mismatchH[t][a][b], (int)I16_INF);
        T.mismatchM[x] = (short)P->mismatchM[t][a][b]; T.mismatch1nI[x] = (short)min(P->mismatch1nI[t][a][b], (int)I16_INF);
        T.mismatch23I[x] = (short)min(P->mismatch23I[t][a][b], (int)I16_INF);
    }
    if (tid == 0) {
        // combination tables of the interior-loop search window (n1 + n2 <= MAXLOOP)
        int g = 0, o = 0;
        for (int u = 6; u <= MAXLOOP; u++)
            for (int n1 = 0; n1 < 32; n1++) {
                int v = 65535;   // inadmissible slot: biased-unsigned ring value (>= 768) + 65535 can never beat the 65535 start value
                if (n1 >= 2 && n1 <= u - 2) { int n2 = u - n1, y = (n1 > n2 ? n1 - n2 : n2 - n1) * P->ninio; v = P->internal_loop[u] + (y < P->MAX_NINIO ? y : P->MAX_NINIO); }
                T.penK[(u - 6) * 34 + n1] = (unsigned short)v;
            }
        g = 375;
        T.ocombo[o++] = 0;                                                                    // class 0: stack
        for (int k = 1; k <= MAXLOOP; k++) T.ocombo[o++] = (unsigned short)(0 | (k << 5) | (1 << 10));   // class 1: bulge, n1 = 0
        for (int k = 1; k <= MAXLOOP; k++) T.ocombo[o++] = (unsigned short)(k | (0 << 5) | (1 << 10));   //          bulge, n2 = 0
        for (int k = 3; k <= MAXLOOP - 1; k++) T.ocombo[o++] = (unsigned short)(1 | (k << 5) | (2 << 10)); // class 2: 1 x n
        for (int k = 3; k <= MAXLOOP - 1; k++) T.ocombo[o++] = (unsigned short)(k | (1 << 5) | (2 << 10)); //          n x 1
        T.ocombo[o++] = (unsigned short)(2 | (3 << 5) | (3 << 10));                           // class 3: 2 x 3
        T.ocombo[o++] = (unsigned short)(3 | (2 << 5) | (3 << 10));                           //          3 x 2
        T.n_gcombo = (short)g; T.n_ocombo = (short)o;
        for (int x = 0; x < 25; x++) T.rt2[x] = (unsigned char)rtype_of(pair_type(x / 5, x % 5));
    }
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
        short* carch = slabs + (size_t)win * 2 * slab_shorts;      // per-window slab: c triangle, then fML triangle (read by fold_lds_epilogue_kernel)
        short* fml_out = carch + slab_shorts;
        if (dbg_cycles && tid == 0) t0 = clock64();
        if (n < 1 || n > LCAP - 2) {   // wave-uniform: empty window, or too long for this kernel (-> generic kernel)
            if (tid == 0) {
                out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; win_state[win] = 0;
                if (n >= 1) { unsigned int k = atomicAdd(fallback_count, 1u); fallback_list[k] = win_base + win; }
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
        for (int x = tid; x < 3 * LCAP; x += LNT) dmlring[x] = (short)I16_INF;
        for (int x = tid; x < 4 * LCAP; x += LNT) acc[x] = INF;
        if (tid == 0) {
            int o = 0;
            for (int d = 4; d <= LDMAX + 1; d++) { off[d] = o; o += (n - d > 0 ? n - d : 0); }
            misc[1] = 0; misc[2] = 0; misc[3] = 0; misc[4] = 0;
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
            }
            spec[x] = s3; spec[nc + x] = s4; spec[2 * nc + x] = s6;
        }
        // paired-cell lists of the first two diagonals (list of diagonal d lives in buffer d % 3)
        for (int dd = 4; dd <= 5 && dd <= D; dd++)
            for (int x = tid; x < n - dd; x += LNT) {
                int i = x + 1;
                if (pair_type(S[i], S[i + dd])) { int k = atomicAdd(&misc[2 + dd % 3], 1); list[(dd % 3) * LCAP + k] = (unsigned short)i; }
            }
        __syncthreads();
        WinCtx X;
        X.P = P; X.S = S; X.seq = seq; X.f3 = f3; X.spec = spec; X.ldspec = nc; X.n = n; X.D = D;

        if (dbg_cycles && tid == 0) { long long t = clock64(); tS += t - t0; t0 = t; }
        // ---- anti-diagonal wavefront, software-pipelined: phase B of diagonal d (one thread per cell) runs in the same barrier
        // interval as phase A of diagonal d+1, which only needs c of diagonals <= d-1 and fML of diagonals <= d-3.
        auto phaseA = [&](const int d) {
            const int ncell = n - d;
            const unsigned short* clist = list + (d % 3) * LCAP;
            const int ncp = misc[2 + d % 3];
            int* cpart = acc + (d & 1) * 2 * LCAP;
            int* mdec = cpart + LCAP;
            // The c ring holds G0(p,q) = c(p,q) + mismatchI[rtype(pq)][S[q+1]][S[p-1]]: the inner-pair part of a generic interior loop
            // is folded in when the cell is finalised, so a generic candidate costs one LDS read.  Plain c = G0 - mismatchI[code].
            // phase A0: the four small interior loops of each paired cell (1x1, 1x2, 2x1, 2x2) read the big int11/int21/int22
            // tables from global memory; their loads are issued here and consumed after A1/A2 so the latency is covered.
            int sp_e[4] = {INF, INF, INF, INF};
            int sp_i = 0;
            if (tid < ncp && !(dbg_flags & 1)) {
                const int i = clist[tid], j = i + d;
                sp_i = i;
                const int type = pair_type(S[i], S[j]);
                const int si1 = S[i + 1], sj1 = S[j - 1];
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) {
                    const int a = 1 + (c4 >> 1), b = 1 + (c4 & 1);       // n1 = a, n2 = b
                    const int p = i + 1 + a, q = j - 1 - b;
                    if (q - p >= TURN + 1) {
                        int t2 = pair_type(S[p], S[q]);
                        if (t2) {
                            t2 = rtype_of(t2);
                            const int sp1 = S[p - 1], sq1 = S[q + 1];
                            int e;
                            if (a == 1 && b == 1) e = P->int11[type][t2][si1][sj1];
                            else if (a == 1 && b == 2) e = P->int21[type][t2][si1][sq1][sj1];
                            else if (a == 2 && b == 1) e = P->int21[t2][type][sq1][si1][sp1];
                            else e = P->int22[type][t2][si1][sp1][sq1][sj1];
                            sp_e[c4] = e + (int)cring[((q - p) & 31) * CSTR + p] - 32768 - (int)T.mismatchI[t2 * 25 + sq1 * 5 + sp1];
                        }
                    }
                }
            }
            // phase A1: 32 lanes per paired cell
            if (!(dbg_flags & 1)) {
                const int sub = tid & 31;
                // the lane's loop size u = 6 + sub is fixed across cells: its 27 penalty taps are loaded once per diagonal (packed pairs)
                unsigned pk2[14];
                {
                    const unsigned* pkp = reinterpret_cast<const unsigned*>(T.penK + (sub < 25 ? sub : 24) * 34 + 2);
#pragma unroll
                    for (int k = 0; k < 14; k++) pk2[k] = pkp[k];
                }
                // lane constants of the four non-generic rounds (the lane's (n1, n2) per round do not depend on the cell)
                const int umax = d - 2 - (TURN + 1);             // n1 + n2 <= umax keeps q - p >= TURN + 1
                int on1[4], on2[4], okc[4], kc[4];
                {
                    const int nin = T.ninio, mxn = T.MAX_NINIO;
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        int n1, n2;
                        if (r4 == 0) { n1 = 0; n2 = sub; }                                   // stack + 3'-side bulges
                        else if (r4 == 1) { n1 = sub + 1; n2 = 0; }                          // 5'-side bulges
                        else if (r4 == 2) { n1 = 1; n2 = sub + 3; }                          // 1 x n
                        else { n1 = sub < 27 ? sub + 3 : sub - 25; n2 = sub < 27 ? 1 : 30 - sub; }   // n x 1, then 2x3 (lane 27), 3x2 (lane 28)
                        const int u = n1 + n2;
                        bool ok = u <= umax && u <= MAXLOOP;
                        if (r4 == 1) ok = ok && sub < 30;
                        if (r4 == 2) ok = ok && sub < 27;
                        if (r4 == 3) ok = ok && sub < 29;
                        on1[r4] = n1; on2[r4] = n2; okc[r4] = ok ? 1 : 0;
                        const int uu = ok ? u : 0;
                        int k;
                        if (r4 <= 1) k = uu == 0 ? 0 : (int)T.bulge[uu];
                        else if (r4 == 2 || sub < 27) { int y = (uu - 2) * nin; y = y < mxn ? y : mxn; k = (int)T.internal_loop[uu] + y; }
                        else k = (int)T.internal_loop[5] + nin;
                        kc[r4] = k;
                    }
                }
                for (int cidx = tid >> 5; cidx < ncp; cidx += LNT / 32) {
                    const int i = clist[cidx], j = i + d;
                    const int type = pair_type(S[i], S[j]);
                    const int o_out = type * 25 + S[i + 1] * 5 + S[j - 1];
                    int best = INF;
                    // generic loops (n1, n2 >= 2, u = n1 + n2 >= 6): il[u] + min(MAX_NINIO, |n1-n2| ninio) + mismatchI(outer) + G0.
                    // The candidates of one size u are a contiguous run of ring row d-2-u; each lane takes the sizes u = 6 + sub, 38 + ... and
                    // walks the run with immediate offsets (straight-line: one LDS read pair + add + min per slot, no index arithmetic).
                    {
                        const int mo = T.mismatchI[o_out];
                        unsigned bg = 65535u;
                        const int um = umax < MAXLOOP ? umax : MAXLOOP;
                        const int u = 6 + sub;
                        if (u <= um && !(dbg_flags & 4)) {
                            const unsigned short* row = cring + ((d - 2 - u) & 31) * CSTR + i + 1;
#pragma unroll
                            for (int n1 = 2; n1 <= 28; n1++) {
                                const unsigned pen = (n1 & 1) ? (pk2[(n1 - 2) >> 1] >> 16) : (pk2[(n1 - 2) >> 1] & 0xffffu);
                                const unsigned e = (unsigned)row[n1] + pen;
                                bg = e < bg ? e : bg;
                            }
                        }
                        if (bg < 65535u) { const int r = (int)bg - 32768 + mo; best = r < best ? r : best; }
                    }
                    // stack, bulges, 1xn, 2x3: four class-homogeneous rounds over the 32 lanes of the cell
                    {
                        const int au1 = type > 2 ? T.TerminalAU : 0;
                        const int m1 = T.mismatch1nI[o_out], m2 = T.mismatch23I[o_out];
                        const int tau = T.TerminalAU;
                        const short* strow = T.stack + type * 8;
#pragma unroll
                        for (int r4 = 0; r4 < ((dbg_flags & 8) ? 0 : 4); r4++) {
                            const int n1 = on1[r4], n2 = on2[r4], u = n1 + n2;
                            bool ok = okc[r4] != 0;
                            const int p = i + 1 + n1, q = j - 1 - n2;
                            // reads are unconditional: for an inadmissible slot p, q stay inside [i, i+31] x [j-31, j], i.e. inside the LDS
                            // arrays (S has slack, the ring is followed by the DML ring), and the result is discarded
                            const int g0u = (int)cring[((q - p) & 31) * CSTR + p];
                            ok = ok && g0u != 65535;
                            const int g0 = g0u - 32768;
                            const int t2 = T.rt2[S[p] * 5 + S[q]];
                            const int code = t2 * 25 + S[q + 1] * 5 + S[p - 1];
                            const int cpq = g0 - (int)T.mismatchI[code];
                            int e;
                            if (r4 <= 1) {
                                const int st = strow[t2];
                                e = u == 0 ? st : kc[r4] + (u == 1 ? st : au1 + (t2 > 2 ? tau : 0));
                            } else if (r4 == 2 || sub < 27) {
                                e = kc[r4] + m1 + (int)T.mismatch1nI[code];
                            } else {
                                e = kc[r4] + m2 + (int)T.mismatch23I[code];
                            }
                            e += cpq;
                            if (ok && e < best) best = e;
                        }
                    }
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) { int t = __shfl_xor(best, o); best = t < best ? t : best; }
                    if (sub == 0 && best < INF) atomicMin(&cpart[i], best);
                }
            }
            // phase A2: multiloop splits DML(i,j) over the finite range of row i / column j.
            // The split point t is wave-uniform (scalar address arithmetic); lanes = consecutive cells.
            {
                const int ncpad = (ncell + 63) & ~63;
                const int nsub = LNT / ncpad;            // >= 2 for ncell <= 512
                const int cell = tid % ncpad;
                const int sub = __builtin_amdgcn_readfirstlane(tid / ncpad);
                if (sub < nsub && !(dbg_flags & 2)) {
                    const int i = cell + 1, j = i + d;
                    // every split t in [4, d-5] is relaxed unconditionally: with the biased uint16 encoding a sum that involves an INF entry
                    // is >= 65535 and any sum of two finite entries is <= 65534, so no per-lane range bookkeeping is needed.
                    // The byte offsets of the two operand diagonals live in SGPRs and advance by second-order recurrences:
                    //   o1(t) = off(t),  o2(t) = off(d-t-1) + t + 1,  off(x) = (x-4) n - (x(x-1)/2 - 6)
                    const int s1 = nsub;
                    int t = 4 + sub;
                    const int uu = d - t - 1;
                    int so1 = __builtin_amdgcn_readfirstlane(2 * ((t - 4) * n - ((t * (t - 1)) / 2 - 6)));
                    int so2 = __builtin_amdgcn_readfirstlane(2 * ((uu - 4) * n - ((uu * (uu - 1)) / 2 - 6) + t + 1));
                    int si1 = __builtin_amdgcn_readfirstlane(2 * (s1 * n - s1 * t - (s1 * (s1 - 1)) / 2));           // off(t+s) - off(t)
                    int si2 = __builtin_amdgcn_readfirstlane(2 * (-(s1 * n) + s1 * uu - (s1 * (s1 + 1)) / 2 + s1));  // off(uu-s) - off(uu) + s
                    const int sss = __builtin_amdgcn_readfirstlane(2 * s1 * s1);
                    const char* fb = reinterpret_cast<const char*>(fml + i);
                    unsigned bu = 65535u;
#define MIRP_SSTEP() asm volatile("s_add_i32 %0, %0, %2\n\ts_sub_i32 %2, %2, %4\n\ts_add_i32 %1, %1, %3\n\ts_sub_i32 %3, %3, %4" : "+s"(so1), "+s"(so2), "+s"(si1), "+s"(si2) : "s"(sss))
#define MIRP_LD(o) ((unsigned)*reinterpret_cast<const unsigned short*>(fb + (o)))
                    for (; t + 3 * s1 <= d - 5; t += 4 * s1) {
                        const unsigned a0 = MIRP_LD(so1), b0 = MIRP_LD(so2);
                        MIRP_SSTEP();
                        const unsigned a1 = MIRP_LD(so1), b1 = MIRP_LD(so2);
                        MIRP_SSTEP();
                        const unsigned a2 = MIRP_LD(so1), b2 = MIRP_LD(so2);
                        MIRP_SSTEP();
                        const unsigned a3 = MIRP_LD(so1), b3 = MIRP_LD(so2);
                        MIRP_SSTEP();
                        unsigned e0 = a0 + b0, e1 = a1 + b1, e2 = a2 + b2, e3 = a3 + b3;
                        e0 = e0 < e1 ? e0 : e1; e2 = e2 < e3 ? e2 : e3; e0 = e0 < e2 ? e0 : e2;
                        bu = e0 < bu ? e0 : bu;
                    }
                    so1 = __builtin_amdgcn_readfirstlane(so1); so2 = __builtin_amdgcn_readfirstlane(so2);
                    si1 = __builtin_amdgcn_readfirstlane(si1); si2 = __builtin_amdgcn_readfirstlane(si2);
                    for (; t <= d - 5; t += s1) {
                        const unsigned e = MIRP_LD(so1) + MIRP_LD(so2);
                        MIRP_SSTEP();
                        bu = e < bu ? e : bu;
                    }
#undef MIRP_SSTEP
#undef MIRP_LD
                    const int best = (cell < ncell && bu < 65535u) ? (int)bu - 2 * FML_BIAS : INF;
                    if (best < INF) atomicMin(&mdec[i], best);
                }
            }
            {
                int e = sp_e[0] < sp_e[1] ? sp_e[0] : sp_e[1];
                int f = sp_e[2] < sp_e[3] ? sp_e[2] : sp_e[3];
                e = e < f ? e : f;
                if (e < INF) atomicMin(&cpart[sp_i], e);
            }
        };
        auto phaseB = [&](const int d) {
            const int ncell = n - d;
            int* cpart = acc + (d & 1) * 2 * LCAP;
            int* mdec = cpart + LCAP;
            const int hp_u = P->hairpinE[d - 1 < MIRP_HP_MAX ? d - 1 : MIRP_HP_MAX - 1];
            for (int x = tid; x < ncell; x += LNT) {
                const int i = x + 1, j = i + d;
                const int type = pair_type(S[i], S[j]);
                int cv = INF;
                const int md = mdec[i];
                if (type) {
                    cv = cpart[i];
                    int h;
                    {
                        const int u = d - 1;
                        int sv = -32768;
                        if (u == 4) sv = spec[nc + i]; else if (u == 6) sv = spec[2 * nc + i]; else if (u == 3) sv = spec[i];
                        if (sv != -32768) h = sv;
                        else if (u == 3) h = hp_u + (type > 2 ? T.TerminalAU : 0);
                        else h = hp_u + T.mismatchH[type * 25 + S[i + 1] * 5 + S[j - 1]];
                    }
                    cv = h < cv ? h : cv;
                    int dml = dmlring[((d + 1) % 3) * LCAP + i + 1];     // (d-2) mod 3
                    if (dml != I16_INF) {
                        int e = dml + T.ML_closing + lds_mlstem(T, P, rtype_of(type), S[j - 1], S[i + 1]);
                        cv = e < cv ? e : cv;
                    }
                }
                int m = INF;
                if (d > 4) {
                    int a = fml[off[d - 1] + i], b = fml[off[d - 1] + i + 1];
                    a = a == 65535 ? INF : a - FML_BIAS; b = b == 65535 ? INF : b - FML_BIAS;
                    m = a < b ? a : b;
                }
                if (type) { int e = cv + lds_mlstem(T, P, type, i > 1 ? (int)S[i - 1] : -1, j < n ? (int)S[j + 1] : -1); m = e < m ? e : m; }
                m = md < m ? md : m;
                if ((cv < INF && (cv > FIN_LIMIT || cv < -FIN_LIMIT)) || (m < INF && (m > FML_MAX || m < -FML_BIAS)) ||
                    (md < INF && (md > FIN_LIMIT || md < -FIN_LIMIT))) misc[1] = 1;
                const short c16 = cv >= INF ? (short)I16_INF : (short)cv;
                const unsigned short m16 = m >= INF ? (unsigned short)65535 : (unsigned short)(m + FML_BIAS);
                {   // G0 = c + mismatchI of (i,j) seen as the inner pair of a generic interior loop
                    unsigned short g16 = 65535;
                    if (cv < INF) g16 = (unsigned short)(cv + T.mismatchI[rtype_of(type) * 25 + S[j + 1] * 5 + S[i - 1]] + 32768);
                    cring[(d & 31) * CSTR + i] = g16;
                }
                carch[off[d] + i] = c16;
                fml[off[d] + i] = m16;
                dmlring[(d % 3) * LCAP + i] = md >= INF ? (short)I16_INF : (short)md;
                cpart[i] = INF; mdec[i] = INF;
                if (d + 2 <= D && i + d + 2 <= n && pair_type(S[i], S[i + d + 2])) {   // paired list of diagonal d+2
                    int k = atomicAdd(&misc[2 + (d + 2) % 3], 1);
                    list[((d + 2) % 3) * LCAP + k] = (unsigned short)i;
                }
            }
        };
        if (D >= 4) phaseA(4);
        __syncthreads();
        if (dbg_cycles && tid == 0) { long long t = clock64(); tA += t - t0; t0 = t; }
        for (int d = 4; d <= D; d++) {
            if (tid == 0) misc[2 + d % 3] = 0;   // list(d) was consumed in the previous interval; the buffer is refilled as list(d+3) in the next one
            phaseB(d);
            if (d + 1 <= D) phaseA(d + 1);
            __syncthreads();
            if (dbg_cycles && tid == 0) { long long t = clock64(); tB += t - t0; t0 = t; }
        }
        const int overflow = misc[1];
        __syncthreads();
        if (dbg_cycles && tid == 0) { long long t = clock64(); tE += t - t0; t0 = t; }
        if (overflow) {   // int16 range exceeded: hand the window to the generic kernel
            if (tid == 0) { unsigned int k = atomicAdd(fallback_count, 1u); fallback_list[k] = win_base + win; out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; win_state[win] = 0; }
        } else {
            // hand the tables to the epilogue kernel: c was archived on the fly, fML is copied out now (coalesced dwords)
            int tri = 0;
            if (D >= 4) tri = off[D] + (n - D) + 1;
            const unsigned int* src = reinterpret_cast<const unsigned int*>(fml);
            unsigned int* dst = reinterpret_cast<unsigned int*>(fml_out);
            for (int x = tid; x < (tri + 1) / 2; x += LNT) dst[x] = src[x];
            if (tid == 0) win_state[win] = 1;
        }
        }   // window fits this kernel
        __syncthreads();
    }
    if (dbg_cycles && tid == 0) {
        atomicAdd((unsigned long long*)&dbg_cycles[0], (unsigned long long)tS); atomicAdd((unsigned long long*)&dbg_cycles[1], (unsigned long long)tA);
        atomicAdd((unsigned long long*)&dbg_cycles[2], (unsigned long long)tB); atomicAdd((unsigned long long*)&dbg_cycles[3], (unsigned long long)tE);
    }
}

// ------------------------------------------------------------------------------------------
// Epilogue kernel: exterior sweep, enumeration, backtracks, output.  Latency-bound pointer chasing with global
// (L2) table reads, so it runs as many small workgroups (256 threads, ~16 KB LDS) per CU instead of sharing the
// fill kernel's one-workgroup-per-CU geometry.
// ------------------------------------------------------------------------------------------
#define ENT 256
__global__ void __launch_bounds__(ENT, 6) fold_lds_epilogue_kernel(
    const FoldParams* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs, const int* __restrict__ win_lens,
    int n_work, int span, const short* __restrict__ slabs, size_t slab_shorts, const int* __restrict__ win_state, unsigned int* __restrict__ work_counter,
    int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines, char* __restrict__ out_ss, int* __restrict__ out_nlines,
    int* __restrict__ out_mfe, int* __restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nc = LCAP + 8;
    int* f3 = (int*)smem;                                            // nc
    int* starts = f3 + nc;                                           // max_lines
    int* lens = starts + max_lines;                                  // max_lines
    int* btstk = lens + max_lines;                                   // (ENT/64)*3*BT_STACK
    int* misc = btstk + (ENT / 64) * 3 * BT_STACK;                   // 16
    int* off = misc + 16;                                            // LDMAX + 2
    short* spec = (short*)(off + LDMAX + 2);                         // 3*nc
    unsigned char* S = (unsigned char*)(spec + 3 * nc);              // nc
    unsigned char* seq = S + nc;                                     // nc
    char* btbuf = (char*)(seq + nc);                                 // (ENT/64)*nc
    EpiTables* EP = (EpiTables*)(smem + ((((size_t)(btbuf - (char*)smem) + (ENT / 64) * nc) + 15) & ~(size_t)15));
    const int tid = threadIdx.x;
    fill_epi_tables(EP, P, tid, ENT);
    __syncthreads();
    for (;;) {
        if (tid == 0) misc[0] = (int)atomicAdd(work_counter, 1u);
        __syncthreads();
        const int win = misc[0];
        __syncthreads();
        if (win >= n_work) break;
        if (win_state[win] == 1) {
            const long long o0 = offs[win];
            const int n = win_lens ? win_lens[win] : (int)(offs[win + 1] - o0);
            const int D = (span - 1 < n - 1) ? span - 1 : n - 1;
            for (int x = tid; x <= n + 1; x += ENT) {
                unsigned char ch = 0;
                if (x >= 1 && x <= n) {
                    ch = seqs[o0 + x - 1];
                    if (ch >= 'a' && ch <= 'z') ch -= 32;
                    if (ch == 'T') ch = 'U';
                }
                seq[x] = ch;
                S[x] = ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'U' ? 4 : 0;
            }
            if (tid == 0) {
                int o = 0;
                for (int d = 4; d <= LDMAX + 1; d++) { off[d] = o; o += (n - d > 0 ? n - d : 0); }
            }
            __syncthreads();
            if (tid == 0) { S[0] = S[n]; S[n + 1] = S[1]; }
            special_hairpins(P, seq, n, spec, nc, tid, ENT);
            __syncthreads();
            WinCtx X;
            X.P = P; X.S = S; X.seq = seq; X.f3 = f3; X.spec = spec; X.ldspec = nc; X.n = n; X.D = D; X.E = EP;
            LTab TB;
            TB.carch = slabs + (size_t)win * 2 * slab_shorts; TB.fml = TB.carch + slab_shorts; TB.off = off;
            fold_epilogue<LTab, ENT>(X, TB, span, f3, starts, lens, btbuf, nc, btstk, misc + 8, win, max_lines, ss_stride, out_lines, out_ss, out_nlines,
                                    out_mfe, out_status);
        }
        __syncthreads();
    }
}

size_t fold_lds_epilogue_bytes(int max_lines) {
    const int nc = LCAP + 8;
    size_t b = sizeof(int) * (nc + 2 * (size_t)max_lines + (ENT / 64) * 3 * BT_STACK + 16 + LDMAX + 2) + sizeof(short) * 3 * nc + 2 * (size_t)nc + (ENT / 64) * (size_t)nc;
    b = (b + 15) & ~(size_t)15;
    return b + sizeof(EpiTables) + 16;
}

size_t fold_lds_bytes(int max_lines) { return lds_layout(max_lines).total; }
int fold_lds_max_n() { return LCAP - 2; }
int fold_lds_max_span() { return LDMAX + 1; }

hipError_t launch_fold_lds(hipStream_t stream, int grid, int grid_epi, const FoldParams* P, const unsigned char* seqs, const long long* offs, const int* lens,
                           int n_work, int win_base, int span, short* slabs, size_t slab_shorts, int* win_state, unsigned int* work_counter, int* fallback_list,
                           unsigned int* fallback_count, int max_lines, int ss_stride, MirpFoldLine* out_lines, char* out_ss, int* out_nlines, int* out_mfe,
                           int* out_status, int dbg_flags, long long* dbg_cycles) {
    size_t lds = fold_lds_bytes(max_lines);
    hipError_t e = hipFuncSetAttribute((const void*)fold_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fold_lds_kernel, dim3(grid), dim3(LNT), lds, stream, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter,
                       fallback_list, fallback_count, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status, dbg_flags, dbg_cycles);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (!(dbg_flags & 16))
        hipLaunchKernelGGL(fold_lds_epilogue_kernel, dim3(grid_epi), dim3(ENT), fold_lds_epilogue_bytes(max_lines), stream, P, seqs, offs, lens, n_work, span,
                           slabs, slab_shorts, win_state, work_counter + 1, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status);
    return hipGetLastError();
}

size_t fold_lds_slab_shorts(int n_cap) {   // triangle of d = 4..LDMAX for windows up to n_cap (+ slack for the dword copy)
    size_t tri = 0;
    for (int d = 4; d <= LDMAX; d++) tri += (size_t)(n_cap - d > 0 ? n_cap - d : 0);
    return (tri + 8 + 7) & ~(size_t)7;
}

}  // namespace mirp
