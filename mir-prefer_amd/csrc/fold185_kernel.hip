// Local fold in the "vienna-1.8.5" flavour: the RNALfold the reference bundles for Linux (dependency/Linux/x64/RNALfold =
// ViennaRNA 1.8.5: Turner-1999 parameters, dangles = 1, full multi-component backtracks; call site
// /root/reference/miR_PREFeR.py:3053-3064).  Compatibility mode next to the default vienna-2.1.2 kernels: one workgroup per
// window, tables (c, fML, DML) as int32 in a global workspace, anti-diagonal wavefront fill, sequential exterior sweep with a
// parallel partner reduction, wave-cooperative first-match backtracks.  Behaviour: SURVEY.md Appendix B (d1 column); checked
// against oracle/lfold185.c, which is pinned to the bundled binary.  Arithmetic is deliberately uncapped (INF = 1,000,000 plus
// small terms, exactly as the original): equality tests in the backtrack see the same numbers.
#include <hip/hip_runtime.h>
#include "fold185_device.h"

namespace mirp {
namespace v185 {

__host__ __device__ inline size_t fold185_table_ints(int n_cap, int span) {
    const size_t per = (size_t)(span + 2) * (size_t)(n_cap + 2);
    return (per + 3) & ~(size_t)3;
}

__host__ __device__ inline size_t fold185_lds_bytes_base(int n_cap, int max_lines) {
    const size_t nc = (size_t)n_cap + 8;
    size_t b = sizeof(int) * (nc + 8 + 2 * (size_t)max_lines + (V_NT / 64) * 3 * V_BT_STACK + V_NT / 64 + 8);
    b += sizeof(short) * nc + 2 * nc + (V_NT / 64) * (nc + 8);
    return (b + 15) & ~(size_t)15;
}

// PHASE 1 = fill, PHASE 2 = exterior sweep + backtracks, launched back to back over batches of `grid` windows (slot = blockIdx.x), as fold_generic_kernel:
// the fill needs half the registers of the epilogue.
template <int PHASE>
__global__ void __launch_bounds__(V_NT) fold185_kernel(
    const FoldParams185* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs, const int* __restrict__ win_lens,
    const int* __restrict__ work_list, int n_work, int span, int n_cap, int* __restrict__ ws, size_t ws_slot_ints, int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines,
    char* __restrict__ out_ss, int* __restrict__ out_nlines, int* __restrict__ out_mfe, int* __restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nc = n_cap + 8;
    int* f3 = (int*)smem;                                  // nc + 8
    int* starts = f3 + nc + 8;                             // max_lines
    int* lens = starts + max_lines;                        // max_lines
    int* btstk = lens + max_lines;                         // (NT/64) * 3 * V_BT_STACK
    int* red = btstk + (V_NT / 64) * 3 * V_BT_STACK;       // NT/64 + 8
    short* tetra = (short*)(red + V_NT / 64 + 8);          // nc
    unsigned char* S = (unsigned char*)(tetra + nc);       // nc
    unsigned char* seq = S + nc;                           // nc
    char* btbuf = (char*)(seq + nc);                       // (NT/64) * (nc + 8)
    int* pcnt = (int*)(smem + fold185_lds_bytes_base(n_cap, max_lines));       // nc: split candidates of every column so far
    int* cbest = pcnt + nc;                                // nc: interior-loop minimum of the diagonal's cells
    unsigned short* plist = (unsigned short*)(cbest + nc); // nc: the diagonal's paired cells
    unsigned char* ctype = (unsigned char*)(plist + nc);   // nc: pair type of the diagonal's cells
    // inner-pair terms of the loop energies ([t2][sq1][sp1] as shorts) and the stacking table out of LDS, as fold_generic_kernel
    short* l_mmI = (short*)(smem + fold185_lds_bytes_base(n_cap, max_lines) + (((size_t)(4 + 4 + 2 + 1) * (size_t)nc + 15) / 16) * 16);
    short* l_stack = l_mmI + 200;
    for (int x = threadIdx.x; x < 200; x += V_NT) l_mmI[x] = (short)min(P->mismatchI[x / 25][(x / 5) % 5][x % 5], 32767);
    for (int x = threadIdx.x; x < 64; x += V_NT) l_stack[x] = (short)min(P->stack[x >> 3][x & 7], 32767);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = span;
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int win = work_list ? work_list[w] : w;
        const long long o0 = offs[win];
        const int n = win_lens ? win_lens[win] : (int)(offs[win + 1] - o0);
        if (n < 1 || n > n_cap) {
            if (PHASE == 2 && tid == 0) { out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = n < 1 ? 0 : -40; }
            continue;
        }
        for (int x = tid; x <= n + 1; x += V_NT) {
            unsigned char ch = 0;
            if (x >= 1 && x <= n) {
                ch = seqs[o0 + x - 1];
                if (ch >= 'a' && ch <= 'z') ch -= 32;
                if (ch == 'T') ch = 'U';
            }
            seq[x] = ch;
            S[x] = ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'U' ? 4 : 0;
        }
        for (int x = tid; x < nc + 8; x += V_NT) f3[x] = 0;
        __syncthreads();
        if (tid == 0) { S[0] = S[n]; S[n + 1] = S[1]; }
        for (int x = tid; x <= n; x += V_NT) {
            short b = 0;
            if (x >= 1 && x + 5 <= n)
                for (int k = 0; k < P->n_tetra; k++) {
                    bool m = true;
                    for (int t = 0; t < 6; t++) m = m && (seq[x + t] == (unsigned char)P->tetra[k][t]);
                    if (m) { b = (short)P->tetraE[k]; break; }
                }
            tetra[x] = b;
        }
        GTab185 T;
        T.ld = n_cap + 2; T.n = n; T.M = M;
        const size_t tab = fold185_table_ints(n_cap, span);
        T.c = ws + (size_t)blockIdx.x * ws_slot_ints;
        T.m = T.c + tab;
        T.dm = T.m + tab;
        // split candidates of every column (round 5, as fold_generic_kernel): the cells (s, j) whose fML is realised strictly by one of the four stem terms
        // (tests/tools/splitcand_gate185.c: identity of the tables with ML_BASE = 0); DML(i,j) = min(DML(i,j-1), min over column j's candidates)
        int2* pool = reinterpret_cast<int2*>(T.dm + tab);          // [ld][pcap]
        const int pcap = span + 2;
        int* gtab = reinterpret_cast<int*>(pool + (size_t)T.ld * pcap);          // c(p,q) + the inner pair's mismatch term, INF where (p,q) is no pair (as fold_generic_kernel's g)
        if constexpr (PHASE == 1) for (int x = tid; x <= n + 1; x += V_NT) pcnt[x] = 0;
        __syncthreads();
        Ctx<FoldParams185> X;
        X.P = P; X.S = S; X.tetra = tetra; X.f3 = f3; X.n = n; X.M = M;

        // ---- anti-diagonal wavefront fill; cells at distance M hold c = INF but a finite fML
        const int Dmax = M < n - 1 ? M : n - 1;
        if constexpr (PHASE == 1) {
        // Three intervals per diagonal (round 5, as fold_generic_kernel): pair types + the list of paired cells; interior loops with LANE = PAIRED CELL and a
        // wave-uniform (n1, n2) shape (task = block of 64 paired cells x n1; no divergence inside loopE, neighbouring reads of c on one diagonal), merged per
        // cell by an LDS atomic minimum; then the cells in groups of V_G lanes: hairpin, multiloop closing, the dense split loop, fML.
        const int sub = tid % V_G;
        for (int d = V_TURN + 1; d <= Dmax; d++) {
            const int ncell = n - d;
            if (tid == 0) red[V_NT / 64 + 7] = 0;
            __syncthreads();
            for (int base = 0; base < ncell; base += V_NT) {
                const int cell = base + tid;
                int type = 0;
                if (cell < ncell) { type = ptype(X, cell + 1, cell + 1 + d); ctype[cell] = (unsigned char)type; cbest[cell] = V_INF; }
                const unsigned long long bal = __ballot(type != 0);
                int wbase = 0;
                if (lane == 0 && bal) wbase = atomicAdd(&red[V_NT / 64 + 7], (int)__popcll(bal));
                wbase = __shfl(wbase, 0);
                if (type) plist[wbase + (int)__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)cell;
            }
            __syncthreads();
            const int np = red[V_NT / 64 + 7];
            const int n1max = (d - 2 - (V_TURN + 1) < V_MAXLOOP) ? d - 2 - (V_TURN + 1) : V_MAXLOOP;
            if (n1max >= 0 && np > 0) {
                const int nblk = (np + 63) >> 6, ntask = nblk * (n1max + 1);
                for (int t = wave; t < ntask; t += V_NT / 64) {
                    const int tu = __builtin_amdgcn_readfirstlane(t);          // the task is the wave's: the loop shape (and loopE's branches) in scalar registers
                    const int blk = tu / (n1max + 1), n1 = tu - blk * (n1max + 1);
                    const int k = blk * 64 + lane;
                    if (k < np) {
                        const int cell = plist[k];
                        const int i = cell + 1, j = i + d, p = i + 1 + n1;
                        const int type = ctype[cell];
                        const int si1 = S[i + 1], sj1 = S[j - 1], sp1 = S[p - 1];
                        const int o_mmI = l_mmI[type * 25 + si1 * 5 + sj1];          // the outer pair's term: once per task
                        int n2max = V_MAXLOOP - n1;
                        if (n2max > d - n1 - 2 - (V_TURN + 1)) n2max = d - n1 - 2 - (V_TURN + 1);
                        int best = V_INF;
                        // the row's first shapes (bulges, 1x1, 1x2, 2x2; every shape when n1 = 0) in the general form, its tail -- generic loops, all of them the
                        // same formula in this model -- as a tight loop over the precombined table: a load, two adds and a compare per candidate
                        const int n2t = n1 == 0 ? n2max + 1 : (n1 <= 2 ? 3 : 1);
                        const int* grow = gtab + (size_t)(d - n1 - 2) * T.ld + p;
#pragma unroll 8
                        for (int n2 = n2t; n2 <= n2max; n2++) {
                            const int x = (n1 > n2 ? n1 - n2 : n2 - n1) * P->ninio;
                            const int e = P->internal_loop[n1 + n2] + (x < P->MAX_NINIO ? x : P->MAX_NINIO) + o_mmI + grow[-(ptrdiff_t)n2 * T.ld];
                            best = e < best ? e : best;
                        }
                        const int n2s = n2t - 1 < n2max ? n2t - 1 : n2max;
                        for (int n2 = 0; n2 <= n2s; n2++) {
                            const int q = j - 1 - n2;
                            const int cv = T.C(p, q);
                            int t2 = ptype(X, p, q);
                            if (!t2) continue;
                            t2 = rtype_of(t2);
                            const int sq1 = S[q + 1];
                            const int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1;          // wave-uniform
                            int e;
                            if (ns >= 1 && !(ns == 1 && nl <= 2) && !(ns == 2 && nl == 2)) {
                                const int x = (nl - ns) * P->ninio;          // every loop but 1x1, 1x2, 2x2 in this model (no 1xn / 2x3 tables)
                                e = P->internal_loop[n1 + n2] + (x < P->MAX_NINIO ? x : P->MAX_NINIO) + o_mmI + l_mmI[t2 * 25 + sq1 * 5 + sp1];
                            } else if (nl == 0) {
                                e = l_stack[type * 8 + t2];
                            } else if (ns == 0) {
                                e = P->bulge[nl] + (nl == 1 ? (int)l_stack[type * 8 + t2] : (AU(X, type) + AU(X, t2)));
                            } else {
                                e = loopE(X, n1, n2, type, t2, si1, sj1, sp1, sq1);
                            }
                            e += cv;
                            best = e < best ? e : best;
                        }
                        if (best < V_INF) atomicMin(&cbest[cell], best);
                    }
                }
            }
            __syncthreads();
            for (int cell = tid / V_G; cell < ((ncell + V_NT / V_G - 1) / (V_NT / V_G)) * (V_NT / V_G); cell += V_NT / V_G) {
                const bool live = cell < ncell;
                const int i = cell + 1, j = i + d;
                int type = 0, best = V_INF, mdec = V_INF;
                if (live) {
                    type = ctype[cell];
                    if (type) {
                        if (sub == 0) { best = hairpin(X, i, j, type); const int il = cbest[cell]; best = il < best ? il : best; }
                        const int si1 = S[i + 1], sj1 = S[j - 1];
                        if (sub == 1 % V_G) {
                            const int tt = rtype_of(type);
                            const int e3 = P->dangle3[tt][si1], e5 = P->dangle5[tt][sj1];
                            int Xm = T.DM(i + 1, j - 1);
                            int v = T.DM(i + 2, j - 1) + e3; Xm = v < Xm ? v : Xm;
                            v = T.DM(i + 1, j - 2) + e5; Xm = v < Xm ? v : Xm;
                            v = T.DM(i + 2, j - 2) + e3 + e5; Xm = v < Xm ? v : Xm;
                            v = P->ML_closing + MLi(X, type) + Xm;
                            best = v < best ? v : best;
                        }
                    }
                    if (sub == 0) mdec = T.DM(i, j - 1);
                    const int pn = pcnt[j];
                    const int2* pj = pool + (size_t)j * pcap;
                    for (int k = sub; k < pn; k += V_G) {
                        const int2 en = pj[k];
                        if (en.x < i + V_TURN + 2) continue;          // fML(i, s-1) does not exist yet
                        const int e = T.Mm(i, en.x - 1) + en.y;
                        mdec = e < mdec ? e : mdec;
                    }
                }
#pragma unroll
                for (int o = V_G / 2; o > 0; o >>= 1) {
                    int t = __shfl_xor(best, o); best = t < best ? t : best;
                    int u = __shfl_xor(mdec, o); mdec = u < mdec ? u : mdec;
                }
                if (live && sub == 0) {
                    const int newc = type ? best : V_INF;
                    int mm = T.Mm(i + 1, j);
                    int v = T.Mm(i, j - 1); mm = v < mm ? v : mm;
                    mm = mdec < mm ? mdec : mm;
                    int stem = newc + MLi(X, type);
                    int t = ptype(X, i + 1, j);
                    v = T.C(i + 1, j) + P->dangle5[t][S[i]] + MLi(X, t); stem = v < stem ? v : stem;
                    t = ptype(X, i, j - 1);
                    v = T.C(i, j - 1) + P->dangle3[t][S[j]] + MLi(X, t); stem = v < stem ? v : stem;
                    t = ptype(X, i + 1, j - 1);
                    v = T.C(i + 1, j - 1) + P->dangle5[t][S[i]] + P->dangle3[t][S[j]] + MLi(X, t); stem = v < stem ? v : stem;
                    if (stem < mm) {          // realised strictly by a stem term: (i, j) is a split candidate of column j
                        mm = stem;
                        const int k = pcnt[j];
                        if (k < pcap) { pool[(size_t)j * pcap + k] = make_int2(i, stem); pcnt[j] = k + 1; }
                    }
                    T.c[(size_t)d * T.ld + i] = newc;
                    gtab[(size_t)d * T.ld + i] = type ? newc + (int)l_mmI[rtype_of(type) * 25 + S[j + 1] * 5 + S[i - 1]] : V_INF;
                    T.m[(size_t)d * T.ld + i] = mm;
                    T.dm[(size_t)d * T.ld + i] = mdec;
                }
            }
            __syncthreads();
        }

        } else {
        epilogue<FoldParams185, GTab185, V_NT>(X, T, f3, starts, lens, btstk, red, btbuf, nc, win, max_lines, ss_stride, out_lines, out_ss, out_nlines,
                                                out_mfe, out_status);
        }
    }
}

}  // namespace v185

size_t fold185_lds_bytes(int n_cap, int max_lines) {
    return v185::fold185_lds_bytes_base(n_cap, max_lines) + (((size_t)(4 + 4 + 2 + 1) * ((size_t)n_cap + 8) + 15) / 16) * 16 + sizeof(short) * (200 + 64) + 16;
}

size_t fold185_ws_slot_ints(int n_cap, int span) {
    // c, fML, DML, the split-candidate pool (two ints per entry, span + 2 entries per column)
    return 4 * v185::fold185_table_ints(n_cap, span) + ((2 * (size_t)(n_cap + 2) * (size_t)(span + 2) + 3) & ~(size_t)3) + 4;          // (+ g)
}

hipError_t launch_fold185(hipStream_t stream, int grid, const FoldParams185* P, const unsigned char* seqs, const long long* offs, const int* lens,
                          const int* work_list, int n_work,
                          int span, int n_cap, int* ws, size_t ws_slot_ints, int max_lines, int ss_stride, MirpFoldLine* out_lines, char* out_ss,
                          int* out_nlines, int* out_mfe, int* out_status) {
    const size_t lds = fold185_lds_bytes(n_cap, max_lines);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)v185::fold185_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)v185::fold185_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    for (int b = 0; b < n_work; b += grid) {          // batches of `grid` windows: window b + k owns workspace slot k in both kernels
        const int nb = n_work - b < grid ? n_work - b : grid;
        const int* wl = work_list ? work_list + b : nullptr;
        const long long* o2 = work_list ? offs : offs + b;
        const int* l2 = (work_list || !lens) ? lens : lens + b;
        MirpFoldLine* ol = work_list ? out_lines : out_lines + (size_t)b * max_lines;
        char* os = work_list ? out_ss : out_ss + (size_t)b * max_lines * ss_stride;
        int* on = work_list ? out_nlines : out_nlines + b;
        int* om = work_list ? out_mfe : out_mfe + b;
        int* ost = work_list ? out_status : out_status + b;
        hipLaunchKernelGGL(v185::fold185_kernel<1>, dim3(nb), dim3(V_NT), lds, stream, P, seqs, o2, l2, wl, nb, span, n_cap, ws, ws_slot_ints, max_lines, ss_stride, ol, os, on,
                           om, ost);
        hipLaunchKernelGGL(v185::fold185_kernel<2>, dim3(nb), dim3(V_NT), lds, stream, P, seqs, o2, l2, wl, nb, span, n_cap, ws, ws_slot_ints, max_lines, ss_stride, ol, os, on,
                           om, ost);
    }
    return hipGetLastError();
}

}  // namespace mirp
