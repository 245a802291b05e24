// Local fold in the "vienna-1.8.5" flavour: the RNALfold the reference bundles for Linux (dependency/Linux/x64/RNALfold =
// ViennaRNA 1.8.5: Turner-1999 parameters, dangles = 1, full multi-component backtracks; call site
// /root/reference/miR_PREFeR.py:3053-3064).  Compatibility mode next to the default vienna-2.1.2 kernels: one workgroup per
// window, tables (c, fML, DML) as int32 in a global workspace, anti-diagonal wavefront fill, sequential exterior sweep with a
// parallel partner reduction, wave-cooperative first-match backtracks.  Behaviour: SURVEY.md Appendix B (d1 column); checked
// against oracle/lfold185.c, which is pinned to the bundled binary.  Arithmetic is deliberately uncapped (INF = 1,000,000 plus
// small terms, exactly as the original): equality tests in the backtrack see the same numbers.
#include <hip/hip_runtime.h>
#include "fold185_device.h"
#include <type_traits>

namespace mirp {
namespace v185 {

#ifndef V_PU
#define V_PU 4                   // split candidates of a column a lane reads together (interval B)
#endif
#define V_PL_MAXN 2500           // longest window whose split candidates get the packed LDS copy (12 bits of position, 20 of energy)
#define V_PL_FLAG 0x40000000
#ifndef V_PLN
#define V_PLN 8                  // candidates of a column kept in LDS
#endif
#define V_STAGE 512               // ints per wave of the interior-loop interval's staging buffer
#define V_FILL_WAVES 5            // waves per SIMD the fill's register allocation aims for
#ifndef V_EPI_WAVES
#define V_EPI_WAVES 8             // (as fold_generic_kernel: the epilogue is latency-bound, eight workgroups per CU with spills beat three without: 0.251 -> 0.242 s at L = 400)
#endif
typedef int v_int2a __attribute__((ext_vector_type(2), aligned(4)));      // consecutive cells of a table row from any 4-byte boundary
typedef int v_int4a __attribute__((ext_vector_type(4), aligned(4)));
typedef int v_int4q __attribute__((ext_vector_type(4), aligned(16)));

// row stride of a workspace table (a multiple of four ints: the interior-loop interval stages row segments with aligned 16-byte loads)
__host__ __device__ inline int fold185_ld(int n_cap) { return (n_cap + 2 + 3) & ~3; }

__host__ __device__ inline size_t fold185_table_ints(int n_cap, int span) {
    const size_t per = (size_t)(span + 2) * (size_t)fold185_ld(n_cap);
    return (per + 3) & ~(size_t)3;
}

__host__ __device__ inline size_t fold185_lds_bytes_base(int n_cap, int max_lines) {
    const size_t nc = (size_t)n_cap + 8;
    size_t b = sizeof(int) * (nc + 8 + 2 * (size_t)max_lines + (V_NT / 64) * 3 * V_BT_STACK + V_NT / 64 + 8);
    b += sizeof(short) * nc + 2 * nc + (V_NT / 64) * (nc + 8);
    return (b + 15) & ~(size_t)15;
}

// the fill kernel's share of the base carve-up: tetraloop bonuses and the sequence codes
__host__ __device__ inline size_t fold185_lds_bytes_base_fill(int n_cap) { return ((size_t)4 * ((size_t)n_cap + 8) + 15) & ~(size_t)15; }

// PHASE 1 = fill, PHASE 2 = exterior sweep + backtracks, launched back to back over batches of `grid` windows (slot = blockIdx.x), as fold_generic_kernel:
// the fill needs half the registers of the epilogue.
template <int PHASE>
__global__ void __launch_bounds__(V_NT, PHASE == 1 ? V_FILL_WAVES : V_EPI_WAVES) fold185_kernel(
    const FoldParams185* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs, const int* __restrict__ win_lens,
    const int* __restrict__ work_list, int n_work, int span, int n_cap, int* __restrict__ ws, size_t ws_slot_ints, int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines,
    char* __restrict__ out_ss, int* __restrict__ out_nlines, int* __restrict__ out_mfe, int* __restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nc = n_cap + 8;
    // (the two kernels have their own layouts, as fold_generic_kernel: the fill needs the sequence codes and the tetraloop bonuses of the base carve-up, nothing else)
    int *f3 = nullptr, *starts = nullptr, *lens = nullptr, *btstk = nullptr, *red = nullptr;
    short* tetra;
    unsigned char *S, *seq;
    char* btbuf = nullptr;
    if constexpr (PHASE == 1) {
        tetra = (short*)smem;                              // nc
        S = (unsigned char*)(tetra + nc);                  // nc
        seq = S + nc;                                      // nc
    } else {
        f3 = (int*)smem;                                   // nc + 8
        starts = f3 + nc + 8;                              // max_lines
        lens = starts + max_lines;                         // max_lines
        btstk = lens + max_lines;                          // (NT/64) * 3 * V_BT_STACK
        red = btstk + (V_NT / 64) * 3 * V_BT_STACK;        // NT/64 + 8
        tetra = (short*)(red + V_NT / 64 + 8);             // nc
        S = (unsigned char*)(tetra + nc);                  // nc
        seq = S + nc;                                      // nc
        btbuf = (char*)(seq + nc);                         // (NT/64) * (nc + 8)
    }
    const size_t base_bytes = PHASE == 1 ? fold185_lds_bytes_base_fill(n_cap) : fold185_lds_bytes_base(n_cap, max_lines);
    int* pcnt = (int*)(smem + base_bytes);       // nc: split candidates of every column so far (| V_PL_FLAG: one of its first four does not fit the packed LDS copy)
    int* cbest = pcnt + nc;                                // nc: interior-loop minimum of the diagonal's cells
    unsigned short* plist = (unsigned short*)(cbest + nc); // nc: the diagonal's paired cells
    unsigned char* ctype = (unsigned char*)(plist + nc);   // nc: pair type of the diagonal's cells
    // inner-pair terms of the loop energies ([t2][sq1][sp1] as shorts) and the stacking table out of LDS, as fold_generic_kernel
    short* l_mmI = (short*)(smem + base_bytes + (((size_t)(4 + 4 + 2 + 1) * (size_t)nc + 15) / 16) * 16);
    short* l_xb = l_mmI + 200;                             // bulge: TerminalAU of the inner pair - its mismatchI (what turns the table's word into c + TerminalAU)
    short* l_stack = l_xb + 200;
    int* stage = reinterpret_cast<int*>(l_stack + 64);     // V_STAGE ints per wave: the row segment a block of paired cells reads for one loop size
    int* wcnt = stage + (V_NT / 64) * V_STAGE;             // 2 * waves: paired cells per wave and half-pass of the list compaction
    int* pl4 = wcnt + 2 * (V_NT / 64);                    // V_PLN nc: the first V_PLN split candidates of every column, packed s << 20 | (fML & 0xfffff) (interval B; windows up to V_PL_MAXN nt)
    if constexpr (PHASE == 1) {
    for (int x = threadIdx.x; x < 200; x += V_NT) {
        l_mmI[x] = (short)min(P->mismatchI[x / 25][(x / 5) % 5][x % 5], 32767);
        l_xb[x] = x >= 25 ? (short)((x / 25 > 2 ? P->TerminalAU : 0) - P->mismatchI[x / 25][(x / 5) % 5][x % 5]) : (short)0;
    }
    for (int x = threadIdx.x; x < 64; x += V_NT) l_stack[x] = (short)min(P->stack[x >> 3][x & 7], 32767);
    }
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
        if constexpr (PHASE == 2) for (int x = tid; x < nc + 8; x += V_NT) f3[x] = 0;
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
        T.ld = fold185_ld(n_cap); T.n = n; T.M = M;
        const size_t tab = fold185_table_ints(n_cap, span);
        T.c = ws + (size_t)blockIdx.x * ws_slot_ints;
        T.m = T.c + tab;
        T.dm = T.m + tab;
        // split candidates of every column (round 5, as fold_generic_kernel): the cells (s, j) whose fML is realised strictly by one of the four stem terms
        // (tests/tools/splitcand_gate185.c: identity of the tables with ML_BASE = 0); DML(i,j) = min(DML(i,j-1), min over column j's candidates)
        int2* pool = reinterpret_cast<int2*>(T.dm + tab);          // [ld][pcap]
        const int pcap = span + 2;
        // the interior-loop interval's view of finished cells, one word per cell (as fold_generic_kernel's w): c(p,q) + mismatchI of (p,q) seen as an inner pair in
        // the low 24 bits (V_INF where (p,q) is no pair), the index of that term, rtype * 25 + S[q+1] * 5 + S[p-1], in the high byte
        int* wtab = reinterpret_cast<int*>(pool + (size_t)T.ld * pcap);
        unsigned short* tbtab = reinterpret_cast<unsigned short*>(wtab + tab);          // trace-back codes, a short per cell
        T.tb = tbtab;
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
            // pair types of the diagonal's cells; the paired ones compacted into a list IN CELL ORDER (a block of 64 list entries is a stretch of the
            // diagonal, whose row segments interval A stages): two cells per thread and pass, the per-wave counts ordered through LDS
            int np = 0;
            for (int base = 0; base < ncell; base += 2 * V_NT) {
                int type[2];
                unsigned long long bal[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int cell = base + h * V_NT + tid;
                    type[h] = 0;
                    if (cell < ncell) { type[h] = ptype(X, cell + 1, cell + 1 + d); ctype[cell] = (unsigned char)type[h]; cbest[cell] = 0x7fffffff; }
                    bal[h] = __ballot(type[h] != 0);
                    if (lane == 0) wcnt[h * (V_NT / 64) + wave] = (int)__popcll(bal[h]);
                }
                __syncthreads();
                int before[2] = {0, 0}, total = 0;
#pragma unroll
                for (int x = 0; x < 2 * (V_NT / 64); x++) {
                    const int v = wcnt[x];
                    if (x < wave) before[0] += v;
                    if (x < V_NT / 64 + wave) before[1] += v;
                    total += v;
                }
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (type[h]) plist[np + before[h] + (int)__popcll(bal[h] & ((1ull << lane) - 1ull))] = (unsigned short)(base + h * V_NT + tid);
                np += total;
                if (base + 2 * V_NT < ncell) __syncthreads();          // (the counts are rewritten by the next pass)
            }
            __syncthreads();
            const int n1max = (d - 2 - (V_TURN + 1) < V_MAXLOOP) ? d - 2 - (V_TURN + 1) : V_MAXLOOP;
#ifdef MIRP_X_GEN_NOA               // timing experiment: no interior loops (tables wrong by construction)
            if (false) {
#else
            if (n1max >= 0 && np > 0) {
#endif
                // Task = (block of 64 paired cells, group of loop sizes s = n1 + n2), as fold_generic_kernel: a size's inner pairs are consecutive cells of
                // ONE diagonal's row (d - 2 - s); the block's segment of that row is loaded once by the wave (aligned 16 bytes per lane, three sizes in
                // flight), staged in LDS, and each lane reads its s + 1 words from there.  A generic candidate -- every shape of this model but stack, bulge,
                // 1 x 1, 1 x 2, 2 x 2 -- is one v_mad_i32_i24 (word's 24-bit energy field + the scalar size / asymmetry term) and a minimum.
                const int smax = n1max;
                const int nblk = (np + 63) >> 6;
                const int ngrp = smax < 7 ? 1 : smax < 18 ? 2 : smax < 25 ? 3 : 4;
                const int ntask = nblk * ngrp;
                for (int t = wave; t < ntask; t += V_NT / 64) {
                    const int tu = __builtin_amdgcn_readfirstlane(t);          // the task is the wave's: loop sizes and shapes in scalar registers
                    const int blk = tu / ngrp, grp = tu - blk * ngrp;
                    const int k = blk * 64 + lane;
                    // (the lanes behind the list's end, in its last block, run along on its last cell: the staged loads are the whole wave's)
                    const bool active = k < np;
                    const int cell = plist[active ? k : np - 1];
                    const int i = cell + 1, j = i + d;
                    const int type = ctype[cell];
                    const int si1 = S[i + 1], sj1 = S[j - 1];
                    const int o_mmI = l_mmI[type * 25 + si1 * 5 + sj1];          // the outer pair's term: once per task
                    const int tau = AU(X, type);
                    // the cell's minimum over the task as ONE 32-bit key, energy * 1024 + (n1 << 5 | n2): among equal energies the loop the backtrack's search
                    // (p ascending, q descending) finds first -- the cell's trace-back code (energies stay below 2^21: V_INF plus small terms)
                    int best = 0x7fffffff;
                    const int c1024 = 1024;
                    auto gen = [&](const int w, const int kterm) {
                        int e;
                        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(e) : "v"(w), "v"(c1024), "s"(kterm));
                        return e;
                    };
                    auto w_e = [](const int w) { return (w << 8) >> 8; };                      // the word's energy (sign-extended 24 bits)
                    auto w_in = [](const int w) { return (int)((unsigned)w >> 24); };          // the word's inner-pair index
                    auto put = [&](const int e, const int shape) { const int key = e * 1024 + shape; best = key < best ? key : best; };
                    auto put_bulge = [&](const int eb, const int w, const int shape) { put(eb + w_e(w) + (int)l_xb[w_in(w)], shape); };
                    const int* wlane = wtab + i + 1;          // candidate n1 of size s = wlane[(d - 2 - s) * ld + n1]
                    // sizes s_lo .. s_hi (>= 7) straight from the table, a load per candidate: the fallback for blocks whose cells lie too far apart for the
                    // staging buffer (sparse pairs in long windows)
                    auto sizes = [&](const int s_lo, const int s_hi) {
                        for (int s = s_lo; s <= s_hi; s++) {
                            const int* wrow = wlane + (d - 2 - s) * T.ld;
                            int kg = 0x7fffffff;
                            for (int n1 = 1; n1 <= s - 1; n1++) { const int e = gen(wrow[n1], P->gen_e[s][n1]); kg = e < kg ? e : kg; }
                            kg += o_mmI * 1024;
                            best = kg < best ? kg : best;
                            const int eb = P->bulge[s] + tau;
                            put_bulge(eb, wrow[0], s); put_bulge(eb, wrow[s], s << 5);
                        }
                    };
                    const int cell_lo = __builtin_amdgcn_readfirstlane(plist[blk * 64]);
                    const int cell_hi = __builtin_amdgcn_readfirstlane(plist[blk * 64 + 63 < np ? blk * 64 + 63 : np - 1]);
                    const int b0a = (cell_lo + 2) & ~3;                     // first staged position (p of the first cell, rounded down to 16 bytes)
                    const int span_w = cell_hi + 2 - b0a;                   // the last cell's offset in the segment
                    int* slot = stage + wave * V_STAGE;
                    const int* sl = slot + (cell + 2 - b0a);                // this lane's candidate n1 = sl[n1]
                    auto sizes_staged = [&](auto two_tag, const int s_lo, const int s_hi) {
                        constexpr bool TWO = decltype(two_tag)::value;          // a second 256-word piece
                        auto issue = [&](const int s, v_int4q& a, v_int4q& b) {
                            const int* src = wtab + ((d - 2 - s) * T.ld + b0a) + 4 * lane;
                            a = *reinterpret_cast<const v_int4q*>(src);
                            if (TWO) b = *reinterpret_cast<const v_int4q*>(src + 256);
                        };
                        // one size: its segment from the registers to the buffer, the registers re-used for the load three sizes on, then the candidates
                        auto step = [&](const int s, v_int4q& a, v_int4q& b) {
                            *reinterpret_cast<v_int4q*>(slot + 4 * lane) = a;
                            if (TWO) *reinterpret_cast<v_int4q*>(slot + 256 + 4 * lane) = b;
                            if (s + 3 <= s_hi) issue(s + 3, a, b);
                            const v_int4q* kq = reinterpret_cast<const v_int4q*>(P->gen_e1[s]);           // terms of n1 = 1 + 4 c .. 4 + 4 c: one scalar load
                            const v_int4q kt = *reinterpret_cast<const v_int4q*>(P->gen_et[s]);            // of n1 = s - 4 .. s - 1
                            int kg = 0x7fffffff;
                            for (int c4 = 0; 4 * c4 + 4 <= s - 1; c4++) {
                                const v_int4q k4 = kq[c4];
#pragma unroll
                                for (int u = 0; u < 4; u++) { const int e = gen(sl[1 + 4 * c4 + u], k4[u]); kg = e < kg ? e : kg; }
                            }
#pragma unroll
                            for (int u = 0; u < 4; u++) { const int e = gen(sl[s - 4 + u], kt[u]); kg = e < kg ? e : kg; }          // (may overlap the last chunk)
                            kg += o_mmI * 1024;
                            best = kg < best ? kg : best;
                            const int eb = P->bulge[s] + tau;
                            put_bulge(eb, sl[0], s); put_bulge(eb, sl[s], s << 5);
                        };
                        v_int4q a0 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};
                        issue(s_lo, a0, b0);
                        if (s_lo + 1 <= s_hi) issue(s_lo + 1, a1, b1);
                        if (s_lo + 2 <= s_hi) issue(s_lo + 2, a2, b2);
                        for (int s = s_lo; s <= s_hi; s += 3) {          // three sizes per turn: each has its own registers, nothing is moved
                            step(s, a0, b0);
                            if (s + 1 <= s_hi) step(s + 1, a1, b1);
                            if (s + 2 <= s_hi) step(s + 2, a2, b2);
                        }
                    };
                    const bool staged = span_w + 31 + 1 <= V_STAGE;
                    if (grp == 0) {
                        // sizes 0 .. 6, straight-line: every load first (rows of sizes beyond smax -- the first diagonals only -- are read at a clamped row and
                        // not used), then the shapes with (n1, n2) as compile-time constants
                        const int sI[4] = {S[i], si1, S[i + 2], S[i + 3]};          // S[i + x]
                        const int sJ[4] = {S[j], sj1, S[j - 2], S[j - 3]};          // S[j - x]
                        const int* wr[7];
#pragma unroll
                        for (int s = 0; s < 7; s++) { const int dd = d - 2 - s; wr[s] = wlane + (dd > 0 ? dd : 0) * T.ld; }
                        const int w0 = wr[0][0];
                        const v_int2a w1 = *reinterpret_cast<const v_int2a*>(wr[1]);
                        const v_int4a w2 = *reinterpret_cast<const v_int4a*>(wr[2]), w3 = *reinterpret_cast<const v_int4a*>(wr[3]), w4 = *reinterpret_cast<const v_int4a*>(wr[4]);
                        const int w44 = wr[4][4];
                        const v_int4a w5 = *reinterpret_cast<const v_int4a*>(wr[5]);
                        const v_int2a w5b = *reinterpret_cast<const v_int2a*>(wr[5] + 4);
                        const v_int4a w6 = *reinterpret_cast<const v_int4a*>(wr[6]), w6b = *reinterpret_cast<const v_int4a*>(wr[6] + 3);
                        // reversed type of the inner pair (p, q) = (i + 1 + n1, j - 1 - n2), 0 = no pair
                        auto t2of = [&](const int n1, const int n2) { return pair_type(sJ[1 + n2], sI[1 + n1]); };
                        const int t00 = t2of(0, 0), t01 = t2of(0, 1), t10 = t2of(1, 0), t11 = t2of(1, 1), t12 = t2of(1, 2), t21 = t2of(2, 1), t22 = t2of(2, 2);
                        // the big tables (sp1 = S[p - 1] = sI[n1], sq1 = S[q + 1] = sJ[n2]); a missing pair reads row 0 and is masked below
                        const int r11 = loopE(X, 1, 1, type, t11, si1, sj1, sI[1], sJ[1]);
                        const int r12 = loopE(X, 1, 2, type, t12, si1, sj1, sI[1], sJ[2]);
                        const int r21 = loopE(X, 2, 1, type, t21, si1, sj1, sI[2], sJ[1]);
                        const int r22 = loopE(X, 2, 2, type, t22, si1, sj1, sI[2], sJ[2]);
                        const int b1 = P->bulge[1];
                        auto cof = [&](const int w) { return w_e(w) - (int)l_mmI[w_in(w)]; };          // c(p,q) of a pair
                        auto put_if = [&](const int t2, const int e, const int shape) { if (t2) put(e, shape); };
                        auto put_gen = [&](const int s, const int n1, const int w) { const int key = (w_e(w) + o_mmI) * 1024 + P->gen_e[s][n1]; best = key < best ? key : best; };
                        put_if(t00, (int)l_stack[type * 8 + t00] + cof(w0), 0);
                        if (smax >= 1) { put_if(t01, b1 + (int)l_stack[type * 8 + t01] + cof(w1[0]), 0 << 5 | 1); put_if(t10, b1 + (int)l_stack[type * 8 + t10] + cof(w1[1]), 1 << 5 | 0); }
                        if (smax >= 2) { const int eb = P->bulge[2] + tau; put_bulge(eb, w2[0], 0 << 5 | 2); put_if(t11, r11 + cof(w2[1]), 1 << 5 | 1); put_bulge(eb, w2[2], 2 << 5 | 0); }
                        if (smax >= 3) {
                            const int eb = P->bulge[3] + tau;
                            put_bulge(eb, w3[0], 0 << 5 | 3); put_if(t12, r12 + cof(w3[1]), 1 << 5 | 2); put_if(t21, r21 + cof(w3[2]), 2 << 5 | 1); put_bulge(eb, w3[3], 3 << 5 | 0);
                        }
                        if (smax >= 4) {
                            const int eb = P->bulge[4] + tau;
                            put_bulge(eb, w4[0], 0 << 5 | 4); put_gen(4, 1, w4[1]); put_if(t22, r22 + cof(w4[2]), 2 << 5 | 2); put_gen(4, 3, w4[3]); put_bulge(eb, w44, 4 << 5 | 0);
                        }
                        if (smax >= 5) {
                            const int eb = P->bulge[5] + tau;
                            put_bulge(eb, w5[0], 0 << 5 | 5); put_gen(5, 1, w5[1]); put_gen(5, 2, w5[2]); put_gen(5, 3, w5[3]); put_gen(5, 4, w5b[0]); put_bulge(eb, w5b[1], 5 << 5 | 0);
                        }
                        if (smax >= 6) {
                            const int eb = P->bulge[6] + tau;
                            put_bulge(eb, w6[0], 0 << 5 | 6); put_gen(6, 1, w6[1]); put_gen(6, 2, w6[2]); put_gen(6, 3, w6[3]); put_gen(6, 4, w6b[1]); put_gen(6, 5, w6b[2]);
                            put_bulge(eb, w6b[3], 6 << 5 | 0);
                        }
                    } else {
                        const int s_lo = grp == 1 ? 7 : grp == 2 ? 18 : 25, s_hi = grp == 1 ? (smax < 17 ? smax : 17) : grp == 2 ? (smax < 24 ? smax : 24) : smax;
                        if (!staged) sizes(s_lo, s_hi);
                        else if (span_w + s_hi + 1 > 256) sizes_staged(std::true_type{}, s_lo, s_hi);
                        else sizes_staged(std::false_type{}, s_lo, s_hi);
                    }
                    if (active && (best >> 10) < V_INF) atomicMin(&cbest[cell], best);
                }
            }
            __syncthreads();
            // the cells in groups of V_G lanes (a thread per cell, as fold_generic_kernel's interval B, is slower here: this model's columns hold more split
            // candidates -- any of four stem terms makes one -- and the serial loop over them then sets the pace: 0.41 -> 0.47 s at L = 400)
            for (int cell = tid / V_G; cell < ((ncell + V_NT / V_G - 1) / (V_NT / V_G)) * (V_NT / V_G); cell += V_NT / V_G) {
                const bool live = cell < ncell;
                const int i = cell + 1, j = i + d;
                int type = 0, best = V_INF, mdec = V_INF, hp = V_INF, il = V_INF, shape = 0;
                if (live) {
                    type = ctype[cell];
                    if (type) {
                        if (sub == 0) {
                            hp = hairpin(X, i, j, type);
                            const int key = cbest[cell];
                            il = key == 0x7fffffff ? V_INF : key >> 10; shape = key & 1023;
                            best = il < hp ? il : hp;
                        }
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
                    const int pnf = pcnt[j], pn = pnf & ~V_PL_FLAG;
                    const int2* pj = pool + (size_t)j * pcap;
                    int kstart = sub;
                    if (n_cap <= V_PL_MAXN && !(pnf & V_PL_FLAG)) {
                        // the column's first V_PLN candidates out of LDS (lane `sub` takes sub, sub + V_G, ...): the fML reads they name go out with the cell's own reads
                        int vv[V_PLN / V_G], ff[V_PLN / V_G];
#pragma unroll
                        for (int u = 0; u < V_PLN / V_G; u++) vv[u] = pl4[V_PLN * j + sub + u * V_G];
#pragma unroll
                        for (int u = 0; u < V_PLN / V_G; u++) { const int s0 = (int)((unsigned)vv[u] >> 20); ff[u] = T.Mm(i, (s0 < i + V_TURN + 2 ? i + V_TURN + 2 : s0) - 1); }
#pragma unroll
                        for (int u = 0; u < V_PLN / V_G; u++) {
                            const int s0 = (int)((unsigned)vv[u] >> 20);
                            if (sub + u * V_G < pn && s0 >= i + V_TURN + 2) { const int e = ff[u] + ((vv[u] << 12) >> 12); mdec = e < mdec ? e : mdec; }
                        }
                        kstart = V_PLN + sub;
                    }
                    for (int k = kstart; k < pn; k += V_G * V_PU) {
                        int2 en[V_PU];
#pragma unroll
                        for (int u = 0; u < V_PU; u++) en[u] = pj[k + u * V_G < pn ? k + u * V_G : pn - 1];
                        int fm[V_PU];
#pragma unroll
                        for (int u = 0; u < V_PU; u++) fm[u] = T.Mm(i, (en[u].x < i + V_TURN + 2 ? i + V_TURN + 2 : en[u].x) - 1);      // (a clamped read where fML(i, s-1) does not exist yet)
#pragma unroll
                        for (int u = 0; u < V_PU; u++) {
                            const int e = fm[u] + en[u].y;
                            if (en[u].x >= i + V_TURN + 2) mdec = e < mdec ? e : mdec;
                        }
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
                        const int kf = pcnt[j], k = kf & ~V_PL_FLAG;
                        if (k < pcap) {
                            pool[(size_t)j * pcap + k] = make_int2(i, stem);
                            int flag = kf & V_PL_FLAG;
                            if (n_cap <= V_PL_MAXN && k < V_PLN) {
                                if (stem >= -(1 << 19) && stem < (1 << 19)) pl4[V_PLN * j + k] = (i << 20) | (stem & 0xfffff);
                                else flag = V_PL_FLAG;          // (this model's arithmetic is uncapped: a stem term of INF-sized parts can be a candidate)
                            }
                            pcnt[j] = (k + 1) | flag;
                        }
                    }
                    T.c[(size_t)d * T.ld + i] = newc;
                    // the loop realises c unless the hairpin does as well or better, or the multiloop strictly better: the order of the backtrack's tests
                    tbtab[(size_t)d * T.ld + i] = (unsigned short)((type && il < hp && il == best) ? 1 + shape : 0);
                    {
                        const int in = rtype_of(type) * 25 + S[j + 1] * 5 + S[i - 1];          // (i,j) as the inner pair of a later loop
                        wtab[(size_t)d * T.ld + i] = type ? (in << 24) | ((newc + (int)l_mmI[in]) & 0xffffff) : V_INF;
                    }
                    T.m[(size_t)d * T.ld + i] = mm;
                    T.dm[(size_t)d * T.ld + i] = mdec;
                }
            }
            __syncthreads();
        }

        } else {
#ifdef MIRP_X_GEN_NOEPI             // timing experiment: fill only (results are empty)
        if (tid == 0) { out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; }
#else
        epilogue<FoldParams185, GTab185, V_NT>(X, T, f3, starts, lens, btstk, red, btbuf, nc, win, max_lines, ss_stride, out_lines, out_ss, out_nlines,
                                                out_mfe, out_status);
#endif
        }
    }
}

}  // namespace v185

static size_t fold185_lds_bytes_fill(int n_cap) {
    return v185::fold185_lds_bytes_base_fill(n_cap) + (((size_t)(4 + 4 + 2 + 1) * ((size_t)n_cap + 8) + 15) / 16) * 16 + sizeof(short) * (2 * 200 + 64) +
           sizeof(int) * (V_NT / 64) * (V_STAGE + 2) + 16 + (n_cap <= V_PL_MAXN ? sizeof(int) * V_PLN * ((size_t)n_cap + 8) : 0);
}
// what the larger of the two kernels takes (the budget check of the caller)
size_t fold185_lds_bytes(int n_cap, int max_lines) {
    const size_t a = fold185_lds_bytes_fill(n_cap), b = v185::fold185_lds_bytes_base(n_cap, max_lines);
    return a > b ? a : b;
}

size_t fold185_ws_slot_ints(int n_cap, int span) {
    // c, fML, DML, the split-candidate pool (two ints per entry, span + 2 entries per column)
    return 4 * v185::fold185_table_ints(n_cap, span) + ((v185::fold185_table_ints(n_cap, span) / 2 + 3) & ~(size_t)3) + ((2 * (size_t)v185::fold185_ld(n_cap) * (size_t)(span + 2) + 3) & ~(size_t)3) + 4;          // (+ w, + the trace-back codes)
}

hipError_t launch_fold185(hipStream_t stream, int grid, const FoldParams185* P, const unsigned char* seqs, const long long* offs, const int* lens,
                          const int* work_list, int n_work,
                          int span, int n_cap, int* ws, size_t ws_slot_ints, int max_lines, int ss_stride, MirpFoldLine* out_lines, char* out_ss,
                          int* out_nlines, int* out_mfe, int* out_status) {
    const size_t lds1 = fold185_lds_bytes_fill(n_cap), lds2 = v185::fold185_lds_bytes_base(n_cap, max_lines);
    if (lds1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)v185::fold185_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return e;
    }
    if (lds2 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)v185::fold185_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
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
        hipLaunchKernelGGL(v185::fold185_kernel<1>, dim3(nb), dim3(V_NT), lds1, stream, P, seqs, o2, l2, wl, nb, span, n_cap, ws, ws_slot_ints, max_lines, ss_stride, ol, os, on,
                           om, ost);
        hipLaunchKernelGGL(v185::fold185_kernel<2>, dim3(nb), dim3(V_NT), lds2, stream, P, seqs, o2, l2, wl, nb, span, n_cap, ws, ws_slot_ints, max_lines, ss_stride, ol, os, on,
                           om, ost);
    }
    return hipGetLastError();
}

}  // namespace mirp
