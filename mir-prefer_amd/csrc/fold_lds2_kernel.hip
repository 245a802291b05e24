// LDS-resident fill kernel of the default fold model (vienna-2.1.2: Turner-2004, dangles 2), TWO anti-diagonals per barrier interval.
// Replaces the RNALfold -L subprocess of /root/reference/miR_PREFeR.py:3053 for windows of n <= 350, span <= 300 (fold_lds_kernel.hip keeps the
// one-diagonal schedule for the vienna-1.8.5 model; the epilogue kernel, the archive layout and the candidate jobs are shared, fold_lds_common.h).
//
// Schedule.  c of diagonals d and d+1 depends on c of diagonals <= d-1 only (the stacked pair of d+1 sits on d-1), fML(d+1) on fML(d) through a
// two-term minimum, and the multiloop splits of d+2, d+3 on fML of diagonals <= d-2.  One interval therefore runs
//     phase B of (d, d+1)   ||   phase A of (d+2, d+3)       -- barrier --
// with half the barriers, half the phase-B chains and half the per-interval set-up of the one-diagonal schedule:
//   * phase A (all 16 waves): interior-loop candidates of the paired cells of BOTH diagonals out of one combined list (lane = paired cell,
//     wave = candidate group, as before; an entry carries which of the two diagonals it is on, which only moves its ring rows by one row and
//     its j by one), and the multiloop splits of both diagonals.  The few candidates whose inner pair is not final while phase A runs --
//     the stacked pair of both diagonals, the two 1-bulges of the second -- are left to phase B, which adds them to the cell's key;
//   * phase B (waves 0-5, one thread per column i): finalises c and fML of cells (i, i+d) and (i, i+d+1).  fML(d+1, i) needs fML(d, i+1) of the
//     neighbouring lane: a wave owns 63 columns, its last lane recomputes the first column of the next wave (no stores), the value moves by DPP;
//   * the multiloop closing term is PUSHED: the thread that finalises DML(i, j) adds the closing energy of the outer pair (i-1, j+1) to that
//     cell's candidate key (code 1023: ranks after every interior loop of equal energy, as the backtrack tests it last), two diagonals ahead.
//     There is no DML ring, and the closing term is off phase B's dependency chain.
// LDS: fML triangle (biased uint16), c ring (G0 = c + inner mismatch, 32 diagonals + mirror row), 4 x candidate keys + 4 x split minima (diagonal
// & 3), two combined paired-cell lists (consumed / being built), parameter tables.  No MFMA: integer min-plus with irregular table look-ups.
#include "fold_lds_common.h"

namespace mirp {

#ifdef MIRP_L2_CLOCKS      // dev build (make L2CLOCKS=1): per-wave phase clocks, printed by fold_lds2_clocks_print()
__device__ unsigned long long g_l2_clk[16 * 4 + 8];
#define L2CLK(x) x
#else
#define L2CLK(x)
#endif
// generic rows (loop sizes U) of the candidate groups 4-7, and the rows the two small-shape groups take on top (their waves also own phase B
// but have the most slack: measured per-wave busy clocks, make VARIANT=clk VFLAGS=-DMIRP_L2_CLOCKS)
#ifndef MIRP_BAL
#define MIRP_BAL 0
#endif
#if MIRP_BAL == 0
#define MIRP_ROWS4 22, 17, 12, 7
#define MIRP_ROWS5 21, 18, 11, 8
#define MIRP_ROWS6 20, 16, 13, 9
#define MIRP_ROWS7 19, 15, 14, 10, 6
#elif MIRP_BAL == 1
#define MIRP_ROWS4 22, 17, 12
#define MIRP_ROWS5 21, 18, 11
#define MIRP_ROWS6 20, 16, 13
#define MIRP_ROWS7 19, 15, 14
#define MIRP_ROWS14 9, 7
#define MIRP_ROWS15 10, 8, 6
#elif MIRP_BAL == 2
#define MIRP_ROWS4 22, 17, 12
#define MIRP_ROWS5 21, 18, 11
#define MIRP_ROWS6 20, 16, 13
#define MIRP_ROWS7 19, 15, 14
#define MIRP_ROWS14 10, 6
#define MIRP_ROWS15 9, 8, 7
#endif
#define L2_LISTCAP 704      // combined list of two diagonals: up to 2 x 352 paired cells
#define L2_BW 63            // columns per phase-B wave (lane 63 shadows the next wave's first column)
#define L2_NBW 6            // phase-B waves: 6 x 63 = 378 >= 346 columns

struct Lds2Layout {
    unsigned fml, ring, acc, S, pax, qbr, lent, loi, tabs, misc, hc, total;
};
__host__ __device__ constexpr Lds2Layout lds2_layout() {
    Lds2Layout L{};
    unsigned o = 0;
    L.fml = o; o += lds_al((tri_off(LDMAX, LCAP - 2) + 2 + 16) * 2);       // fML triangle, d = 4..LDMAX-1 at n = LCAP - 2 (+ slack for the copy-out's over-reads)
    L.ring = o; o += lds_al(CRING_ROWS * CSTR * 2);                        // c ring
    L.acc = o; o += lds_al(8 * LCAP * 4);                                  // ckey[4][LCAP] | mdec[4][LCAP]; the staging copy of the window's characters borrows it during set-up
    L.S = o; o += lds_al(LCAP + 8);
    L.pax = o; o += lds_al((LCAP + 8) * 2);
    L.qbr = o; o += lds_al(LCAP + 8);
    L.lent = o; o += lds_al(2 * L2_LISTCAP * 2);                           // entries: i | type << 9 | second diagonal << 12
    L.loi = o; o += lds_al(2 * L2_LISTCAP);                                // outer-pair table index of the entry
    L.tabs = o; o += lds_al((unsigned)sizeof(LdsTables));
    L.misc = o; o += lds_al((16 + ARCH_RB) * 4);
    L.hc = o; o += 32;                                                     // code of the mismatch bonus of an outer pair by its two neighbour bases (MIRP_E1)
    L.total = o;
    return L;
}
static_assert(lds2_layout().total <= 160 * 1024, "two-diagonal fill kernel LDS budget");

// stack-free / bulge-free variant of a1_small14f: the stacked pair is never final while phase A runs, the two 1-bulges only for a cell of the
// first diagonal of the pair (`second` = false).  1x1 and 1x2 as before.
__device__ __forceinline__ unsigned a1_small14f2(const A1& a, int i, int j, int type, bool second) {
    lds_vu8 Sv = (lds_vu8)a.S;
    const int s_i = Sv[i], s_i1 = Sv[i + 1], s_i2 = Sv[i + 2], s_j3 = Sv[j - 3], s_j2 = Sv[j - 2], s_j1 = Sv[j - 1], s_j = Sv[j];
    const unsigned short* rb = a.cring;
    const unsigned g01 = rb[((a.r0 - 1) & 31) * CSTR + i + 1], g10 = rb[((a.r0 - 1) & 31) * CSTR + i + 2];
    const unsigned g11 = rb[((a.r0 - 2) & 31) * CSTR + i + 2], g12 = rb[((a.r0 - 3) & 31) * CSTR + i + 2];
    const int t01 = rtype_of(pair_type(s_i1, s_j2)), t10 = rtype_of(pair_type(s_i2, s_j1));
    const int t11 = rtype_of(pair_type(s_i2, s_j2)), t12 = rtype_of(pair_type(s_i2, s_j3));
    const LdsTables& T = *a.T;
    const int m01 = T.mismatchI[t01 * 25 + s_j1 * 5 + s_i], m10 = T.mismatchI[t10 * 25 + s_j * 5 + s_i1];
    const int m11 = T.mismatchI[t11 * 25 + s_j1 * 5 + s_i1], m12 = T.mismatchI[t12 * 25 + s_j2 * 5 + s_i1];
    const int st01 = T.stack[type * 8 + t01], st10 = T.stack[type * 8 + t10], b1 = T.bulge[1];
    const int r11 = a.P->int11[type][t11][s_i1][s_j1];
    const int r12 = a.P->int21[type][t12][s_i1][s_j2][s_j1];
    unsigned res = a1_small_key(g11, m11, r11, 1u << 5 | 1u);
    unsigned k = a1_small_key(g12, m12, r12, 1u << 5 | 2u); res = k < res ? k : res;
    const unsigned k01 = a1_small_key(g01, m01, b1 + st01, 0u << 5 | 1u), k10 = a1_small_key(g10, m10, b1 + st10, 1u << 5 | 0u);
    k = k01 < k10 ? k01 : k10;
    if (!second) res = k < res ? k : res;
    return res;
}

__global__ void __launch_bounds__(LNT) fold_lds2_kernel(
    const FoldParams* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs, const int* __restrict__ win_lens,
    int n_work, int win_base, int span, short* __restrict__ slabs, size_t slab_shorts, int* __restrict__ win_state,
    unsigned int* __restrict__ work_counter, int* __restrict__ fallback_list, unsigned int* __restrict__ fallback_count,
    int* __restrict__ out_nlines, int* __restrict__ out_mfe, int* __restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr Lds2Layout LY = lds2_layout();
    unsigned short* fml = (unsigned short*)(smem + LY.fml);        // biased uint16 (FML_BIAS), 65535 = INF
    unsigned short* cring = (unsigned short*)(smem + LY.ring);     // [33][CSTR] G0 + 32768 as uint16, 65535 = INF
    int* acc = (int*)(smem + LY.acc);                              // ckey[d & 3][LCAP], then mdec[d & 3][LCAP]
    unsigned char* seq = smem + LY.acc;                            // set-up only (before acc is initialised)
    unsigned char* S = smem + LY.S;
    unsigned short* pax = (unsigned short*)(smem + LY.pax);
    unsigned char* qbr = smem + LY.qbr;
    unsigned short* lent = (unsigned short*)(smem + LY.lent);      // [2][L2_LISTCAP]: combined list of diagonals (e, e+1), e even, in buffer (e >> 1) & 1
    unsigned char* loi = smem + LY.loi;                            // [2][L2_LISTCAP]: type * 25 + S[i+1] * 5 + S[j-1] of the entry's cell (outer pair of its interior loops)
    LdsTables& T = *(LdsTables*)(smem + LY.tabs);
    int* misc = (int*)(smem + LY.misc);                            // 0: next window, 1: overflow flag, 4..7: list lengths
    int* lcnt = misc + 4;                                          // [4]: entries of the list of pair (e, e+1) at (e >> 1) & 3
    int* rbt = misc + 16;                                          // [ARCH_RB]: row-block offsets of the window's archive slabs
    short* spec = (short*)(cring + 29 * CSTR);                     // special-hairpin energies by start position, read on diagonals 4, 5, 7 only (rows first written on diagonal 29)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nc = CSTR;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    typedef __attribute__((address_space(3))) int* lds_i32p;

    // ---- one-time: hot parameter tables into LDS
    if (tid == 0) misc[2] = 0;          // set when the compiled-in parameter structure does not hold (MIRP_E1)
    __syncthreads();
    for (int x = tid; x < 64; x += LNT) T.stack[x] = (short)min(P->stack[x >> 3][x & 7], (int)I16_INF);
    for (int x = tid; x < 31; x += LNT) { T.bulge[x] = (short)min(P->bulge[x], (int)I16_INF); T.internal_loop[x] = (short)min(P->internal_loop[x], (int)I16_INF); }
    for (int x = tid; x < 200; x += LNT) {
        int t = x / 25, a = (x % 25) / 5, b = x % 5;
        T.mismatchI[x] = (short)min(P->mismatchI[t][a][b], (int)I16_INF); T.mismatchH[x] = (short)min(P->mismatchH[t][a][b], (int)I16_INF);
        T.mismatchM[x] = (short)P->mismatchM[t][a][b]; T.mismatch1nI[x] = (short)min(P->mismatch1nI[t][a][b], (int)I16_INF);
        T.mismatch23I[x] = (short)min(P->mismatch23I[t][a][b], (int)I16_INF);
    }
    xt_fill(T, P, tid, LNT);
    if (tid < 40) { T.dangle5[tid] = (short)P->dangle5[tid / 5][tid % 5]; T.dangle3[tid] = (short)P->dangle3[tid / 5][tid % 5]; }
    if (tid < 25) T.rt2[tid] = (unsigned char)rtype_of(pair_type(tid / 5, tid % 5));
    if (tid == 0) { T.ML_closing = (short)P->ML_closing; T.ML_intern = (short)P->ML_intern; T.TerminalAU = (short)P->TerminalAU; T.ninio = (short)P->ninio; T.MAX_NINIO = (short)P->MAX_NINIO; }
    __syncthreads();

#ifdef MIRP_E1
    // Outer-pair terms without a table read: in the Turner-2004 set mismatchI[t][a][b] = F(t) + H(a, b) with F = 0 for CG / GC and 70 for the AU / GU
    // classes, H one of {0, -80, -100, -60} (A.G, G.A / G.G, U.U mismatches), and mismatch1nI[t][a][b] = F(t).  The constants are compiled in
    // (no registers held across the kernel) and CHECKED against the parameters in use when the workgroup starts: a set that differs sends
    // every window to the generic kernel.  A list entry carries the 2-bit code of H for its cell, so phase A1 needs no second LDS round trip
    // in front of its ring reads.
    constexpr int E1_F = 70, E1_TAU = 50, E1_H1 = -80, E1_H2 = -100, E1_H3 = -60;
    unsigned char* hcd = smem + LY.hc;
    if (tid < 25) {
        const int a = tid / 5, b = tid % 5;
        const int code = (a == 1 && b == 3) ? 1 : ((a == 3 && b == 1) || (a == 3 && b == 3)) ? 2 : (a == 4 && b == 4) ? 3 : 0;
        const int hv = code == 1 ? E1_H1 : code == 2 ? E1_H2 : code == 3 ? E1_H3 : 0;
        hcd[tid] = (unsigned char)code;
        int ok = P->TerminalAU == E1_TAU;
        for (int t = 1; t <= 6; t++) ok = ok && P->mismatchI[t][a][b] == (t > 2 ? E1_F : 0) + hv && P->mismatch1nI[t][a][b] == (t > 2 ? E1_F : 0);
        if (!ok) misc[2] = 1;          // (misc[2] starts at 0: see below)
    }
    __syncthreads();
#endif
    // phase-B column of this thread: waves 0..5 own 63 columns each, lane 63 shadows the first column of the next wave
    const bool bwave = wave < L2_NBW;
    const int bx = wave * L2_BW + lane;                  // column index, i = bx + 1
    const bool shadow = lane == L2_BW;

    for (;;) {
        if (tid == 0) misc[0] = (int)atomicAdd(work_counter, 1u);
        __syncthreads();
        const int win = misc[0];
        __syncthreads();
        if (win >= n_work) break;
        const long long o0 = offs[win];
        const int n = win_lens ? win_lens[win] : (int)(offs[win + 1] - o0);
        short* carch = slabs + (size_t)win * 3 * slab_shorts;      // per-window slab: c, fML and trace-back triangles (read by fold_lds_epilogue_kernel)
        short* fml_out = carch + slab_shorts;
        unsigned short* tb_out = reinterpret_cast<unsigned short*>(carch + 2 * slab_shorts);
#ifdef MIRP_E1
        const bool unsupported = misc[2] != 0;
#else
        const bool unsupported = false;
#endif
        if (n < 1 || n > LCAP - 2 || unsupported) {   // wave-uniform: empty window, too long for this kernel, or parameters it cannot use (-> generic kernel)
            if (tid == 0) {
                out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; win_state[win] = 0;
                if (n >= 1) { unsigned int k = atomicAdd(fallback_count, 1u); fallback_list[k] = win_base + win; }
            }
        } else {
        const int D = (span - 1 < n - 1) ? span - 1 : n - 1;      // largest pair distance = last diagonal of the fill
        // ---- stage sequence, codes, special hairpins, pair-code arrays
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
        if (tid == 0) {
            misc[1] = 0;
            for (int x = 0; x < 4; x++) lcnt[x] = 0;
        }
        if (tid >= 64 && tid < 64 + ARCH_RB) rbt[tid - 64] = arch_rowblk_off(tid - 64, n, span);
        __syncthreads();
        if (tid == 0) { S[0] = S[n]; S[n + 1] = S[1]; }
        for (int x = tid; x <= n; x += LNT) {
            short s3 = -32768, s4 = -32768, s6 = -32768;
            if (x >= 1) {
                if (x + 4 <= n)
                    for (int k = 0; k < P->n_tri; k++) { bool m = true; for (int t = 0; t < 5; t++) m = m && (seq[x + t] == (unsigned char)P->tri[k][t]); if (m && s3 == -32768) s3 = (short)P->triE[k]; }
                if (x + 5 <= n)
                    for (int k = 0; k < P->n_tetra; k++) { bool m = true; for (int t = 0; t < 6; t++) m = m && (seq[x + t] == (unsigned char)P->tetra[k][t]); if (m && s4 == -32768) s4 = (short)P->tetraE[k]; }
                if (x + 7 <= n)
                    for (int k = 0; k < P->n_hexa; k++) { bool m = true; for (int t = 0; t < 8; t++) m = m && (seq[x + t] == (unsigned char)P->hexa[k][t]); if (m && s6 == -32768) s6 = (short)P->hexaE[k]; }
            }
            spec[x] = s3; spec[nc + x] = s4; spec[2 * nc + x] = s6;
            // combined pair codes (only interior positions are ever read: p - 1 >= 1, q + 1 <= n)
            if (x >= 1) {
                pax[x] = (unsigned short)xt_pcode(S[x], x > 1 ? (int)S[x - 1] : 0);
                qbr[n + 1 - x] = (unsigned char)xt_qcode(S[x], x < n ? (int)S[x + 1] : 0);
            }
        }
        __syncthreads();      // the character copy is dead: its bytes become the key / split-minimum arrays
        for (int x = tid; x < 8 * LCAP; x += LNT) acc[x] = x >= 4 * LCAP ? INF : (int)KEY_NONE;   // ckey x 4 | mdec x 4

        // Appends this thread's cells (i, i + e) and (i, i + e + 1) to the combined list of the pair (e, e + 1), e even: ballot compaction
        // inside the wave, one LDS atomic per wave for its range.  Nothing depends on the order of a list.
        auto list_pair = [&](const int e, const int lt0, const int oi0, const int lt1, const int oi1) {
            const unsigned long long b0 = __ballot(lt0 != 0), b1 = __ballot(lt1 != 0);
            int base = 0;
            if ((b0 | b1) && lane == 0) {
                const unsigned la = (unsigned)(size_t)(lds_i32p)&lcnt[(e >> 1) & 3];
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(la), "v"((int)(__popcll(b0) + __popcll(b1))) : "memory");
            }
            base = __builtin_amdgcn_readfirstlane(base);
            const int lb = ((e >> 1) & 1) * L2_LISTCAP + base;
#ifdef MIRP_E1
            // oi = type * 25 + S[i+1] * 5 + S[j-1]: the entry keeps the H code of the two neighbour bases instead
            if (lt0) { const int k = lb + (int)__popcll(b0 & lane_lt); lent[k] = (unsigned short)((bx + 1) | (lt0 << 9) | ((int)hcd[oi0 - lt0 * 25] << 13)); }
            if (lt1) { const int k = lb + (int)__popcll(b0) + (int)__popcll(b1 & lane_lt); lent[k] = (unsigned short)((bx + 1) | (lt1 << 9) | (1 << 12) | ((int)hcd[oi1 - lt1 * 25] << 13)); }
#else
            if (lt0) { const int k = lb + (int)__popcll(b0 & lane_lt); lent[k] = (unsigned short)((bx + 1) | (lt0 << 9)); loi[k] = (unsigned char)oi0; }
            if (lt1) { const int k = lb + (int)__popcll(b0) + (int)__popcll(b1 & lane_lt); lent[k] = (unsigned short)((bx + 1) | (lt1 << 9) | (1 << 12)); loi[k] = (unsigned char)oi1; }
#endif
        };
        // list of the first pair with interior loops, (6, 7)
        if (bwave && D >= 6) {
            int lt0 = 0, oi0 = 0, lt1 = 0, oi1 = 0;
            if (!shadow) {
                const int i = bx + 1;
                if (i + 6 <= n) { lt0 = pair_type(S[i], S[i + 6]); oi0 = lt0 * 25 + S[i + 1] * 5 + S[i + 5]; }
                if (D >= 7 && i + 7 <= n) { lt1 = pair_type(S[i], S[i + 7]); oi1 = lt1 * 25 + S[i + 1] * 5 + S[i + 6]; }
            }
            list_pair(6, lt0, oi0, lt1, oi1);
        }
        __syncthreads();

        // split loop state carried across diagonals (see splits below)
        int sp_ncpad = 0, sp_nsub = 0, sp_pair = 0, sp_sub = 0, sp_so1 = 0, sp_si1 = 0, sp_so2 = 0, sp_si2 = 0;
        const int abase = (bwave && bx < 8 * ARCH_RB) ? rbt[bx >> 3] + (bx & 7) - 32 : 0;   // archive offset of (d, i = bx + 1) is abase + 8 d

        // phase A2: multiloop splits DML(i,j) = min_t fML(i, i+t) + fML(i+t+1, j) of diagonal d into mdec[d & 3].  Called once per diagonal in
        // ascending order (the lane mapping and the operand offsets are carried across diagonals in registers and advanced with two scalar adds).
        // The split point t is wave-uniform; every lane owns TWO consecutive cells (i, i+1), i odd.  Operand a (diagonal t, cells i, i+1) is one
        // aligned 32-bit word; operand b (diagonal d-t-1, cells i+t+1, i+t+2) is one aligned word for odd t and straddles two words for even t
        // (one v_alignbit).  The step between the splits of a wave is even, so that parity is wave-uniform.  One packed saturating add and one
        // packed min then relax both cells; with the biased uint16 encoding a sum that involves an INF entry saturates at 65535 and any sum of
        // two finite entries is <= 65534, so every split t in [4, d-5] is relaxed unconditionally.
        auto splits = [&](const int d) {
            const int ncell = n - d;
            int* mdec = acc + (4 + (d & 3)) * LCAP;
            const int npair = (ncell + 1) >> 1;
            const int ncpad = (npair + 63) & ~63;
            if (ncpad != sp_ncpad) {
                sp_ncpad = ncpad;
                sp_nsub = (LNT / ncpad) & ~1;     // even, >= 4 for ncell <= 384
                sp_pair = tid % ncpad;
                sp_sub = __builtin_amdgcn_readfirstlane(tid / ncpad);
                const int t0 = 4 + sp_sub, u0 = d - t0 - 1, s1 = sp_nsub;
                sp_so1 = __builtin_amdgcn_readfirstlane(2 * tri_off(t0, n));
                sp_si1 = __builtin_amdgcn_readfirstlane(2 * (tri_off(t0 + s1, n) - tri_off(t0, n)));
                sp_so2 = __builtin_amdgcn_readfirstlane(2 * (tri_off(u0, n) + t0 + 1 - ((t0 & 1) ? 0 : 1)));
                sp_si2 = __builtin_amdgcn_readfirstlane(2 * (tri_off(u0 - s1, n) - tri_off(u0, n) + s1));
            } else {
                // d advanced by one since the last call: operand b starts one diagonal higher (tri_off(u + 1) = tri_off(u) + tri_len(u)), and its
                // first difference grows by s1 cells (s1 is even, so the paddings of the two diagonals involved cancel)
                const int uprev = d - 1 - (4 + sp_sub) - 1;
                sp_so2 += 2 * tri_len_any(uprev, n);
                sp_si2 += 2 * sp_nsub;
            }
            const int nsub = sp_nsub, pair = sp_pair, sub = sp_sub;
            if (sub < nsub) {
                const int i = 2 * pair + 1;
                const int s1 = nsub;
                int t = 4 + sub;
                const int odd = t & 1;
                int so1 = sp_so1, so2 = sp_so2, si1 = sp_si1, si2 = sp_si2;
                const int sss = 2 * s1 * s1;
                us2 bu = {65535, 65535};
                typedef const __attribute__((address_space(3))) unsigned* lds_cu32;
                const unsigned fb0 = (unsigned)(size_t)(lds_cu32)reinterpret_cast<const unsigned*>(fml + i);
                unsigned va = fb0 + (unsigned)so1, vb = fb0 + (unsigned)so2;
#define MIRP_SSTEP() do { va += (unsigned)si1; vb += (unsigned)si2; asm volatile("s_sub_i32 %0, %0, %2\n\ts_sub_i32 %1, %1, %2" : "+s"(si1), "+s"(si2) : "s"(sss) : "scc"); } while (0)
#define MIRP_LDA() (*(lds_cu32)(va))
#define MIRP_LDB(o) (*(lds_cu32)(vb + (o)))
                auto group = [&](auto ODD, auto KK) {
                    constexpr bool kOdd = decltype(ODD)::value;
                    constexpr int K = decltype(KK)::value;
                    unsigned a[K], b[K], c[K];
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        a[k] = MIRP_LDA(); b[k] = MIRP_LDB(0);
                        if (!kOdd) c[k] = MIRP_LDB(4);
                        MIRP_SSTEP();
                    }
                    us2 e[K];
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        const unsigned bw = kOdd ? b[k] : __builtin_amdgcn_alignbit(c[k], b[k], 16);
                        us2 av, bv;
                        __builtin_memcpy(&av, &a[k], 4); __builtin_memcpy(&bv, &bw, 4);
                        e[k] = __builtin_elementwise_add_sat(av, bv);
                    }
#pragma unroll
                    for (int w = 1; w < K; w *= 2)
#pragma unroll
                        for (int k = 0; k + w < K; k += 2 * w) e[k] = __builtin_elementwise_min(e[k], e[k + w]);
                    bu = __builtin_elementwise_min(bu, e[0]);
                    t += K * s1;
                };
                auto relax = [&](auto ODD) {
                    while (t + 7 * s1 <= d - 5) group(ODD, std::integral_constant<int, 8>{});     // 16 (24) reads in flight
                    if (t + 3 * s1 <= d - 5) group(ODD, std::integral_constant<int, 4>{});
                    if (t + s1 <= d - 5) group(ODD, std::integral_constant<int, 2>{});
                    if (t <= d - 5) group(ODD, std::integral_constant<int, 1>{});
                };
                if (odd) relax(std::true_type{}); else relax(std::false_type{});
#undef MIRP_SSTEP
#undef MIRP_LDA
#undef MIRP_LDB
                const unsigned r0 = bu[0], r1 = bu[1];
                if (i <= ncell && r0 < 65535u) atomicMin(&mdec[i], (int)r0 - 2 * FML_BIAS);
                if (i + 1 <= ncell && r1 < 65535u) atomicMin(&mdec[i + 1], (int)r1 - 2 * FML_BIAS);
            }
        };

        // phase A1 of the pair (e, e+1), e even, e >= 6: interior-loop candidates of the cells of the combined list.  The c ring holds
        // G0(p,q) = c(p,q) + mismatchI[rtype(pq)][S[q+1]][S[p-1]] (+ 32768).  Everything a cell of the second diagonal does differently is per
        // lane: j one further, its ring rows are "the rows after" those of the first diagonal (hence the mirror row), its keys go to the
        // other key array.  Not evaluated here (inner pair not final yet): the stacked pair of both, the 1-bulges of the second (phase B).
        auto interior = [&](const int e) {
            const bool two = e + 1 <= D;
            const int q = e >> 1;
            const unsigned short* clist = lent + (q & 1) * L2_LISTCAP;
            const unsigned char* coi = loi + (q & 1) * L2_LISTCAP;
            const int ncp = __builtin_amdgcn_readfirstlane(lcnt[q & 3]);
            const int nblk = (ncp + 63) >> 6;
            unsigned* ckey0 = reinterpret_cast<unsigned*>(acc + (e & 3) * LCAP);
            unsigned* ckey1 = reinterpret_cast<unsigned*>(acc + ((e + 1) & 3) * LCAP);
            // roles (0-7: generic rows, 8-13: bulges / 1xn, 14-15: small shapes); waves 0-5 also own phase B, so they take the cheapest jobs
            const int role = wave < 4 ? wave : wave < 6 ? wave + 10 : wave < 12 ? wave + 2 : wave - 8;
            A1 a;
            a.P = P; a.T = &T; a.S = S; a.cring = cring; a.pax = pax; a.qbr = qbr; a.n = n;
            // steady state: every loop size is admissible on both diagonals (um = MAXLOOP), both diagonals share the blocks.  Before that the
            // admissible sizes differ between the two diagonals: one pass per diagonal over the same blocks, the other diagonal's lanes idle.
            const bool steady = e - 2 - (TURN + 1) >= MAXLOOP;
            const int npass = steady ? 1 : (two ? 2 : 1);
            for (int pass = 0; pass < npass; pass++) {
                const int dd = e + pass;
                for (int blk = 0; blk < nblk; blk++) {
                    {   // re-materialise the wave-uniform loop parameters per block: keeps the admissibility tests and row offsets as plain
                        // scalar compares inside the block instead of dozens of hoisted masks (SGPR spills)
                        int r0 = dd - 2, um = dd - 2 - (TURN + 1) < MAXLOOP ? dd - 2 - (TURN + 1) : MAXLOOP;
                        asm volatile("" : "+s"(r0), "+s"(um));
                        a.r0 = r0; a.um = um;
                        a.rowtab = P->ring_rowoff[r0 & 31];
                    }
                    const int k = blk * 64 + lane;
                    const bool inl = k < ncp;
                    const unsigned ent = inl ? (unsigned)clist[k] : (1u | (1u << 9));      // idle lanes: harmless dummy cell
#ifndef MIRP_E1
                    const int oi = inl ? (int)coi[k] : 25;
#endif
                    const bool second = (ent >> 12) & 1;
                    const bool act = inl && (steady || (int)second == pass);
                    const bool shift = steady && second;             // ring rows and j of the second diagonal relative to dd
                    const int i = ent & 511, type = (ent >> 9) & 7, j = i + dd + (shift ? 1 : 0);
                    a.cring = cring + (shift ? CSTR : 0);
                    unsigned* ck = second ? ckey1 : ckey0;
                    unsigned res = KEY_NONE;
                    int au1 = 0, mmo = 0, mm1 = 0;
#ifdef MIRP_E1
                    {
                        const int hcode = (ent >> 13) & 3;
                        const int hval = hcode == 0 ? 0 : hcode == 1 ? E1_H1 : hcode == 2 ? E1_H2 : E1_H3;
                        const int f = type > 2 ? E1_F : 0;
                        au1 = type > 2 ? E1_TAU : 0;
                        mmo = f + hval; mm1 = f;
                    }
#else
                    if (role < 14) {
                        au1 = type > 2 ? (int)T.TerminalAU : 0;
                        mmo = T.mismatchI[oi]; mm1 = T.mismatch1nI[oi];
                    }
#ifdef MIRP_ROWS14
                    else mmo = T.mismatchI[oi];
#endif
#endif
                    if (role < 8) {
#define MIRP_GEN(CK)                                                                      \
    switch (role) {                                                                       \
    case 0: res = a1_generic<CK, 30, 23>(a, i, j, mmo); if (CK) a1_i1<CK, 28, 29>(a, i, j, xi); else a1_i1f<28, 29>(a, i, j, xi); break;      \
    case 1: res = a1_generic<CK, 29, 24>(a, i, j, mmo); if (CK) a1_i1<CK, 25, 27>(a, i, j, xi); else a1_i1f<25, 27>(a, i, j, xi); break;      \
    case 2: res = a1_generic<CK, 28, 25>(a, i, j, mmo); if (CK) a1_i0<CK, 26, 29>(a, i, j, xi); else a1_i0f<26, 29>(a, i, j, xi); break;      \
    case 3: res = a1_generic<CK, 27, 26>(a, i, j, mmo); if (CK) a1_b1<CK, 26, 30>(a, i, j, xb); else a1_b1f<26, 30>(a, i, j, xb); break;      \
    case 4: res = a1_generic<CK, MIRP_ROWS4>(a, i, j, mmo); break;                    \
    case 5: res = a1_generic<CK, MIRP_ROWS5>(a, i, j, mmo); break;                    \
    case 6: res = a1_generic<CK, MIRP_ROWS6>(a, i, j, mmo); break;                    \
    default: res = a1_generic<CK, MIRP_ROWS7>(a, i, j, mmo); break;               \
    }
                        unsigned xb = KEY_INF, xi = KEY_INF;
                        if (a.um >= MAXLOOP) { MIRP_GEN(false) } else { MIRP_GEN(true) }
                        if (role < 4) {
                            const unsigned rb = a1_key(xb, -32768 - OTH_BIAS + au1);
                            const unsigned ri = a1_key(xi, -32768 - OTH_BIAS + mm1);
                            res = rb < res ? rb : res;
                            res = ri < res ? ri : res;
                        }
#undef MIRP_GEN
                    } else if (role < 14) {
                        unsigned bb = KEY_INF, bi = KEY_INF;
#define MIRP_OTH(CK)                                                                      \
    switch (role) {                                                                       \
    case 8: a1_b0<CK, 2, 18>(a, i, j, bb); break;                                         \
    case 9: a1_b0<CK, 19, 30>(a, i, j, bb); a1_b1<CK, 2, 6>(a, i, j, bb); break;          \
    case 10: a1_b1<CK, 7, 22>(a, i, j, bb); break;                                        \
    case 11: a1_b1<CK, 23, 25>(a, i, j, bb); a1_i0<CK, 3, 15>(a, i, j, bi); break;        \
    case 12: a1_i0<CK, 16, 25>(a, i, j, bi); a1_i1<CK, 3, 8>(a, i, j, bi); break;         \
    default: a1_i1<CK, 9, 24>(a, i, j, bi); break;                                        \
    }
                        if (a.um >= MAXLOOP) {
                            switch (role) {
                            case 8: a1_b0f<2, 18>(a, i, j, bb); break;
                            case 9: a1_b0f<19, 30>(a, i, j, bb); a1_b1f<2, 6>(a, i, j, bb); break;
                            case 10: a1_b1f<7, 22>(a, i, j, bb); break;
                            case 11: a1_b1f<23, 25>(a, i, j, bb); a1_i0f<3, 15>(a, i, j, bi); break;
                            case 12: a1_i0f<16, 25>(a, i, j, bi); a1_i1f<3, 8>(a, i, j, bi); break;
                            default: a1_i1f<9, 24>(a, i, j, bi); break;
                            }
                        } else { MIRP_OTH(true) }
#undef MIRP_OTH
                        const unsigned rb = a1_key(bb, -32768 - OTH_BIAS + au1);
                        const unsigned ri = a1_key(bi, -32768 - OTH_BIAS + mm1);
                        res = rb < ri ? rb : ri;
                    } else if (a.um >= MAXLOOP) {
                        res = role == 14 ? a1_small14f2(a, i, j, type, second) : a1_small15f(a, i, j, type);
#ifdef MIRP_ROWS14
                        const unsigned rg = role == 14 ? a1_generic<false, MIRP_ROWS14>(a, i, j, mmo) : a1_generic<false, MIRP_ROWS15>(a, i, j, mmo);
                        res = rg < res ? rg : res;
#endif
                    } else {
                        const int si1 = S[i + 1], sj1 = S[j - 1];
                        int ra, ca, rb2, cb2;
                        unsigned ka, kb2;
                        if (role == 14) {
                            a1_small_g<1, 1>(a, i, j, type, si1, sj1, ra, ca); a1_small_g<1, 2>(a, i, j, type, si1, sj1, rb2, cb2);
                            unsigned rbl = KEY_NONE;
                            a1_small<0, 1>(a, i, j, type, si1, sj1, rbl); a1_small<1, 0>(a, i, j, type, si1, sj1, rbl);
                            if (!second) res = rbl;          // the 1-bulges of a cell of the second diagonal are not final yet (phase B adds them)
                            ka = 1 << 5 | 1; kb2 = 1 << 5 | 2;
                        } else {
                            a1_small_g<2, 1>(a, i, j, type, si1, sj1, ra, ca); a1_small_g<2, 2>(a, i, j, type, si1, sj1, rb2, cb2);
                            a1_small<2, 3>(a, i, j, type, si1, sj1, res); a1_small<3, 2>(a, i, j, type, si1, sj1, res);
                            ka = 2 << 5 | 1; kb2 = 2 << 5 | 2;
                        }
                        if (ca < INF) { const unsigned kx = ((unsigned)(ra + ca + KEY_BIAS) << 10) | ka; res = kx < res ? kx : res; }
                        if (cb2 < INF) { const unsigned kx = ((unsigned)(rb2 + cb2 + KEY_BIAS) << 10) | kb2; res = kx < res ? kx : res; }
#ifdef MIRP_ROWS14
                        const unsigned rg = role == 14 ? a1_generic<true, MIRP_ROWS14>(a, i, j, mmo) : a1_generic<true, MIRP_ROWS15>(a, i, j, mmo);
                        res = rg < res ? rg : res;
#endif
                    }
                    if (act && res != KEY_NONE) atomicMin(&ck[i], res);
                }
            }
        };
        L2CLK(long long wB = 0; long long wI = 0; long long wS = 0; long long wW = 0; long long wt = 0;)
        auto phaseA = [&](const int e) {      // pair (e, e+1), e even
            const bool two = e + 1 <= D;
            // Half of the waves run the splits before the interior loops: the split loop loads the LDS pipe much more than the interior loops do,
            // so the two halves even out the LDS load of the interval (the phases are independent: both only feed phase B of the next interval).
            const bool swap_order = wave & 1;
            if (swap_order) { splits(e); if (two) splits(e + 1); }
            L2CLK(if (lane == 0) { const long long t = clock64(); if (swap_order) wS += t - wt; wt = t; })
            if (e >= 6) interior(e);
            L2CLK(if (lane == 0) { const long long t = clock64(); wI += t - wt; wt = t; })
            if (!swap_order) { splits(e); if (two) splits(e + 1); }
            L2CLK(if (lane == 0) { const long long t = clock64(); if (!swap_order) wS += t - wt; wt = t; })
        };

        // phase B of the pair (d, d+1), d even: the thread of column i finalises cells A = (i, i+d) and B = (i, i+d+1).
        auto phaseB = [&](const int d) {
            if (!bwave) return;
            const bool two = d + 1 <= D;
            const int ncA = n - d;                     // cells of diagonal d; diagonal d+1 has one less
            const int hp_uA = P->hairpinE[d - 1 < MIRP_HP_MAX ? d - 1 : MIRP_HP_MAX - 1];
            const int hp_uB = P->hairpinE[d < MIRP_HP_MAX ? d : MIRP_HP_MAX - 1];
            const int odA = tri_off(d, n), od1 = tri_off(d - 1, n), odB = odA + tri_len_any(d, n);
            unsigned* ckA = reinterpret_cast<unsigned*>(acc + (d & 3) * LCAP);
            unsigned* ckB = reinterpret_cast<unsigned*>(acc + ((d + 1) & 3) * LCAP);
            unsigned* ckA2 = reinterpret_cast<unsigned*>(acc + ((d + 2) & 3) * LCAP);     // targets of the pushed multiloop closings
            unsigned* ckB2 = reinterpret_cast<unsigned*>(acc + ((d + 3) & 3) * LCAP);
            int* mdA_p = acc + (4 + (d & 3)) * LCAP;
            int* mdB_p = acc + (4 + ((d + 1) & 3)) * LCAP;
            if (tid == 0) lcnt[((d + 6) >> 1) & 3] = 0;      // counter of the list the NEXT interval builds (its previous list was consumed two intervals ago)
            const bool hasA = bx < ncA;
            const bool hasB = two && !shadow && bx < ncA - 1;
            const bool own = hasA && !shadow;
            int mA = INF;                 // fML(d, i): also read by the lane below
            int lt0 = 0, oi0 = 0, lt1 = 0, oi1 = 0;
            // cell B's inputs, loaded with cell A's
            int mdB = INF, svB = -32768, typeB = 0, mmHB = 0, mmMsB = 0, dg5B = 0, dg3B = 0, mmIB = 0, tOB = 0, mmMcB = 0;
            unsigned kkB = KEY_NONE;
            int s_im1 = 0, s_jp2 = 0;
            const int i = bx + 1, j = i + d;
            if (hasA) {
                // ---- round 1: everything whose address depends on (i, d) only
                lds_vu8 Sv = (lds_vu8)S;
                s_im1 = Sv[i - 1];
                const int s_i = Sv[i], s_ip1 = Sv[i + 1], s_ip2 = Sv[i + 2], s_jm1 = Sv[j - 1], s_j = Sv[j], s_jp1 = Sv[j + 1];
                s_jp2 = Sv[j + 2 <= n + 1 ? j + 2 : n + 1];
                const int s_j3 = Sv[j + 3 <= n ? j + 3 : n], s_j4 = Sv[j + 4 <= n ? j + 4 : n], s_j5 = Sv[j + 5 <= n ? j + 5 : n];
                const int mdA = mdA_p[i];
                const unsigned kkA0 = ckA[i];
                if (hasB) { mdB = mdB_p[i]; kkB = ckB[i]; }
                int fa = 65535, fb = 65535;
                if (d > 4) { fa = fml[od1 + i]; fb = fml[od1 + i + 1]; }
                int svA = -32768;
                if (d == 4) { svA = spec[i]; svB = spec[nc + i]; } else if (d == 6) svB = spec[2 * nc + i];
                unsigned gS = 65535u, gSB = 65535u, g10 = 65535u;      // inner pairs of A's stack (= B's 0x1 bulge), B's stack, B's 1x0 bulge
                if (d >= 6) {
                    gS = cring[((d - 2) & 31) * CSTR + i + 1];
                    gSB = cring[((d - 1) & 31) * CSTR + i + 1];
                    g10 = cring[((d - 2) & 31) * CSTR + i + 2];
                }
                // paired cells of the pair (d+4, d+5) for the list the next interval consumes
                if (!shadow) {
                    if (d + 4 <= D && j + 4 <= n) { lt0 = pair_type(s_i, s_j4); oi0 = lt0 * 25 + s_ip1 * 5 + s_j3; }
                    if (d + 5 <= D && j + 5 <= n) { lt1 = pair_type(s_i, s_j5); oi1 = lt1 * 25 + s_ip1 * 5 + s_j4; }
                }
                // ---- pair types (arithmetic) and round 2: parameter tables; a type-0 row of a table is valid memory, its value is never used
                const int typeA = pair_type(s_i, s_j), rtA = rtype_of(typeA);
                typeB = pair_type(s_i, s_jp1);
                const int rtB = rtype_of(typeB);
                const int tin = rtype_of(pair_type(s_ip1, s_jm1));          // inner pair (i+1, j-1)
                const int tinB = rtype_of(pair_type(s_ip1, s_j));           // inner pair (i+1, j)
                const int t10 = rtype_of(pair_type(s_ip2, s_j));            // inner pair (i+2, j)
                const int tOA = pair_type(s_im1, s_jp1);                     // outer pair (i-1, j+1) of cell A
                tOB = pair_type(s_im1, s_jp2);                               // outer pair (i-1, j+2) of cell B
                const int tau = T.TerminalAU, mli = T.ML_intern, mlc = T.ML_closing, b1 = T.bulge[1];
                const int mmHA = T.mismatchH[typeA * 25 + s_ip1 * 5 + s_jm1];
                const int mmMsA = T.mismatchM[typeA * 25 + s_im1 * 5 + s_jp1];
                const int dg5A = T.dangle5[typeA * 5 + s_im1], dg3A = T.dangle3[typeA * 5 + s_jp1];
                const int mmIA = T.mismatchI[rtA * 25 + s_jp1 * 5 + s_im1];
                const int mmMcA = T.mismatchM[rtype_of(tOA) * 25 + s_j * 5 + s_i];
                const int m00 = T.mismatchI[tin * 25 + s_j * 5 + s_i];
                const int st00 = T.stack[typeA * 8 + tin];
                mmHB = T.mismatchH[typeB * 25 + s_ip1 * 5 + s_j];
                mmMsB = T.mismatchM[typeB * 25 + s_im1 * 5 + s_jp2];
                dg5B = T.dangle5[typeB * 5 + s_im1]; dg3B = T.dangle3[typeB * 5 + s_jp2];
                mmIB = T.mismatchI[rtB * 25 + s_jp2 * 5 + s_im1];
                mmMcB = T.mismatchM[rtype_of(tOB) * 25 + s_jp1 * 5 + s_i];
                const int mSB = T.mismatchI[tinB * 25 + s_jp1 * 5 + s_i];
                const int stB = T.stack[typeB * 8 + tinB], st01 = T.stack[typeB * 8 + tin];
                const int m10 = T.mismatchI[t10 * 25 + s_jp1 * 5 + s_ip1];
                const int st10 = T.stack[typeB * 8 + t10];
                // the candidates phase A left out, for cell B (its key is completed here, the cell itself follows the DPP exchange below)
                {
                    unsigned k = a1_small_key(gSB, mSB, stB, 0u); kkB = k < kkB ? k : kkB;
                    k = a1_small_key(gS, m00, b1 + st01, 0u << 5 | 1u); kkB = k < kkB ? k : kkB;
                    k = a1_small_key(g10, m10, b1 + st10, 1u << 5 | 0u); kkB = k < kkB ? k : kkB;
                }
                // ---- cell A
                const int auA = typeA > 2 ? tau : 0;
                int cv = INF, tb = 0;
                if (typeA) {
                    unsigned kk = kkA0;
                    { const unsigned k = a1_small_key(gS, m00, st00, 0u); kk = k < kk ? k : kk; }
                    const int cint = kk == KEY_NONE ? INF : (int)(kk >> 10) - KEY_BIAS;
                    int h;
                    if (svA != -32768) h = svA;
                    else if (d == 4) h = hp_uA + auA;
                    else h = hp_uA + mmHA;
                    cv = h < cint ? h : cint;
                    // the backtrack tests the hairpin first, then the interior loops in key order, then the multiloop (code 1023)
                    if (cint < INF && cint == cv && h != cv) { const int code = (int)(kk & 1023u); tb = code == 1023 ? 0 : code + 1; }
                }
                {
                    const int a = fa == 65535 ? INF : fa - FML_BIAS, b = fb == 65535 ? INF : fb - FML_BIAS;
                    mA = a < b ? a : b;
                }
                if (typeA) {
                    const int stem = mli + auA + ((i > 1 && j < n) ? mmMsA : (i > 1) ? dg5A : (j < n) ? dg3A : 0);
                    const int e = cv + stem;
                    mA = e < mA ? e : mA;
                }
                mA = mdA < mA ? mdA : mA;
                if ((cv < INF && (cv > FIN_LIMIT || cv < -FIN_LIMIT)) || (mA < INF && (mA > FML_MAX || mA < -FML_BIAS)) ||
                    (mdA < INF && (mdA > FIN_LIMIT || mdA < -FIN_LIMIT))) misc[1] = 1;
                if (own) {
                    const short c16 = cv >= INF ? (short)I16_INF : (short)cv;
                    const unsigned short m16 = mA >= INF ? (unsigned short)65535 : (unsigned short)(mA + FML_BIAS);
                    const unsigned short g16 = cv < INF ? (unsigned short)(cv + mmIA + 32768) : (unsigned short)65535;
                    cring[(d & 31) * CSTR + i] = g16;
                    if ((d & 31) == 0) cring[32 * CSTR + i] = g16;
                    carch[abase + 8 * d] = c16; tb_out[abase + 8 * d] = (unsigned short)tb;
                    fml[odA + i] = m16;
                    ckA[i] = KEY_NONE; mdA_p[i] = INF;
                    // multiloop closed by (i-1, j+1), two diagonals ahead
                    if (mdA < INF && tOA && i > 1 && j < n && d + 2 <= D) {
                        const int e = mdA + mlc + mli + (rtype_of(tOA) > 2 ? tau : 0) + mmMcA;
                        atomicMin(&ckA2[i - 1], ((unsigned)(e + KEY_BIAS) << 10) | 1023u);
                    }
                }
            }
            // fML(d, i+1) of the next column: lane + 1 (the wave's last lane shadows the next wave's first column)
            const int mAn = __builtin_amdgcn_update_dpp(INF, mA, 0x130, 0xf, 0xf, false);     // wave_shl:1
            if (hasB) {
                const int jB = j + 1;
                const int tau = T.TerminalAU, mli = T.ML_intern, mlc = T.ML_closing;
                const int auB = typeB > 2 ? tau : 0;
                int cv = INF, tb = 0;
                if (typeB) {
                    const int cint = kkB == KEY_NONE ? INF : (int)(kkB >> 10) - KEY_BIAS;
                    int h;
                    if (svB != -32768) h = svB;
                    else h = hp_uB + mmHB;                         // d + 1 >= 5: never a 3-loop
                    cv = h < cint ? h : cint;
                    if (cint < INF && cint == cv && h != cv) { const int code = (int)(kkB & 1023u); tb = code == 1023 ? 0 : code + 1; }
                }
                int mB = mA < mAn ? mA : mAn;
                if (typeB) {
                    const int stem = mli + auB + ((i > 1 && jB < n) ? mmMsB : (i > 1) ? dg5B : (jB < n) ? dg3B : 0);
                    const int e = cv + stem;
                    mB = e < mB ? e : mB;
                }
                mB = mdB < mB ? mdB : mB;
                if ((cv < INF && (cv > FIN_LIMIT || cv < -FIN_LIMIT)) || (mB < INF && (mB > FML_MAX || mB < -FML_BIAS)) ||
                    (mdB < INF && (mdB > FIN_LIMIT || mdB < -FIN_LIMIT))) misc[1] = 1;
                const short c16 = cv >= INF ? (short)I16_INF : (short)cv;
                const unsigned short m16 = mB >= INF ? (unsigned short)65535 : (unsigned short)(mB + FML_BIAS);
                const unsigned short g16 = cv < INF ? (unsigned short)(cv + mmIB + 32768) : (unsigned short)65535;
                cring[((d + 1) & 31) * CSTR + i] = g16;            // d + 1 is odd: never row 0, no mirror copy
                carch[abase + 8 * (d + 1)] = c16; tb_out[abase + 8 * (d + 1)] = (unsigned short)tb;
                fml[odB + i] = m16;
                ckB[i] = KEY_NONE; mdB_p[i] = INF;
                if (mdB < INF && tOB && i > 1 && jB < n && d + 3 <= D) {
                    const int e = mdB + mlc + mli + (rtype_of(tOB) > 2 ? tau : 0) + mmMcB;
                    atomicMin(&ckB2[i - 1], ((unsigned)(e + KEY_BIAS) << 10) | 1023u);
                }
            }
            if (d + 4 <= D) list_pair(d + 4, lt0, oi0, lt1, oi1);
        };

        if (D >= 4) { splits(4); if (D >= 5) splits(5); }      // (no split exists below diagonal 9: this only starts the carried split state)
        __syncthreads();
        for (int d = 4; d <= D; d += 2) {
            L2CLK(if (lane == 0) wt = clock64();)
            phaseB(d);
            L2CLK(if (lane == 0) { const long long t = clock64(); wB += t - wt; wt = t; })
            if (d + 2 <= D) phaseA(d + 2);
            __syncthreads();
            L2CLK(if (lane == 0) { const long long t = clock64(); wW += t - wt; wt = t; })
        }
        L2CLK(if (lane == 0) { atomicAdd(&g_l2_clk[wave * 4 + 0], (unsigned long long)wB); atomicAdd(&g_l2_clk[wave * 4 + 1], (unsigned long long)wI);
                               atomicAdd(&g_l2_clk[wave * 4 + 2], (unsigned long long)wS); atomicAdd(&g_l2_clk[wave * 4 + 3], (unsigned long long)wW); })
        const int overflow = misc[1];
        __syncthreads();
        if (overflow) {   // int16 range exceeded: hand the window to the generic kernel
            if (tid == 0) { unsigned int k = atomicAdd(fallback_count, 1u); fallback_list[k] = win_base + win; out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; win_state[win] = 0; }
        } else {
            // hand the tables to the epilogue kernel: c and the trace-back codes were archived on the fly, fML is copied out now into the same tiled
            // layout.  A wave takes whole row blocks; lane = diagonal, so the 8 rows of a row block on one diagonal are one 16-byte store and a
            // wave stores contiguous kilobytes; all of a row block's LDS reads are issued before the first store.
            if (D >= 4) {
                constexpr int NGD = (LDMAX + 1 - 4) / 64 + 1;
                for (int rb = wave; 8 * rb + 1 + 4 <= n; rb += LNW) {
                    const int dmax_rb = D < n - 1 - 8 * rb ? D : n - 1 - 8 * rb;      // the block's first row reaches furthest
                    short* dst = fml_out + rbt[rb] - 32;
                    unsigned v[NGD][8];
#pragma unroll
                    for (int g = 0; g < NGD; g++) {
                        const int d = 4 + 64 * g + lane;
                        const int o = tri_off(d <= dmax_rb ? d : 4, n) + 8 * rb + 1;
#pragma unroll
                        for (int k = 0; k < 8; k++) v[g][k] = fml[o + k];      // past a diagonal's end: some other cell, never read back
                    }
#pragma unroll
                    for (int g = 0; g < NGD; g++) {
                        const int d = 4 + 64 * g + lane;
                        if (d <= dmax_rb) {
                            uint4 w;
                            w.x = v[g][0] | v[g][1] << 16; w.y = v[g][2] | v[g][3] << 16; w.z = v[g][4] | v[g][5] << 16; w.w = v[g][6] | v[g][7] << 16;
                            *reinterpret_cast<uint4*>(dst + 8 * d) = w;
                        }
                    }
                }
            }
            if (tid == 0) win_state[win] = 1;
        }
        }   // window fits this kernel
        __syncthreads();
    }
}

size_t fold_lds2_bytes() { return lds2_layout().total; }

#ifdef MIRP_L2_CLOCKS
void fold_lds2_clocks_print() {
    unsigned long long h[16 * 4 + 8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_l2_clk), sizeof(h));
    for (int w = 0; w < 16; w++)
        std::fprintf(stderr, "[mirp fill2 clocks] wave %2d: phaseB=%llu interior=%llu splits=%llu barrier=%llu\n", w, h[4 * w], h[4 * w + 1], h[4 * w + 2], h[4 * w + 3]);
    unsigned long long z[16 * 4 + 8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_l2_clk), z, sizeof(z));
}
#endif

hipError_t launch_fold_lds2_fill(hipStream_t stream, int grid, const FoldParams* P, const unsigned char* seqs, const long long* offs, const int* lens, int n_work,
                                 int win_base, int span, short* slabs, size_t slab_shorts, int* win_state, unsigned int* work_counter, int* fallback_list,
                                 unsigned int* fallback_count, int* out_nlines, int* out_mfe, int* out_status) {
    const size_t lds = lds2_layout().total;
    hipError_t e = hipFuncSetAttribute((const void*)fold_lds2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fold_lds2_kernel, dim3(grid), dim3(LNT), lds, stream, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter,
                       fallback_list, fallback_count, out_nlines, out_mfe, out_status);
    return hipGetLastError();
}

}  // namespace mirp
