// LDS-resident batched L-bounded Zuker local fold for precursor windows (n <= 350, span <= 300):
// the production case PRECURSOR_LEN = 300 (/root/reference/miR_PREFeR.py:90, RNALfold -L at :3053).
//
// One window per workgroup (1024 threads = 16 wavefronts), one workgroup per CU:
//   * fML lives entirely in LDS as a triangular biased-uint16 table, diagonal-major: (d,i) -> off(d)+i, so
//     the two operands of a multiloop split are read at consecutive addresses by consecutive lanes;
//   * c keeps its last 32 anti-diagonals in an LDS ring (interior loops reach back MAXLOOP+2) and is
//     archived once, coalesced, as int16 to a per-window global slab for the exterior (f3) sweep
//     and the backtracks (fold_lds_epilogue_kernel);
//   * per anti-diagonal, phase A1 = interior-loop candidates with LANE = PAIRED CELL and WAVE = CANDIDATE GROUP:
//     every wave walks its own fixed share of the 496 (n1, n2) shapes for up to 64 paired cells at once, so the
//     loop shape is wave-uniform -- ring rows are scalar offsets, candidates are immediate offsets, size penalties
//     are scalar loads -- and one candidate costs one LDS read + add + min (generic shapes, ring stores
//     G0 = c + inner mismatch) or three LDS reads + four VALU ops (bulges and 1xn loops, through combined
//     per-window pair-code arrays).  Waves 0-7: generic rows (47 shapes each), 8-13: bulges / 1xn (18-19 each),
//     14: stack, 1-bulges, 2x3, 15: the 1x1/1x2/2x2 loops that read the big tables from global memory;
//   * phase A2 = multiloop splits (lane = cell, wave-uniform split point, scalar offsets); phase B = one thread per
//     cell finalises c, fML, DML, builds the ordered paired-cell list of diagonal d+2 (ballot compaction) and runs
//     in the same barrier interval as phase A of the next diagonal;
//   * INF needs no predicates: tables are biased unsigned 16-bit with INF = 65535, so any sum that involves
//     INF is >= 65535 and can never beat the 65535 start value of a running minimum;
//   * windows whose energies leave the 16-bit ranges are flagged and re-run by the generic kernel.
// No MFMA: integer min-plus DP with irregular table lookups.
#define MIRP_A1_CODES4 1
#include "fold_lds_common.h"
// In-place compaction of the split-candidate pool, period in diagonals per model (0 = never) and the number of 64-entry rounds a wave holds in registers.
#ifndef MIRP_CPERIOD0
#define MIRP_CPERIOD0 0       // default model: never (925 entries: a compaction's two barriers cost more than the dead lanes; measured 60.9 / 61.7 / 62.1 / 62.9 ms at never / 128 / 64 / 32)
#endif
#ifndef MIRP_CPERIOD1
#define MIRP_CPERIOD1 64      // vienna-1.8.5 (1,283 pair entries, 70 instructions per visit): 89.5 ms without, 87.3 / 85.9 / 85.6 / 85.0 at 8 / 16 / 32 / 64
#endif
#define CPOOL_ROUNDS 4

namespace mirp {

// MODEL 0: vienna-2.1.2 (Turner-2004, dangles 2).  MODEL 1: vienna-1.8.5 (Turner-1999 values in the same parameter layout, dangles 1: four-way
// dangle minima in the multiloop closing and the fML pair terms, fML also on the diagonal d = span; SURVEY.md Appendix B, d1 column).
//
// SPARSE (the product's first pass): the multiloop splits DML(i,j) = min_s fML(i,s-1) + fML(s,j) run over split CANDIDATES only.  With ML_BASE = 0
//     DML(i,j) = min( DML(i,j-1),  min over s in Cand(j), i+TURN+2 <= s <= j-TURN-1, of fML(i,s-1) + fML(s,j) ),
// Cand(j) = { s : fML(s,j) is realised STRICTLY by its pair term c(s,j) + MLstem } -- an entry that equals fML(s+1,j) is dominated by the split at
// s+1 (fML(i,s) <= fML(i,s-1)), one that equals fML(s,j-1) is a split of (i,j-1), one that equals DML(s,j) = fML(s,u-1) + fML(u,j) is dominated by
// the split at u.  The tables stay bit-identical (tests/tools/splitcand_gate.c checks the identity cell by cell on the CPU oracle's tables and
// counts: 925 candidates per benchmark window, 2.5 % of the dense loop's relaxations).  Phase B appends the candidates it finds to a pool
// {s-1, j, fML(s,j)} behind the window's fML triangle; phase A2 maps LANE = POOL ENTRY: the entry's column j holds exactly one cell of the
// diagonal at hand, (j-d, j), which it relaxes with one gather + one LDS atomic minimum.  A row's thread carries DML(i,j-1) in a register.
// A window whose pool overflows (tandem repeats) or whose length leaves no room for one is handed to the dense instantiation (second launch).
// MIRP_FILL_ATTR (dev builds): extra attributes of the fill kernel, e.g. -DMIRP_FILL_ATTR='__attribute__((amdgpu_num_vgpr(52)))' -- gfx950 doubles the
// request (unified register file), 52 caps the kernel at 104 VGPRs
#ifndef MIRP_FILL_ATTR
#define MIRP_FILL_ATTR
#endif
template <int MODEL, bool SPARSE>
__global__ void __launch_bounds__(LNT) MIRP_FILL_ATTR fold_lds_kernel(
    const FoldParams* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs, const int* __restrict__ win_lens,
    int n_work, int win_base, int span, short* __restrict__ slabs, size_t slab_shorts, int* __restrict__ win_state,
    unsigned int* __restrict__ work_counter, int* __restrict__ fallback_list,
    unsigned int* __restrict__ fallback_count, int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines, char* __restrict__ out_ss,
    int* __restrict__ out_nlines, int* __restrict__ out_mfe, int* __restrict__ out_status, int dbg_flags_arg, long long* __restrict__ dbg_cycles_arg,
    const int* __restrict__ todo_list, const unsigned int* __restrict__ todo_count, int* __restrict__ dense_list, unsigned int* __restrict__ dense_count) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr LdsLayout LY = lds_layout<MODEL, SPARSE>();
    // phase ablation flags and phase clocks exist in the diagnostics build only (make DIAG=1); the product kernel carries none of that code
#ifdef MIRP_DIAG
    const int dbg_flags = dbg_flags_arg;
    long long* const dbg_cycles = dbg_cycles_arg;
#elif defined(MIRP_LITE_CLOCKS)      // dev build: product code + per-wave busy / barrier-wait clocks only (two clock reads per wave and interval)
    constexpr int dbg_flags = 1 << 20;
    long long* const dbg_cycles = dbg_cycles_arg;
    (void)dbg_flags_arg;
#else
#ifndef MIRP_ABLATE
#define MIRP_ABLATE 0      // dev builds only (make ABLATE=<flags>): phases removed at compile time, to time product-like code without them
#endif
    constexpr int dbg_flags = MIRP_ABLATE;
    constexpr long long* dbg_cycles = nullptr;
    (void)dbg_flags_arg; (void)dbg_cycles_arg;
#endif
    constexpr int DMLR = MODEL ? 5 : 3;      // depth of the DML ring
    // outer-pair terms of a list entry (default model): two 10-bit signed fields above the 12 bits of i and type; the host checks that the tables fit
#define ENT_OUTER(mmo, mm1) ((unsigned)(((mmo) & 1023) | (((mm1) & 1023) << 10)))
    const int tau_s = __builtin_amdgcn_readfirstlane(P->TerminalAU);
    constexpr int GEN_WD = 5;                // default model: |n1 - n2| from which the asymmetry term of a generic loop is saturated (checked on the host: FoldParams::gen_wing_d)
    long long tA = 0, tB = 0, tS = 0, tE = 0, t0 = 0;   // diagnostic phase clocks (thread 0 only, dbg_cycles != nullptr)
    long long wB = 0, wA1 = 0, wA2 = 0, wW = 0, wt = 0; // per-wave: phase B, interior loops, multiloop splits, barrier wait (lane 0 of each wave)
    unsigned short* fml = (unsigned short*)(smem + LY.fml);   // biased uint16 (see FML_BIAS)
    unsigned short* cring = (unsigned short*)(smem + LY.aux);       // [32][CSTR] G0 + 32768 as uint16, 65535 = INF
    short* dmlring = (short*)(cring + CRING_ROWS * CSTR);           // [DMLR][LCAP] int16
    int* acc = (int*)(dmlring + DMLR * LCAP);                          // ckey[3 (diagonal % 3)][LCAP], then mdec[2 (diagonal parity)][LCAP] (SPARSE: [3 (diagonal % 3)])
    constexpr int NACC = SPARSE ? 6 : 5;
    auto mdec_of = [&](int d) -> int* { return acc + (3 + (SPARSE ? d % 3 : (d & 1))) * LCAP; };
    unsigned char* S = smem + LY.S;
    unsigned char* seq = smem + LY.seq;
    pax_t* pax = (pax_t*)(smem + LY.pax);
    unsigned char* qbr = smem + LY.qb2;
    unsigned char* code4 = SPARSE ? smem + LY.code4 : nullptr;      // [2][4][CODE_STR]: shifted byte copies of the q codes, then of the p codes (a1_codes4)
    // special-hairpin energies by start position (tri-, tetra-, hexaloops): only read on diagonals 4, 5 and 7, so they borrow the ring rows
    // of diagonals 29-31, which are first written on diagonal 29
    short* spec = (short*)(cring + 29 * CSTR);
    // [3][LSEG]: paired cells of diagonal d in buffer d % 3 (compact, unordered).  Default model: i | type << 9 | mmo << 12 | mm1 << 22, mmo / mm1 = the cell's
    // outer-pair terms mismatchI / mismatch1nI [type][S[i+1]][S[j-1]] as 10-bit signed values (ENT_OUTER below): phase B, which has the time, looks them up
    // when it builds the entry, and a block's prologue in phase A1 goes from the entry straight to arithmetic -- no dependent table read in front
    // of every block of every wave (timing build -DMIRP_X_NOOUTER: worth 2 ms).  (Before: oi = type * 25 + S[i+1] * 5 + S[j-1]
    // indexes the outer pair's mismatch tables: it rides in the entry so that phase A1 goes from the entry straight to the tables (reading the two
    // bases first was one more LDS round trip in front of every block of every wave)
    using list_t = unsigned;
    list_t* list = (list_t*)(smem + LY.list);
    LdsTables& T = *(LdsTables*)(smem + LY.tabs);
    int* misc = (int*)(smem + LY.misc);                             // 0: next window, 1: overflow flag, 2: candidate pool overflow, 3: candidates in the pool, 16..21: list lengths
    int* lcnt = misc + 16;                                          // [6]: entries in the list of diagonal d at d % 6
    int* rbt = misc + 48;                                           // [ARCH_RB]: row-block offsets of the window's archive slabs (arch_rowblk_off)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Issue priority (s_setprio, round 6): the four waves with the longest interior-loop roles (12-15: the four- and five-row generic jobs) yield to the
    // other twelve when a SIMD's arbiter has a choice -- every SIMD holds exactly one of them.  The interval's critical path runs through the waves that
    // own phase-B cells and then their own roles, not through the longest role: raising waves 12-15 instead costs 7 % (63.5 ms), a graded map 14 %,
    // switching the priority around phase B costs more than it gains; this map: 59.55 -> 59.12 ms (profiles/experiments/r6_fill_setprio.txt).
    if (wave < 12) __builtin_amdgcn_s_setprio(1);
    const int nc = CSTR;
    // Appends this thread's cell (i, pair type t; t = 0: none) to the paired-cell list of diagonal dd: ballot compaction inside the wave, one
    // LDS atomic per wave for its range.  The order of the ranges depends on which wave arrives first; nothing depends on the order of a
    // list, only on it staying fixed once built.
    auto list_append = [&](int dd, int i, int t, int oi) {
        const unsigned long long bal = __ballot(t != 0);
        if (bal) {      // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&lcnt[dd % 6], (int)__popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (t) list[(dd % 3) * LSEG + base + __popcll(bal & ((1ull << lane) - 1ull))] = (list_t)((unsigned)i | ((unsigned)t << 9) | ((unsigned)oi << 12));
        }
    };

    // ---- one-time: hot parameter tables into LDS
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

    const int n_todo = todo_count ? (int)*todo_count : n_work;      // second pass: the windows the sparse pass handed over
    if (todo_count && blockIdx.x == 0 && tid == 0 && n_todo) atomicAdd(const_cast<unsigned int*>(todo_count) + 2, (unsigned)n_todo);   // ctl[5]: running total for mirp_last_fold_dense
    for (;;) {
        if (tid == 0) misc[0] = (int)atomicAdd(work_counter, 1u);
        __syncthreads();
        const int wk = misc[0];
        __syncthreads();
        if (wk >= n_todo) break;
        const int win = todo_list ? todo_list[wk] : wk;
        const long long o0 = offs[win];
        const int n = win_lens ? win_lens[win] : (int)(offs[win + 1] - o0);
        // sparse splits: the candidate pool (u32 {s-1, j << 9} + u16 fML(s,j) per entry) takes what the window's triangle leaves of the fml region
        const int pool_off = SPARSE ? (int)lds_al(2u * (unsigned)(tri_off(((span < n - 1) ? span : n - 1) + 1, n > 5 ? n : 5) + 2)) : 0;
        // (vienna-1.8.5: 8-byte entries, one per PAIR -- see "pair pool" at splits_sparse185)
        // A model whose pool is compacted in place (compact_pool: every wave keeps its slice in CPOOL_ROUNDS x 64 registers) cannot hold more than
        // LNW x CPOOL_ROUNDS x 64 entries: short windows leave room for more behind their triangle, the capacity stops there and a larger pool takes
        // the overflow hand-off to the dense instantiation (misc[2]) like any other.
        const int pool_room = SPARSE ? ((((int)LY.fml_bytes - pool_off) / (MODEL ? 8 : 6)) & ~63) : 0;
        const int pool_cap = (SPARSE && (MODEL ? MIRP_CPERIOD1 : MIRP_CPERIOD0) > 0 && pool_room > LNW * CPOOL_ROUNDS * 64) ? LNW * CPOOL_ROUNDS * 64 : pool_room;
        unsigned* poolA = (unsigned*)(smem + LY.fml + pool_off);
        unsigned short* poolB = (unsigned short*)(poolA + (pool_cap > 0 ? pool_cap : 0));
        unsigned* poolB32 = poolA + (pool_cap > 0 ? pool_cap : 0);
        unsigned* pbits = (unsigned*)(misc + 48 + ARCH_RB);      // [4][11]: pair (p, q) of diagonal dd is in the pool: bit p of row dd & 3
        short* carch = slabs + (size_t)win * 3 * slab_shorts;      // per-window slab: c, fML and trace-back triangles (read by fold_lds_epilogue_kernel)
        short* fml_out = carch + slab_shorts;
        unsigned short* tb_out = reinterpret_cast<unsigned short*>(carch + 2 * slab_shorts);
        if (dbg_cycles && tid == 0) t0 = clock64();
        if (n < 1 || n > LCAP - 2) {   // wave-uniform: empty window, or too long for this kernel (-> generic kernel)
            if (tid == 0) {
                out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; win_state[win] = 0;
                if (n >= 1) { unsigned int k = atomicAdd(fallback_count, 1u); fallback_list[k] = win_base + win; }
            }
        } else if (SPARSE && pool_cap < POOL_MIN_CAP) {   // wave-uniform: no room for a candidate pool behind this window's triangle (-> dense instantiation)
            if (tid == 0) { out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; win_state[win] = 0; dense_list[atomicAdd(dense_count, 1u)] = win; }
        } else {
        const int D = (span - 1 < n - 1) ? span - 1 : n - 1;      // largest pair distance
        const int Dm = MODEL ? ((span < n - 1) ? span : n - 1) : D;   // last diagonal of the fill (vienna-1.8.5: fML exists at distance span, c does not)
        // ---- stage sequence, codes, special hairpins, pair-code arrays, triangular offsets
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
        for (int x = tid; x < DMLR * LCAP; x += LNT) dmlring[x] = (short)I16_INF;
        for (int x = tid; x < NACC * LCAP; x += LNT) acc[x] = x >= 3 * LCAP ? INF : (int)KEY_NONE;   // ckey x 3 | mdec x 2 (3)
        if (tid == 0) {
            misc[1] = 0; misc[2] = 0; misc[3] = 0;
            if constexpr (SPARSE && MODEL != 0) for (int x = 0; x < 44; x++) pbits[x] = 0;
            for (int x = 0; x < 6; x++) lcnt[x] = 0;
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
            if (MODEL) s4 = s4 == -32768 ? (short)0 : s4;     // vienna-1.8.5: a bonus added to the hairpin energy, not a total
            spec[x] = s3; spec[nc + x] = s4; spec[2 * nc + x] = s6;
            // combined pair codes (only interior positions are ever read: p - 1 >= 1, q + 1 <= n)
            if (x >= 1) {
                pax[x] = (pax_t)xt_pcode(S[x], x > 1 ? (int)S[x - 1] : 0);
                qbr[n + 1 - x] = (unsigned char)xt_qcode(S[x], x < n ? (int)S[x + 1] : 0);
            }
        }
        // paired-cell lists of the first three diagonals (list of diagonal d lives in buffer d % 3, its length in lcnt[d % 6])
        for (int dd = 4; dd <= 6 && dd <= D; dd++) {
            int t = 0, oi = 0;
            if (tid < n - dd) {
                t = pair_type(S[tid + 1], S[tid + 1 + dd]);
                const int x = t * 25 + S[tid + 2] * 5 + S[tid + dd];
                oi = ENT_OUTER((int)T.mismatchI[x], (int)T.mismatch1nI[x]);
            }
            list_append(dd, tid + 1, t, oi);
        }
        __syncthreads();
        // byte-shifted copies 1 - 3 of both pair-code arrays (copy 0 = the arrays themselves, a1_codes4); entries past the ends are never used as codes
        if constexpr (SPARSE) {
            for (int x = tid; x < 8 * CODE_STR; x += LNT) {
                const int which = x / (4 * CODE_STR), c = (x / CODE_STR) & 3, y = x % CODE_STR + c;
                if (c) code4[x] = y < CODE_STR ? (which ? (unsigned char)pax[y] : qbr[y]) : (unsigned char)0;
            }
            __syncthreads();
        }

        if (dbg_cycles && tid == 0) { long long t = clock64(); tS += t - t0; t0 = t; }
        // ---- anti-diagonal wavefront, software-pipelined: phase B of diagonal d (one thread per cell) runs in the same barrier
        // interval as phase A of diagonal d+1, which only needs c of diagonals <= d-1 and fML of diagonals <= d-3.
        // split loop state carried across diagonals (see splits below)
        int sp_ncpad = 0, sp_nsub = 0, sp_pair = 0, sp_sub = 0, sp_so1 = 0, sp_si1 = 0, sp_so2 = 0, sp_si2 = 0;
        int sp_snap = 0;          // sparse splits: pool size as read one interval ago (wave-uniform)
        int dml_carry = INF;      // sparse splits: DML(i, j-1) of this thread's row i = tid + 1 (phase B carries it from diagonal to diagonal)
#ifndef MIRP_PB_BASES
#define MIRP_PB_BASES 1
#endif
        // (MIRP_PB_BASES, round 6) phase B of the default model takes its bases out of two registers: the row's own three once per window, the far side's five as a window that slides by one
        // base per diagonal (a row's thread owns cell (i, i + d) in interval d) -- one byte read per cell and interval instead of eight
        int pb_si = 0;            // S[i-1] | S[i] << 3 | S[i+1] << 6
        int pb_sj = 0;            // S[j-1] | S[j] << 3 | S[j+1] << 6 | S[j+2] << 9 | S[j+3] << 12 of the cell of the coming phase B
        int a1_done = 0;      // phase A1: cells of the next diagonal's list already relaxed (wave-uniform)
        const int abase = tid < 8 * ARCH_RB ? rbt[tid >> 3] + (tid & 7) - 32 : 0;   // archive offset of (d, i = tid + 1) is abase + 8 d
        int a1_ncp = __builtin_amdgcn_readfirstlane(lcnt[0]);   // phase A1: length of the next diagonal's list (first: diagonal 6)
        int lc_pre = 0;           // list length of diagonal d+1 for phaseA(d), read at the top of the interval (see the main loop)
        bool lc_have = false;
        auto phaseA = [&](const int d) {
            const int ncell = n - d;
            unsigned* ckey = reinterpret_cast<unsigned*>(acc + MIRP_CK(d) * LCAP);   // best interior-loop candidate key per cell
            int* mdec = mdec_of(d);
            // phase A2: multiloop splits DML(i,j) = min_t fML(i, i+t) + fML(i+t+1, j).
            // The split point t is wave-uniform (scalar address arithmetic); every lane owns TWO consecutive cells (i, i+1), i odd.  Operand a
            // (diagonal t, cells i, i+1) is one aligned 32-bit word; operand b (diagonal d-t-1, cells i+t+1, i+t+2) is one aligned word for odd t
            // and straddles two words for even t (one v_alignbit).  The step between the splits of a wave is even, so that parity is
            // wave-uniform.  One packed saturating add and one packed min then relax both cells.
            // Sparse splits: lane = pool entry.  Every candidate (s, j) found up to diagonal d-5 relaxes the one cell of diagonal d in its column,
            // (i, j) with i = j - d: DML(i,j) <- fML(i, s-1) + fML(s, j).  An entry younger than that has its left operand on a diagonal t < 4
            // and is skipped by the same test that skips dead columns (i < 1).  The pool size is the one read an interval ago: phase B of
            // the current interval may have claimed entries it has not written yet.  The blocks of 64 entries go to the waves from the top
            // (waves 0-5 own phase B).
            auto splits_sparse = [&]() {
                const int lim = sp_snap;
                // pool size for the NEXT interval: an LDS read issued here and first looked at behind the loop.  (Through the LDS address space on
                // purpose: a volatile read through the generic pointer compiles to flat_load + s_waitcnt vmcnt(0), a stall in front of the loop.)
                int pnv;
                {
                    const unsigned pa = (unsigned)(size_t)(__attribute__((address_space(3))) int*)&misc[3];
                    asm volatile("ds_read_b32 %0, %1" : "=v"(pnv) : "v"(pa) : "memory");
                }
                if (!(dbg_flags & 2))
                for (int k = (LNW - 1 - wave) * 64 + lane; k < lim; k += LNT) {
                    const unsigned ea = poolA[k];
                    const unsigned vb = poolB[k];
                    const int s1 = (int)(ea & 511u), j = (int)(ea >> 9);
                    const int i = j - d, t = s1 - i;
                    const bool ok = i >= 1 && t >= TURN + 1;
                    const int o = 7 - 4 * n + (__mul24(t, 2 * n + 1 - t) >> 1) + ((t - 4 + (n & 1)) >> 1) + i;     // tri_off(t, n) + i
                    const unsigned sum = (unsigned)fml[ok ? o : 1] + vb;
                    if (ok && sum < 65535u) atomicMin(&mdec[i], (int)sum - 2 * FML_BIAS);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pnv) : : "memory");
                { const int pn = __builtin_amdgcn_readfirstlane(pnv); sp_snap = pn < pool_cap ? pn : pool_cap; }
            };
            // vienna-1.8.5 (dangles 1): the pool holds PAIRS.  A pair (p, q) reaches the multiloop through four cells -- (p,q) plain, (p-1,q) with
            // its 5' dangle, (p,q+1) with its 3' dangle, (p-1,q+1) with both -- so one entry {p, q, c(p,q) + MLintern, d5, d3} relaxes the two cells
            // of the diagonal in columns q and q+1, each over s = p and s = p-1: a term built from the pair is never below the true split value
            // (fML(s,j) is the minimum over its variants) and equals it for every strict candidate, whose realising pair phase B pooled
            // (tests/tools: gate185b -- 1,283 pairs per benchmark window against 3,142 candidate cells, identity checked cell by cell).
            auto splits_sparse185 = [&]() {
                const int lim = sp_snap;
                int pnv;
                {
                    const unsigned pa = (unsigned)(size_t)(__attribute__((address_space(3))) int*)&misc[3];
                    asm volatile("ds_read_b32 %0, %1" : "=v"(pnv) : "v"(pa) : "memory");
                }
                if (!(dbg_flags & 2))
                for (int k = (LNW - 1 - wave) * 64 + lane; k < lim; k += LNT) {
                    const unsigned lo = poolA[k], hi = poolB32[k];
                    const int p = (int)(lo & 511u), q = (int)((lo >> 9) & 511u);
                    const int valb = (int)(hi & 0xffffu), e5 = -(int)((hi >> 16) & 255u), e3 = -(int)(hi >> 24);
                    const int i0 = q - d, t0 = p - 1 - i0;
                    const int oA0 = 7 - 4 * n + (__mul24(t0, 2 * n + 1 - t0) >> 1) + ((t0 - 4 + (n & 1)) >> 1) + i0;      // tri_off(t0, n) + i0
                    const int oA1 = oA0 - tri_len_any(t0 - 1, n);                                                     // tri_off(t0 - 1, n) + i0
                    const int oB1 = oA1 - tri_len_any(t0 - 2, n) + 1;                                                 // tri_off(t0 - 2, n) + i0 + 1
                    const bool c0 = i0 >= 1, c1 = i0 >= 0 && q + 1 <= n;
                    const bool vA0 = c0 && t0 >= TURN + 1, vA1 = c0 && t0 - 1 >= TURN + 1, vB0 = c1 && t0 - 1 >= TURN + 1, vB1 = c1 && t0 - 2 >= TURN + 1;
                    const int a0 = fml[vA0 ? oA0 : 1], a1 = fml[vA1 ? oA1 : 1], b0 = fml[vB0 ? oA1 + 1 : 1], b1 = fml[vB1 ? oB1 : 1];
                    int best0 = INF, best1 = INF;
                    if (vA0 && a0 != 65535) best0 = a0 + valb;
                    if (vA1 && a1 != 65535) { const int v = a1 + valb + e5; best0 = v < best0 ? v : best0; }
                    if (vB0 && b0 != 65535) best1 = b0 + valb + e3;
                    if (vB1 && b1 != 65535) { const int v = b1 + valb + e5 + e3; best1 = v < best1 ? v : best1; }
                    if (best0 < INF) atomicMin(&mdec[i0], best0 - 2 * FML_BIAS);
                    if (best1 < INF) atomicMin(&mdec[i0 + 1], best1 - 2 * FML_BIAS);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pnv) : : "memory");
                { const int pn = __builtin_amdgcn_readfirstlane(pnv); sp_snap = pn < pool_cap ? pn : pool_cap; }
            };
            auto splits_dense = [&]() {
                const int npair = (ncell + 1) >> 1;
                const int ncpad = (npair + 63) & ~63;
                // The lane mapping (pair, sub) and the start offsets / first differences of the two operand walks depend on d only through the
                // diagonal of operand b, which moves up by one per diagonal: they are carried across diagonals in registers and advanced with
                // two scalar adds; everything is set up again only when the number of cell pairs crosses a multiple of 64 (the mapping changes).
                // Recomputing the closed forms (integer division by ncpad, four triangle offsets) on every diagonal cost more scalar and vector
                // instructions than the split loop itself.
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
                if (sub < nsub && !(dbg_flags & 2)) {
                    const int i = 2 * pair + 1;
                    // every split t in [4, d-5] is relaxed unconditionally: with the biased uint16 encoding a sum that involves an INF entry
                    // saturates at 65535 and any sum of two finite entries is <= 65534, so no per-lane range bookkeeping is needed.
                    // The byte offsets of the two operand diagonals advance by second-order recurrences (tri_off above):
                    //   o1(t) = off(t),  o2(t) = off(d-t-1) + t + 1   (minus one short for even t: the aligned word below the pair)
                    const int s1 = nsub;
                    int t = 4 + sub;
                    const int odd = t & 1;
                    int so1 = sp_so1, so2 = sp_so2, si1 = sp_si1, si2 = sp_si2;
                    const int sss = 2 * s1 * s1;
                    us2 bu = {65535, 65535};
                    // the two operand addresses run in VGPRs (LDS byte addresses of this lane's pair): per split one vector add each, and only the
                    // second-order terms of the recurrences stay on the scalar unit, which is the busiest pipe of this kernel
                    typedef const __attribute__((address_space(3))) unsigned* lds_cu32;
                    const unsigned fb0 = (unsigned)(size_t)(lds_cu32)reinterpret_cast<const unsigned*>(fml + i);
                    unsigned va = fb0 + (unsigned)so1, vb = fb0 + (unsigned)so2;
#define MIRP_SSTEP() do { va += (unsigned)si1; vb += (unsigned)si2; asm volatile("s_sub_i32 %0, %0, %2\n\ts_sub_i32 %1, %1, %2" : "+s"(si1), "+s"(si2) : "s"(sss) : "scc"); } while (0)
#ifdef MIRP_X_NOSPLITLDS          // timing experiment: the split loop's address arithmetic and packed min-plus without its LDS reads
#define MIRP_LDA() (va | 0x40004000u)
#define MIRP_LDB(o) ((vb + (o)) | 0x40004000u)
#else
#define MIRP_LDA() (*(lds_cu32)(va))
#define MIRP_LDB(o) (*(lds_cu32)(vb + (o)))
#endif
                    // K splits with all their reads in flight before the first use.  The tail of a wave's split range (up to 7 splits) goes through
                    // the 4-, 2- and 1-deep groups: at most three LDS round trips instead of one per split.
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
#ifdef MIRP_X_SPLITODD          // timing experiment: every wave takes the aligned (odd t) path: no third read, no v_alignbit
                    relax(std::true_type{});
#else
                    if (odd) relax(std::true_type{}); else relax(std::false_type{});
#endif
#undef MIRP_SSTEP
#undef MIRP_LDA
#undef MIRP_LDB
                    const unsigned r0 = bu[0], r1 = bu[1];
#ifdef MIRP_X_NOSPLITATOM      // timing experiment: plain stores instead of the split loop's two atomic minima
                    if (i <= ncell && r0 < 65535u) mdec[i] = (int)r0 - 2 * FML_BIAS;
                    if (i + 1 <= ncell && r1 < 65535u) mdec[i + 1] = (int)r1 - 2 * FML_BIAS;
#else
                    if (i <= ncell && r0 < 65535u) atomicMin(&mdec[i], (int)r0 - 2 * FML_BIAS);
                    if (i + 1 <= ncell && r1 < 65535u) atomicMin(&mdec[i + 1], (int)r1 - 2 * FML_BIAS);
#endif
                }
            };
            // Half of the waves run the splits before the interior loops: the split loop loads the LDS pipe much more than the interior loops do,
            // so the two halves even out the LDS load of the interval (the phases are independent: both only feed phase B of this diagonal).
            const bool swap_order = !SPARSE && (wave & 1) && !(dbg_flags & 2048);      // (the sparse splits are too short to matter: measured 0.5 ms better behind the interior loops)
            auto splits = [&]() { if constexpr (SPARSE && MODEL != 0) splits_sparse185(); else if constexpr (SPARSE) splits_sparse(); else splits_dense(); };
            if (swap_order) splits();
            if (dbg_cycles && lane == 0 && !(dbg_flags & (1 << 20))) wt = clock64();
            // phase A1: interior-loop candidates.  The c ring holds G0(p,q) = c(p,q) + mismatchI[rtype(pq)][S[q+1]][S[p-1]] (+ 32768).
            if (!(dbg_flags & (1 | 64)) && d >= 6 && d <= D) {
                const list_t* clist = list + (d % 3) * LSEG;
                // Lane fill: the blocks of 64 paired cells of diagonal d are topped up with the first cells of diagonal d+1.  All candidates of
                // a cell of d+1 except the stacked pair have their inner pair on diagonals <= d-2, which are final in this interval; the stacked
                // pair follows one interval later (`done` cells below).  Such a lane differs only in j = i + d + 1 and in its ring rows, which
                // are the rows after those of diagonal d (CRING_ROWS).  Only once every loop size is admissible (um = MAXLOOP for both).
                const bool mix = d - 2 - (TURN + 1) >= MAXLOOP && d + 1 <= D && !(dbg_flags & 256);
                // the length of this list is known since the previous interval (a1_ncp); the first block's entries are fetched before anything
                // else: every dependent LDS access in front of the shape code costs hundreds of cycles when the pipe is loaded
                const int ncp = a1_ncp;                    // = lcnt[d % 6], read one interval ago as ncp2: no LDS round trip in front of the first block
                const int done = a1_done;                  // leading cells of this diagonal's list that were relaxed in the previous interval
                const int rem = ncp - done;
                const int nblk = (rem + 63) >> 6;
                int ncp2;
                if (lc_have) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lc_pre) : : "memory"); ncp2 = __builtin_amdgcn_readfirstlane(lc_pre); }
                else ncp2 = __builtin_amdgcn_readfirstlane(lcnt[(d + 1) % 6]);
                a1_ncp = ncp2;
                const int room = nblk * 64 - rem;          // < 64: idle lanes of the last block
                const int take2 = mix ? (room < ncp2 ? room : ncp2) : 0;
                a1_done = take2;
                unsigned* ckey2 = reinterpret_cast<unsigned*>(acc + MIRP_CK(d + 1) * LCAP);
                unsigned aent = 0;                         // list entry of a lane that goes ahead in the last block (0: none)
                {
                    const int ka = lane - (64 - room);
                    if (ka >= 0 && ka < take2) aent = list[((d + 1) % 3) * LSEG + ka];
                }
#ifndef MIRP_ROLEMAP
#define MIRP_ROLEMAP 1
#endif
#define MIRP_XROWS15 9, 7
#define MIRP_XROWS8 8
#define MIRP_XROWS9 10
                // roles (0-7: generic rows, 8-13: bulges / 1xn, 14-15: small shapes), measured job costs (MIRP_FOLD_CLOCKS): generic 2-row < small
                // shapes < generic 4-row < bulges / 1xn.  Phase B of the previous diagonal runs on waves 0-5 (one thread per cell, wave 0 always,
                // wave 5 rarely), so those waves take the cheapest jobs.
                const int role = wave < 4 ? wave : wave < 6 ? wave + 10 : wave < 12 ? wave + 2 : wave - 8;
                A1 a;
                a.P = P; a.T = &T; a.S = S; a.cring = cring; a.pax = pax; a.qbr = qbr; a.code4 = code4; a.n = n;
                const bool slow = (dbg_flags & 8192) != 0;     // diagnostics build: the one-round-trip-per-candidate versions of the jobs
                {
                for (int blk = 0; blk < nblk; blk++) {
                    {   // re-materialise the wave-uniform loop parameters per block: keeps the admissibility tests and row offsets as plain
                        // scalar compares inside the block instead of dozens of hoisted masks (SGPR spills)
                        int r0 = d - 2, um = d - 2 - (TURN + 1) < MAXLOOP ? d - 2 - (TURN + 1) : MAXLOOP;
                        asm volatile("" : "+s"(r0), "+s"(um));
                        a.r0 = r0; a.um = um;
                        a.rowtab = P->ring_rowoff[r0 & 31];
                    }
                    const int k = blk * 64 + lane;
                    const bool own = k < rem, ahead = !own && aent != 0;       // k >= rem only happens in the last block
                    const bool act = own || ahead;
#ifdef MIRP_X_NOENT           // timing experiment: no list-entry read in front of a block
                    const unsigned ent = own ? (unsigned)(((done + k) * 3 + 1) % 300 + 1) | (1u << 9) : ahead ? aent : (1u | (1u << 9));
#else
                    const unsigned ent = own ? clist[done + k] : ahead ? aent : (1u | (1u << 9));   // idle lanes: harmless dummy cell
#endif
                    const int i = ent & 511, type = (ent >> 9) & 7, j = i + d + (ahead ? 1 : 0);
                    a.cring = cring + (ahead ? CSTR : 0);
                    unsigned* ck = ahead ? ckey2 : ckey;
                    unsigned res = KEY_NONE;
                    // the terms of the outer pair that turn a job's running minimum into the cell's key: fetched before the shape code, so that
                    // their two round trips (bases, then tables) overlap the job's own reads instead of following them
                    int au1 = 0, mmo = 0, mm1 = 0;
                    if (role < 14 || (MIRP_ROLEMAP && MODEL == 0)) {
                        au1 = type > 2 ? tau_s : 0;
                        mmo = ((int)(ent << 10)) >> 22; mm1 = ((int)ent) >> 22;          // the 10-bit signed fields of the entry
                    }
                    if (role < 8) {
                        if (!(dbg_flags & 4)) {
#define MIRP_GEN(CK)                                                                      \
    switch (role) {                                                                       \
    case 0: res = MIRP_A1G<CK MIRP_A1WD, 30, 23>(a, i, j, mmo); if (CK || slow) a1_i1<CK, 28, 29>(a, i, j, xi); else a1_i1f<28, 29>(a, i, j, xi); break;      \
    case 1: res = MIRP_A1G<CK MIRP_A1WD, 29, 24>(a, i, j, mmo); if (CK || slow) a1_i1<CK, 25, 27>(a, i, j, xi); else a1_i1f<25, 27>(a, i, j, xi); break;      \
    case 2: res = MIRP_A1G<CK MIRP_A1WD, 28, 25>(a, i, j, mmo); if (CK || slow) a1_i0<CK, 26, 29>(a, i, j, xi); else a1_i0f<26, 29>(a, i, j, xi); break;      \
    case 3: res = MIRP_A1G<CK MIRP_A1WD, 27, 26>(a, i, j, mmo); if (CK || slow) a1_b1<CK, 26, 30>(a, i, j, xb); else a1_b1f<26, 30>(a, i, j, xb); break;      \
    case 4: res = MIRP_A1G<CK MIRP_A1WD, MIRP_ROWS4>(a, i, j, mmo); break;                    \
    case 5: res = MIRP_A1G<CK MIRP_A1WD, MIRP_ROWS5>(a, i, j, mmo); break;                    \
    case 6: res = MIRP_A1G<CK MIRP_A1WD, MIRP_ROWS6>(a, i, j, mmo); break;                    \
    default: res = MIRP_A1G<CK MIRP_A1WD, MIRP_ROWS7>(a, i, j, mmo); break;               \
    }
// generic rows that ride on the waves of other jobs (MIRP_ROLEMAP 1): with the split loop sparse, the 4-5-row generic groups were the busiest waves of an
// interval (90 % busy against 56-65 % on the small-shape and bulge waves, profiles/tools/wave_busy.sh); a row can run anywhere -- the key carries the shape
#define MIRP_XGEN(CK)                                                                     \
    switch (role) {                                                                       \
    case 15: rx = MIRP_A1G<CK MIRP_A1WD, MIRP_XROWS15>(a, i, j, mmo); break;                         \
    case 8: rx = MIRP_A1G<CK MIRP_A1WD, MIRP_XROWS8>(a, i, j, mmo); break;                           \
    case 9: rx = MIRP_A1G<CK MIRP_A1WD, MIRP_XROWS9>(a, i, j, mmo); break;                           \
    default: break;                                                                       \
    }
                            // the 2-row generic groups run on the phase-B waves, which have slack left: they also take a few bulge / 1xn shapes
                            unsigned xb = KEY_INF, xi = KEY_INF;
                            // default model: the saturated-asymmetry candidates of a row go through one minimum (a1_gen_row_w)
                            if constexpr (MODEL == 0) {
#define MIRP_A1G a1_generic_w
#define MIRP_A1WD , GEN_WD
#if MIRP_ROLEMAP              // rows 7 - 10 ride on the small-shape and bulge waves (MIRP_XGEN below)
#define MIRP_ROWS4 22, 17, 12
#define MIRP_ROWS5 21, 18, 11
#define MIRP_ROWS6 20, 16, 13
#define MIRP_ROWS7 19, 15, 14, 6
#else
#define MIRP_ROWS4 22, 17, 12, 7
#define MIRP_ROWS5 21, 18, 11, 8
#define MIRP_ROWS6 20, 16, 13, 9
#define MIRP_ROWS7 19, 15, 14, 10, 6
#endif
                                if (a.um >= MAXLOOP) { MIRP_GEN(false) } else { MIRP_GEN(true) }
#undef MIRP_A1G
#undef MIRP_A1WD
#undef MIRP_ROWS4
#undef MIRP_ROWS5
#undef MIRP_ROWS6
#undef MIRP_ROWS7
                            } else {          // vienna-1.8.5 keeps all generic rows on roles 4 - 7 (the other map measured +0.6 ms there)
#define MIRP_A1G a1_generic
#define MIRP_A1WD
#define MIRP_ROWS4 22, 17, 12, 7
#define MIRP_ROWS5 21, 18, 11, 8
#define MIRP_ROWS6 20, 16, 13, 9
#define MIRP_ROWS7 19, 15, 14, 10, 6
                                if (a.um >= MAXLOOP) { MIRP_GEN(false) } else { MIRP_GEN(true) }
#undef MIRP_A1G
#undef MIRP_A1WD
#undef MIRP_ROWS4
#undef MIRP_ROWS5
#undef MIRP_ROWS6
#undef MIRP_ROWS7
                            }
                            if (role < 4) {
                                const unsigned rb = a1_key(xb, -32768 - OTH_BIAS + au1);
                                const unsigned ri = a1_key(xi, -32768 - OTH_BIAS + mm1);
                                res = rb < res ? rb : res;
                                res = ri < res ? ri : res;
                            }
#undef MIRP_GEN
                        }
                    } else if (role < 14) {
                        if (!(dbg_flags & 8)) {
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
                            if (a.um >= MAXLOOP && !slow) {
                                switch (role) {
                                case 8: a1_b0f<2, 18>(a, i, j, bb); break;
                                case 9: a1_b0f<19, 30>(a, i, j, bb); a1_b1f<2, 6>(a, i, j, bb); break;
                                case 10: a1_b1f<7, 22>(a, i, j, bb); break;
                                case 11: a1_b1f<23, 25>(a, i, j, bb); a1_i0f<3, 15>(a, i, j, bi); break;
                                case 12: a1_i0f<16, 25>(a, i, j, bi); a1_i1f<3, 8>(a, i, j, bi); break;
                                default: a1_i1f<9, 24>(a, i, j, bi); break;
                                }
                            } else if (a.um >= MAXLOOP) { MIRP_OTH(false) } else { MIRP_OTH(true) }
#undef MIRP_OTH
                            const unsigned rb = a1_key(bb, -32768 - OTH_BIAS + au1);
                            const unsigned ri = a1_key(bi, -32768 - OTH_BIAS + mm1);
                            res = rb < ri ? rb : ri;
                        }
                    } else if (!(dbg_flags & 32) && a.um >= MAXLOOP && !slow) {
#ifdef MIRP_X_NOSMALLGLOBAL     // timing experiment: the small-shape jobs without their global table loads
                        res = role == 14 ? a1_small14f(a, i, j, type, ahead, true) : a1_small15f(a, i, j, type, true);
#elif defined(MIRP_X_NOSMALL)        // timing experiment: no small-shape jobs at all
                        res = KEY_NONE;
#else
                        res = role == 14 ? a1_small14f(a, i, j, type, ahead, (dbg_flags & 16384) != 0) : a1_small15f(a, i, j, type, (dbg_flags & 16384) != 0);
#endif
                    } else if (!(dbg_flags & 32)) {
                        const int si1 = S[i + 1], sj1 = S[j - 1];
                        int ra, ca, rb2, cb2;
                        unsigned ka, kb2;
                        if (role == 14) {
                            a1_small_g<1, 1>(a, i, j, type, si1, sj1, ra, ca); a1_small_g<1, 2>(a, i, j, type, si1, sj1, rb2, cb2);
                            unsigned r00 = KEY_NONE;
                            a1_small<0, 0>(a, i, j, type, si1, sj1, r00); a1_small<0, 1>(a, i, j, type, si1, sj1, res); a1_small<1, 0>(a, i, j, type, si1, sj1, res);
                            if (!ahead) res = r00 < res ? r00 : res;          // the stacked pair of a lane of diagonal d+1 is not final yet
                            ka = 1 << 5 | 1; kb2 = 1 << 5 | 2;
                        } else {
                            a1_small_g<2, 1>(a, i, j, type, si1, sj1, ra, ca); a1_small_g<2, 2>(a, i, j, type, si1, sj1, rb2, cb2);
                            a1_small<2, 3>(a, i, j, type, si1, sj1, res); a1_small<3, 2>(a, i, j, type, si1, sj1, res);
                            ka = 2 << 5 | 1; kb2 = 2 << 5 | 2;
                        }
                        if (ca < INF) { const unsigned k = ((unsigned)(ra + ca + KEY_BIAS) << 10) | ka; res = k < res ? k : res; }
                        if (cb2 < INF) { const unsigned k = ((unsigned)(rb2 + cb2 + KEY_BIAS) << 10) | kb2; res = k < res ? k : res; }
                    }
#if MIRP_ROLEMAP
                    if constexpr (MODEL == 0) {
                        if (role >= 8 && !(dbg_flags & 4)) {
                            unsigned rx = KEY_NONE;
#define MIRP_A1G a1_generic_w
#define MIRP_A1WD , GEN_WD
                            if (a.um >= MAXLOOP) { MIRP_XGEN(false) } else { MIRP_XGEN(true) }
#undef MIRP_A1G
#undef MIRP_A1WD
                            res = rx < res ? rx : res;
                        }
                    }
#endif
#ifdef MIRP_X_NOATOM            // timing experiment: a plain store instead of the block's atomic minimum
                    if (act && res != KEY_NONE) ck[i] = res;
#else
                    if (act && res != KEY_NONE) atomicMin(&ck[i], res);
#endif
                }
                if (role == 14 && done > 0 && !(dbg_flags & 32)) {   // stacked pairs of the cells that went ahead in the previous interval (done <= 63)
                    const bool act = lane < done;
                    const unsigned ent = act ? clist[lane] : (1u | (1u << 9));
                    const int i = ent & 511, type = (ent >> 9) & 7, j = i + d;
                    int r0 = d - 2, um = MAXLOOP;
                    asm volatile("" : "+s"(r0), "+s"(um));
                    a.r0 = r0; a.um = um; a.cring = cring;
                    unsigned res = KEY_NONE;
                    a1_small<0, 0>(a, i, j, type, S[i + 1], S[j - 1], res);
                    if (act && res != KEY_NONE) atomicMin(&ckey[i], res);
                }
                }
                if (dbg_cycles && lane == 0 && wave == 9 && !(dbg_flags & (1 << 20))) {   // diagnostics: interior-loop time of one wave by number of blocks
                    const int b = nblk < 3 ? nblk : 3;
                    atomicAdd((unsigned long long*)&dbg_cycles[68 + b], (unsigned long long)(clock64() - wt));
                    atomicAdd((unsigned long long*)&dbg_cycles[72 + b], 1ull);
                }
            }
            if (dbg_cycles && lane == 0 && !(dbg_flags & (1 << 20))) { const long long t = clock64(); wA1 += t - wt; wt = t; }
            if (!swap_order) splits();
        };
        auto phaseB = [&](const int d) {
            const int ncell = n - d;
            unsigned* ckey = reinterpret_cast<unsigned*>(acc + MIRP_CK(d) * LCAP);
            int* mdec = mdec_of(d);
            int cand = 0; unsigned cent = 0, cval = 0;      // sparse splits: this cell as a split candidate
            int rp = 0, rq = 0, rval = 0, rtp = 0;          // vienna-1.8.5 candidate pass: the pair whose term realises fML of this cell, its plain term and type
            if constexpr (SPARSE && MODEL != 0) { if (tid < 11) pbits[((d + 1) & 3) * 11 + tid] = 0; }      // the row of diagonal d-3 serves diagonal d+1 from the next interval on
            const int hp_u = P->hairpinE[d - 1 < MIRP_HP_MAX ? d - 1 : MIRP_HP_MAX - 1];
            const int od = tri_off(d, n), od1 = tri_off(d - 1, n);     // scalar arithmetic instead of a table read on the cell's dependency chain
            const int x = tid;
            // paired-cell list of diagonal d+3 (phase A1 of this interval reads those of d+1 and d+2): the range is claimed first, the entry is
            // written at the end, so that the atomic's latency is covered by the cell work in between
            int lt = 0, lbase = 0, loi = 0;
            unsigned long long lbal = 0;
            if (d + 3 <= D && wave < 6) {
                if (x + 1 + d + 3 <= n) { lt = pair_type(S[x + 1], S[x + 1 + d + 3]); loi = lt * 25 + S[x + 2] * 5 + S[x + d + 3]; }
                lbal = __ballot(lt != 0);
                if (lbal && lane == 0) lbase = atomicAdd(&lcnt[(d + 3) % 6], (int)__popcll(lbal));
            }
            if (tid == 0) lcnt[(d + 4) % 6] = 0;   // the list of diagonal d-2 is dead: its counter serves diagonal d+4 in the next interval
            if (x < ncell) {
                const int i = x + 1, j = i + d;
                const int type = (MODEL && d > D) ? 0 : pair_type(S[i], S[j]);
                int cv = INF;
                int md = mdec[i];
                if constexpr (SPARSE) { md = dml_carry < md ? dml_carry : md; dml_carry = md; }      // DML(i,j) = min(DML(i,j-1), candidate splits)
                int tb = 0;          // trace-back code: 0 = hairpin / multiloop / unpaired, else 1 + (n1 << 5 | n2) of the interior loop the backtrack takes
                if (type) {
                    const unsigned kk = ckey[i];
                    const int cint = kk == KEY_NONE ? INF : (int)(kk >> 10) - KEY_BIAS;
                    cv = cint;
                    int h;
                    const int u = d - 1;
                    if (MODEL) {
                        h = hp_u + (u == 3 ? (type > 2 ? (int)T.TerminalAU : 0) : (int)T.mismatchH[type * 25 + S[i + 1] * 5 + S[j - 1]]);
                        if (u == 4) h += spec[nc + i];
                    } else {
                        int sv = -32768;
                        if (u == 4) sv = spec[nc + i]; else if (u == 6) sv = spec[2 * nc + i]; else if (u == 3) sv = spec[i];
                        if (sv != -32768) h = sv;
                        else if (u == 3) h = hp_u + (type > 2 ? T.TerminalAU : 0);
                        else h = hp_u + T.mismatchH[type * 25 + S[i + 1] * 5 + S[j - 1]];
                    }
                    cv = h < cv ? h : cv;
                    if (MODEL) {
                        // multiloop closed by (i,j), dangles 1: min over { DML(i+1,j-1), DML(i+2,j-1)+d3, DML(i+1,j-2)+d5, DML(i+2,j-2)+d3+d5 }
                        const int tt = rtype_of(type);
                        const int e3 = T.dangle3[tt * 5 + S[i + 1]], e5 = T.dangle5[tt * 5 + S[j - 1]];
                        int X = INF, v;
                        v = dmlring[((d + DMLR - 2) % DMLR) * LCAP + i + 1]; if (v != I16_INF) X = v;
                        v = dmlring[((d + DMLR - 3) % DMLR) * LCAP + i + 2]; if (v != I16_INF && v + e3 < X) X = v + e3;
                        v = dmlring[((d + DMLR - 3) % DMLR) * LCAP + i + 1]; if (v != I16_INF && v + e5 < X) X = v + e5;
                        v = dmlring[((d + DMLR - 4) % DMLR) * LCAP + i + 2]; if (v != I16_INF && v + e3 + e5 < X) X = v + e3 + e5;
                        if (X < INF) {
                            const int e = X + T.ML_closing + T.ML_intern + (type > 2 ? (int)T.TerminalAU : 0);
                            cv = e < cv ? e : cv;
                        }
                    } else {
                        int dml = dmlring[((d + DMLR - 2) % DMLR) * LCAP + i + 1];
                        if (dml != I16_INF) {
                            int e = dml + T.ML_closing + lds_mlstem(T, P, rtype_of(type), S[j - 1], S[i + 1]);
                            cv = e < cv ? e : cv;
                        }
                    }
                    // the backtrack tests the hairpin first, then the interior loops in key order, then the multiloop
                    if (cint < INF && cint == cv && h != cv) tb = (int)(kk & 1023u) + 1;
                }
                int m = INF;
                if (d > 4) {
                    int a = fml[od1 + i], b = fml[od1 + i + 1];
                    a = a == 65535 ? INF : a - FML_BIAS; b = b == 65535 ? INF : b - FML_BIAS;
                    m = a < b ? a : b;
                }
                const int mab = m;
                if (MODEL) {
                    // fML pair terms, dangles 1: (i,j) plain, (i+1,j) with a 5' dangle, (i,j-1) with a 3' dangle, (i+1,j-1) with both.  Plain c of the
                    // neighbouring cells comes out of the G0 ring (G0 = c + mismatchI of the pair seen as an inner pair).
                    const int mli = T.ML_intern, tau = T.TerminalAU;
                    if (type) { const int e = cv + mli + (type > 2 ? tau : 0); if (e < m) { m = e; rp = i; rq = j; rval = e; rtp = type; } }
                    auto plain = [&](int dd, int ii, int& tp) -> int {
                        tp = 0;
                        if (dd < 4) return INF;
                        const unsigned g = cring[(dd & 31) * CSTR + ii];
                        if (g == 65535u) return INF;
                        tp = pair_type(S[ii], S[ii + dd]);
                        return (int)g - 32768 - (int)T.mismatchI[rtype_of(tp) * 25 + S[ii + dd + 1] * 5 + S[ii - 1]];
                    };
                    int tp;
                    int cc = plain(d - 1, i + 1, tp);
                    if (cc < INF) { const int pl = cc + mli + (tp > 2 ? tau : 0), e = pl + T.dangle5[tp * 5 + S[i]]; if (e < m) { m = e; rp = i + 1; rq = j; rval = pl; rtp = tp; } }
                    cc = plain(d - 1, i, tp);
                    if (cc < INF) { const int pl = cc + mli + (tp > 2 ? tau : 0), e = pl + T.dangle3[tp * 5 + S[j]]; if (e < m) { m = e; rp = i; rq = j - 1; rval = pl; rtp = tp; } }
                    cc = plain(d - 2, i + 1, tp);
                    if (cc < INF) { const int pl = cc + mli + (tp > 2 ? tau : 0), e = pl + T.dangle5[tp * 5 + S[i]] + T.dangle3[tp * 5 + S[j]]; if (e < m) { m = e; rp = i + 1; rq = j - 1; rval = pl; rtp = tp; } }
                } else if (type) {
                    int e = cv + lds_mlstem(T, P, type, i > 1 ? (int)S[i - 1] : -1, j < n ? (int)S[j + 1] : -1); m = e < m ? e : m;
                }
                if constexpr (SPARSE) cand = m < mab && m < md;      // fML(i,j) strictly realised by a pair term: a split candidate of column j
                m = md < m ? md : m;
                if ((cv < INF && (cv > FIN_LIMIT || cv < -FIN_LIMIT)) || (m < INF && (m > FML_MAX || m < -FML_BIAS)) ||
                    (md < INF && (md > FIN_LIMIT || md < -FIN_LIMIT))) misc[1] = 1;
                const short c16 = cv >= INF ? (short)I16_INF : (short)cv;
                const unsigned short m16 = m >= INF ? (unsigned short)65535 : (unsigned short)(m + FML_BIAS);
                {   // G0 = c + mismatchI of (i,j) seen as the inner pair of a generic interior loop
                    unsigned short g16 = 65535;
                    if (cv < INF) g16 = (unsigned short)(cv + T.mismatchI[rtype_of(type) * 25 + S[j + 1] * 5 + S[i - 1]] + 32768);
                    cring[(d & 31) * CSTR + i] = g16;
                    if ((d & 31) == 0) cring[32 * CSTR + i] = g16;
                }
                carch[abase + 8 * d] = c16;
                tb_out[abase + 8 * d] = (unsigned short)tb;
                fml[od + i] = m16;
                dmlring[(d % DMLR) * LCAP + i] = md >= INF ? (short)I16_INF : (short)md;
                ckey[i] = KEY_NONE;
                if constexpr (SPARSE) { mdec_of(d + 2)[i] = INF; cval = m16; cent = (unsigned)(i - 1) | ((unsigned)j << 9); }
                else mdec[i] = INF;
                if constexpr (SPARSE && MODEL != 0) {
                    if (cand) {      // the realising pair goes to the pool once: the first of its (up to four) candidate cells claims its bit
                        const unsigned bit = 1u << (rp & 31);
                        const unsigned old = atomicOr(&pbits[((rq - rp) & 3) * 11 + (rp >> 5)], bit);
                        cand = (old & bit) ? 0 : 1;
                        cent = (unsigned)rp | ((unsigned)rq << 9);
                        cval = (unsigned)(rval + FML_BIAS) | ((unsigned)(-(int)T.dangle5[rtp * 5 + S[rp - 1]]) << 16) | ((unsigned)(-(int)T.dangle3[rtp * 5 + S[rq + 1]]) << 24);
                        if (rval + FML_BIAS < 0 || rval + FML_BIAS > 65534) misc[1] = 1;      // (cannot happen inside the fML range check above; kept as a guard)
                    }
                }
            }
            if constexpr (SPARSE) {
                const unsigned long long cbal = __ballot(cand != 0);
                if (cbal) {      // wave-uniform
                    int cbase = 0;
                    if (lane == (int)__builtin_ctzll(cbal)) cbase = atomicAdd(&misc[3], (int)__popcll(cbal));
                    const int at = __builtin_amdgcn_readlane(cbase, (int)__builtin_ctzll(cbal)) + (int)__popcll(cbal & ((1ull << lane) - 1ull));
                    if (cand) {
                        if (at < pool_cap) { poolA[at] = cent; if constexpr (MODEL != 0) poolB32[at] = cval; else poolB[at] = (unsigned short)cval; }
                        else misc[2] = 1;
                    }
                }
            }
            const int lb = __builtin_amdgcn_readfirstlane(lbase);      // lane 0 holds the claimed range
            if (lt) list[(d % 3) * LSEG + lb + __popcll(lbal & ((1ull << lane) - 1ull))] = (list_t)((unsigned)(x + 1) | ((unsigned)lt << 9) | (ENT_OUTER((int)T.mismatchI[loi], (int)T.mismatch1nI[loi]) << 12));
        };
        // Phase B of the default model, written as two rounds of loads and then arithmetic: every LDS read whose address depends on (i, j, d)
        // only is issued first (round 1), the parameter-table reads that need the pair type and the neighbouring bases follow together
        // (round 2), and nothing is read inside a branch.  The straightforward version (phaseB above, still the vienna-1.8.5 path) compiles
        // to a chain of a dozen read-wait pairs, which is what the waves that own cells spend their interval on.
        auto phaseB0 = [&](const int d) {
            const int ncell = n - d;
            unsigned* ckey = reinterpret_cast<unsigned*>(acc + MIRP_CK(d) * LCAP);
            int* mdec = mdec_of(d);
            int cbase = 0, cand = 0; unsigned cent = 0, cval = 0;      // sparse splits: this cell as a split candidate
            unsigned long long cbal = 0;
            const int hp_u = P->hairpinE[d - 1 < MIRP_HP_MAX ? d - 1 : MIRP_HP_MAX - 1];
            const int od = tri_off(d, n), od1 = tri_off(d - 1, n);
            const int x = tid;
            int lt = 0, lbase = 0, loi = 0, ent_terms = 0;
            unsigned long long lbal = 0;
            const bool do_list = d + 3 <= D && !(dbg_flags & 32768);
            if (tid == 0) lcnt[(d + 4) % 6] = 0;
            if (x < ncell && !(dbg_flags & 131072)) {
                const int i = x + 1, j = i + d, u = d - 1;
                // ---- round 1
                lds_vu8 Sv = (lds_vu8)S;
#if MIRP_PB_BASES
                const int s_im1 = pb_si & 7, s_i = (pb_si >> 3) & 7, s_ip1 = (pb_si >> 6) & 7, s_jm1 = pb_sj & 7, s_j = (pb_sj >> 3) & 7, s_jp1 = (pb_sj >> 6) & 7;
                const int s_j2 = (pb_sj >> 9) & 7, s_j3 = (pb_sj >> 12) & 7;
                pb_sj = (pb_sj >> 3) | ((int)Sv[j + 4] << 12);      // (S holds LCAP + 8 bytes: in range for every j <= n)
#else
                const int s_im1 = Sv[i - 1], s_i = Sv[i], s_ip1 = Sv[i + 1], s_jm1 = Sv[j - 1], s_j = Sv[j], s_jp1 = Sv[j + 1];
                const int s_j3 = Sv[j + 3 <= n ? j + 3 : n];          // far end of cell (i, j+3): the paired-cell list of diagonal d+3
                const int s_j2 = Sv[j + 2 <= n ? j + 2 : n];          // its 5' neighbour, for the entry's outer-pair table index
#endif
                int md = mdec[i];
                if constexpr (SPARSE) { md = dml_carry < md ? dml_carry : md; dml_carry = md; }      // DML(i,j) = min(DML(i,j-1), candidate splits)
                const unsigned kk = ckey[i];
                const int dmlv = dmlring[((d + DMLR - 2) % DMLR) * LCAP + i + 1];
                int fa = 65535, fb = 65535;
                if (d > 4) { fa = fml[od1 + i]; fb = fml[od1 + i + 1]; }
                int sv = -32768;
                if (u == 4) sv = spec[nc + i]; else if (u == 6) sv = spec[2 * nc + i]; else if (u == 3) sv = spec[i];
                // the list range is claimed here, between the two rounds: the atomic's return is first looked at after the cell's stores, so its
                // round trip is not in front of anything (issued ahead of round 1 it put two LDS round trips in front of the whole chain)
                if (do_list) {
                    if (j + 3 <= n) { lt = pair_type(s_i, s_j3); loi = lt * 25 + s_ip1 * 5 + s_j2; }          // index of the entry's outer-pair terms: read in round 2
                    lbal = __ballot(lt != 0);
                    // hand-issued: the compiler's atomic optimizer wraps atomicAdd in a wave reduction whose readfirstlane waits right here
                    if (lbal && lane == 0) {
                        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) int*)&lcnt[(d + 3) % 6];
                        asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(lbase) : "v"(la), "v"((int)__popcll(lbal)) : "memory");
                    }
                }
                // ---- pair type (arithmetic) and round 2: parameter tables; a type-0 row of a table is valid memory, its value is never used
                const int type = pair_type(s_i, s_j);
                const int rt = rtype_of(type);
                const int tau = T.TerminalAU, mli = T.ML_intern, mlc = T.ML_closing;
#ifdef MIRP_X_NOPBTAB          // timing experiment: phase B without its second round of LDS reads (what precomputing the sequence-only terms would remove)
                const int mmH = -(type * 7 + s_ip1), mmMc = -(rt * 3 + s_jm1), mmMs = -(type + s_im1 * 5), dg5 = -s_im1, dg3 = -s_jp1, mmI = rt * 10 - s_jp1;
#else
                const int mmH = T.mismatchH[type * 25 + s_ip1 * 5 + s_jm1];
                const int mmMc = T.mismatchM[rt * 25 + s_jm1 * 5 + s_ip1];
                const int mmMs = T.mismatchM[type * 25 + s_im1 * 5 + s_jp1];
                const int dg5 = T.dangle5[type * 5 + s_im1], dg3 = T.dangle3[type * 5 + s_jp1];
                const int mmI = T.mismatchI[rt * 25 + s_jp1 * 5 + s_im1];
#endif
                ent_terms = ENT_OUTER((int)T.mismatchI[loi], (int)T.mismatch1nI[loi]);          // (a type-0 row is valid memory, the value is not used)
                // ---- arithmetic
                const int au = type > 2 ? tau : 0;
                int cv = INF, tb = 0;
                if (type) {
                    const int cint = kk == KEY_NONE ? INF : (int)(kk >> 10) - KEY_BIAS;
                    int h;
                    if (sv != -32768) h = sv;
                    else if (u == 3) h = hp_u + au;
                    else h = hp_u + mmH;
                    cv = h < cint ? h : cint;
                    if (dmlv != I16_INF) {
                        const int e = dmlv + mlc + mli + (rt > 2 ? tau : 0) + mmMc;
                        cv = e < cv ? e : cv;
                    }
                    if (cint < INF && cint == cv && h != cv) tb = (int)(kk & 1023u) + 1;
                }
                int m = INF;
                {
                    const int a = fa == 65535 ? INF : fa - FML_BIAS, b = fb == 65535 ? INF : fb - FML_BIAS;
                    m = a < b ? a : b;
                }
                if (type) {
                    // lds_mlstem(type, i > 1 ? S[i-1] : -1, j < n ? S[j+1] : -1)
                    const int stem = mli + au + ((i > 1 && j < n) ? mmMs : (i > 1) ? dg5 : (j < n) ? dg3 : 0);
                    const int e = cv + stem;
                    if constexpr (SPARSE) cand = cv < INF && e < m && e < md;      // fML(i,j) strictly realised by the pair term: a split candidate of column j
                    m = e < m ? e : m;
                }
                m = md < m ? md : m;
                if constexpr (SPARSE) {
                    cbal = __ballot(cand != 0);
                    if (cbal && lane == 0) {      // claim the wave's pool range (hand-issued like the list claim above; its return is first needed after the stores)
                        const unsigned pa = (unsigned)(size_t)(__attribute__((address_space(3))) int*)&misc[3];
                        asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(cbase) : "v"(pa), "v"((int)__popcll(cbal)) : "memory");
                    }
                    cent = (unsigned)(i - 1) | ((unsigned)j << 9);
                }
#ifndef MIRP_TIMING_ONLY          // (timing experiments compute garbage on purpose: no hand-over to the generic kernel)
                if ((cv < INF && (cv > FIN_LIMIT || cv < -FIN_LIMIT)) || (m < INF && (m > FML_MAX || m < -FML_BIAS)) ||
                    (md < INF && (md > FIN_LIMIT || md < -FIN_LIMIT))) misc[1] = 1;
#endif
                const short c16 = cv >= INF ? (short)I16_INF : (short)cv;
                const unsigned short m16 = m >= INF ? (unsigned short)65535 : (unsigned short)(m + FML_BIAS);
                const unsigned short g16 = cv < INF ? (unsigned short)(cv + mmI + 32768) : (unsigned short)65535;
                cring[(d & 31) * CSTR + i] = g16;
                if ((d & 31) == 0) cring[32 * CSTR + i] = g16;
#ifndef MIRP_X_NOHBM           // (timing experiment: no archive stores)
                if (!(dbg_flags & 65536)) { carch[abase + 8 * d] = c16; tb_out[abase + 8 * d] = (unsigned short)tb; }
#endif
                fml[od + i] = m16;
                dmlring[(d % DMLR) * LCAP + i] = md >= INF ? (short)I16_INF : (short)md;
                ckey[i] = KEY_NONE;
                if constexpr (SPARSE) { mdec_of(d + 2)[i] = INF; cval = m16; }      // the buffer of diagonal d-1 is dead: it serves diagonal d+2 from the next interval on
                else mdec[i] = INF;
            }
            if constexpr (SPARSE) {
                if (cbal) {      // wave-uniform
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cbase) : : "memory");
                    const int at = __builtin_amdgcn_readfirstlane(cbase) + (int)__popcll(cbal & ((1ull << lane) - 1ull));
                    if (cand) {
                        if (at < pool_cap) { poolA[at] = cent; poolB[at] = (unsigned short)cval; }
                        else misc[2] = 1;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lbase) : : "memory");   // the atomic's return is first needed here
            const int lb = __builtin_amdgcn_readfirstlane(lbase);
            if (lt) list[(d % 3) * LSEG + lb + __popcll(lbal & ((1ull << lane) - 1ull))] = (list_t)((unsigned)(x + 1) | ((unsigned)lt << 9) | ((unsigned)ent_terms << 12));
        };
        // Phase B of the vienna-1.8.5 model in the same two-round form (round 4): the straightforward phaseB above reads inside branches and lambdas --
        // a dozen read-wait pairs per cell -- and cost the model 22 ms against the default model's phase B.  Same arithmetic, same order of the
        // minima (the first of equal terms wins wherever the order matters: the realising pair of the candidate pass).
        auto phaseB1 = [&](const int d) {
            const int ncell = n - d;
            unsigned* ckey = reinterpret_cast<unsigned*>(acc + MIRP_CK(d) * LCAP);
            int* mdec = mdec_of(d);
            int cand = 0; unsigned cent = 0, cval = 0;
            if constexpr (SPARSE) { if (tid < 11) pbits[((d + 1) & 3) * 11 + tid] = 0; }
            const int hp_u = P->hairpinE[d - 1 < MIRP_HP_MAX ? d - 1 : MIRP_HP_MAX - 1];
            const int od = tri_off(d, n), od1 = tri_off(d - 1, n);
            const int x = tid;
            int lt = 0, lbase = 0, loi = 0, ent_terms = 0;
            unsigned long long lbal = 0;
            const bool do_list = d + 3 <= D;
            const bool has1 = d - 1 >= 4, has2 = d - 2 >= 4;      // the ring rows of diagonals d-1 / d-2 hold cells of this window
            if (tid == 0) lcnt[(d + 4) % 6] = 0;
            if (x < ncell) {
                const int i = x + 1, j = i + d, u = d - 1;
                // ---- round 1: everything addressed by (i, j, d) alone
                lds_vu8 Sv = (lds_vu8)S;
#if MIRP_PB_BASES
                const int s_im1 = pb_si & 7, s_i = (pb_si >> 3) & 7, s_ip1 = (pb_si >> 6) & 7, s_jm1 = pb_sj & 7, s_j = (pb_sj >> 3) & 7, s_jp1 = (pb_sj >> 6) & 7;
                const int s_j2 = (pb_sj >> 9) & 7, s_j3 = (pb_sj >> 12) & 7;
                pb_sj = (pb_sj >> 3) | ((int)Sv[j + 4] << 12);
#else
                const int s_im1 = Sv[i - 1], s_i = Sv[i], s_ip1 = Sv[i + 1], s_jm1 = Sv[j - 1], s_j = Sv[j], s_jp1 = Sv[j + 1];
                const int s_j3 = Sv[j + 3 <= n ? j + 3 : n], s_j2 = Sv[j + 2 <= n ? j + 2 : n];
#endif
                int md = mdec[i];
                if constexpr (SPARSE) { md = dml_carry < md ? dml_carry : md; dml_carry = md; }
                const unsigned kk = ckey[i];
                const int q11 = dmlring[((d + DMLR - 2) % DMLR) * LCAP + i + 1], q21 = dmlring[((d + DMLR - 3) % DMLR) * LCAP + i + 2];
                const int q12 = dmlring[((d + DMLR - 3) % DMLR) * LCAP + i + 1], q22 = dmlring[((d + DMLR - 4) % DMLR) * LCAP + i + 2];
                int fa = 65535, fb = 65535;
                if (d > 4) { fa = fml[od1 + i]; fb = fml[od1 + i + 1]; }
                const int tetra = u == 4 ? (int)spec[nc + i] : 0;
                const unsigned g1 = has1 ? (unsigned)cring[((d - 1) & 31) * CSTR + i + 1] : 65535u;      // G0 of (i+1, j)
                const unsigned g2 = has1 ? (unsigned)cring[((d - 1) & 31) * CSTR + i] : 65535u;          // G0 of (i, j-1)
                const unsigned g3 = has2 ? (unsigned)cring[((d - 2) & 31) * CSTR + i + 1] : 65535u;      // G0 of (i+1, j-1)
                if (do_list) {
                    if (j + 3 <= n) { lt = pair_type(s_i, s_j3); loi = lt * 25 + s_ip1 * 5 + s_j2; }
                    lbal = __ballot(lt != 0);
                    if (lbal && lane == 0) {
                        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) int*)&lcnt[(d + 3) % 6];
                        asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(lbase) : "v"(la), "v"((int)__popcll(lbal)) : "memory");
                    }
                }
                // ---- pair types (arithmetic) and round 2: parameter tables (a type-0 row is valid memory, its value is never used)
                const int type = d > D ? 0 : pair_type(s_i, s_j);
                const int rt = rtype_of(type);
                const int tp1 = pair_type(s_ip1, s_j), tp2 = pair_type(s_i, s_jm1), tp3 = pair_type(s_ip1, s_jm1);
                const int tau = T.TerminalAU, mli = T.ML_intern, mlc = T.ML_closing;
                const int mmH = T.mismatchH[type * 25 + s_ip1 * 5 + s_jm1];
                const int e3 = T.dangle3[rt * 5 + s_ip1], e5 = T.dangle5[rt * 5 + s_jm1];
                const int mmI = T.mismatchI[rt * 25 + s_jp1 * 5 + s_im1];
                const int mi1 = T.mismatchI[rtype_of(tp1) * 25 + s_jp1 * 5 + s_i];
                const int mi2 = T.mismatchI[rtype_of(tp2) * 25 + s_j * 5 + s_im1];
                const int mi3 = T.mismatchI[rtype_of(tp3) * 25 + s_j * 5 + s_i];
                const int d5_1 = T.dangle5[tp1 * 5 + s_i], d3_2 = T.dangle3[tp2 * 5 + s_j], d5_3 = T.dangle5[tp3 * 5 + s_i], d3_3 = T.dangle3[tp3 * 5 + s_j];
                ent_terms = ENT_OUTER((int)T.mismatchI[loi], (int)T.mismatch1nI[loi]);
                // ---- arithmetic
                const int au = type > 2 ? tau : 0;
                int cv = INF, tb = 0;
                if (type) {
                    const int cint = kk == KEY_NONE ? INF : (int)(kk >> 10) - KEY_BIAS;
                    const int h = hp_u + (u == 3 ? au : mmH) + tetra;
                    cv = h < cint ? h : cint;
                    // multiloop closed by (i,j), dangles 1: min over { DML(i+1,j-1), DML(i+2,j-1)+d3, DML(i+1,j-2)+d5, DML(i+2,j-2)+d3+d5 }
                    int X = INF;
                    if (q11 != I16_INF) X = q11;
                    if (q21 != I16_INF && q21 + e3 < X) X = q21 + e3;
                    if (q12 != I16_INF && q12 + e5 < X) X = q12 + e5;
                    if (q22 != I16_INF && q22 + e3 + e5 < X) X = q22 + e3 + e5;
                    if (X < INF) { const int e = X + mlc + mli + au; cv = e < cv ? e : cv; }
                    if (cint < INF && cint == cv && h != cv) tb = (int)(kk & 1023u) + 1;
                }
                int m = INF;
                {
                    const int a = fa == 65535 ? INF : fa - FML_BIAS, b = fb == 65535 ? INF : fb - FML_BIAS;
                    m = a < b ? a : b;
                }
                const int mab = m;
                int rp = 0, rq = 0, rval = 0, rtp = 0;
                if (type) { const int e = cv + mli + au; if (e < m) { m = e; rp = i; rq = j; rval = e; rtp = type; } }
                if (g1 != 65535u) { const int pl = (int)g1 - 32768 - mi1 + mli + (tp1 > 2 ? tau : 0), e = pl + d5_1; if (e < m) { m = e; rp = i + 1; rq = j; rval = pl; rtp = tp1; } }
                if (g2 != 65535u) { const int pl = (int)g2 - 32768 - mi2 + mli + (tp2 > 2 ? tau : 0), e = pl + d3_2; if (e < m) { m = e; rp = i; rq = j - 1; rval = pl; rtp = tp2; } }
                if (g3 != 65535u) { const int pl = (int)g3 - 32768 - mi3 + mli + (tp3 > 2 ? tau : 0), e = pl + d5_3 + d3_3; if (e < m) { m = e; rp = i + 1; rq = j - 1; rval = pl; rtp = tp3; } }
                if constexpr (SPARSE) cand = m < mab && m < md;
                m = md < m ? md : m;
                if ((cv < INF && (cv > FIN_LIMIT || cv < -FIN_LIMIT)) || (m < INF && (m > FML_MAX || m < -FML_BIAS)) ||
                    (md < INF && (md > FIN_LIMIT || md < -FIN_LIMIT))) misc[1] = 1;
                const short c16 = cv >= INF ? (short)I16_INF : (short)cv;
                const unsigned short m16 = m >= INF ? (unsigned short)65535 : (unsigned short)(m + FML_BIAS);
                const unsigned short g16 = cv < INF ? (unsigned short)(cv + mmI + 32768) : (unsigned short)65535;
                cring[(d & 31) * CSTR + i] = g16;
                if ((d & 31) == 0) cring[32 * CSTR + i] = g16;
                carch[abase + 8 * d] = c16;
                tb_out[abase + 8 * d] = (unsigned short)tb;
                fml[od + i] = m16;
                dmlring[(d % DMLR) * LCAP + i] = md >= INF ? (short)I16_INF : (short)md;
                ckey[i] = KEY_NONE;
                if constexpr (SPARSE) {
                    mdec_of(d + 2)[i] = INF;
                    if (cand) {      // the realising pair goes to the pool once: the first of its (up to four) candidate cells claims its bit
                        const unsigned bit = 1u << (rp & 31);
                        const unsigned old = atomicOr(&pbits[((rq - rp) & 3) * 11 + (rp >> 5)], bit);
                        cand = (old & bit) ? 0 : 1;
                        cent = (unsigned)rp | ((unsigned)rq << 9);
                        cval = (unsigned)(rval + FML_BIAS) | ((unsigned)(-(int)T.dangle5[rtp * 5 + Sv[rp - 1]]) << 16) | ((unsigned)(-(int)T.dangle3[rtp * 5 + Sv[rq + 1]]) << 24);
                        if (rval + FML_BIAS < 0 || rval + FML_BIAS > 65534) misc[1] = 1;
                    }
                } else mdec[i] = INF;
            }
            if constexpr (SPARSE) {
                const unsigned long long cbal = __ballot(cand != 0);
                if (cbal) {      // wave-uniform
                    int cbase = 0;
                    if (lane == (int)__builtin_ctzll(cbal)) cbase = atomicAdd(&misc[3], (int)__popcll(cbal));
                    const int at = __builtin_amdgcn_readlane(cbase, (int)__builtin_ctzll(cbal)) + (int)__popcll(cbal & ((1ull << lane) - 1ull));
                    if (cand) {
                        if (at < pool_cap) { poolA[at] = cent; poolB32[at] = cval; }
                        else misc[2] = 1;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lbase) : : "memory");
            const int lb = __builtin_amdgcn_readfirstlane(lbase);
            if (lt) list[(d % 3) * LSEG + lb + __popcll(lbal & ((1ull << lane) - 1ull))] = (list_t)((unsigned)(x + 1) | ((unsigned)lt << 9) | ((unsigned)ent_terms << 12));
        };
        // Candidate pool: an entry is dead once its column has left the diagonal (j <= d + 1; pairs: q <= d), and dead entries still cost the
        // readers a lane each.  Every MIRP_CPERIOD diagonals the pool is compacted in place: every wave keeps its slice in registers across a barrier, the
        // survivors move left behind the survivors of the lower waves.  (Between two barriers of its own: phase B of this interval appends after it.)
        auto compact_pool = [&](const int d) {
            constexpr int CR = CPOOL_ROUNDS;       // rounds of 64 entries per wave: 16 x 4 x 64 = 4096 = the capacity pool_cap is clamped to
            int* cnts = misc + 22;                 // [16]
            int total = misc[3];
            total = total < pool_cap ? total : pool_cap;
            const int per = ((total + LNW * 64 - 1) / (LNW * 64)) * 64;      // entries per wave (multiple of 64)
            unsigned ea[CR], eb[CR];
            int nal = 0;
            unsigned long long al[CR];
#pragma unroll
            for (int r = 0; r < CR; r++) {
                const int k = wave * per + r * 64 + lane;
                bool alive = false;
                ea[r] = 0; eb[r] = 0;
                if (r * 64 < per && k < total) {
                    ea[r] = poolA[k];
                    if constexpr (MODEL != 0) eb[r] = poolB32[k]; else eb[r] = poolB[k];
                    const int col = (int)((ea[r] >> 9) & 511u);
                    alive = MODEL ? (col >= d + 1) : (col >= d + 2);
                }
                al[r] = __ballot(alive);
                nal += (int)__popcll(al[r]);
            }
            if (lane == 0) cnts[wave] = nal;
            __syncthreads();
            int pre = 0, tot = 0;
            for (int w = 0; w < LNW; w++) { const int c = cnts[w]; pre += w < wave ? c : 0; tot += c; }
#pragma unroll
            for (int r = 0; r < CR; r++) {
                if ((al[r] >> lane) & 1ull) {
                    const int at = pre + (int)__popcll(al[r] & ((1ull << lane) - 1ull));
                    poolA[at] = ea[r];
                    if constexpr (MODEL != 0) poolB32[at] = eb[r]; else poolB[at] = (unsigned short)eb[r];
                }
                pre += (int)__popcll(al[r]);
            }
            if (tid == 0) misc[3] = tot;
            sp_snap = __builtin_amdgcn_readfirstlane(tot);
            __syncthreads();
        };
        if constexpr (MIRP_PB_BASES != 0) {
            if (tid < n - 4) {
                const int i = tid + 1, j = i + 4;
                lds_vu8 Sv = (lds_vu8)S;
                pb_si = (int)Sv[i - 1] | ((int)Sv[i] << 3) | ((int)Sv[i + 1] << 6);
                pb_sj = (int)Sv[j - 1] | ((int)Sv[j] << 3) | ((int)Sv[j + 1] << 6) | ((int)Sv[j + 2] << 9) | ((int)Sv[j + 3] << 12);
            }
        }
        if (Dm >= 4) phaseA(4);
        __syncthreads();
        if (dbg_cycles && tid == 0) { long long t = clock64(); tA += t - t0; t0 = t; }
        for (int d = 4; d <= Dm; d++) {
            if constexpr (SPARSE) { constexpr int CP = MODEL ? MIRP_CPERIOD1 : MIRP_CPERIOD0; if (CP > 0 && (d & (CP - 1)) == 0 && d >= 32) compact_pool(d); }
            if (dbg_cycles && lane == 0) wt = clock64();
            if constexpr (MODEL != 0) {   // the length of the list phase A1 of this interval looks ahead to (diagonal d+2, built in the previous interval): read
                // now, first looked at in phaseA -- on the waves that own cells the phase-B chain covers the round trip.  (vienna-1.8.5: -0.9 ms; the
                // default model measured +0.4 ms with it and keeps the read in phaseA.)
                const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) int*)&lcnt[(d + 2) % 6];
                asm volatile("ds_read_b32 %0, %1" : "=v"(lc_pre) : "v"(la) : "memory");
                lc_have = true;
            }
            if constexpr (MODEL == 0) { if (!SPARSE && (dbg_flags & 4096)) phaseB(d); else phaseB0(d); } else { if (dbg_flags & 4096) phaseB(d); else phaseB1(d); }
            if (dbg_cycles && lane == 0 && !(dbg_flags & (1 << 20))) { const long long t = clock64(); wB += t - wt; wt = t; }   // bit 20: light mode, busy / barrier only
            if (d + 1 <= Dm) phaseA(d + 1);
            if (dbg_cycles && lane == 0) { const long long t = clock64(); wA2 += t - wt; wt = t; }
            __syncthreads();
            if (dbg_cycles && lane == 0) { const long long t = clock64(); wW += t - wt; wt = t; }
            if (dbg_cycles && tid == 0 && !(dbg_flags & (1 << 20))) { long long t = clock64(); tB += t - t0; t0 = t; }
        }
        const int overflow = misc[1];
        __syncthreads();
        if (dbg_cycles && tid == 0) { long long t = clock64(); tE += t - t0; t0 = t; }
        const int pool_over = SPARSE ? misc[2] : 0;
        if (overflow) {   // int16 range exceeded: hand the window to the generic kernel
            if (tid == 0) { unsigned int k = atomicAdd(fallback_count, 1u); fallback_list[k] = win_base + win; out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; win_state[win] = 0; }
        } else if (pool_over) {   // more split candidates than the pool holds: the dense instantiation folds this window
            if (tid == 0) { out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; win_state[win] = 0; dense_list[atomicAdd(dense_count, 1u)] = win; }
        } else {
            // hand the tables to the epilogue kernel: c and the trace-back codes were archived on the fly, fML is copied out now into the same tiled
            // layout.  A wave takes whole row blocks; lane = diagonal, so the 8 rows of a row block on one diagonal are one 16-byte store and a
            // wave stores contiguous kilobytes; all of a row block's LDS reads are issued before the first store.
            if (Dm >= 4 && !(dbg_flags & 262144)) {
                constexpr int NGD = (LDMAX + 1 - 4) / 64 + 1;
                for (int rb = wave; 8 * rb + 1 + 4 <= n; rb += LNW) {
                    const int dmax_rb = Dm < n - 1 - 8 * rb ? Dm : n - 1 - 8 * rb;      // the block's first row reaches furthest
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
    if (dbg_cycles && lane == 0) {
        atomicAdd((unsigned long long*)&dbg_cycles[4 + wave * 4 + 0], (unsigned long long)wB); atomicAdd((unsigned long long*)&dbg_cycles[4 + wave * 4 + 1], (unsigned long long)wA1);
        atomicAdd((unsigned long long*)&dbg_cycles[4 + wave * 4 + 2], (unsigned long long)wA2); atomicAdd((unsigned long long*)&dbg_cycles[4 + wave * 4 + 3], (unsigned long long)wW);
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
#ifndef ENT
#define ENT 256
#endif
#ifndef MIRP_EPI_WGS
#define MIRP_EPI_WGS 8
#endif
__global__ void __launch_bounds__(ENT, MIRP_EPI_WGS) fold_lds_epilogue_kernel(
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
    short* xtab = (short*)(EP + 1);                                  // XTAB_N
    unsigned char* pq2 = (unsigned char*)(xtab + XTAB_N);            // nc
    short* ppart = (short*)(pq2 + nc);                               // nc
    const int tid = threadIdx.x;
    fill_epi_tables(EP, P, tid, ENT);
    fill_ext_table(xtab, P, tid, ENT);
    __syncthreads();
    EPI_INIT();
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
            if (tid < ARCH_RB) off[tid] = arch_rowblk_off(tid, n, span);
            __syncthreads();
            if (tid == 0) { S[0] = S[n]; S[n + 1] = S[1]; }
            for (int x = tid + 1; x <= n; x += ENT) pq2[x] = (unsigned char)((S[x] * 6 + (x < n ? (int)S[x + 1] : 5)) * 2);
            special_hairpins(P, seq, n, spec, nc, tid, ENT);
            __syncthreads();
            WinCtx X;
            X.P = P; X.S = S; X.seq = seq; X.f3 = f3; X.spec = spec; X.ldspec = nc; X.n = n; X.D = D; X.E = EP; X.xtab = xtab; X.pq2 = pq2; X.pp = ppart;
            LTab TB;
            TB.carch = slabs + (size_t)win * 3 * slab_shorts; TB.fml = TB.carch + slab_shorts; TB.off = off;
            TB.tb = reinterpret_cast<const unsigned short*>(TB.carch + 2 * slab_shorts);
            fold_epilogue<LTab, ENT>(X, TB, span, f3, starts, lens, btbuf, nc, btstk, misc + 8, win, max_lines, ss_stride, out_lines, out_ss, out_nlines,
                                    out_mfe, out_status);
        }
        __syncthreads();
    }
    EPI_FLUSH();
}

// Epilogue of the vienna-1.8.5 model on the slabs of fold_lds_kernel<1>: exterior sweep, enumeration, full backtracks (interior loops follow the
// trace-back codes), output.  Shares its device code with the generic vienna-1.8.5 kernel (fold185_device.h).
struct LTab185 {
    static constexpr bool kTiled = true;     // 8 x 8-tiled archive with trace-back codes: the patch backtrack applies
    const short* carch;
    const short* fml;
    const unsigned short* tb;
    const int* off;         // LDS: rowblk_off of the tiled archive layout
    int n, D, Dm;
    __device__ __forceinline__ int at(int d, int i) const { return off[(i - 1) >> 3] + ((i - 1) & 7) + 8 * (d - 4); }
    __device__ __forceinline__ int C(int i, int j) const {
        const int d = j - i;
        if (d <= TURN || d > D || i < 1 || j > n) return V_INF;
        const int v = carch[at(d, i)];
        return v == I16_INF ? V_INF : v;
    }
    __device__ __forceinline__ int Mm(int i, int j) const {
        const int d = j - i;
        if (d <= TURN || d > Dm || i < 1 || j > n) return V_INF;
        const int v = (unsigned short)fml[at(d, i)];
        return v == 65535 ? V_INF : v - FML_BIAS;
    }
    __device__ __forceinline__ int TB(int i, int j) const { return tb[at(j - i, i)]; }
};

#ifndef MIRP_EPI185_WGS
#define MIRP_EPI185_WGS 8
#endif
__global__ void __launch_bounds__(ENT, MIRP_EPI185_WGS) fold185_lds_epilogue_kernel(
    const FoldParams* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs, const int* __restrict__ win_lens,
    int n_work, int span, const short* __restrict__ slabs, size_t slab_shorts, const int* __restrict__ win_state, unsigned int* __restrict__ work_counter,
    int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines, char* __restrict__ out_ss, int* __restrict__ out_nlines,
    int* __restrict__ out_mfe, int* __restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nc = LCAP + 8;
    int* f3 = (int*)smem;                                            // nc + 8
    int* starts = f3 + nc + 8;                                       // max_lines
    int* lens = starts + max_lines;                                  // max_lines
    int* btstk = lens + max_lines;                                   // (ENT/64) * 3 * V_BT_STACK
    int* red = btstk + (ENT / 64) * 3 * V_BT_STACK;                  // ENT/64 + 8
    int* misc = red + ENT / 64 + 8;                                  // 4
    int* off = misc + 4;                                             // LDMAX + 2
    short* tetra = (short*)(off + LDMAX + 2);                        // nc
    unsigned char* S = (unsigned char*)(tetra + nc);                 // nc
    unsigned char* seq = S + nc;                                     // nc
    char* btbuf = (char*)(seq + nc);                                 // (ENT/64) * (nc + 8)
    short* dg = (short*)(smem + ((((size_t)(btbuf - (char*)smem) + (ENT / 64) * (nc + 8)) + 15) & ~(size_t)15));   // 80: dangle5 | dangle3
    const int tid = threadIdx.x;
    if (tid < 40) { dg[tid] = (short)P->dangle5[tid / 5][tid % 5]; dg[40 + tid] = (short)P->dangle3[tid / 5][tid % 5]; }
    __syncthreads();
    EPI_INIT();
    for (;;) {
        if (tid == 0) misc[0] = (int)atomicAdd(work_counter, 1u);
        __syncthreads();
        const int win = misc[0];
        __syncthreads();
        if (win >= n_work) break;
        if (win_state[win] == 1) {
            const long long o0 = offs[win];
            const int n = win_lens ? win_lens[win] : (int)(offs[win + 1] - o0);
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
            for (int x = tid; x < nc + 8; x += ENT) f3[x] = 0;
            if (tid < ARCH_RB) off[tid] = arch_rowblk_off(tid, n, span);
            __syncthreads();
            if (tid == 0) { S[0] = S[n]; S[n + 1] = S[1]; }
            for (int x = tid; x <= n; x += ENT) {
                short b = 0;
                if (x >= 1 && x + 5 <= n)
                    for (int k = 0; k < P->n_tetra; k++) {
                        bool m = true;
                        for (int t = 0; t < 6; t++) m = m && (seq[x + t] == (unsigned char)P->tetra[k][t]);
                        if (m) { b = (short)P->tetraE[k]; break; }
                    }
                tetra[x] = b;
            }
            __syncthreads();
            v185::Ctx<FoldParams> X;
            X.P = P; X.S = S; X.tetra = tetra; X.f3 = f3; X.n = n; X.M = span; X.dg = dg;
            LTab185 T;
            T.carch = slabs + (size_t)win * 3 * slab_shorts; T.fml = T.carch + slab_shorts;
            T.tb = reinterpret_cast<const unsigned short*>(T.carch + 2 * slab_shorts);
            T.off = off; T.n = n; T.D = (span - 1 < n - 1) ? span - 1 : n - 1; T.Dm = (span < n - 1) ? span : n - 1;
            v185::epilogue<FoldParams, LTab185, ENT>(X, T, f3, starts, lens, btstk, red, btbuf, nc, win, max_lines, ss_stride, out_lines, out_ss, out_nlines,
                                                     out_mfe, out_status);
        }
        __syncthreads();
    }
    EPI_FLUSH();
}

size_t fold185_lds_epilogue_bytes(int max_lines) {
    const size_t nc = LCAP + 8;
    size_t b = sizeof(int) * (nc + 8 + 2 * (size_t)max_lines + (ENT / 64) * 3 * V_BT_STACK + ENT / 64 + 8 + 4 + LDMAX + 2);
    b += sizeof(short) * nc + 2 * nc + (ENT / 64) * (nc + 8);
    return ((b + 15) & ~(size_t)15) + 16 + 80 * sizeof(short);
}

size_t fold_lds_epilogue_bytes(int max_lines) {
    const int nc = LCAP + 8;
    size_t b = sizeof(int) * (nc + 2 * (size_t)max_lines + (ENT / 64) * 3 * BT_STACK + 16 + LDMAX + 2) + sizeof(short) * 3 * nc + 2 * (size_t)nc + (ENT / 64) * (size_t)nc;
    b = (b + 15) & ~(size_t)15;
    return b + sizeof(EpiTables) + 16 + sizeof(short) * XTAB_N + nc + sizeof(short) * nc;
}

size_t fold_lds_bytes(int max_lines) { (void)max_lines; return lds_layout<0, true>().total; }
int fold_lds_max_n() { return LCAP - 2; }
int fold_lds_gen_wing_d() { return 5; }      // GEN_WD of fold_lds_kernel<0>
int fold_lds_max_span() { return LSPAN; }

// ctl (= work_counter): [0] work counter of the fill kernel, [1] of the epilogue, [2] work counter of the dense second pass, [3] windows handed to it
// (dense_list: their indices), [4] windows handed to the generic kernel, [5] running total of [3] over the sub-batches
hipError_t launch_fold_lds(hipStream_t stream, int model, int grid, int grid_epi, const FoldParams* P, const unsigned char* seqs, const long long* offs, const int* lens,
                           int n_work, int win_base, int span, short* slabs, size_t slab_shorts, int* win_state, unsigned int* work_counter, int* fallback_list,
                           unsigned int* fallback_count, int max_lines, int ss_stride, MirpFoldLine* out_lines, char* out_ss, int* out_nlines, int* out_mfe,
                           int* out_status, int dbg_flags, long long* dbg_cycles, hipEvent_t ev_between, int* dense_list, int force_dense) {
    const size_t lds = model ? lds_layout<1>().total : lds_layout<0>().total;
    const size_t lds_sp = lds_layout<0, true>().total, lds_sp1 = lds_layout<1, true>().total;
    hipError_t e = hipFuncSetAttribute((const void*)fold_lds_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_layout<1>().total);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)fold_lds_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_layout<0>().total);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)fold_lds_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)fold_lds_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_layout<1, true>().total);
    if (e != hipSuccess) return e;
    const int* no_list = nullptr; const unsigned int* no_count = nullptr;
    if (model) {
        // vienna-1.8.5: with dangles 1 every pair gives up to four strictly pair-realised fML cells ((i,j), (i-1,j), (i,j+1), (i-1,j+1)), i.e. about four
        // times the candidate cells of the default model (3,142 per benchmark window: a cell pool overflowed for most windows, 161 ms = both passes);
        // the pool of this instantiation holds PAIRS instead (1,283 per window, 8-byte entries: splits_sparse185).
        const bool sparse185 = !force_dense;
        if (sparse185) {      // as the default model below: candidate-pool pass, then the dense instantiation over what it handed over
            hipLaunchKernelGGL((fold_lds_kernel<1, true>), dim3(grid), dim3(LNT), lds_sp1, stream, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter,
                               fallback_list, fallback_count, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status, dbg_flags, dbg_cycles,
                               no_list, no_count, dense_list, work_counter + 3);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fold_lds_kernel<1, false>), dim3(grid), dim3(LNT), lds, stream, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter + 2,
                               fallback_list, fallback_count, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status, dbg_flags, dbg_cycles,
                               (const int*)dense_list, (const unsigned int*)(work_counter + 3), dense_list, work_counter + 3);
        } else
            hipLaunchKernelGGL((fold_lds_kernel<1, false>), dim3(grid), dim3(LNT), lds, stream, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter,
                               fallback_list, fallback_count, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status, dbg_flags, dbg_cycles,
                               no_list, no_count, dense_list, work_counter + 3);
    } else {
#if !defined(MIRP_FILL2)
        // product: first pass with sparse multiloop splits (candidate pool), then the dense instantiation over the windows the first pass handed over
        // (pool overflow, no room for a pool: zero on the benchmark inputs; the launch then finds an empty list).  force_dense (tests, A/B timing): the
        // dense instantiation folds everything.  -DMIRP_FILL2 (dev builds, make VARIANT=...) selects the two-diagonal schedule of
        // fold_lds2_kernel.hip, which is parity-green but measured no faster
        if (!force_dense) {
            hipLaunchKernelGGL((fold_lds_kernel<0, true>), dim3(grid), dim3(LNT), lds_sp, stream, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter,
                               fallback_list, fallback_count, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status, dbg_flags, dbg_cycles,
                               no_list, no_count, dense_list, work_counter + 3);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fold_lds_kernel<0, false>), dim3(grid), dim3(LNT), lds, stream, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter + 2,
                               fallback_list, fallback_count, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status, dbg_flags, dbg_cycles,
                               (const int*)dense_list, (const unsigned int*)(work_counter + 3), dense_list, work_counter + 3);
        } else
            hipLaunchKernelGGL((fold_lds_kernel<0, false>), dim3(grid), dim3(LNT), lds, stream, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter,
                               fallback_list, fallback_count, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status, dbg_flags, dbg_cycles,
                               no_list, no_count, dense_list, work_counter + 3);
#else
        e = launch_fold_lds2_fill(stream, grid, P, seqs, offs, lens, n_work, win_base, span, slabs, slab_shorts, win_state, work_counter, fallback_list, fallback_count,
                                  out_nlines, out_mfe, out_status);
        if (e != hipSuccess) return e;
#endif
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ev_between) { e = hipEventRecord(ev_between, stream); if (e != hipSuccess) return e; }   // fill | epilogue boundary (mirp_last_fold_kernel_ms)
    if (!(dbg_flags & 16)) {
        if (model) {
            const size_t el = fold185_lds_epilogue_bytes(max_lines);
            if (el > 64 * 1024) {
                e = hipFuncSetAttribute((const void*)fold185_lds_epilogue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)el);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(fold185_lds_epilogue_kernel, dim3(grid_epi), dim3(ENT), el, stream, P, seqs, offs, lens, n_work, span, slabs, slab_shorts, win_state,
                               work_counter + 1, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status);
        } else {
            hipLaunchKernelGGL(fold_lds_epilogue_kernel, dim3(grid_epi), dim3(ENT), fold_lds_epilogue_bytes(max_lines), stream, P, seqs, offs, lens, n_work, span,
                               slabs, slab_shorts, win_state, work_counter + 1, max_lines, ss_stride, out_lines, out_ss, out_nlines, out_mfe, out_status);
        }
    }
    return hipGetLastError();
}

#ifdef MIRP_EPI_CLOCKS
void fold_lds_epi_clocks_print() {
    unsigned long long h[32];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_epi_clk), sizeof(h));
    const char* nm[14] = {"f3 sweep", "enumeration", "partner scan", "backtrack", "output", "loop tail", "barrier", "containment",
                          "sweep: issue+init+barrier", "sweep: load wait", "sweep: step 1", "sweep: barrier 2", "sweep: step 2", "sweep: barrier 3"};
    for (int k = 0; k < 14; k++) std::fprintf(stderr, "[mirp epi clocks] %-14s %llu\n", nm[k], h[k]);
    const char* cn[10] = {"ext partner scan rounds", "ml segment pops", "ml pair checks", "ml split rounds (segment)", "helix line fetches", "line-end c fetches",
                          "line-end code fetches", "ml split rounds (closing)", "short-backtrack scan rounds", "structures"};
    for (int k = 0; k < 10; k++) std::fprintf(stderr, "[mirp epi counts] %-28s %llu\n", cn[k], h[16 + k]);
    unsigned long long z[32] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_epi_clk), z, sizeof(z));
}
#endif

size_t fold_lds_slab_shorts(int n_cap) {   // tiled archive of one table for windows up to n_cap at the largest span (whole 128-byte tiles)
    return (size_t)arch_rowblk_off(ARCH_RB, n_cap < LCAP ? n_cap : LCAP, LDMAX + 1) + 64;
}

}  // namespace mirp
