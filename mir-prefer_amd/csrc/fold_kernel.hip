// Batched L-bounded Zuker MFE local fold with enumeration of locally optimal structures.
// One precursor window per workgroup.  Replaces `RNALfold -L <PRECURSOR_LEN>`
// (/root/reference/miR_PREFeR.py:3047-3119, command line :3053) for gfx950.
//
// This file holds the GENERIC kernels: DP tables (c, fML, trace-back codes, a DML ring and the split-candidate pool) live in a
// per-window global workspace slot laid out diagonal-major ((d, i) -> d*ld + i) so that the reads of the anti-diagonal wavefront
// fill are neighbours across the lanes that own neighbouring cells.  Any window length and any span (PRECURSOR_LEN 60 .. 3000,
// MP:167-184): the path of every window when span > 300, and the fallback for windows the LDS-resident kernels
// (fold_lds_kernel.hip) hand back (longer than 350 nt, energies outside the 16-bit range).  Two kernels per batch of windows:
// fold_generic_kernel<1> fills the tables, <2> runs the exterior sweep, the enumeration and the backtracks (fold_epilogue.h).
#include <hip/hip_runtime.h>
#include "fold_epilogue.h"

namespace mirp {

#define TURN MIRP_TURN
#define MAXLOOP MIRP_MAXLOOP
#define INF MIRP_INF

#include <type_traits>
typedef int int2a __attribute__((ext_vector_type(2), aligned(4)));      // consecutive cells of a table row from any 4-byte boundary
typedef int int4a __attribute__((ext_vector_type(4), aligned(4)));
typedef int int4q __attribute__((ext_vector_type(4), aligned(16)));

struct GTab {
    static constexpr bool kTiled = false;
    int* __restrict__ c;
    int* __restrict__ m;
    // The fill's own read-only view of finished cells (round 5): one word per cell = c(p,q) + mismatchI of (p,q) seen as the INNER pair of a generic interior
    // loop in the low 24 bits (signed; GEN_PINF where (p,q) is no pair) and the index of that term, rtype * 25 + S[q+1] * 5 + S[p-1], in the high byte --
    // from which the other classes' terms follow through small LDS tables (bulge: TerminalAU - mismatchI, 1 x n: mismatch1nI - mismatchI, the rest: c itself).
    // The interior-loop interval reads nothing else: a 32-diagonal ring of it is 4 bytes per cell.
    int* __restrict__ w;
    unsigned short* __restrict__ tb;
    int ld;
    __device__ __forceinline__ int C(int d, int i) const { return c[(size_t)d * ld + i]; }
    __device__ __forceinline__ int M(int d, int i) const { return m[(size_t)d * ld + i]; }
    // trace-back code of a pair (round 5): 1 + (n1 << 5 | n2) = the interior loop the backtrack's search (p ascending, q descending) would find first,
    // 0 = the hairpin realises c(i,j), or no interior loop does (multiloop)
    __device__ __forceinline__ int TB(int d, int i) const { return tb[(size_t)d * ld + i]; }
};

// ------------------------------------------------------------------------------------------
// Generic kernel: tables in global workspace.
// ------------------------------------------------------------------------------------------
#ifndef GEN_NT
#define GEN_NT 256
#endif
#ifndef GEN_FILL_WAVES
#define GEN_FILL_WAVES 6
#endif
#ifndef GEN_EPI_WAVES
#define GEN_EPI_WAVES 8                   // (the epilogue is a chain of round trips to the workspace: 148 VGPRs and three workgroups per CU take 0.142 s at L = 400, 64 with spills and eight 0.133)
#endif
#define GEN_MIN_WAVES(phase) ((phase) == 1 ? GEN_FILL_WAVES : GEN_EPI_WAVES)      // waves per SIMD the register allocation aims for
#ifndef GEN_PU
#define GEN_PU 4                  // split candidates of a column whose reads are in flight together (interval B)
#endif
#define GEN_PL_MAXN 2500            // longest window whose split candidates fit the packed LDS copy (12 bits of position would allow 4,095; 20 bits of energy: 340 per pair)
#define GEN_STAGE 512              // ints per wave of the interior-loop interval's staging buffer
#define GEN_PINF 1500000           // 'no pair' in the 24-bit energy field of GTab::w: with every loop term added it stays below 2^21, so that energy * 1024 + shape is an int
#define GEN_EMAX 1000000           // a candidate energy at or above this came from a GEN_PINF entry (real energies: a few hundred per nucleotide pair at most)
#define GEN_AUX_BYTES(nc) ((((size_t)(8 + 2 + 2 + 2) * (nc) + 8) + 15) / 16 * 16)      // cbest (32-bit keys; two diagonals), pcnt (short), plist (short), ctype (byte; two diagonals) per position

__host__ __device__ size_t fold_generic_lds_bytes_base(int n_cap, int max_lines) {
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

// row stride of a workspace table (a multiple of four ints: the interior-loop interval stages row segments with aligned 16-byte loads)
__host__ __device__ inline int fold_generic_ld(int n_cap) { return (n_cap + 2 + 3) & ~3; }

// the fill kernel's share of the base carve-up: the sequence codes and the special-hairpin table (everything else above is the epilogue's)
__host__ __device__ size_t fold_generic_lds_bytes_base_fill(int n_cap) {
    const int nc = n_cap + 8;
    return ((size_t)(sizeof(int) * 8 + sizeof(short) * 3 * nc + nc) + 15) & ~(size_t)15;
}

// one table (c or fML) of a workspace slot, in ints
__host__ __device__ size_t fold_generic_table_ints(int n_cap, int span) {
    size_t D = (size_t)(span < n_cap ? span : n_cap) + 1;
    size_t per = D * (size_t)fold_generic_ld(n_cap);
    return (per + 63) & ~(size_t)63;
}

// split candidates a column can hold: one per diagonal at most
__host__ __device__ int fold_generic_pool_cap(int n_cap, int span) { return (span < n_cap ? span : n_cap) + 1; }

// PHASE 1 = fill (tables into the window's workspace slot), PHASE 2 = exterior sweep + structure enumeration + backtracks out of the slot.  Two
// instantiations, launched back to back over a batch of at most `grid` windows (slot = blockIdx.x): the fill needs 74 VGPRs, the epilogue 163 -- as one
// kernel the fill ran at the epilogue's occupancy (3 workgroups per CU instead of 6).
template <int PHASE>
__global__ void __launch_bounds__(GEN_NT, GEN_MIN_WAVES(PHASE)) fold_generic_kernel(
    const FoldParams* __restrict__ P, const unsigned char* __restrict__ seqs, const long long* __restrict__ offs,
    const int* __restrict__ win_lens, const int* __restrict__ work_list, int n_work, int span, int n_cap, int* __restrict__ ws, size_t ws_slot_ints,
    int max_lines, int ss_stride, MirpFoldLine* __restrict__ out_lines, char* __restrict__ out_ss,
    int* __restrict__ out_nlines, int* __restrict__ out_mfe, int* __restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS carve-up (n_cap = max window length this launch supports)
    const int nc = n_cap + 8;
    // (the two kernels have their own layouts: the epilogue needs none of the fill's lists, tables and staging buffers, the fill none of the epilogue's
    // line and backtrack buffers -- 24 KB instead of 33 at n = 425 lets five of its workgroups share a CU's LDS)
    int *f3 = nullptr, *starts = nullptr, *lens = nullptr, *btstk = nullptr, *sh_misc;
    short* spec;
    unsigned char *S, *seq;
    char* btbuf = nullptr;
    if constexpr (PHASE == 1) {
        sh_misc = (int*)smem;                              // 8
        spec = (short*)(sh_misc + 8);                      // 3*nc
        S = (unsigned char*)(spec + 3 * nc);               // nc
        seq = nullptr;                                     // (set below: the window's letters are needed for the special-hairpin table only, and sit in the staging buffer meanwhile)
    } else {
        f3 = (int*)smem;                                   // nc ints
        starts = f3 + nc;                                  // max_lines
        lens = starts + max_lines;                         // max_lines
        btstk = lens + max_lines;                          // (NT/64)*3*BT_STACK
        sh_misc = btstk + (GEN_NT / 64) * 3 * BT_STACK;    // 8
        spec = (short*)(sh_misc + 8);                      // 3*nc
        S = (unsigned char*)(spec + 3 * nc);               // nc
        seq = S + nc;                                      // nc
        btbuf = (char*)(seq + nc);                         // (NT/64)*nc
    }
    const size_t base_bytes = PHASE == 1 ? fold_generic_lds_bytes_base_fill(n_cap) : fold_generic_lds_bytes_base(n_cap, max_lines);
    int* cbest = (int*)(smem + base_bytes);                // 2 nc: interior-loop minimum of the diagonal's cells as the tasks' key, energy * 1024 + (n1 << 5 | n2): the
                                                           // minimum names the first loop in the backtrack's search order (0x7fffffff: none)  (from here on: the fill's alone)
    unsigned short* pcnt = (unsigned short*)(cbest + 2 * nc);      // nc: split candidates of every column so far
    unsigned short* plist = pcnt + nc;                     // nc: the diagonal's paired cells  (cbest, ctype: [diagonal & 1][nc] -- interval A of d + 1 runs beside interval B of d)
    unsigned char* ctype = (unsigned char*)(plist + nc);   // 2 nc: pair type of the diagonal's cells
    // inner-pair terms of the interior-loop energies, [t2][sq1][sp1] as shorts, and the stacking table: read per candidate -- out of LDS, not through
    // the texture addresser (the interval was bound by vector-memory instructions, one per table look-up and lane: DESIGN.md 4, "Generic kernels")
    short* l_mmI = (short*)(smem + base_bytes + GEN_AUX_BYTES((size_t)nc));
    short* l_mm1n = l_mmI + 200;
    short* l_mm23 = l_mm1n + 200;
    short* l_xb = l_mm23 + 200;                            // bulge: TerminalAU of the inner pair - its mismatchI (what turns the table's word into c + TerminalAU)
    short* l_x1 = l_xb + 200;                              // 1 x n: mismatch1nI - mismatchI
    short* l_stack = l_x1 + 200;                           // 64
    int* stage = reinterpret_cast<int*>(l_stack + 64);     // GEN_STAGE ints per wave: the row segment a block of paired cells reads for one loop size (interval A)
    int* wcnt = stage + (GEN_NT / 64) * GEN_STAGE;         // 2 * waves: paired cells per wave and half-pass of the list compaction
    if constexpr (PHASE == 1) seq = reinterpret_cast<unsigned char*>(stage);          // nc <= 4 * (GEN_NT / 64) * GEN_STAGE bytes (checked by the host: windows up to 8,184 nt)
    int* pl4 = wcnt + 2 * (GEN_NT / 64);                  // 4 nc: the first four split candidates of every column, packed s << 20 | (fML & 0xfffff) (interval B; windows up to GEN_PL_MAXN nt)
    if constexpr (PHASE == 1) {
    for (int x = threadIdx.x; x < 200; x += GEN_NT) {
        const int t = x / 25, a = (x / 5) % 5, b = x % 5;
        // (the rows of pair type 0 hold INF and are never read: an interior candidate has a pair on both sides)
        l_mmI[x] = (short)min(P->mismatchI[t][a][b], 32767); l_mm1n[x] = (short)min(P->mismatch1nI[t][a][b], 32767); l_mm23[x] = (short)min(P->mismatch23I[t][a][b], 32767);
        l_xb[x] = t ? (short)((t > 2 ? P->TerminalAU : 0) - P->mismatchI[t][a][b]) : (short)0;
        l_x1[x] = t ? (short)(P->mismatch1nI[t][a][b] - P->mismatchI[t][a][b]) : (short)0;
    }
    for (int x = threadIdx.x; x < 64; x += GEN_NT) l_stack[x] = (short)min(P->stack[x >> 3][x & 7], 32767);
    }
    __syncthreads();

    const int tid = threadIdx.x;
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int win = work_list ? work_list[w] : w;
        const long long o0 = offs[win];
        const int n = win_lens ? win_lens[win] : (int)(offs[win + 1] - o0);
        if (n < 1 || n > n_cap) {
            if (PHASE == 2 && tid == 0) { out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = n < 1 ? 0 : -40; }
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
        T.ld = fold_generic_ld(n_cap);
        const size_t tab_ints = fold_generic_table_ints(n_cap, span);
        T.c = ws + (size_t)blockIdx.x * ws_slot_ints;
        T.m = T.c + tab_ints;
        // Multiloop splits (round 5): DML(i,j) = min_s fML(i,s-1) + fML(s,j) is kept for the last four diagonals (the cell's own fML needs DML(i,j), the
        // closing term of (i,j) is DML(i+1,j-1) of diagonal d-2 -- the dense kernel recomputed that loop), and a diagonal's DML comes from
        // DML(i,j-1) and the SPLIT CANDIDATES of column j only: the cells (s,j) whose fML is realised strictly by their pair term (exact with
        // ML_BASE = 0: any other split is dominated, DESIGN.md 4 / tests/tools/splitcand_gate.c).  Column j's candidates {s, fML(s,j)} are appended to
        // pool[j] by the one cell of the column on each diagonal (no contention), counts in LDS.
        int* dml = T.m + tab_ints;                              // [4][ld]
        int2* pool = reinterpret_cast<int2*>(dml + 4 * (size_t)T.ld);      // [ld][pcap]
        const int pcap = fold_generic_pool_cap(n_cap, span);
        T.tb = reinterpret_cast<unsigned short*>(pool + (size_t)T.ld * pcap);      // [D + 1][ld] shorts
        T.w = reinterpret_cast<int*>(T.tb) + ((tab_ints / 2 + 63) & ~(size_t)63);      // [D + 1][ld] ints, behind the trace-back codes (tab_ints shorts = half a table)
        if constexpr (PHASE == 1) {
            // diagonal TURN of fML must read as INF
            for (int x = tid; x <= n; x += GEN_NT) T.m[(size_t)TURN * T.ld + x] = INF;
            for (int x = tid; x < 4 * T.ld; x += GEN_NT) dml[x] = INF;
            for (int x = tid; x <= n + 1; x += GEN_NT) pcnt[x] = 0;
        }
        __syncthreads();
        WinCtx X;
        X.P = P; X.S = S; X.seq = seq; X.f3 = f3; X.spec = spec; X.ldspec = nc; X.n = n; X.D = D;
        if constexpr (PHASE == 1) {

        // ---- anti-diagonal wavefront fill: all cells with the same d = j - i are independent.  Three intervals per diagonal:
        //  0  pair types of the diagonal's cells, the paired ones compacted into a list;
        //  A  interior loops with LANE = PAIRED CELL and a wave-uniform loop shape: a task is (block of 64 paired cells, n1), the n2 loop runs in lockstep,
        //     so every lane of a wave evaluates the same (n1, n2) form of the loop energy (no divergence inside e_intloop) and the lanes' reads of
        //     c(p, q) are neighbours on ONE diagonal (d - n1 - n2 - 2); the minimum per cell is merged with an LDS atomic;
        //  B  a thread per cell: hairpin, multiloop closing from the DML ring, DML(i,j) from DML(i,j-1) and the column's split candidates, fML.
        const int lane = tid & 63, wave = tid >> 6;
#ifdef MIRP_X_GEN_CLOCKS              // dev: where a wave's time goes (printed by wave 0 of block 0 for its first window)
        long long ck_0 = 0, ck_small = 0, ck_large = 0, ck_b = 0, ck_wait = 0, ck_t = clock64();
        int ck_nsmall = 0, ck_nlarge = 0;
#define CK(acc) do { const long long now_ = clock64(); acc += now_ - ck_t; ck_t = now_; } while (0)
#else
#define CK(acc) do { } while (0)
#endif
        // A turn of the loop: pair types and the paired-cell list of diagonal d, then -- no barrier between them: they touch different diagonals -- the interior
        // loops of d (interval A: reads rows <= d - 2 of the packed table) and hairpin / multiloop / fML of d - 1 (interval B: writes row d - 1).  Each wave runs
        // its share of both, so the turn takes the slowest wave's SUM, not the slowest of A plus the slowest of B, and a diagonal costs two barriers, not three.
        for (int d = TURN + 1; d <= D + 1; d++) {
            unsigned char* const ctA = ctype + (d & 1) * nc;
            int* const cbA = cbest + (d & 1) * nc;
            if (d <= D) {
            const int ncell = n - d;
            // pair types of the diagonal's cells; the paired ones compacted into a list IN CELL ORDER (a block of 64 list entries is a stretch of the
            // diagonal: interval A stages that stretch's row segments): two cells per thread and pass, the eight per-wave counts ordered through LDS
            int np_run = 0;
            for (int base = 0; base < ncell; base += 2 * GEN_NT) {
                int type[2];
                unsigned long long bal[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int cell = base + h * GEN_NT + tid;
                    type[h] = 0;
                    if (cell < ncell) { type[h] = pair_type(S[cell + 1], S[cell + 1 + d]); ctA[cell] = (unsigned char)type[h]; cbA[cell] = 0x7fffffff; }
                    bal[h] = __ballot(type[h] != 0);
                    if (lane == 0) wcnt[h * (GEN_NT / 64) + wave] = (int)__popcll(bal[h]);
                }
                __syncthreads();
                int before[2] = {0, 0}, total = 0;
#pragma unroll
                for (int x = 0; x < 2 * (GEN_NT / 64); x++) {
                    const int v = wcnt[x];
                    if (x < wave) before[0] += v;
                    if (x < GEN_NT / 64 + wave) before[1] += v;
                    total += v;
                }
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (type[h]) plist[np_run + before[h] + (int)__popcll(bal[h] & ((1ull << lane) - 1ull))] = (unsigned short)(base + h * GEN_NT + tid);
                np_run += total;
                if (base + 2 * GEN_NT < ncell) __syncthreads();          // (the counts are rewritten by the next pass)
            }
            CK(ck_0);
            __syncthreads();
            CK(ck_wait);
            const int np = np_run;
            const int n1max = (d - 2 - (TURN + 1) < MAXLOOP) ? d - 2 - (TURN + 1) : MAXLOOP;      // q - p = d - n1 - n2 - 2 >= TURN + 1
#ifdef MIRP_X_GEN_NOA               // timing experiment: no interior loops (tables wrong by construction)
            if (false) {
#else
            if (n1max >= 0) {
#endif
                // Task = (block of 64 paired cells, group of loop sizes s = n1 + n2).  All candidates of one size have their inner pair on ONE diagonal,
                // dd = d - 2 - s, and a lane's candidates n1 = 0 .. s are the CONSECUTIVE cells p = i + 1 + n1 of that diagonal's row: four of them per 16-byte
                // load, and every load of a size is issued before the first is used (the earlier order -- n1 outer, n2 inner, one 4-byte gather per
                // candidate on 31 different diagonals, each waited for -- was bound by memory round trips).  Every candidate but nine reads ONE precombined
                // value (c + the inner pair's term of its class; INF where (p,q) is no pair): size and asymmetry terms are scalar (P->gen_key), the outer
                // pair's term a register, so a candidate is a multiply-add and a minimum.  (n1, n2) are wave-uniform.  Groups: sizes 0..9 (0..6: the shapes with
                // their own tables, straight-line), 10..17, 18..24, 25..30 -- about equal work by the waves' clocks (a wave keeps its group: the task stride is the group count).
                const int smax = n1max;
                const int nblk = (np + 63) >> 6;
                const int ngrp = smax < 10 ? 1 : smax < 18 ? 2 : smax < 25 ? 3 : 4;
                const int ntask = nblk * ngrp;
                const int ninio = P->ninio, max_ninio = P->MAX_NINIO;
                for (int t = wave; t < ntask; t += GEN_NT / 64) {
                    const int tu = __builtin_amdgcn_readfirstlane(t);          // the task is the wave's: loop sizes and shapes in scalar registers
                    const int blk = tu / ngrp, grp = tu - blk * ngrp;
                    const int k = blk * 64 + lane;
                    {
                        // (the lanes behind the list's end, in its last block, run along on its last cell: the staged loads below are the whole wave's)
                        const bool active = k < np;
                        const int cell = plist[active ? k : np - 1];
                        const int i = cell + 1, j = i + d;
                        const int type = ctA[cell];
                        const int si1 = S[i + 1], sj1 = S[j - 1];
                        const int ij = type * 25 + si1 * 5 + sj1;
                        const int o_mmI = l_mmI[ij], o_mm1n = l_mm1n[ij], o_mm23 = l_mm23[ij];      // the outer pair's terms: once per task
                        const int tau = type > 2 ? P->TerminalAU : 0;
                        // The cell's minimum over the task as ONE 32-bit key, energy * 1024 + (n1 << 5 | n2): among equal energies the smallest n1, then the smallest
                        // n2 -- the loop the backtrack's search (p ascending, q descending) finds first.  (Energies of the generic path stay below 2^20 in
                        // magnitude: GEN_PINF.)  A generic candidate is one v_mad_i32_i24 -- the instruction takes the word's 24-bit energy field as it is --
                        // with the scalar key term of its shape (P->gen_key: size and asymmetry terms and the shape), and a minimum.
                        int kmin = 0x7fffffff;
                        const int c1024 = 1024;
                        auto gen = [&](const int w, const int kterm) {
                            int key;
                            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(key) : "v"(w), "v"(c1024), "s"(kterm));
                            return key;
                        };
                        auto w_e = [](const int w) { return (w << 8) >> 8; };                      // the word's energy (sign-extended 24 bits)
                        auto w_in = [](const int w) { return (int)((unsigned)w >> 24); };          // the word's inner-pair index
                        auto put = [&](const int e, const int shape) { const int key = e * 1024 + shape; kmin = key < kmin ? key : kmin; };
                        auto put_bulge = [&](const int eb, const int w, const int shape) { put(eb + w_e(w) + (int)l_xb[w_in(w)], shape); };
                        auto put_1n = [&](const int e1, const int w, const int shape) { put(e1 + w_e(w) + (int)l_x1[w_in(w)], shape); };
                        const int* wlane = T.w + i + 1;          // candidate n1 of size s = wlane[(d - 2 - s) * ld + n1]
                        // sizes s_lo .. s_hi (>= 7) straight from the table, a load per candidate: the fallback for blocks whose cells lie too far apart for the staging
                        // buffer below (sparse pairs in long windows)
                        auto sizes = [&](const int s_lo, const int s_hi) {
                            for (int s = s_lo; s <= s_hi; s++) {
                                const int* wrow = wlane + (d - 2 - s) * T.ld;
                                const int* kc = reinterpret_cast<const int*>(P->gen_key[s - 6]);
                                int kg = 0x7fffffff;
                                for (int n1 = 2; n1 <= s - 2; n1++) { const int key = gen(wrow[n1], kc[n1]); kg = key < kg ? key : kg; }
                                kg += o_mmI * 1024;
                                kmin = kg < kmin ? kg : kmin;
                                const int eb = P->bulge[s] + tau;
                                const int x1 = (s - 2) * ninio;
                                const int e1 = P->internal_loop[s] + (x1 < max_ninio ? x1 : max_ninio) + o_mm1n;
                                put_bulge(eb, wrow[0], s); put_1n(e1, wrow[1], 1 << 5 | (s - 1)); put_1n(e1, wrow[s - 1], (s - 1) << 5 | 1); put_bulge(eb, wrow[s], s << 5);
                            }
                        };
                        // The same through LDS (the form that runs): the row segment the block's
                        // cells read for one size -- from the first cell's p to the last cell's p + s, 200 words or so for 64 paired cells -- is loaded ONCE by the
                        // wave (aligned 16 bytes per lane), written to the wave's staging buffer, and each lane reads its s + 1 consecutive words from there.  Lane by
                        // lane the same words cost 17 cache-line accesses per 16-byte load and kept the CU's vector-memory path busy the whole time (counters:
                        // profiles/EXPERIMENT_LOG.md); the loads of the next two sizes are in flight while a size is reduced.
                        const int cell_lo = __builtin_amdgcn_readfirstlane(plist[blk * 64]);
                        const int cell_hi = __builtin_amdgcn_readfirstlane(plist[blk * 64 + 63 < np ? blk * 64 + 63 : np - 1]);
                        const int b0a = (cell_lo + 2) & ~3;                     // first staged position (p of the first cell, rounded down to 16 bytes)
                        const int span_w = cell_hi + 2 - b0a;                   // the last cell's offset in the segment
                        int* slot = stage + wave * GEN_STAGE;
                        const int* sl = slot + (cell + 2 - b0a);                // this lane's candidate n1 = sl[n1]
                        auto sizes_staged = [&](auto two_tag, const int s_lo, const int s_hi) {
                            constexpr bool TWO = decltype(two_tag)::value;          // a second 256-word piece
                            auto issue = [&](const int s, int4q& a, int4q& b) {
                                const int* src = T.w + ((d - 2 - s) * T.ld + b0a) + 4 * lane;
                                a = *reinterpret_cast<const int4q*>(src);
                                if (TWO) b = *reinterpret_cast<const int4q*>(src + 256);
                            };
                            // one size: its segment from the registers to the buffer, the registers re-used for the load three sizes on, then the candidates
                            auto step = [&](const int s, int4q& a, int4q& b) {
                                *reinterpret_cast<int4q*>(slot + 4 * lane) = a;
                                if (TWO) *reinterpret_cast<int4q*>(slot + 256 + 4 * lane) = b;
                                if (s + 3 <= s_hi) issue(s + 3, a, b);
                                const int4q* kq = reinterpret_cast<const int4q*>(P->gen_key2[s - 6]);          // key terms of n1 = 2 + 4 c .. 5 + 4 c: one scalar load
                                const int4q kt = *reinterpret_cast<const int4q*>(P->gen_keyt[s - 6]);           // of n1 = s - 5 .. s - 2
                                int kg = 0x7fffffff;
                                // (rolled on purpose: with the chunk count a compile-time constant per size -- every read of a size under way before the first
                                // candidate -- the kernel is seven bodies longer and 7 % slower)
                                for (int c4 = 0; 4 * c4 + 5 <= s - 2; c4++) {
                                    const int4q k4 = kq[c4];
#pragma unroll
                                    for (int u = 0; u < 4; u++) { const int key = gen(sl[2 + 4 * c4 + u], k4[u]); kg = key < kg ? key : kg; }
                                }
#pragma unroll
                                for (int u = 0; u < 4; u++) { const int key = gen(sl[s - 5 + u], kt[u]); kg = key < kg ? key : kg; }          // (may overlap the last chunk)
                                kg += o_mmI * 1024;
                                kmin = kg < kmin ? kg : kmin;
                                const int eb = P->bulge[s] + tau;
                                const int x1 = (s - 2) * ninio;
                                const int e1 = P->internal_loop[s] + (x1 < max_ninio ? x1 : max_ninio) + o_mm1n;
                                put_bulge(eb, sl[0], s); put_1n(e1, sl[1], 1 << 5 | (s - 1)); put_1n(e1, sl[s - 1], (s - 1) << 5 | 1); put_bulge(eb, sl[s], s << 5);
                            };
                            int4q a0 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};
                            issue(s_lo, a0, b0);
                            if (s_lo + 1 <= s_hi) issue(s_lo + 1, a1, b1);
                            if (s_lo + 2 <= s_hi) issue(s_lo + 2, a2, b2);
                            for (int s = s_lo; s <= s_hi; s += 3) {          // three sizes per turn: each has its own registers, nothing is moved
                                step(s, a0, b0);
                                if (s + 1 <= s_hi) step(s + 1, a1, b1);
                                if (s + 2 <= s_hi) step(s + 2, a2, b2);
                            }
                        };
                        const bool staged = span_w + 31 + 1 <= GEN_STAGE;
                        if (grp == 0) {
                            // sizes 0 .. 6, straight-line: every load first (rows of sizes beyond smax -- the first diagonals only -- are read at a clamped row and
                            // not used), then the shapes with (n1, n2) as compile-time constants
                            const int sI[5] = {S[i], si1, S[i + 2], S[i + 3], S[i + 4]};          // S[i + x]
                            const int sJ[5] = {S[j], sj1, S[j - 2], S[j - 3], S[j - 4]};          // S[j - x]
                            const int* wr[7];
#pragma unroll
                            for (int s = 0; s < 7; s++) { const int dd = d - 2 - s; wr[s] = wlane + (dd > 0 ? dd : 0) * T.ld; }
                            const int w0 = wr[0][0];
                            const int2a w1 = *reinterpret_cast<const int2a*>(wr[1]);
                            const int4a w2 = *reinterpret_cast<const int4a*>(wr[2]), w3 = *reinterpret_cast<const int4a*>(wr[3]);
                            // reversed type of the inner pair (p, q) = (i + 1 + n1, j - 1 - n2), 0 = no pair
                            auto t2of = [&](const int n1, const int n2) { return pair_type(sJ[1 + n2], sI[1 + n1]); };
                            const int t00 = t2of(0, 0), t01 = t2of(0, 1), t10 = t2of(1, 0), t11 = t2of(1, 1), t12 = t2of(1, 2), t21 = t2of(2, 1), t22 = t2of(2, 2),
                                      t23 = t2of(2, 3), t32 = t2of(3, 2);
                            // the big tables (sq1 = S[q + 1] = sJ[n2], sp1 = S[p - 1] = sI[n1]); a missing pair reads row 0 and is masked below
                            const int r11 = P->int11[type][t11][si1][sj1];
                            const int r12 = P->int21[type][t12][si1][sJ[2]][sj1];
                            const int r21 = P->int21[t21][type][sJ[1]][si1][sI[2]];
                            const int r22 = P->int22[type][t22][si1][sI[2]][sJ[2]][sj1];
                            const int b1 = P->bulge[1];
                            auto cof = [&](const int w) { return w_e(w) - (int)l_mmI[w_in(w)]; };          // c(p,q) of a pair
                            auto put_if = [&](const int t2, const int e, const int shape) { const int key = t2 ? e * 1024 + shape : 0x7fffffff; kmin = key < kmin ? key : kmin; };
                            put_if(t00, (int)l_stack[type * 8 + t00] + cof(w0), 0);
                            if (smax >= 1) {
                                put_if(t01, b1 + (int)l_stack[type * 8 + t01] + cof(w1[0]), 0 << 5 | 1);
                                put_if(t10, b1 + (int)l_stack[type * 8 + t10] + cof(w1[1]), 1 << 5 | 0);
                            }
                            if (smax >= 2) {
                                const int eb = P->bulge[2] + tau;
                                put_bulge(eb, w2[0], 0 << 5 | 2); put_if(t11, r11 + cof(w2[1]), 1 << 5 | 1); put_bulge(eb, w2[2], 2 << 5 | 0);
                            }
                            if (smax >= 3) {
                                const int eb = P->bulge[3] + tau;
                                put_bulge(eb, w3[0], 0 << 5 | 3); put_if(t12, r12 + cof(w3[1]), 1 << 5 | 2); put_if(t21, r21 + cof(w3[2]), 2 << 5 | 1); put_bulge(eb, w3[3], 3 << 5 | 0);
                            }
                            // (the words of the sizes 4 .. 6 are asked for here, behind the sizes 0 .. 3: all ten loads up front were the kernel's register peak -- with
                            // them in two halves it fits 79 VGPRs without new spills, six waves per SIMD: six workgroups per CU wherever LDS allows, windows up to 400 nt)
                            const int4a w4 = *reinterpret_cast<const int4a*>(wr[4]);
                            const int w44 = wr[4][4];
                            const int4a w5 = *reinterpret_cast<const int4a*>(wr[5]);
                            const int2a w5b = *reinterpret_cast<const int2a*>(wr[5] + 4);
                            const int4a w6 = *reinterpret_cast<const int4a*>(wr[6]), w6b = *reinterpret_cast<const int4a*>(wr[6] + 3);
                            if (smax >= 4) {
                                const int eb = P->bulge[4] + tau;
                                const int x1 = 2 * ninio;
                                const int e1 = P->internal_loop[4] + (x1 < max_ninio ? x1 : max_ninio) + o_mm1n;
                                put_bulge(eb, w4[0], 0 << 5 | 4); put_1n(e1, w4[1], 1 << 5 | 3); put_if(t22, r22 + cof(w4[2]), 2 << 5 | 2); put_1n(e1, w4[3], 3 << 5 | 1);
                                put_bulge(eb, w44, 4 << 5 | 0);
                            }
                            if (smax >= 5) {
                                const int eb = P->bulge[5] + tau;
                                const int x1 = 3 * ninio;
                                const int e1 = P->internal_loop[5] + (x1 < max_ninio ? x1 : max_ninio) + o_mm1n;
                                const int e23 = P->internal_loop[5] + ninio + o_mm23;
                                put_bulge(eb, w5[0], 0 << 5 | 5); put_1n(e1, w5[1], 1 << 5 | 4);
                                put_if(t23, e23 + (int)l_mm23[t23 * 25 + sJ[3] * 5 + sI[2]] + cof(w5[2]), 2 << 5 | 3);
                                put_if(t32, e23 + (int)l_mm23[t32 * 25 + sJ[2] * 5 + sI[3]] + cof(w5[3]), 3 << 5 | 2);
                                put_1n(e1, w5b[0], 4 << 5 | 1); put_bulge(eb, w5b[1], 5 << 5 | 0);
                            }
                            if (smax >= 6) {
                                const int eb = P->bulge[6] + tau;
                                const int x1 = 4 * ninio;
                                const int e1 = P->internal_loop[6] + (x1 < max_ninio ? x1 : max_ninio) + o_mm1n;
                                const int* kc = reinterpret_cast<const int*>(P->gen_key[0]);
                                put_bulge(eb, w6[0], 0 << 5 | 6); put_1n(e1, w6[1], 1 << 5 | 5);
                                int kg = gen(w6[2], kc[2]);
                                { const int k3 = gen(w6[3], kc[3]), k4 = gen(w6b[1], kc[4]); kg = k3 < kg ? k3 : kg; kg = k4 < kg ? k4 : kg; }
                                kg += o_mmI * 1024;
                                kmin = kg < kmin ? kg : kmin;
                                put_1n(e1, w6b[2], 5 << 5 | 1); put_bulge(eb, w6b[3], 6 << 5 | 0);
                            }
                        }
                        if (grp != 0 || smax >= 7) {          // the sizes from 7 on: 7 .. 9 ride with the small group
                            const int s_lo = grp == 0 ? 7 : grp == 1 ? 10 : grp == 2 ? 18 : 25;
                            const int s_top = grp == 0 ? 9 : grp == 1 ? 17 : grp == 2 ? 24 : 30, s_hi = smax < s_top ? smax : s_top;
                            if (!staged) sizes(s_lo, s_hi);
                            else if (span_w + s_hi + 1 > 256) sizes_staged(std::true_type{}, s_lo, s_hi);
                            else sizes_staged(std::false_type{}, s_lo, s_hi);
                        }
#ifdef MIRP_X_GEN_CLOCKS
                        if (grp == 0) { CK(ck_small); ck_nsmall++; } else { CK(ck_large); ck_nlarge++; }
#endif
                        const int best = kmin >> 10;
                        if (active && best < GEN_EMAX) atomicMin(&cbA[cell], kmin);
                    }
                }
            }
            CK(ck_large);
            }          // (d <= D)
            if (d - 1 >= TURN + 1) {
            const int d_turn = d;
            const int d = d_turn - 1;          // interval B works on the diagonal before the one interval A just did
            const int ncell = n - d;
            const unsigned char* const ctB = ctype + (d & 1) * nc;
            const int* const cbB = cbest + (d & 1) * nc;
            // (two cells per thread a pass, each step for both cells before the next, was tried: 14 spilled VGPRs under the five-wave budget and 0.168 -> 0.192 s)
            for (int cell = tid; cell < ncell; cell += GEN_NT) {
                const int i = cell + 1, j = i + d;
                const int type = ctB[cell];
                // everything that depends on the cell alone is asked for first -- the column's first GEN_PU split candidates, the DML ring, fML of diagonal
                // d - 1 -- so that the reads the candidates name are the second round trip of the cell, not the third
                const int pn = pcnt[j];
                const int2* pj = pool + (size_t)j * pcap;
                int2 en0[GEN_PU];
                static_assert(GEN_PU == 4, "the LDS copy of a column's first candidates is four packed words");
                if (n_cap <= GEN_PL_MAXN) {          // (launch-uniform) the column's first four candidates out of LDS: the reads they name go out with the cell's own
                    const int4q pk = *reinterpret_cast<const int4q*>(pl4 + 4 * j);
#pragma unroll
                    for (int u = 0; u < GEN_PU; u++) { const int v = pk[u < pn ? u : (pn > 0 ? pn - 1 : 0)]; en0[u] = make_int2((int)((unsigned)v >> 20), (v << 12) >> 12); }
                } else {
#pragma unroll
                    for (int u = 0; u < GEN_PU; u++) en0[u] = pj[u < pn ? u : (pn > 0 ? pn - 1 : 0)];          // (entries behind the last one repeat it: a minimum does not mind)
                }
                int mdec = dml[(size_t)((d - 1) & 3) * T.ld + i];
                const int m1 = T.M(d - 1, i + 1), m2 = T.M(d - 1, i);
                int best = INF;
                int code = 0;
                if (type) {
                    const int hp = e_hairpin(X, i, j, type);
                    best = hp;
                    const int key = cbB[cell];
                    const int il = key == 0x7fffffff ? INF : key >> 10;
                    if (il < best) { best = il; code = 1 + (key & 1023); }          // the hairpin does not realise c and this loop does (unless the multiloop below wins)
                    // multiloop closed by (i,j): DML(i+1, j-1), kept from diagonal d-2
                    int dec = dml[(size_t)((d - 2) & 3) * T.ld + i + 1];
                    dec += P->ML_closing + e_mlstem(P, rtype_of(type), S[j - 1], S[i + 1]);
                    if (dec < best) { best = dec; code = 0; }          // strictly better than every interior loop: the backtrack finds no loop and searches the split
                    if (best > INF) best = INF;
                }
                // DML(i,j): DML(i,j-1) and the candidates of column j (those far enough from i for fML(i,s-1) to exist), GEN_PU a turn: their entries, then
                // their fML reads, are in flight together (one at a time the loop was a chain of 2 pn dependent round trips, and interval B the longest part of
                // a window's diagonal)
                for (int k = 0; k < pn; k += GEN_PU) {
                    int2 en[GEN_PU];
#pragma unroll
                    for (int u = 0; u < GEN_PU; u++) en[u] = k == 0 ? en0[u] : pj[k + u < pn ? k + u : pn - 1];
                    int fm[GEN_PU];
#pragma unroll
                    for (int u = 0; u < GEN_PU; u++) { const int s0 = en[u].x < i + TURN + 2 ? i + TURN + 2 : en[u].x; fm[u] = T.M(s0 - 1 - i, i); }      // (a clamped read where fML(i, s-1) does not exist yet)
#pragma unroll
                    for (int u = 0; u < GEN_PU; u++) {
                        const int e = fm[u] + en[u].y;
                        if (en[u].x >= i + TURN + 2) mdec = e < mdec ? e : mdec;
                    }
                }
                if (mdec > INF) mdec = INF;
                int mm = m1;
                mm = m2 < mm ? m2 : mm;
                mm = mdec < mm ? mdec : mm;
                if (type) {
                    const int e = best + ml_term(X, i, j, type);
                    if (e < mm) {          // realised strictly by the pair term: (i,j) is a split candidate of column j
                        mm = e;
                        if (pn < pcap) {
                            pool[(size_t)j * pcap + pn] = make_int2(i, e);
                            if (n_cap <= GEN_PL_MAXN && pn < 4) pl4[4 * j + pn] = (i << 20) | (e & 0xfffff);
                            pcnt[j] = (unsigned short)(pn + 1);
                        }
                    }
                }
                if (mm > INF) mm = INF;
                T.c[(size_t)d * T.ld + i] = type ? best : INF;
                T.tb[(size_t)d * T.ld + i] = (unsigned short)code;
                {
                    const int in = rtype_of(type) * 25 + S[j + 1] * 5 + S[i - 1];          // (i,j) as the inner pair of a later loop
                    const int gq = best + (int)l_mmI[in];
                    T.w[(size_t)d * T.ld + i] = type ? (in << 24) | ((gq < GEN_PINF ? gq : GEN_PINF) & 0xffffff) : GEN_PINF;
                }
                T.m[(size_t)d * T.ld + i] = mm;
                dml[(size_t)(d & 3) * T.ld + i] = mdec;
            }
            CK(ck_b);
            }          // (interval B of d - 1)
            __syncthreads();
            CK(ck_wait);
        }
#ifdef MIRP_X_GEN_CLOCKS
        if (blockIdx.x == 0 && w == 0 && lane == 0)
            printf("[gen clocks] wave %d n %d: phase0 %lld small %lld (%d tasks) large %lld (%d tasks) B %lld barrier-wait %lld\n", wave, n, ck_0, ck_small, ck_nsmall, ck_large, ck_nlarge, ck_b, ck_wait);
#endif
        } else {
#ifdef MIRP_X_GEN_NOEPI             // timing experiment: fill only (results are empty)
        if (tid == 0) { out_nlines[win] = 0; out_mfe[win] = 0; out_status[win] = 0; }
#else
        fold_epilogue<GTab, GEN_NT>(X, T, span, f3, starts, lens, btbuf, nc, btstk, sh_misc, win, max_lines, ss_stride,
                                   out_lines, out_ss, out_nlines, out_mfe, out_status);
#endif
        }
    }
}

// split-candidate counts, interior-loop minima, paired-cell list and pair types of a diagonal behind the base carve-up
static size_t fold_generic_lds_bytes_fill(int n_cap) {
    return fold_generic_lds_bytes_base_fill(n_cap) + GEN_AUX_BYTES((size_t)(n_cap + 8)) + sizeof(short) * (5 * 200 + 64) + sizeof(int) * (GEN_NT / 64) * (GEN_STAGE + 2) + 16 +
           (n_cap <= GEN_PL_MAXN ? sizeof(int) * 4 * (size_t)(n_cap + 8) : 0);
}
// what the larger of the two kernels takes (the budget check of the caller)
size_t fold_generic_lds_bytes(int n_cap, int max_lines) {
    const size_t a = fold_generic_lds_bytes_fill(n_cap), b = fold_generic_lds_bytes_base(n_cap, max_lines);
    return a > b ? a : b;
}

size_t fold_generic_ws_slot_ints(int n_cap, int span) {
    // c, fML, four diagonals of DML, the candidate pool (two ints per entry), the trace-back codes (a short per cell)
    const size_t ld = (size_t)fold_generic_ld(n_cap);
    return 3 * fold_generic_table_ints(n_cap, span) + ((4 * ld + 2 * ld * (size_t)fold_generic_pool_cap(n_cap, span) + 63) & ~(size_t)63) +
           ((fold_generic_table_ints(n_cap, span) / 2 + 63) & ~(size_t)63) + 64;          // (+ w: the interior-loop interval's view of c)
}

void launch_fold_generic(hipStream_t stream, int grid, const FoldParams* P, const unsigned char* seqs, const long long* offs,
                         const int* lens, const int* work_list, int n_work, int span, int n_cap, int* ws, size_t ws_slot_ints, int max_lines,
                         int ss_stride, MirpFoldLine* out_lines, char* out_ss, int* out_nlines, int* out_mfe, int* out_status) {
    // batches of `grid` windows: window b + k of the batch owns workspace slot k in both kernels (without a work list the kernels index the windows
    // directly, so the batch is addressed by shifted array bases; with one, by the shifted list)
    const size_t lds1 = fold_generic_lds_bytes_fill(n_cap), lds2 = fold_generic_lds_bytes_base(n_cap, max_lines);
    // windows of some 2,000 nt and more (PRECURSOR_LEN up to 3000, MP:167-184): one or two workgroups per CU
    if (lds1 > 64 * 1024) (void)hipFuncSetAttribute((const void*)fold_generic_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (lds2 > 64 * 1024) (void)hipFuncSetAttribute((const void*)fold_generic_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    for (int b = 0; b < n_work; b += grid) {
        const int nb = n_work - b < grid ? n_work - b : grid;
        const int* wl = work_list ? work_list + b : nullptr;
        const long long* o2 = work_list ? offs : offs + b;
        const int* l2 = (work_list || !lens) ? lens : lens + b;
        MirpFoldLine* ol = work_list ? out_lines : out_lines + (size_t)b * max_lines;
        char* os = work_list ? out_ss : out_ss + (size_t)b * max_lines * ss_stride;
        int* on = work_list ? out_nlines : out_nlines + b;
        int* om = work_list ? out_mfe : out_mfe + b;
        int* ost = work_list ? out_status : out_status + b;
        hipLaunchKernelGGL(fold_generic_kernel<1>, dim3(nb), dim3(GEN_NT), lds1, stream, P, seqs, o2, l2, wl, nb, span, n_cap, ws, ws_slot_ints, max_lines, ss_stride,
                           ol, os, on, om, ost);
        hipLaunchKernelGGL(fold_generic_kernel<2>, dim3(nb), dim3(GEN_NT), lds2, stream, P, seqs, o2, l2, wl, nb, span, n_cap, ws, ws_slot_ints, max_lines, ss_stride,
                           ol, os, on, om, ost);
    }
}

}  // namespace mirp
