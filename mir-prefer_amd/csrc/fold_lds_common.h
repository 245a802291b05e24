// Shared definitions of the LDS-resident fill kernels (fold_lds_kernel.hip: one diagonal per barrier interval, vienna-1.8.5 model;
// fold_lds2_kernel.hip: two diagonals per barrier interval, default model): LDS table copies, triangle / archive layouts and the
// phase-A1 candidate jobs (lane = paired cell, wave = candidate group).  Reference work unit: RNALfold -L, /root/reference/miR_PREFeR.py:3053.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdio>
#include "fold_epilogue.h"
#include "fold185_device.h"

namespace mirp {

#define LNT 1024
#define LNW (LNT / 64)
#define LCAP 352            // window length capacity
#define LDMAX 300           // diagonals 4..LDMAX are allocated: pair distances up to span-1 = 299, plus the fML-only diagonal d = span of the vienna-1.8.5 model
#define LSPAN 300           // largest span (-L) this kernel supports
#define I16_INF 0x7fff
#define FIN_LIMIT 28000      // finite c must stay in [-28000, 28000]: G0 + 32768 + any loop term then stays below 65535
#define FML_BIAS 31500       // fML is kept in LDS as uint16 (value + FML_BIAS), 65535 = INF; finite fML must stay in [-31500, 1267] so that the sum of
#define FML_MAX 1267         // two finite entries (<= 65534) can never be mistaken for a sum that involves INF (>= 65535)
#define LSEG 384             // paired-cell list: 6 producer waves x 64 entries
#define KEY_BIAS 40000        // candidate keys: (energy + KEY_BIAS) << 10 | n1 << 5 | n2, 0xffffffff = none
#define KEY_NONE 0xffffffffu
#define KEY_INF (65535u << 10) // running-minimum start inside a job: any key that involves an INF ring entry is >= this
#define OTH_BIAS 2048        // keeps (table delta + size term) of a bulge / 1xn candidate non-negative (the FoldParams key tables carry it)

struct LdsTables {          // int16 copies of the hot parameter tables
    short stack[64];
    short bulge[32];
    short internal_loop[32];
    short mismatchI[200], mismatchH[200], mismatchM[200], mismatch1nI[200], mismatch23I[200];
    // inner-pair terms relative to G0, as bytes (term + FoldParams::xb_bias / x1_bias), at index xt_pcode(p) + xt_qcode(q) (below): 175 bytes = 44
    // dwords per table, fewer than the LDS has banks -- a wave's gather never has two lanes on different dwords of one bank.  (Before: 625
    // shorts at PA(p) * 25 + QB(q); the 3- to 4-way conflicts of those gathers cost 3.3 ms of the kernel: timing build -DMIRP_X_XBBCAST.)
    unsigned char XB[176];  // TerminalAU(inner) - mismatchI(inner)           (bulges of size >= 2)
    unsigned char X1[176];  // mismatch1nI(inner) - mismatchI(inner)          (1 x n loops, n >= 3)
    short dangle5[40], dangle3[40];   // [type*5 + base], clamped <= 0 (vienna-1.8.5 model)
    unsigned char rt2[28];  // rtype(pair_type(a, b)) at [a*5+b]
    short ML_closing, ML_intern, TerminalAU, ninio, MAX_NINIO, pad[3];
};
// Index of an inner pair (p, q) in XB / X1: slot(S[p], S[q]) * 25 + S[q+1] * 5 + S[p-1], where the slot is a SUM of a p part and a q part that
// is distinct for the six pair types (A-U 1, C-G 2, G-C 3, G-U 4, U-A 5, U-G 6; what other base combinations land on does not matter, their ring
// entry is INF) -- so the index is pcode(p) + qcode(q) of two per-position codes, as before, but the table has 7 x 25 entries instead of 625.
__host__ __device__ constexpr int xt_pslot(int s) { return s == 3 ? 3 : s == 4 ? 4 : 0; }                      // A 0, C 0, G 3, U 4
__host__ __device__ constexpr int xt_qslot(int s) { return s == 1 ? 1 : s == 3 ? 2 : s == 4 ? 1 : 0; }        // A 1, C 0, G 2, U 1
__host__ __device__ constexpr int xt_pcode(int sp, int sp_m1) { return xt_pslot(sp) * 25 + sp_m1; }
__host__ __device__ constexpr int xt_qcode(int sq, int sq_p1) { return xt_qslot(sq) * 25 + sq_p1 * 5; }
static_assert(xt_pslot(1) + xt_qslot(4) == 1 && xt_pslot(2) + xt_qslot(3) == 2 && xt_pslot(3) + xt_qslot(2) == 3 && xt_pslot(3) + xt_qslot(4) == 4 &&
              xt_pslot(4) + xt_qslot(1) == 5 && xt_pslot(4) + xt_qslot(3) == 6, "inner-pair table slots");
template <class TAB>
__device__ inline void xt_fill(TAB& T, const FoldParams* __restrict__ P, int tid, int nt) {
    for (int x = tid; x < 175; x += nt) {
        const int slot = x / 25, a = (x % 25) / 5, b = x % 5;          // a = S[q+1], b = S[p-1]
        const int sp = slot == 1 ? 1 : slot == 2 ? 2 : (slot == 3 || slot == 4) ? 3 : 4, sq = slot == 1 ? 4 : slot == 2 ? 3 : slot == 3 ? 2 : slot == 4 ? 4 : slot == 5 ? 1 : 3;
        const int t2 = slot ? rtype_of(pair_type(sp, sq)) : 0;
        int xb = 0, x1 = 0;
        if (t2) {
            const int mi = P->mismatchI[t2][a][b];
            xb = (t2 > 2 ? P->TerminalAU : 0) - mi;
            x1 = P->mismatch1nI[t2][a][b] - mi;
        }
        T.XB[x] = (unsigned char)(xb + P->xb_bias); T.X1[x] = (unsigned char)(x1 + P->x1_bias);
    }
}
#define CSTR 354            // c-ring row stride in shorts (177 dwords: odd, spreads LDS banks)
static_assert(CSTR == MIRP_RING_CSTR, "FoldParams::ring_rowoff is built for this row stride");
#define MIRP_CK(d) ((d) % 3)
#define CRING_ROWS 33       // diagonal dd lives in row dd & 31; row 32 mirrors row 0, so "the row after row r" is always r + 1 (phase A1 mixes lanes of two diagonals)

struct LTab {               // table accessors for the shared epilogue/backtrack
    static constexpr bool kTiled = true;   // 8 x 8 tiles over (row, diagonal): see the archive layout below
    const short* fml;       // fML slab (per-window, global), tiled like the other two
    const int* off;         // LDS: rowblk_off of the tiled archive layout (arch_rowblk_off)
    const short* carch;     // global archive of c
    const unsigned short* tb;   // trace-back codes written by the fill kernel
    __device__ __forceinline__ int at(int d, int i) const { return off[(i - 1) >> 3] + ((i - 1) & 7) + 8 * (d - 4); }
    __device__ __forceinline__ int TB(int d, int i) const { return tb[at(d, i)]; }
    __device__ __forceinline__ int C(int d, int i) const { int v = carch[at(d, i)]; return v == I16_INF ? INF : v; }
    // the three tables share their offsets: a patch computes at() once
    __device__ __forceinline__ int TBat(int o) const { return tb[o]; }
    __device__ __forceinline__ int Cat(int o) const { int v = carch[o]; return v == I16_INF ? INF : v; }
    __device__ __forceinline__ int Mat(int o) const { const int v = (unsigned short)fml[o]; return v == 65535 ? INF : v - FML_BIAS; }
    __device__ __forceinline__ int M(int d, int i) const {
        if (d < 4) return INF;
        const int v = (unsigned short)fml[at(d, i)];
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

// Diagonal-major triangle of fML / c / trace-back: cell (d, i), i = 1..n-d, lives at off(d) + i.  Every diagonal starts at an ODD offset and
// is padded to an even length, so a pair of cells (i, i+1) with odd i is one naturally aligned 32-bit word (the split loop reads pairs; a
// misaligned 32-bit DS read is replayed on gfx950).  off(d) = 1 + (d-4) n - (d(d-1)/2 - 6) + #{odd lengths among diagonals 4..d-1}.
__host__ __device__ constexpr int tri_off(int d, int n) { return 1 + (d - 4) * n - (d * (d - 1) / 2 - 6) + (((d - 4) + (n & 1)) >> 1); }
__host__ __device__ constexpr int tri_len(int d, int n) { return n - d > 0 ? (n - d) + ((n - d) & 1) : 0; }
// tri_off(d + 1, n) - tri_off(d, n) of the closed form, for any d (also below the first diagonal, where tri_len is cut off)
__host__ __device__ constexpr int tri_len_any(int d, int n) { return (n - d) + ((n - d) & 1); }
__device__ inline void fill_tri_off(int* off, int n) {
    int o = 1;
    for (int d = 4; d <= LDMAX + 1; d++) { off[d] = o; o += tri_len(d, n); }
}

// Archive layout in HBM (the c, fML and trace-back slabs a window hands to its epilogue): 8 x 8 tiles over (row i, diagonal d), one tile =
// one 128-byte line.  Rows are cut into row blocks of 8 (block b = rows 8b+1 .. 8b+8); a block holds the diagonals 4 .. min(dcap, n-1-8b) its
// first row can have (rounded up to whole tiles), its tiles follow each other by diagonal:
//     arch(d, i) = rowblk_off[(i-1) >> 3] + ((i-1) & 7) + 8 (d - 4),      rowblk_off[b] = 8 * sum of arch_nd over the blocks below b.
// The epilogue's accesses are runs along a row (partner scans), along a column (multiloop splits), along a helix (i+l, j-l) and the exterior
// sweep's (row block x all diagonals) streams; in a diagonal-major triangle each of those touched one line per cell.  The fill kernel writes one
// diagonal per interval: 16-byte runs that the L2 merges into whole lines over the next seven diagonals.
#define ARCH_RB ((LCAP + 7) / 8)
__host__ __device__ constexpr int arch_nd(int b, int n, int dcap) {
    int m = n - 1 - 8 * b;
    if (m > dcap) m = dcap;
    m -= 3;
    return m > 0 ? (m + 7) & ~7 : 0;
}
__host__ __device__ inline int arch_rowblk_off(int b, int n, int dcap) {
    int o = 0;
    for (int x = 0; x < b; x++) o += 8 * arch_nd(x, n, dcap);
    return o;
}

struct LdsLayout {
    unsigned fml, aux, S, seq, pax, qb2, list, tabs, misc, total, fml_bytes, code4;
};
#ifndef MIRP_A1_CODES4
#define MIRP_A1_CODES4 0      // fold_lds_kernel.hip sets it: the bulge / 1xn jobs read their pair codes four per ds_read_b32 (see a1_codes4)
#endif
#define CODE_STR 368          // bytes per shifted copy of a pair-code array (LCAP + 8 codes + the last job's over-read of 3), a multiple of 16
__host__ __device__ constexpr unsigned lds_al(unsigned x) { return (x + 15u) & ~15u; }
// SPARSE: the multiloop splits run over a pool of split candidates (see "sparse splits" in fold_lds_kernel.hip): a third mdec buffer, and the pool
// lives behind the window's fML triangle inside the fml region, which then takes everything the other arrays leave of the 160 KB.
#define FML_REGION_BYTES (lds_al((tri_off(LDMAX + 1, LCAP) + 2) * 2))      // fML triangle, d = 4..LDMAX at n = LCAP
#define POOL_MIN_CAP 1024    // a window whose length leaves room for fewer candidates goes to the dense kernel right away
template <int MODEL, bool SPARSE = false>
__host__ __device__ constexpr LdsLayout lds_layout() {
    LdsLayout L{};
    unsigned o = 0;
    if (!SPARSE) { L.fml = o; L.fml_bytes = FML_REGION_BYTES; o += L.fml_bytes; }
    L.aux = o; o += lds_al(CRING_ROWS * CSTR * 2 + (MODEL ? 5 : 3) * LCAP * 2 + (SPARSE ? 6 : 5) * LCAP * 4);   // c ring (32 diagonals + mirror row), DML ring (3; 5 in the vienna-1.8.5 model), 3 x ckey, 2 (sparse: 3) x mdec
    L.S = o; o += lds_al(LCAP + 8);
    L.seq = o; o += lds_al(LCAP + 8);
    if (MIRP_A1_CODES4 && SPARSE) {      // 4 byte-shifted copies of the q codes, then 4 of the p codes (bytes); copy 0 of each IS the array
        L.code4 = o; L.qb2 = o; L.pax = o + 4 * CODE_STR; o += 8 * CODE_STR;
    } else if (MIRP_A1_CODES4) {         // the dense instantiations (overflow pass) have no room for the copies: byte arrays, codes read one by one
        L.pax = o; o += CODE_STR;
        L.qb2 = o; o += CODE_STR;
    } else {
        L.pax = o; o += lds_al((LCAP + 8) * 2);
        L.qb2 = o; o += lds_al(LCAP + 8);
    }
    L.list = o; o += lds_al(3 * LSEG * 4);      // 32-bit entries (see `list` in the kernel)
    L.tabs = o; o += lds_al((unsigned)sizeof(LdsTables));
    L.misc = o; o += lds_al((48 + ARCH_RB + ((SPARSE && MODEL) ? 48 : 0)) * 4);      // (vienna-1.8.5 candidate pass: + the 4 x 11-word bitmap of pooled pairs)
    if (SPARSE) { L.fml = o; L.fml_bytes = 160u * 1024u - o; o += L.fml_bytes; }
    L.total = o;
    return L;
}
static_assert(lds_layout<0>().total <= 160 * 1024 && lds_layout<1>().total <= 160 * 1024, "fill kernel LDS budget");

// ---- phase A1 building blocks.  All take the lane's paired cell (i, j = i + d) and wave-uniform d; r0 = d - 2 (ring row of the
// stacked pair), um = largest admissible n1 + n2 (inner pair keeps q - p >= TURN + 1).  Running minima are biased uint (65535 = none).
typedef const volatile __attribute__((address_space(3))) unsigned short* lds_vu16;   // LDS reads that must stay narrow (see a1_gen_row)
typedef const volatile __attribute__((address_space(3))) unsigned char* lds_vu8;
typedef const volatile __attribute__((address_space(3))) unsigned* lds_vu32;
#ifndef MIRP_A1_SMALLPACK
#define MIRP_A1_SMALLPACK 10     // short generic rows (U < 13) from this size on through realigned dwords as well (8: no gain, 10: -0.3 %)
#endif
#ifndef MIRP_A1_WPACK
#define MIRP_A1_WPACK 2      // 0: 16-bit reads of the generic rows (rounds 3-5), 1: the wings as aligned dwords, 2: whole rows as aligned dwords (round 6)
#endif
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#if MIRP_A1_CODES4
typedef unsigned char pax_t;          // the p codes (xt_pcode < 256) as bytes: copy 0 of their four shifted copies
typedef lds_vu8 lds_vpax;
#else
typedef unsigned short pax_t;
typedef lds_vu16 lds_vpax;
#endif
struct A1 {
    const FoldParams* __restrict__ P;
    const LdsTables* T;
    const unsigned char* S;
    const unsigned short* cring;
    const pax_t* pax;              // xt_pcode(S[x], S[x-1])
    const unsigned char* qbr;      // xt_qcode(S[x], S[x+1]) at n + 1 - x: the q side is walked downwards, so it is stored reversed (ascending immediates)
    const unsigned char* code4 = nullptr;      // MIRP_A1_CODES4: [4][CODE_STR] copy c of qbr shifted by c bytes, then the same of the p codes as bytes
    int r0, um, n;
    const int* __restrict__ rowtab;    // FoldParams::ring_rowoff[r0 & 31]: ring-row offset (shorts) of loop size u, i.e. ((r0 - u) & 31) * CSTR
    const unsigned short* rba = nullptr;       // a1_gen_row_w: cring + i + 1 rounded down to a dword, and the end masks of the lane's parity
    unsigned w_lo = 0, w_hie = 0, w_hio = 0, w_sh = 0;
};

// generic loops (n1, n2 >= 2) of size U: one contiguous run of ring row r0 - U.  The reads stay 16-bit on purpose (volatile keeps the
// compiler from fusing neighbours into b64/b128 reads): a lane's run starts at an arbitrary 2-byte boundary, and a wide DS read off its
// natural alignment is replayed at 64 cycles per wave-instruction on gfx950, against 2 cycles for a 16-bit read.
template <bool CHECK, int U>
__device__ __forceinline__ void a1_gen_row(const A1& a, const unsigned short* rb, unsigned& bg) {
    if constexpr (MIRP_A1_WPACK == 2 && !CHECK && U >= 8) {
        // (round 6) the row as aligned dwords, every entry out of a realigned register (see a1_gen_row_w): half the LDS passes of 16-bit reads
        lds_vu32 rq = (lds_vu32)(a.rba + a.rowtab[U]);
        constexpr int m1 = (U - 2) >> 1;
        unsigned dw[m1 + 2];
#pragma unroll
        for (int k = 1; k <= m1 + 1; k++) dw[k] = rq[k];
        const unsigned c1024 = 1024u;
#pragma unroll
        for (int m = 1; m <= m1; m++) {
            const unsigned re = __builtin_amdgcn_alignbit(dw[m + 1], dw[m], a.w_sh);      // entries 2 m, 2 m + 1
            unsigned e;
            asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(e) : "v"(re), "v"(c1024), "s"(a.P->gen_key[U - 6][2 * m]));
            bg = e < bg ? e : bg;
            if (2 * m + 1 <= U - 2) {
                asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(e) : "v"(re), "v"(c1024), "s"(a.P->gen_key[U - 6][2 * m + 1]));
                bg = e < bg ? e : bg;
            }
        }
    } else
    if (!CHECK || U <= a.um) {
        lds_vu16 rp = (lds_vu16)(rb + a.rowtab[U]);
        unsigned v[U - 3];      // all reads of the run in flight before the first use
#pragma unroll
        for (int n1 = 2; n1 <= U - 2; n1++) v[n1 - 2] = rp[n1];
#pragma unroll
        for (int n1 = 2; n1 <= U - 2; n1++) {
            const unsigned e = (v[n1 - 2] << 10) + a.P->gen_key[U - 6][n1];
            bg = e < bg ? e : bg;
        }
    }
}
// job-local key (term << 10 | code) -> cell key ((energy + KEY_BIAS) << 10 | code); `adj` turns the job's term into the loop energy
__device__ __forceinline__ unsigned a1_key(unsigned b, int adj) {
    return b >= KEY_INF ? KEY_NONE : ((unsigned)((int)(b >> 10) + adj + KEY_BIAS) << 10) | (b & 1023u);
}
// The same row when the asymmetry term saturates at |n1 - n2| >= WD (FoldParams::gen_wing_d; WD = 5 with Turner-2004): the candidates
// beyond that -- the two wings of the row -- share one term, so their ring entries are minimised as they are (two per v_min3_u32) and the term
// is added once; only the 2 WD - 1 or fewer candidates around n1 = n2 pay an add of their own.  The wing minimum names no shape: its key
// carries code 63, and a cell that ends up with that code has its loop found by the epilogue (TB_GENERIC, bt_search_unnamed) -- rarely,
// MFE structures seldom hold loops that lopsided.  A row of U has U - 3 candidates: 54 VALU instructions before, 20 now at U = 30.
// Round 6: the wings as ALIGNED 32-bit reads, two ring entries each.  A wing's minimum does not care which entry came from where, so the reads need not start
// at the lane's own (2-byte) boundary: they cover the wing from the aligned dword below it -- an entry more at the inner end lands on a centre candidate,
// whose own key is smaller than the wing key of the same entry, at the outer ends the lane's parity masks (A1::w_lo / w_hie / w_hio) turn the 1 x n / bulge
// neighbours into INF -- and v_pk_min_u16 folds two entries per instruction as v_min3_u32 did.  A 16-bit read moves 2 of the 4 bytes a lane's bank slot
// carries: 27 reads of row 30 become 17.  Only rows whose wings hold three entries or more (U >= 14), not on the first diagonals (CHECK).
template <bool CHECK, int WD, int U>
__device__ __forceinline__ void a1_gen_row_w(const A1& a, const unsigned short* rb, unsigned& bg) {
    constexpr int nL = (U - WD) / 2, nR = (U + WD + 1) / 2;      // last entry of the left wing, first of the right one
    if constexpr (MIRP_A1_WPACK == 2 && !CHECK && nL < 4 && U >= MIRP_A1_SMALLPACK) {
        // short rows (wings of at most two entries): every entry out of a realigned register, the wing entries with the wing's key term
        lds_vu32 rq = (lds_vu32)(a.rba + a.rowtab[U]);
        constexpr int m1 = (U - 2) >> 1;
        unsigned dw[m1 + 2];
#pragma unroll
        for (int k = 1; k <= m1 + 1; k++) dw[k] = rq[k];
        const unsigned c1024 = 1024u;
#pragma unroll
        for (int m = 1; m <= m1; m++) {
            const unsigned re = __builtin_amdgcn_alignbit(dw[m + 1], dw[m], a.w_sh);      // entries 2 m, 2 m + 1
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int n1 = 2 * m + h, dd = 2 * n1 - U;
                if (n1 <= U - 2) {
                    const unsigned kt = (dd <= -WD || dd >= WD) ? a.P->gen_wing_key[U - 6] : a.P->gen_key[U - 6][n1];
                    unsigned e;
                    if (h) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(e) : "v"(re), "v"(c1024), "s"(kt));
                    else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(e) : "v"(re), "v"(c1024), "s"(kt));
                    bg = e < bg ? e : bg;
                }
            }
        }
    } else
    if constexpr (MIRP_A1_WPACK == 2 && !CHECK && nL >= 4) {
        // the whole row as contiguous aligned dwords; the centre's entries come out of the same registers: v_alignbit by the lane's parity puts entries
        // (2 m, 2 m + 1) into one register, v_mad_u32_u16 takes either half (op_sel) times 1024 plus the candidate's scalar key term
        lds_vu32 rq = (lds_vu32)(a.rba + a.rowtab[U]);
        constexpr int kL1 = (1 + nL) >> 1, kR0 = nR >> 1, kR1 = (U - 1) >> 1;
        constexpr int m0 = (nL + 1) >> 1, m1 = (nR - 1) >> 1;
        static_assert(m1 + 1 <= kR1, "the centre's realigned dwords lie inside the row's reads");
        unsigned dw[kR1 + 1];
#pragma unroll
        for (int k = 1; k <= kR1; k++) dw[k] = rq[k];
        unsigned re[m1 - m0 + 1];
#pragma unroll
        for (int m = m0; m <= m1; m++) re[m - m0] = __builtin_amdgcn_alignbit(dw[m + 1], dw[m], a.w_sh);
        us2 mm, t;
        { const unsigned f = dw[1] | a.w_lo; __builtin_memcpy(&mm, &f, 4); }
#pragma unroll
        for (int k = 2; k <= kL1; k++) { __builtin_memcpy(&t, &dw[k], 4); mm = __builtin_elementwise_min(mm, t); }
#pragma unroll
        for (int k = kR0; k < kR1; k++) { __builtin_memcpy(&t, &dw[k], 4); mm = __builtin_elementwise_min(mm, t); }
        { const unsigned l = dw[kR1] | ((U & 1) ? a.w_hio : a.w_hie); __builtin_memcpy(&t, &l, 4); mm = __builtin_elementwise_min(mm, t); }
        const unsigned w0 = mm[0], w1 = mm[1], w = w0 < w1 ? w0 : w1;
        { const unsigned k = (w << 10) + a.P->gen_wing_key[U - 6]; bg = k < bg ? k : bg; }
        const unsigned c1024 = 1024u;
#pragma unroll
        for (int n1 = nL + 1; n1 < nR; n1++) {
            unsigned e;
            if (n1 & 1) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(e) : "v"(re[(n1 >> 1) - m0]), "v"(c1024), "s"(a.P->gen_key[U - 6][n1]));
            else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(e) : "v"(re[(n1 >> 1) - m0]), "v"(c1024), "s"(a.P->gen_key[U - 6][n1]));
            bg = e < bg ? e : bg;
        }
    } else
    if constexpr (MIRP_A1_WPACK && !CHECK && nL >= 4) {
        lds_vu16 rp = (lds_vu16)(rb + a.rowtab[U]);
        lds_vu32 rq = (lds_vu32)(a.rba + a.rowtab[U]);           // the aligned dword at or below entry 0 (row offsets are whole dwords)
        constexpr int kL0 = 1, kL1 = (1 + nL) >> 1, kR0 = nR >> 1, kR1 = (U - 1) >> 1;
        static_assert(kL1 < kR0 && nR - nL - 1 >= 2, "the wings' dwords stay apart and over-cover centre candidates only");
        unsigned wl[kL1 - kL0 + 1], wr[kR1 - kR0 + 1], c[nR - nL - 1];
#pragma unroll
        for (int k = kL0; k <= kL1; k++) wl[k - kL0] = rq[k];
#pragma unroll
        for (int n1 = nL + 1; n1 < nR; n1++) c[n1 - nL - 1] = rp[n1];
#pragma unroll
        for (int k = kR0; k <= kR1; k++) wr[k - kR0] = rq[k];
        wl[0] |= a.w_lo;
        wr[kR1 - kR0] |= (U & 1) ? a.w_hio : a.w_hie;
        us2 m, t;
        __builtin_memcpy(&m, &wl[0], 4);
#pragma unroll
        for (int k = 1; k <= kL1 - kL0; k++) { __builtin_memcpy(&t, &wl[k], 4); m = __builtin_elementwise_min(m, t); }
#pragma unroll
        for (int k = 0; k <= kR1 - kR0; k++) { __builtin_memcpy(&t, &wr[k], 4); m = __builtin_elementwise_min(m, t); }
        const unsigned w0 = m[0], w1 = m[1], w = w0 < w1 ? w0 : w1;
        { const unsigned k = (w << 10) + a.P->gen_wing_key[U - 6]; bg = k < bg ? k : bg; }
#pragma unroll
        for (int n1 = nL + 1; n1 < nR; n1++) { const unsigned e = (c[n1 - nL - 1] << 10) + a.P->gen_key[U - 6][n1]; bg = e < bg ? e : bg; }
    } else
    if (!CHECK || U <= a.um) {
        lds_vu16 rp = (lds_vu16)(rb + a.rowtab[U]);
        unsigned v[U - 3];      // all reads of the run in flight before the first use
#ifdef MIRP_X_NOGENLDS          // timing experiment: the row's values without touching LDS
#pragma unroll
        for (int n1 = 2; n1 <= U - 2; n1++) v[n1 - 2] = 50000u + (((unsigned)(size_t)rp + (unsigned)n1) & 1023u);
#else
#pragma unroll
        for (int n1 = 2; n1 <= U - 2; n1++) v[n1 - 2] = rp[n1];
#endif
#ifdef MIRP_X_NOGENVALU         // timing experiment: the reads, folded with the fewest instructions that keep them alive
        { unsigned x = 0;
#pragma unroll
          for (int n1 = 2; n1 <= U - 2; n1++) x |= v[n1 - 2];
          bg = (x << 10) < bg ? (x << 10) : bg; return; }
#endif
        unsigned w = 65535u;    // wing minimum, ring units
        unsigned e[2 * WD];     // keys of the centre candidates
        int ne = 0;
#pragma unroll
        for (int n1 = 2; n1 <= U - 2; n1++) {
            const int dd = 2 * n1 - U;      // n1 - n2
            if (dd <= -WD || dd >= WD) w = v[n1 - 2] < w ? v[n1 - 2] : w;
            else e[ne++] = (v[n1 - 2] << 10) + a.P->gen_key[U - 6][n1];
        }
        if (U - 4 >= WD) { const unsigned k = (w << 10) + a.P->gen_wing_key[U - 6]; bg = k < bg ? k : bg; }      // the row has wings
#pragma unroll
        for (int k = 0; k < ne; k++) bg = e[k] < bg ? e[k] : bg;
    }
}
template <bool CHECK, int... Us>
__device__ __forceinline__ unsigned a1_generic(const A1& a0, int i, int j, int mm_outer) {
    unsigned bg = KEY_INF;
    const unsigned short* rb = a0.cring + i + 1;
    A1 a = a0;
    if constexpr (MIRP_A1_WPACK == 2 && !CHECK) {
        const unsigned par = ((unsigned)(size_t)(lds_vu16)rb >> 1) & 1u;
        a.rba = rb - par; a.w_sh = par * 16u;
    }
    (a1_gen_row<CHECK, Us>(a, rb, bg), ...);
    return a1_key(bg, -32768 + mm_outer);      // mm_outer = mismatchI of the outer pair (i, j), fetched by the caller ahead of the rows
}
template <bool CHECK, int WD, int... Us>
__device__ __forceinline__ unsigned a1_generic_w(const A1& a0, int i, int j, int mm_outer) {
    unsigned bg = KEY_INF;
#ifdef MIRP_X_GENBCAST          // timing experiment: every lane reads the same ring columns (no bank conflicts in the generic rows)
    const unsigned short* rb = a0.cring + 40 + (i & 1);
#else
    const unsigned short* rb = a0.cring + i + 1;
#endif
    A1 a = a0;
    if constexpr (MIRP_A1_WPACK && !CHECK) {
        // entry n1 of a row sits at halfword (par + n1) of the aligned base: the dword below the left wing holds entry 1 (1 x n) when par is set, the dword
        // at the right wing's end entry U - 1 (U even, par clear), entries U - 1 and U (U odd, par clear) or entry U - 1 (U odd, par set)
        const unsigned par = ((unsigned)(size_t)(lds_vu16)rb >> 1) & 1u;
        a.rba = rb - par;
        a.w_lo = par ? 0xffffu : 0u; a.w_hie = par ? 0u : 0xffff0000u; a.w_hio = par ? 0xffff0000u : 0xffffffffu; a.w_sh = par * 16u;
    }
    (a1_gen_row_w<CHECK, WD, Us>(a, rb, bg), ...);
    return a1_key(bg, -32768 + mm_outer);
}

// bulges, n1 = 0, n2 = U in [LO, HI]: p = i+1, q = j-1-U
template <bool CHECK, int LO, int HI>
__device__ __forceinline__ void a1_b0(const A1& a, int i, int j, unsigned& best) {
    const unsigned idxp = a.pax[i + 1];
    lds_vu8 ql = (lds_vu8)(a.qbr + (a.n + 2 - j));
    const unsigned short* rb = a.cring + i + 1;
    const unsigned char* xb = a.T->XB;
#pragma unroll
    for (int U = LO; U <= HI; U++) {
        if (!CHECK || U <= a.um) {
            const unsigned idx2 = idxp + ql[U];
            const int x = xb[idx2];
            const unsigned g = rb[((a.r0 - U) & 31) * CSTR];
            const unsigned e = ((g + (unsigned)x) << 10) + a.P->kb0_key[U];
            best = e < best ? e : best;
        }
    }
}
// bulges, n2 = 0, n1 = U: p = i+1+U, q = j-1
template <bool CHECK, int LO, int HI>
__device__ __forceinline__ void a1_b1(const A1& a, int i, int j, unsigned& best) {
    const unsigned idxq = a.qbr[a.n + 2 - j];
    lds_vpax pl = (lds_vpax)(a.pax + i + 1);
    const unsigned short* rb = a.cring + i + 1;
    const unsigned char* xb = a.T->XB;
#pragma unroll
    for (int U = LO; U <= HI; U++) {
        if (!CHECK || U <= a.um) {
            const unsigned idx2 = idxq + pl[U];
            const int x = xb[idx2];
            const unsigned g = rb[((a.r0 - U) & 31) * CSTR + U];
            const unsigned e = ((g + (unsigned)x) << 10) + a.P->kb1_key[U];
            best = e < best ? e : best;
        }
    }
}
// 1 x K loops, n1 = 1, n2 = K in [LO, HI]: p = i+2, q = j-1-K
template <bool CHECK, int LO, int HI>
__device__ __forceinline__ void a1_i0(const A1& a, int i, int j, unsigned& best) {
    const unsigned idxp = a.pax[i + 2];
    lds_vu8 ql = (lds_vu8)(a.qbr + (a.n + 2 - j));
    const unsigned short* rb = a.cring + i + 2;
    const unsigned char* xb = a.T->X1;
#pragma unroll
    for (int K = LO; K <= HI; K++) {
        if (!CHECK || K + 1 <= a.um) {
            const unsigned idx2 = idxp + ql[K];
            const int x = xb[idx2];
            const unsigned g = rb[((a.r0 - K - 1) & 31) * CSTR];
            const unsigned e = ((g + (unsigned)x) << 10) + a.P->k1n0_key[K];
            best = e < best ? e : best;
        }
    }
}
// K x 1 loops, n2 = 1, n1 = K: p = i+1+K, q = j-2
template <bool CHECK, int LO, int HI>
__device__ __forceinline__ void a1_i1(const A1& a, int i, int j, unsigned& best) {
    const unsigned idxq = a.qbr[a.n + 3 - j];
    lds_vpax pl = (lds_vpax)(a.pax + i + 1);
    const unsigned short* rb = a.cring + i + 1;
    const unsigned char* xb = a.T->X1;
#pragma unroll
    for (int K = LO; K <= HI; K++) {
        if (!CHECK || K + 1 <= a.um) {
            const unsigned idx2 = idxq + pl[K];
            const int x = xb[idx2];
            const unsigned g = rb[((a.r0 - K - 1) & 31) * CSTR + K];
            const unsigned e = ((g + (unsigned)x) << 10) + a.P->k1n1_key[K];
            best = e < best ? e : best;
        }
    }
}

// the nine small shapes: full energy function, branch-free so the global table loads of one wave are issued together
template <int N1, int N2>
__device__ __forceinline__ void a1_small(const A1& a, int i, int j, int type, int si1, int sj1, unsigned& best) {
    if (N1 + N2 <= a.um) {
        const int p = i + 1 + N1, q = j - 1 - N2;
        const unsigned g = a.cring[((a.r0 - N1 - N2) & 31) * CSTR + p];
        const int sp1 = a.S[p - 1], sq1 = a.S[q + 1];
        const int t2 = a.T->rt2[a.S[p] * 5 + a.S[q]];
        const int c = (int)g - 32768 - (int)a.T->mismatchI[t2 * 25 + sq1 * 5 + sp1];
        const int e = lds_intloop(*a.T, a.P, N1, N2, type, t2, si1, sj1, sp1, sq1) + c;
        const unsigned k = g == 65535u ? KEY_NONE : ((unsigned)(e + KEY_BIAS) << 10) | (unsigned)(N1 << 5 | N2);
        best = k < best ? k : best;
    }
}

// the four shapes whose energies come from the big int11 / int21 / int22 tables in global memory: the load is issued here and consumed
// by the caller after its LDS-only shapes, so the L2 latency is covered
template <int N1, int N2>
__device__ __forceinline__ void a1_small_g(const A1& a, int i, int j, int type, int si1, int sj1, int& raw, int& cc) {
    raw = 0; cc = INF;
    if (N1 + N2 <= a.um) {
        const int p = i + 1 + N1, q = j - 1 - N2;
        const unsigned g = a.cring[((a.r0 - N1 - N2) & 31) * CSTR + p];
        const int sp1 = a.S[p - 1], sq1 = a.S[q + 1];
        const int t2 = a.T->rt2[a.S[p] * 5 + a.S[q]];
        cc = g == 65535u ? INF : (int)g - 32768 - (int)a.T->mismatchI[t2 * 25 + sq1 * 5 + sp1];
        if (N1 == 1 && N2 == 1) raw = a.P->int11[type][t2][si1][sj1];
        else if (N1 == 1 && N2 == 2) raw = a.P->int21[type][t2][si1][sq1][sj1];
        else if (N1 == 2 && N2 == 1) raw = a.P->int21[t2][type][sq1][si1][sp1];
        else raw = a.P->int22[type][t2][si1][sp1][sq1][sj1];
    }
}

// ---- "fast" variants of the jobs above for the steady state (every loop size admissible: um = MAXLOOP).  Same candidates, same keys; the
// difference is the order of the loads: everything whose address is known up front (pair codes, ring entries, bases) is issued as one batch,
// the table reads that depend on it as a second one, then the arithmetic.  The straightforward versions interleave a volatile read (kept
// narrow on purpose, see a1_gen_row) with the reads that depend on it, and a volatile access is an ordering point for the scheduler: they
// compile to one LDS round trip per candidate.
#ifdef MIRP_X_XBBCAST          // timing experiment: every lane gathers the same table entry (no bank conflicts in the XB / X1 reads)
#define MIRP_XIDX(e) (((e) & 0u))
#elif defined(MIRP_X_NOCODE)      // timing experiment: the table gather without the dependent code read in front of it
#define MIRP_XIDX(e) ((e) & 127u)
#define MIRP_NOCODE_READ 1
#else
#define MIRP_XIDX(e) (e)
#endif
#define A1_CHUNK 10
// N consecutive pair codes starting at byte A of a code array, four per read.  A lane's A has any alignment (consecutive cells have consecutive
// columns) and a ds_read_b32 off its alignment costs several aligned ones (profiles/tools/lds_unaligned.hip), so the array exists four times, copy c
// shifted by c bytes: the dword at (A - c) of copy c = A & 3 holds codes A .. A + 3 and is aligned.  The byte selects fold into the adds that use
// the codes (v_add_u32_sdwa).  which = 0: q codes (qbr), 1: p codes (pax; they fit a byte).
// base + byte K of w in one instruction (the compiler extracts with v_bfe_u32 first when the sum is an LDS address)
template <int K>
__device__ __forceinline__ unsigned a1_add_byte(unsigned base, unsigned w) {
    unsigned r;
    if constexpr (K == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(base), "v"(w));
    else if constexpr (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(base), "v"(w));
    else if constexpr (K == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(base), "v"(w));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(base), "v"(w));
    return r;
}
// x[k] = tab[idx + code k] for the N codes starting at byte A of code array `which`
template <int N, int K = 0>
__device__ __forceinline__ void a1_gather4(unsigned tab_idx, const unsigned (&dw)[(N + 3) / 4], int (&x)[N]) {
    if constexpr (K < N) {
        x[K] = *(lds_vu8)(size_t)a1_add_byte<K & 3>(tab_idx, dw[K >> 2]);
        a1_gather4<N, K + 1>(tab_idx, dw, x);
    }
}
template <int N>
__device__ __forceinline__ void a1_codes4(const A1& a, int which, int A, const unsigned char* tab, unsigned idx, int (&x)[N]) {
    typedef const volatile __attribute__((address_space(3))) unsigned* lds_vu32;
    const int c = A & 3;
    lds_vu32 p = (lds_vu32)(a.code4 + which * 4 * CODE_STR + c * (CODE_STR - 1) + A);      // copy c, byte A - c
    unsigned dw[(N + 3) / 4];
#pragma unroll
    for (int t = 0; t < (N + 3) / 4; t++) dw[t] = p[t];
    const unsigned tab_idx = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)tab + idx;      // LDS address of tab[idx]
    a1_gather4<N>(tab_idx, dw, x);
}
template <int LO, int HI>
__device__ __forceinline__ void a1_b0f(const A1& a, int i, int j, unsigned& best) {
    if constexpr (HI - LO + 1 > A1_CHUNK) { a1_b0f<LO, LO + A1_CHUNK - 1>(a, i, j, best); a1_b0f<LO + A1_CHUNK, HI>(a, i, j, best); }
    else {
        constexpr int N = HI - LO + 1;
        const unsigned idxp = a.pax[i + 1];
        lds_vu8 ql = (lds_vu8)(a.qbr + (a.n + 2 - j));
        const unsigned short* rb = a.cring + i + 1;
        const unsigned char* xb = a.T->XB;
        unsigned code[N], g[N];
        int x[N];
#pragma unroll
        for (int k = 0; k < N; k++) g[k] = rb[a.rowtab[LO + k]];
#ifdef MIRP_X_RINGONLY          // timing experiment: what a ring of c + bulge / 1xn term would leave of these jobs (no code read, no table gather)
#pragma unroll
        for (int k = 0; k < N; k++) { code[k] = 0; x[k] = (int)idxp; }
#else
#pragma unroll
#ifdef MIRP_NOCODE_READ
        for (int k = 0; k < N; k++) code[k] = 3 * (LO + k);
#else
        for (int k = 0; k < N; k++) if (!(MIRP_A1_CODES4 && a.code4)) code[k] = ql[LO + k];
#endif
        if (MIRP_A1_CODES4 && a.code4) a1_codes4<N>(a, 0, a.n + 2 - j + LO, xb, idxp, x);
        else {
#pragma unroll
            for (int k = 0; k < N; k++) x[k] = xb[MIRP_XIDX(idxp + code[k])];
        }
#endif
#pragma unroll
        for (int k = 0; k < N; k++) { const unsigned e = ((g[k] + (unsigned)x[k]) << 10) + a.P->kb0_key[LO + k]; best = e < best ? e : best; }
    }
}
template <int LO, int HI>
__device__ __forceinline__ void a1_b1f(const A1& a, int i, int j, unsigned& best) {
    if constexpr (HI - LO + 1 > A1_CHUNK) { a1_b1f<LO, LO + A1_CHUNK - 1>(a, i, j, best); a1_b1f<LO + A1_CHUNK, HI>(a, i, j, best); }
    else {
        constexpr int N = HI - LO + 1;
        const unsigned idxq = a.qbr[a.n + 2 - j];
        lds_vpax pl = (lds_vpax)(a.pax + i + 1);
        const unsigned short* rb = a.cring + i + 1;
        const unsigned char* xb = a.T->XB;
        unsigned code[N], g[N];
        int x[N];
#pragma unroll
        for (int k = 0; k < N; k++) g[k] = rb[a.rowtab[LO + k] + (LO + k)];
#ifdef MIRP_X_RINGONLY
#pragma unroll
        for (int k = 0; k < N; k++) { code[k] = 0; x[k] = (int)idxq; }
#else
#pragma unroll
#ifdef MIRP_NOCODE_READ
        for (int k = 0; k < N; k++) code[k] = 3 * (LO + k);
#else
        for (int k = 0; k < N; k++) if (!(MIRP_A1_CODES4 && a.code4)) code[k] = pl[LO + k];
#endif
        if (MIRP_A1_CODES4 && a.code4) a1_codes4<N>(a, 1, i + 1 + LO, xb, idxq, x);
        else {
#pragma unroll
            for (int k = 0; k < N; k++) x[k] = xb[MIRP_XIDX(idxq + code[k])];
        }
#endif
#pragma unroll
        for (int k = 0; k < N; k++) { const unsigned e = ((g[k] + (unsigned)x[k]) << 10) + a.P->kb1_key[LO + k]; best = e < best ? e : best; }
    }
}
template <int LO, int HI>
__device__ __forceinline__ void a1_i0f(const A1& a, int i, int j, unsigned& best) {
    if constexpr (HI - LO + 1 > A1_CHUNK) { a1_i0f<LO, LO + A1_CHUNK - 1>(a, i, j, best); a1_i0f<LO + A1_CHUNK, HI>(a, i, j, best); }
    else {
        constexpr int N = HI - LO + 1;
        const unsigned idxp = a.pax[i + 2];
        lds_vu8 ql = (lds_vu8)(a.qbr + (a.n + 2 - j));
        const unsigned short* rb = a.cring + i + 2;
        const unsigned char* xb = a.T->X1;
        unsigned code[N], g[N];
        int x[N];
#pragma unroll
        for (int k = 0; k < N; k++) g[k] = rb[a.rowtab[LO + k + 1]];
#ifdef MIRP_X_RINGONLY          // timing experiment: what a ring of c + bulge / 1xn term would leave of these jobs (no code read, no table gather)
#pragma unroll
        for (int k = 0; k < N; k++) { code[k] = 0; x[k] = (int)idxp; }
#else
#pragma unroll
#ifdef MIRP_NOCODE_READ
        for (int k = 0; k < N; k++) code[k] = 3 * (LO + k);
#else
        for (int k = 0; k < N; k++) if (!(MIRP_A1_CODES4 && a.code4)) code[k] = ql[LO + k];
#endif
        if (MIRP_A1_CODES4 && a.code4) a1_codes4<N>(a, 0, a.n + 2 - j + LO, xb, idxp, x);
        else {
#pragma unroll
            for (int k = 0; k < N; k++) x[k] = xb[MIRP_XIDX(idxp + code[k])];
        }
#endif
#pragma unroll
        for (int k = 0; k < N; k++) { const unsigned e = ((g[k] + (unsigned)x[k]) << 10) + a.P->k1n0_key[LO + k]; best = e < best ? e : best; }
    }
}
template <int LO, int HI>
__device__ __forceinline__ void a1_i1f(const A1& a, int i, int j, unsigned& best) {
    if constexpr (HI - LO + 1 > A1_CHUNK) { a1_i1f<LO, LO + A1_CHUNK - 1>(a, i, j, best); a1_i1f<LO + A1_CHUNK, HI>(a, i, j, best); }
    else {
        constexpr int N = HI - LO + 1;
        const unsigned idxq = a.qbr[a.n + 3 - j];
        lds_vpax pl = (lds_vpax)(a.pax + i + 1);
        const unsigned short* rb = a.cring + i + 1;
        const unsigned char* xb = a.T->X1;
        unsigned code[N], g[N];
        int x[N];
#pragma unroll
        for (int k = 0; k < N; k++) g[k] = rb[a.rowtab[LO + k + 1] + (LO + k)];
#ifdef MIRP_X_RINGONLY
#pragma unroll
        for (int k = 0; k < N; k++) { code[k] = 0; x[k] = (int)idxq; }
#else
#pragma unroll
#ifdef MIRP_NOCODE_READ
        for (int k = 0; k < N; k++) code[k] = 3 * (LO + k);
#else
        for (int k = 0; k < N; k++) if (!(MIRP_A1_CODES4 && a.code4)) code[k] = pl[LO + k];
#endif
        if (MIRP_A1_CODES4 && a.code4) a1_codes4<N>(a, 1, i + 1 + LO, xb, idxq, x);
        else {
#pragma unroll
            for (int k = 0; k < N; k++) x[k] = xb[MIRP_XIDX(idxq + code[k])];
        }
#endif
#pragma unroll
        for (int k = 0; k < N; k++) { const unsigned e = ((g[k] + (unsigned)x[k]) << 10) + a.P->k1n1_key[LO + k]; best = e < best ? e : best; }
    }
}

// key of one small shape from its ring entry g (G0 + 32768, 65535 = none), the inner pair's mismatchI term (G0 = c + that term) and the loop energy
__device__ __forceinline__ unsigned a1_small_key(unsigned g, int mm_inner, int e_loop, unsigned code) {
    const int e = e_loop + (int)g - 32768 - mm_inner;
    return g == 65535u ? KEY_NONE : ((unsigned)(e + KEY_BIAS) << 10) | code;
}
// stack, the two 1-bulges, 1x1, 1x2 of the lane's cell (i, j): the seven bases and five ring entries first, pair types by arithmetic, then the table reads
__device__ __forceinline__ unsigned a1_small14f(const A1& a, int i, int j, int type, bool ahead, bool noglobal = false) {
    lds_vu8 Sv = (lds_vu8)a.S;
    const int s_i = Sv[i], s_i1 = Sv[i + 1], s_i2 = Sv[i + 2], s_j3 = Sv[j - 3], s_j2 = Sv[j - 2], s_j1 = Sv[j - 1], s_j = Sv[j];
    const unsigned short* rb = a.cring;
    const unsigned g00 = rb[((a.r0) & 31) * CSTR + i + 1], g01 = rb[((a.r0 - 1) & 31) * CSTR + i + 1], g10 = rb[((a.r0 - 1) & 31) * CSTR + i + 2];
    const unsigned g11 = rb[((a.r0 - 2) & 31) * CSTR + i + 2], g12 = rb[((a.r0 - 3) & 31) * CSTR + i + 2];
    const int t00 = rtype_of(pair_type(s_i1, s_j1)), t01 = rtype_of(pair_type(s_i1, s_j2)), t10 = rtype_of(pair_type(s_i2, s_j1));
    const int t11 = rtype_of(pair_type(s_i2, s_j2)), t12 = rtype_of(pair_type(s_i2, s_j3));
    const LdsTables& T = *a.T;
    const int m00 = T.mismatchI[t00 * 25 + s_j * 5 + s_i], m01 = T.mismatchI[t01 * 25 + s_j1 * 5 + s_i], m10 = T.mismatchI[t10 * 25 + s_j * 5 + s_i1];
    const int m11 = T.mismatchI[t11 * 25 + s_j1 * 5 + s_i1], m12 = T.mismatchI[t12 * 25 + s_j2 * 5 + s_i1];
    const int st00 = T.stack[type * 8 + t00], st01 = T.stack[type * 8 + t01], st10 = T.stack[type * 8 + t10], b1 = T.bulge[1];
    const int r11 = noglobal ? 0 : a.P->int11[type][t11][s_i1][s_j1];
    const int r12 = noglobal ? 0 : a.P->int21[type][t12][s_i1][s_j2][s_j1];
    unsigned res = a1_small_key(g01, m01, b1 + st01, 0u << 5 | 1u);
    unsigned k = a1_small_key(g10, m10, b1 + st10, 1u << 5 | 0u); res = k < res ? k : res;
    k = a1_small_key(g00, m00, st00, 0u); if (!ahead) res = k < res ? k : res;     // the stacked pair of a lane of diagonal d+1 is not final yet
    k = a1_small_key(g11, m11, r11, 1u << 5 | 1u); res = k < res ? k : res;
    k = a1_small_key(g12, m12, r12, 1u << 5 | 2u); res = k < res ? k : res;
    return res;
}
// 2x1, 2x2, 2x3, 3x2
__device__ __forceinline__ unsigned a1_small15f(const A1& a, int i, int j, int type, bool noglobal = false) {
    lds_vu8 Sv = (lds_vu8)a.S;
    const int s_i1 = Sv[i + 1], s_i2 = Sv[i + 2], s_i3 = Sv[i + 3], s_i4 = Sv[i + 4], s_j4 = Sv[j - 4], s_j3 = Sv[j - 3], s_j2 = Sv[j - 2], s_j1 = Sv[j - 1];
    const unsigned short* rb = a.cring;
    const unsigned g21 = rb[((a.r0 - 3) & 31) * CSTR + i + 3], g22 = rb[((a.r0 - 4) & 31) * CSTR + i + 3], g23 = rb[((a.r0 - 5) & 31) * CSTR + i + 3];
    const unsigned g32 = rb[((a.r0 - 5) & 31) * CSTR + i + 4];
    // (n1, n2): p = i + 1 + n1, q = j - 1 - n2; sp1 = S[p - 1], sq1 = S[q + 1]
    const int t21 = rtype_of(pair_type(s_i3, s_j2)), t22 = rtype_of(pair_type(s_i3, s_j3)), t23 = rtype_of(pair_type(s_i3, s_j4)), t32 = rtype_of(pair_type(s_i4, s_j3));
    const LdsTables& T = *a.T;
    const int m21 = T.mismatchI[t21 * 25 + s_j1 * 5 + s_i2], m22 = T.mismatchI[t22 * 25 + s_j2 * 5 + s_i2], m23 = T.mismatchI[t23 * 25 + s_j3 * 5 + s_i2];
    const int m32 = T.mismatchI[t32 * 25 + s_j2 * 5 + s_i3];
    const int o23 = T.mismatch23I[type * 25 + s_i1 * 5 + s_j1], i23 = T.mismatch23I[t23 * 25 + s_j3 * 5 + s_i2], i32 = T.mismatch23I[t32 * 25 + s_j2 * 5 + s_i3];
    const int base23 = T.internal_loop[5] + T.ninio;
    const int r21 = noglobal ? 0 : a.P->int21[t21][type][s_j1][s_i1][s_i2];
    const int r22 = noglobal ? 0 : a.P->int22[type][t22][s_i1][s_i2][s_j2][s_j1];
    unsigned res = a1_small_key(g23, m23, base23 + o23 + i23, 2u << 5 | 3u);
    unsigned k = a1_small_key(g32, m32, base23 + o23 + i32, 3u << 5 | 2u); res = k < res ? k : res;
    k = a1_small_key(g21, m21, r21, 2u << 5 | 1u); res = k < res ? k : res;
    k = a1_small_key(g22, m22, r22, 2u << 5 | 2u); res = k < res ? k : res;
    return res;
}

}  // namespace mirp
