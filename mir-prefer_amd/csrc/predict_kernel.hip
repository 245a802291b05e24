// Data-parallel predict-stage filter: dot-bracket hairpin classification, mature/star duplex
// rules, expression rules and the per-window locus decision.  One wavefront per precursor window.
// Replaces the per-window Python loop of filter_next_loci / check_loci
// (/root/reference/miR_PREFeR.py = MP: a8 :1541-1724, a9 :1727-1999, a10 :2002-2163, a11 :2206-2347)
// and its two `samtools view` subprocesses per window.
//
// Phase 1 (lane per RNALfold line): structure list of the window (MP:1541-1599).
// Phase 2 (lane per structure, loop over matures): get_maturestar_info + check_expression_new.
// Phase 3 (lane 0): the sequential "lowest normalised energy that passes" rule (MP:2246-2343).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>
#include <vector>
#include "mirp_internal.h"

namespace mirp {

// Capacities of one launch (PredictCaps, mirp_internal.h): p_cap pieces per structure line (filter_ss keeps pieces longer than 55 characters: at most
// len / 56 of them), s_cap structures per window (lines x pieces), m_cap candidate matures per window.  A window that needs more is flagged
// (status 2 / 3) with what it needs, and run_predict_launch re-runs just those windows with capacities sized for them: the reference has no
// such limits (MP:1541-1599, 2241), so neither does the result.
#define PW_MIN_STRUCTS 192

// One structure piece of a window: line index, offset and length inside the line, kind (0 stem-loop, 1 good bifurcation).
// start and normalised energy are recomputed from the line record.
struct PStruct { unsigned short line, off, len, type; };

// Dot-bracket text staged in LDS at 2 bits per character (16 characters per word): a window's lines then take <= 9 KB instead of
// 34 KB, which is what bounds the number of resident windows per CU.  0 '.', 1 '(', 2 ')'.
struct SS {
    const unsigned* w;
    int o;
    __device__ __forceinline__ char operator[](int i) const {
        const int k = o + i;
        const unsigned c = (w[k >> 4] >> ((k & 15) * 2)) & 3u;
        return c == 0 ? '.' : (c == 1 ? '(' : ')');
    }
    __device__ __forceinline__ SS operator+(int d) const { SS r; r.w = w; r.o = o + d; return r; }
};

__device__ __forceinline__ int d_find(SS s, char c, int a, int b) { for (int i = a; i < b; i++) if (s[i] == c) return i; return -1; }
__device__ __forceinline__ int d_rfind(SS s, char c, int a, int b) { for (int i = b - 1; i >= a; i--) if (s[i] == c) return i; return -1; }
__device__ __forceinline__ int d_count(SS s, char c, int a, int b) { int n = 0; for (int i = a; i < b; i++) n += (s[i] == c); return n; }
__device__ __forceinline__ void d_clip(int len, int& a, int& b) {   // Python slice bounds
    if (a < 0) { a += len; if (a < 0) a = 0; }
    if (b < 0) { b += len; if (b < 0) b = 0; }
    if (a > len) a = len;
    if (b > len) b = len;
}

// partner of bracket at x inside s[0,len); -1 if unmatched
__device__ int d_partner(SS s, int len, int x) {
    int depth = 0;
    if (s[x] == '(') {
        for (int i = x; i < len; i++) { if (s[i] == '(') depth++; else if (s[i] == ')') { if (--depth == 0) return i; } }
    } else if (s[x] == ')') {
        for (int i = x; i >= 0; i--) { if (s[i] == ')') depth++; else if (s[i] == '(') { if (--depth == 0) return i; } }
    }
    return -1;
}

__device__ bool d_is_stem_loop(SS s, int len) {   // MP:1602-1608, minloop 3
    int lo = d_rfind(s, '(', 0, len), fc = d_find(s, ')', 0, len);
    return fc - lo - 1 >= 3;
}

// MP:1611-1659
__device__ bool d_good_bifurcation(SS s, int len) {
    int depth = 0, bif = 0, last_pop = 0, first_bi_last = 0, second_bi_first = 0, last_pos = 0;
    for (int idx = 0; idx < len; idx++) {
        char ch = s[idx];
        if (ch == '(') {
            if (idx != 0) {
                if (depth == 0) { if (last_pop) return false; }
                else if (last_pop) {
                    if (bif >= 1) return false;
                    bif = 1; first_bi_last = last_pos; second_bi_first = idx;
                }
            }
            depth++; last_pop = 0; last_pos = idx;
        } else if (ch == ')') {
            last_pop = 1;
            if (depth == 0) return false;
            depth--; last_pos = idx;
        }
    }
    int a = d_partner(s, len, second_bi_first), b = d_partner(s, len, first_bi_last);
    if (a < 0 || b < 0) return false;
    if ((double)(a - b) / len < 0.5)
        if ((double)b / len > 0.25)
            if ((double)a / len < 0.75) return true;
    return false;
}

// stat_duplex + pass_stat_duplex (MP:1815-1873) on mature_duplex = s[m0,m1), star_duplex = s[s0,s1). Returns MS code.
__device__ int d_duplex_code(SS s, int m0, int m1, int s0, int s1) {
    const int ml = m1 - m0, L = ml + (s1 - s0);
#define DCH(i) ((i) < ml ? s[m0 + (i)] : s[s0 + (i) - ml])
    int openpos = -1, closepos = -1;
    for (int i = 0; i < L && (openpos < 0 || closepos < 0); i++) {
        char ch = DCH(i);
        if (ch == '(' && openpos < 0) openpos = i;
        if (ch == ')' && closepos < 0) closepos = i;
    }
    char oc = '(', cc = ')';
    if (openpos > closepos) { oc = ')'; cc = '('; }
    // Pairs (open idx -> close idx), visited in increasing open idx.  Opens are matched LIFO.
    // Walk the opens in order and find each one's close with a depth counter (no arrays needed).
    int nloops = 0, nbulges = 0, totalloop = 0, maxbulge = 0;
    int prev_o = -1, prev_c = -1;
    // first make sure no close pops an empty stack (the reference would raise IndexError)
    {
        int depth = 0;
        for (int i = 0; i < L; i++) { char ch = DCH(i); if (ch == oc) depth++; else if (ch == cc) { if (depth == 0) return 12; depth--; } }
    }
    for (int i = 0; i < L; i++) {
        if (DCH(i) != oc) continue;
        int depth = 0, c = -1;
        for (int j = i; j < L; j++) { char ch = DCH(j); if (ch == oc) depth++; else if (ch == cc) { if (--depth == 0) { c = j; break; } } }
        if (c < 0) continue;   // never closed: not in dict_bp
        if (prev_o >= 0) {
            if (!((i - prev_o == 1) && (prev_c - c == 1))) {
                int mb = i - prev_o - 1, sb = prev_c - c - 1;
                if (mb == sb) { nloops++; totalloop += mb; }
                else { nbulges++; maxbulge = max(maxbulge, max(mb, sb)); }
            }
        }
        prev_o = i; prev_c = c;
    }
#undef DCH
    if (nloops + nbulges > 5) return 8;
    if (maxbulge > 2) return 9;
    if (totalloop > 5) return 10;
    if (nbulges > 2) return 11;
    return 0;
}

struct MStar { int code, star_s, star_e, fold_s, fold_e; };

// get_maturestar_info (MP:1876-1999). strand 0 '+', 1 '-'.
__device__ void d_maturestar(SS ss, int len, int m0, int m1, int foldstart, int rs, int re, int strand, MStar& o) {
    o.code = 0; o.star_s = o.star_e = 0;
    { int depth = 0; for (int i = 0; i < len; i++) { if (ss[i] == '(') depth++; else if (ss[i] == ')') { if (depth == 0) { o.code = 1; return; } depth--; } } }
    int l0, l1, fg0, fg1;
    if (strand == 0) { l0 = m0 - rs - foldstart + 1; l1 = m1 - rs - foldstart + 1; fg0 = rs + foldstart - 1; fg1 = fg0 + len; }
    else { l0 = re - m1 - foldstart + 1; l1 = re - m0 - foldstart + 1; fg0 = re - foldstart - len + 1; fg1 = re - foldstart + 1; }
    o.fold_s = fg0; o.fold_e = fg1;
    if (!(m0 >= fg0 && m1 <= fg1)) { o.code = 2; return; }
    int a = l0, b = l1; d_clip(len, a, b);
    bool has_o = d_find(ss, '(', a, b) != -1, has_c = d_find(ss, ')', a, b) != -1;
    if (has_o && has_c) { o.code = 3; return; }
    if ((b > a ? b - a : 0) - d_count(ss, '.', a, b) < 14) { o.code = 4; return; }
    char sym = '(';
    int firstbp = d_find(ss, '(', a, b), lastbp = d_rfind(ss, '(', a, b);
    if (firstbp == -1) { firstbp = d_find(ss, ')', a, b); lastbp = d_rfind(ss, ')', a, b); sym = ')'; }
    if (firstbp == -1) { o.code = 4; return; }
    int p_last = d_partner(ss, len, lastbp), p_first = d_partner(ss, len, firstbp);
    if (p_last < 0 || p_first < 0) { o.code = 12; return; }
    int star_start = p_last - (l1 - 1 - lastbp) + 2, star_end = p_first + (firstbp - l0) + 2 + 1;
    if (l0 <= star_start) {
        if (star_start - l1 < 3) { o.code = 5; return; }
        if (star_end > len) { o.code = 6; return; }
    }
    if (star_start <= l0) {
        if (l0 - star_end < 3) { o.code = 5; return; }
        if (star_start < 0) { o.code = 6; return; }
    }
    int ra = l0, rb = l1 - 2; d_clip(len, ra, rb);
    int mend = d_rfind(ss, sym, ra, rb);
    int p_mend = mend >= 0 ? d_partner(ss, len, mend) : -1;
    if (mend < 0 || p_mend < 0) { o.code = 12; return; }
    int md0 = l0, md1 = mend + 1, sd0 = p_mend, sd1 = p_first + 1;
    d_clip(len, md0, md1); d_clip(len, sd0, sd1);
    if (md1 < md0) md1 = md0;
    if (sd1 < sd0) sd1 = sd0;
    int total_bps = (md1 - md0) - d_count(ss, '.', md0, md1);
    if (total_bps < 14) { o.code = 4; return; }
    int sa = star_start, sb = star_end; d_clip(len, sa, sb);
    if (d_find(ss, '(', sa, sb) != -1 && d_find(ss, ')', sa, sb) != -1) { o.code = 7; return; }
    int dc = d_duplex_code(ss, md0, md1, sd0, sd1);
    if (dc) { o.code = dc; return; }
    if (strand == 0) { o.star_s = rs + foldstart - 1 + star_start; o.star_e = rs + foldstart - 1 + star_end; }
    else { o.star_s = re - foldstart - star_end + 1; o.star_e = re - foldstart - star_start + 1; }
}

__device__ __forceinline__ long long aln_lower_bound(const MirpAln* __restrict__ a, long long n, int tid, long long pos) {
    long long lo = 0, hi = n;
    while (lo < hi) {
        long long mid = (lo + hi) >> 1;
        MirpAln r = a[mid];
        if (r.tid < tid || (r.tid == tid && r.pos < pos)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// lower bound inside [lo, hi) (the window's own records: tens to hundreds, found once per window by the whole wave)
__device__ __forceinline__ long long aln_lower_bound_in(const MirpAln* __restrict__ a, long long lo, long long hi, int tid, long long pos) {
    while (lo < hi) {
        long long mid = (lo + hi) >> 1;
        MirpAln r = a[mid];
        if (r.tid < tid || (r.tid == tid && r.pos < pos)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// the same lower bound over the whole record array with all 64 lanes probing at once: 3 round trips for 10^5 records, 5 for 10^7 (a lane's own binary
// search takes 17 .. 25 dependent ones, and every (mature, structure) pair of the window used to pay two of them)
__device__ __forceinline__ long long wave_lower_bound(const MirpAln* __restrict__ a, long long n, int tid, long long pos) {
    const int lane = threadIdx.x & 63;
    long long lo = 0, hi = n;
    while (hi - lo > 64) {
        const long long step = (hi - lo + 63) / 64;
        const long long idx = lo + step * (lane + 1) - 1;
        bool less = false;
        if (idx < hi) { const MirpAln r = a[idx]; less = r.tid < tid || (r.tid == tid && r.pos < pos); }
        const long long cnt = (long long)__popcll(__ballot(less));          // the probes below the key are a prefix of the lanes
        const long long nlo = lo + step * cnt, nhi = lo + step * (cnt + 1) - 1;
        lo = nlo;
        hi = nhi < hi ? nhi : hi;
    }
    bool less = false;
    if (lo + lane < hi) { const MirpAln r = a[lo + lane]; less = r.tid < tid || (r.tid == tid && r.pos < pos); }
    return lo + (long long)__popcll(__ballot(less));
}

struct ExprRes {
    long long total_this, total_mature, total_iso, total_star; // total_star = after max with imperfect (when key present)
    long long raw_star, total_anti, imp[3];                    // reasons mode: star before the imperfect maximum, antisense depth, the three imperfect-star depths
    int distance, has_imp_key, imp_start, imp_end, imp_which /* -1 none */;
    double ratio_total, ratio_iso;
    bool too_many_start, expressed_all, exception;
};

// check_expression_new (MP:2037-2163) on reads kept by gen_mapinfo_each_sample (MP:2021)
// Sample sets as bit masks over the 8-bit sample index of a record (any number of ALIGNMENT_FILEs up to MIRP_MAX_SAMPLES, MP:3300-3308): eight words,
// indexed through compile-time selects so that they stay in registers.
struct SampleSet {
    unsigned int w[8];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = 0u;
    }
    __device__ __forceinline__ bool test_and_set(int s) {          // -> was the bit clear?
        const unsigned int bit = 1u << (s & 31);
        const int wi = s >> 5;
        bool fresh = false;
#pragma unroll
        for (int i = 0; i < 8; i++) if (i == wi) { fresh = !(w[i] & bit); w[i] |= bit; }
        return fresh;
    }
    __device__ __forceinline__ bool holds_first(int n) const {       // bits 0 .. n-1 all set
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int lo = 32 * i;
            if (n > lo) { const unsigned int want = (n - lo >= 32) ? 0xffffffffu : ((1u << (n - lo)) - 1u); ok = ok && ((w[i] & want) == want); }
        }
        return ok;
    }
};

// mature_each (reasons mode): where the per-sample mature depths go -- the tail of the thread's own reasons record, already zeroed; nullptr otherwise
template <bool REASONS>
__device__ void d_expression(const MirpAln* __restrict__ a, long long na, int n_samples, int tid, int ws, int we, int fold_s, int fold_e,
                             int m0, int m1, int star_s, int star_e, int strand, int allow_3nt, ExprRes& o, int* mature_each, int mature_each_cap,
                             long long wk0, long long wk1) {
    const int mature_len = m1 - m0, star_len = star_e - star_s, pre_len = fold_e - fold_s;
    // [wk0, wk1) = the records that start inside the window [ws, we); a record that starts at or behind we fails the test below anyway
    (void)na;
    long long k0 = aln_lower_bound_in(a, wk0, wk1, tid, fold_s > ws ? fold_s : ws), k1 = aln_lower_bound_in(a, wk0, wk1, tid, fold_e);
    long long tot_pre = 0, tot_mat = 0, tot_iso = 0, tot_star = 0, imp[3] = {0, 0, 0};
    SampleSet mature_set;          // samples with reads_mature > 0
    SampleSet seen;                // samples with a read on this strand starting at cur_pos (the records are sorted by position)
    mature_set.clear(); seen.clear();
    int starts = 0, cur_pos = -1;
    long long tot_anti = 0;
    for (long long k = k0; k < k1; k++) {
        MirpAln r = a[k];
        if (r.pos < ws || r.pos + (int)r.len > we) continue;
        if ((int)r.strand != strand) { if (REASONS) tot_anti += (int)r.depth; continue; }   // antisense reads do not enter any rule used by check_loci
        int d = (int)r.depth, rl = r.len, sp = r.pos, sm = r.sample;
        if (sp != cur_pos) { cur_pos = sp; seen.clear(); }
        if (seen.test_and_set(sm)) starts++;          // once per (start position, sample with reads there), MP:2066-2068
        tot_pre += d;
        if (sp == m0 && rl == mature_len) {
            tot_mat += d;
            if (d > 0) (void)mature_set.test_and_set(sm);
            if (REASONS) { if (mature_each && sm < mature_each_cap) mature_each[sm] += d; }
        }
        if (sp == star_s && rl == star_len) tot_star += d;
        int dx = sp - m0, dl = rl - mature_len;
        if (dx >= -3 && dx <= 3 && dl >= -3 && dl <= 3) tot_iso += d;
        if (allow_3nt) {
            if (sp == star_s && rl == star_len + 1) imp[0] += d;
            else if (sp - 1 == star_s && rl == star_len - 1) imp[1] += d;
            else if (sp - 1 == star_s && rl == star_len) imp[2] += d;
        }
    }
    o.total_this = tot_pre; o.total_mature = tot_mat; o.total_iso = tot_iso;
    o.raw_star = tot_star; o.total_anti = tot_anti; o.imp[0] = imp[0]; o.imp[1] = imp[1]; o.imp[2] = imp[2];
    o.distance = (star_s > m1) ? star_s - m1 : m0 - star_e;
    long long max_imp = 0;
    o.has_imp_key = 0; o.imp_start = 0; o.imp_end = 0; o.imp_which = -1;
    if (tot_star == 0 && allow_3nt) {
        o.has_imp_key = 1;
        long long mx = imp[0] > imp[1] ? imp[0] : imp[1]; mx = mx > imp[2] ? mx : imp[2];
        if (mx > 0) {
            int which = (imp[0] == mx) ? 0 : (imp[1] == mx) ? 1 : 2;
            if (which == 0) { o.imp_start = star_s; o.imp_end = star_e + 1; }
            else if (which == 1) { o.imp_start = star_s + 1; o.imp_end = star_e; }
            else { o.imp_start = star_s + 1; o.imp_end = star_e + 1; }
            o.imp_which = which;
            max_imp = mx;
        }
    }
    long long st = tot_star > max_imp ? tot_star : max_imp;
    o.total_star = st;
    o.exception = (tot_pre == 0);
    o.ratio_total = o.exception ? 0.0 : (double)(tot_mat + st) / (double)tot_pre;
    o.ratio_iso = o.exception ? 0.0 : (double)(tot_iso + st) / (double)tot_pre;
    // only the LAST sample's start counter is ever non-zero (stale loop variable, MP:2046/2068)
    o.too_many_start = ((double)starts / pre_len) > 0.5;
    o.expressed_all = mature_set.holds_first(n_samples);
}

// REASONS: the -d mode of the reference (check_loci's dict_why_not_miRNA_reasons, MP:2206-2347): every evaluated (mature, structure) pair is
// appended to a record pool (rstride ints per record, layout in mirp_pipeline.cpp) together with one record per window; nothing else changes.
// GTEXT: the staged structure text lives in a global scratch area of the workgroup instead of LDS -- windows whose lines do not fit there (PRECURSOR_LEN in
// the thousands: hundreds of lines of thousands of characters); everything else is the same code.
template <bool REASONS, bool GTEXT>
#ifndef MIRP_PRED_WPS
#define MIRP_PRED_WPS 4
#endif
__global__ void __launch_bounds__(64, MIRP_PRED_WPS) predict_kernel(
    const MirpWindow* __restrict__ windows, int n_windows, const MirpMature* __restrict__ matures,
    const MirpAln* __restrict__ alns, long long n_alns, const MirpFoldLine* __restrict__ lines, const char* __restrict__ ss,
    int ss_stride, int max_lines, const int* __restrict__ n_lines, MirpPredictParams pp,
    MirpMirna* __restrict__ out /* [n_windows * MIRP_MAX_MIRNA_PER_WINDOW] */, int* __restrict__ n_out, int* __restrict__ status,
    unsigned int* __restrict__ rcount, int* __restrict__ rpool, unsigned int rcap, int rstride,
    const int* __restrict__ wsel, int n_sel, const int* __restrict__ skip, const int* __restrict__ wslot, PredictCaps caps, int* __restrict__ need,
    unsigned* __restrict__ gtext) {
    // wsel == nullptr: every window w in [0, n_windows) with its fold output at slot w, except those with skip[w] >= 0 (folded again at full
    // line capacity: a second launch handles them); wsel != nullptr: the windows wsel[k], k in [0, n_sel), with their fold output at slot
    // wslot[k] (wslot == nullptr: slot k).  need (optional): need[3 w] = structures, need[3 w + 1] = pieces of one line, need[3 w + 2] = staged lines the window has.
    extern __shared__ __align__(16) unsigned char smem[];
    const int wpl = (ss_stride + 15) >> 4;                                // packed words per line
    const int max_structs = caps.s_cap, PW_MAX_PIECES = caps.p_cap, PW_MAX_MATURES = caps.m_cap, L_CAP = caps.l_cap;
    unsigned* textw = GTEXT ? gtext + (size_t)blockIdx.x * L_CAP * wpl : (unsigned*)smem;      // l_cap * wpl: only the lines phase 1 looks at (printed, >= minlen) are staged
    PStruct* sts = (PStruct*)(smem + (GTEXT ? (size_t)0 : (((size_t)L_CAP * wpl * 4 + 15) & ~(size_t)15))); // max_structs
    PStruct* slot = sts + max_structs;                                // 64 * p_cap
    int* cnts = (int*)(slot + 64 * PW_MAX_PIECES);                       // 64
    int* morder = cnts + 64;                                             // m_cap: mature indices in stable depth-descending order (MP:2241)
    unsigned short* lslot = (unsigned short*)(morder + PW_MAX_MATURES);  // max_lines: staged slot of a line, 0xffff = not staged
    unsigned short* lline = lslot + max_lines;                           // l_cap: line of a staged slot
    const int lane = threadIdx.x;
    const int n_iter = wsel ? n_sel : n_windows;
    for (int it = blockIdx.x; it < n_iter; it += gridDim.x) {
        const int w = wsel ? wsel[it] : it;
        if (!wsel && skip && skip[it] >= 0) continue;      // block-uniform
        const int sl = wslot ? wslot[it] : it;             // slot of the window's fold output
        const MirpWindow W = windows[w];
        const int nl = n_lines[sl] < max_lines ? n_lines[sl] : max_lines;
        const MirpFoldLine* wl = lines + (size_t)sl * max_lines;
        int st_flag = 0, need_pieces = 0;
        // which lines are staged: the printed ones of at least minlen characters (MP:1568-1570), in line order; about half of a window's lines
        // (23 of 42 on the benchmark input), which is what lets 16 instead of 11 windows be resident per CU
        int n_used = 0;
        for (int lb = 0; lb < nl; lb += 64) {
            const int k = lb + lane;
            bool used = false;
            if (k < nl) { const MirpFoldLine ln = wl[k]; used = ln.printed && ln.len >= pp.minlen; }
            const unsigned long long bal = __ballot(used);
            const int u = n_used + (int)__popcll(bal & ((1ull << lane) - 1ull));
            if (k < nl) lslot[k] = (used && u < L_CAP) ? (unsigned short)u : (unsigned short)0xffff;
            if (used && u < L_CAP) lline[u] = (unsigned short)k;
            if (used && u >= L_CAP) st_flag = 2;          // more lines than this launch stages: the window is run again with room for all of them
            n_used += (int)__popcll(bal);
        }
        __syncthreads();
        // stage their text, 2 bits per character (lane = one packed word = 16 characters)
        {
            const char* src = ss + (size_t)sl * max_lines * ss_stride;
            const int nu = n_used < L_CAP ? n_used : L_CAP;
            for (int x = lane; x < nu * wpl; x += 64) {
                const int us = x / wpl, wi = x - us * wpl, ln = lline[us];
                const char* p = src + (size_t)ln * ss_stride + wi * 16;
                const int lim = ss_stride - wi * 16;       // bytes of this line left
                unsigned v = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    unsigned four = 0;
                    if (q * 4 + 4 <= lim) four = *reinterpret_cast<const unsigned*>(p + q * 4);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const unsigned ch = (four >> (8 * b)) & 0xffu;
                        v |= (ch == '(' ? 1u : (ch == ')' ? 2u : 0u)) << ((q * 4 + b) * 2);
                    }
                }
                textw[x] = v;
            }
        }
        __syncthreads();
        // ---- phase 1: structures (lane per line, chunks of 64 lines)
        int nst = 0;
        for (int lb = 0; lb < nl; lb += 64) {
            int k = lb + lane, cnt = 0;
            if (k < nl) {
                MirpFoldLine ln = wl[k];
                const int us = lslot[k];
                SS s; s.w = textw + (size_t)(us == 0xffff ? 0 : us) * wpl; s.o = 0;
                if (ln.printed && ln.len >= pp.minlen && us != 0xffff) {
                    if (d_is_stem_loop(s, ln.len)) {
                        PStruct p; p.line = (unsigned short)k; p.off = 0; p.len = (unsigned short)ln.len; p.type = 0;
                        slot[lane * PW_MAX_PIECES + cnt++] = p;
                    } else {
                        // filter_ss (MP:1685-1724): one piece per top-level stem, [previous gap start, next stem start)
                        int len = ln.len;
                        int prefirst = d_find(s, '(', 0, len);
                        if (prefirst >= 0) {
                            int prelast = d_partner(s, len, prefirst), curfirst = prefirst, curlast = prelast, pregap0 = 0;
                            while (curfirst != -1) {
                                curfirst = d_find(s, '(', prelast < 0 ? len : prelast, len);
                                int after1 = len;
                                if (curfirst != -1) { after1 = curfirst; curlast = d_partner(s, len, curfirst); }
                                int ps = pregap0, pl = after1 - pregap0;
                                if (pl > 55) {
                                    int type = -1;
                                    if (d_is_stem_loop(s + ps, pl)) type = 0;
                                    else if (d_good_bifurcation(s + ps, pl)) type = 1;
                                    if (type >= 0) {
                                        if (cnt < PW_MAX_PIECES) {
                                            PStruct p; p.line = (unsigned short)k; p.off = (unsigned short)ps; p.len = (unsigned short)pl; p.type = (unsigned short)type;
                                            slot[lane * PW_MAX_PIECES + cnt++] = p;
                                        } else st_flag = 2;
                                        need_pieces++;
                                    }
                                }
                                pregap0 = prelast + 1;
                                prelast = curlast;
                            }
                        }
                    }
                }
            }
            cnts[lane] = cnt;
            __syncthreads();
            // ordered compaction (line order, piece order)
            int base = nst;
            for (int l = 0; l < 64; l++) { if (l < lane) base += cnts[l]; }
            for (int c = 0; c < cnt; c++) { if (base + c < max_structs) sts[base + c] = slot[lane * PW_MAX_PIECES + c]; else st_flag = 2; }
            int tot = 0;
            for (int l = 0; l < 64; l++) tot += cnts[l];
            nst += tot;
            __syncthreads();
        }
        if (need) {          // what this window needs, for the re-run of flagged windows: structures in all, pieces of its richest line
            int np = need_pieces;
            for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(np, o); np = t > np ? t : np; }
            if (lane == 0) { need[3 * (size_t)w] = nst; need[3 * (size_t)w + 1] = np; need[3 * (size_t)w + 2] = n_used; }
        }
        if (nst > max_structs) nst = max_structs;
        // ---- phases 2+3 per mature, depth-descending stable order (MP:2241)
        int nm = W.n_matures < PW_MAX_MATURES ? W.n_matures : PW_MAX_MATURES;
        if (W.n_matures > PW_MAX_MATURES) st_flag = 3;
        for (int k = lane; k < nm; k += 64) {          // rank of mature k among the first nm: lane-parallel, then the order is a table
            const int dk = matures[W.mature_off + k].depth;
            int r = 0;
            for (int j = 0; j < nm; j++) { const int dj = matures[W.mature_off + j].depth; if (dj > dk || (dj == dk && j < k)) r++; }
            morder[r] = k;
        }
        __syncthreads();
        int nout = 0;
        const long long wk0 = wave_lower_bound(alns, n_alns, W.tid, W.ws), wk1 = wave_lower_bound(alns, n_alns, W.tid, W.we);
        bool any_in_range = false;
        for (int k = 0; k < nm; k++) { MirpMature m = matures[W.mature_off + k]; int l = m.end - m.start; if (!(l < pp.min_mature_len || l > pp.max_mature_len)) any_in_range = true; }
        if (nst > 0 && any_in_range) {
            for (int rank = 0; rank < nm; rank++) {
                const int mi = morder[rank];          // rank-th mature in stable depth-descending order
                MirpMature m = matures[W.mature_off + mi];
                int ml = m.end - m.start;
                if (ml < pp.min_mature_len || ml > pp.max_mature_len) continue;
                // Each lane evaluates its structures and keeps its own winner of the sequential rule of MP:2246-2343
                // ("for s in order: skip if ne > lowest; if it passes: best = s, lowest = ne", lowest starting at 0.0), which selects the
                // passing structure with the smallest normalised energy <= 0 and, among equals, the LAST one in order.
                double b_ne = 0.0;
                int b_s = -1, b_has_star = 0, b_star_s = 0, b_star_e = 0, b_fold_s = 0, b_fold_e = 0, b_tm = 0, b_ts = 0, b_imp = 0;
                for (int sb = 0; sb < nst; sb += 64) {
                    int s = sb + lane;
                    if (s < nst) {
                        const PStruct p = sts[s];
                        const MirpFoldLine ln = wl[p.line];
                        const double ne = ((double)ln.energy / 100.0) / (double)ln.len;
                        SS str; str.w = textw + (size_t)lslot[p.line] * wpl; str.o = p.off;
                        MStar ms;
                        d_maturestar(str, p.len, m.start, m.end, ln.start + p.off, W.ws, W.we, m.strand, ms);
                        int pass = 0, has_star = 0, star_s = ms.star_s, star_e = ms.star_e, impf = 0;
                        long long tm = 0, ts = 0;
                        ExprRes rx;
                        rx.exception = true;
                        int* rrec = nullptr;          // reasons mode: this (mature, structure) pair's record, claimed before the expression test fills its tail
                        if (REASONS) {
                            const unsigned int ridx = atomicAdd(rcount, 1u);
                            if (ridx < rcap) { rrec = rpool + (size_t)ridx * rstride; for (int q = 12; q < rstride; q++) rrec[q] = 0; }
                        }
                        if (ms.code == 0) {
                            ExprRes& ex = rx;
                            d_expression<REASONS>(alns, n_alns, pp.n_samples, W.tid, W.ws, W.we, ms.fold_s, ms.fold_e, m.start, m.end, ms.star_s, ms.star_e,
                                         m.strand, pp.allow_3nt, ex, (REASONS && rrec && rstride > 21) ? rrec + 21 : nullptr, rstride - 21, wk0, wk1);
                            tm = ex.total_mature; ts = ex.total_star;
                            // 'max_imperfect_star' in exprinfo (MP:2161, 2631-2635): bit0 key present, bits1-2 which+1, bit3 max > 0
                            impf = (ex.has_imp_key ? 1 : 0) | ((ex.imp_which + 1) << 1) | ((ex.imp_which >= 0) ? 8 : 0);
                            if (!ex.exception && ex.distance > 4) {
                                if (ex.total_star > 0) {
                                    if (!(ex.ratio_total < 0.2)) {
                                        pass = 1; has_star = 1;
                                        if (ex.has_imp_key) { star_s = ex.imp_start; star_e = ex.imp_end; }
                                    }
                                } else if (pp.allow_no_star && !ex.too_many_start) {
                                    if (ex.ratio_iso >= 0.8 && (ex.expressed_all || ex.total_mature >= 1000)) pass = 1;
                                }
                            }
                        }
                        if (REASONS) {
                            int flags = 0;
                            ExprRes* exp_ = nullptr;
                            (void)exp_;
                            if (rrec) {
                                int* r = rrec;
                                r[0] = w; r[1] = mi; r[2] = s; r[3] = p.line; r[4] = p.off; r[5] = p.len; r[6] = ms.code; r[8] = ms.fold_s; r[9] = ms.fold_e;
                                r[10] = ms.star_s; r[11] = ms.star_e;
                                if (ms.code == 0) {
                                    r[12] = (int)rx.total_this; r[13] = (int)rx.total_anti; r[14] = (int)rx.total_mature; r[15] = (int)rx.total_iso; r[16] = (int)rx.raw_star;
                                    r[17] = (int)rx.imp[0]; r[18] = (int)rx.imp[1]; r[19] = (int)rx.imp[2]; r[20] = rx.distance;
                                    if (rx.exception) flags |= 256;
                                    else if (rx.distance <= 4) flags |= 1;
                                    else if (rx.total_star > 0) { if (rx.ratio_total < 0.2) flags |= 2; else flags |= 128; }
                                    else if (!pp.allow_no_star) flags |= 4;
                                    else if (rx.too_many_start) flags |= 8;
                                    else if (rx.ratio_iso >= 0.8 && (rx.expressed_all || rx.total_mature >= 1000)) flags |= 128;
                                    else {
                                        if (rx.ratio_iso < 0.8) flags |= 16;
                                        if (rx.total_mature <= 100) flags |= 32;
                                        if (!rx.expressed_all) flags |= 64;
                                    }
                                }
                                r[7] = flags;
                            }
                        }
                        if (pass && ne <= b_ne) {   // later structures win ties (this lane's s only grows)
                            b_ne = ne; b_s = s; b_has_star = has_star; b_star_s = star_s; b_star_e = star_e; b_fold_s = ms.fold_s; b_fold_e = ms.fold_e;
                            b_tm = (int)(tm > 0x7fffffffLL ? 0x7fffffffLL : tm); b_ts = (int)(ts > 0x7fffffffLL ? 0x7fffffffLL : ts); b_imp = impf;
                        }
                    }
                }
                // wave arg-min over (ne ascending, s descending) among lanes that have a candidate
                double r_ne = b_ne;
                int r_s = b_s;
                for (int o = 32; o > 0; o >>= 1) {
                    const double t_ne = __shfl_xor(r_ne, o);
                    const int t_s = __shfl_xor(r_s, o);
                    const bool take = t_s >= 0 && (r_s < 0 || t_ne < r_ne || (t_ne == r_ne && t_s > r_s));
                    if (take) { r_ne = t_ne; r_s = t_s; }
                }
                if (r_s >= 0 && nout < MIRP_MAX_MIRNA_PER_WINDOW) {
                    if (b_s == r_s) {   // exactly one lane owns the winner
                        const PStruct bp = sts[r_s];
                        MirpMirna r;
                        r.window = w; r.tid = W.tid; r.fold_s = b_fold_s; r.fold_e = b_fold_e; r.mat_s = m.start; r.mat_e = m.end;
                        r.star_s = b_star_s; r.star_e = b_star_e; r.strand = m.strand; r.has_star = b_has_star;
                        r.line = bp.line; r.ss_off = bp.off; r.ss_len = bp.len; r.reserved = b_imp;
                        r.total_depth_mature = b_tm; r.total_depth_star = b_ts;
                        out[(size_t)w * MIRP_MAX_MIRNA_PER_WINDOW + nout] = r;
                    }
                    nout++;
                }
            }
        }
        if (lane == 0) { n_out[w] = nout; }
        if (REASONS && lane == 0) {
            const unsigned int ridx = atomicAdd(rcount, 1u);
            if (ridx < rcap) {
                int* r = rpool + (size_t)ridx * rstride;
                for (int q = 0; q < rstride; q++) r[q] = 0;
                r[0] = w; r[1] = -1; r[2] = nst; r[3] = any_in_range ? 1 : 0; r[4] = nout;
            }
        }
        // any lane may have raised a capacity flag
        {
            int f = st_flag;
            for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(f, o); f = t > f ? t : f; }
            if (lane == 0) status[w] = f;
        }
        __syncthreads();
    }
}

#ifndef MIRP_PRED_LCAP
#define MIRP_PRED_LCAP 48
#endif
PredictCaps predict_default_caps(int max_lines, int ss_stride) {
    PredictCaps c;
    c.p_cap = std::max(6, ss_stride / 56 + 1);
#ifdef MIRP_PRED_SCAP_BY_LCAP          // (experiment: structures sized from the staged lines)
    { const int lc = max_lines <= 96 ? std::min(max_lines, MIRP_PRED_LCAP) : max_lines; c.s_cap = 2 * lc > 96 ? 2 * lc : 96; }
#else
    c.s_cap = 2 * max_lines > PW_MIN_STRUCTS ? 2 * max_lines : PW_MIN_STRUCTS;
#endif
    c.m_cap = 64;
    // lines staged per window: the main launch (max_lines = 96) takes 40 -- 99.98 % of the benchmark's windows have fewer printed lines of >= 55
    // characters (profiles/tools/nl_hist.py), the rest is run again; launches over the full-capacity side buffers stage everything
    c.l_cap = max_lines <= 96 ? std::min(max_lines, MIRP_PRED_LCAP) : max_lines;
    return c;
}

size_t predict_lds_bytes(int max_lines, int ss_stride, PredictCaps caps, bool text_in_lds) {
    size_t b = text_in_lds ? (((size_t)caps.l_cap * ((ss_stride + 15) >> 4) * 4 + 15) & ~(size_t)15) : 0;
    b += sizeof(PStruct) * ((size_t)caps.s_cap + 64 * (size_t)caps.p_cap);
    b += sizeof(int) * (64 + (size_t)caps.m_cap);
    b += sizeof(unsigned short) * ((size_t)max_lines + (size_t)caps.l_cap);
    return (b + 15) & ~(size_t)15;
}
size_t predict_lds_bytes(int max_lines, int ss_stride, PredictCaps caps) { return predict_lds_bytes(max_lines, ss_stride, caps, true); }
size_t predict_lds_bytes(int max_lines, int ss_stride) { return predict_lds_bytes(max_lines, ss_stride, predict_default_caps(max_lines, ss_stride)); }
// what has to fit whatever the lines' number and length (structures, pieces, matures): the staged text itself moves to global memory when it does not
size_t predict_lds_bytes_min(int max_lines, int ss_stride, PredictCaps caps) { return predict_lds_bytes(max_lines, ss_stride, caps, false); }
size_t predict_lds_bytes_min(int max_lines, int ss_stride) { return predict_lds_bytes(max_lines, ss_stride, predict_default_caps(max_lines, ss_stride), false); }

template <bool REASONS, bool GTEXT>
static hipError_t launch_predict_as(hipStream_t stream, int grid, size_t lds, const MirpWindow* windows, int n_windows, const MirpMature* matures,
                                    const MirpAln* alns, long long n_alns, const MirpFoldLine* lines, const char* ss, int ss_stride, int max_lines,
                                    const int* n_lines, MirpPredictParams pp, MirpMirna* out, int* n_out, int* status, unsigned int* rcount, int* rpool,
                                    unsigned int rcap, int rstride, const int* wsel, int n_sel, const int* skip, const int* wslot, PredictCaps caps, int* need, unsigned* gtext) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)predict_kernel<REASONS, GTEXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((predict_kernel<REASONS, GTEXT>), dim3(grid), dim3(64), lds, stream, windows, n_windows, matures, alns, n_alns, lines, ss, ss_stride,
                       max_lines, n_lines, pp, out, n_out, status, REASONS ? rcount : nullptr, REASONS ? rpool : nullptr, REASONS ? rcap : 0u, REASONS ? rstride : 0,
                       wsel, n_sel, skip, wslot, caps, need, gtext);
    return hipGetLastError();
}

hipError_t launch_predict(hipStream_t stream, int grid, const MirpWindow* windows, int n_windows, const MirpMature* matures,
                          const MirpAln* alns, long long n_alns, const MirpFoldLine* lines, const char* ss, int ss_stride, int max_lines,
                          const int* n_lines, MirpPredictParams pp, MirpMirna* out, int* n_out, int* status, unsigned int* rcount, int* rpool,
                          unsigned int rcap, int rstride, const int* wsel, int n_sel, const int* skip, const int* wslot, const PredictCaps* caps_in, int* need) {
    const PredictCaps caps = caps_in ? *caps_in : predict_default_caps(max_lines, ss_stride);
    size_t lds = predict_lds_bytes(max_lines, ss_stride, caps);
#ifdef MIRP_PRED_LDS_PAD
    lds += MIRP_PRED_LDS_PAD;      // timing experiment: fewer resident windows per CU
#endif
    if (lds <= 160 * 1024) {
        if (rpool)
            return launch_predict_as<true, false>(stream, grid, lds, windows, n_windows, matures, alns, n_alns, lines, ss, ss_stride, max_lines, n_lines, pp, out, n_out, status,
                                                  rcount, rpool, rcap, rstride, wsel, n_sel, skip, wslot, caps, need, nullptr);
        return launch_predict_as<false, false>(stream, grid, lds, windows, n_windows, matures, alns, n_alns, lines, ss, ss_stride, max_lines, n_lines, pp, out, n_out, status,
                                               rcount, rpool, rcap, rstride, wsel, n_sel, skip, wslot, caps, need, nullptr);
    }
    // the staged text does not fit in LDS: a scratch area per workgroup in global memory (a few workgroups: the area is l_cap lines of ss_stride / 4 bytes each),
    // allocated and released around the launch -- the path of PRECURSOR_LEN in the thousands, not of the benchmark
    lds = predict_lds_bytes(max_lines, ss_stride, caps, false);
    const size_t per = (size_t)caps.l_cap * ((ss_stride + 15) >> 4) * 4;
    grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)grid, ((size_t)2 << 30) / std::max<size_t>(per, 1)));
    unsigned* gtext = nullptr;
    hipError_t e = hipMalloc((void**)&gtext, per * (size_t)grid + 16);
    if (e != hipSuccess) return e;
    if (rpool)
        e = launch_predict_as<true, true>(stream, grid, lds, windows, n_windows, matures, alns, n_alns, lines, ss, ss_stride, max_lines, n_lines, pp, out, n_out, status,
                                          rcount, rpool, rcap, rstride, wsel, n_sel, skip, wslot, caps, need, gtext);
    else
        e = launch_predict_as<false, true>(stream, grid, lds, windows, n_windows, matures, alns, n_alns, lines, ss, ss_stride, max_lines, n_lines, pp, out, n_out, status,
                                           rcount, rpool, rcap, rstride, wsel, n_sel, skip, wslot, caps, need, gtext);
    const hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(gtext);
    return e != hipSuccess ? e : e2;
}

// records of the first pass that belong to windows which are run again: r[0] = -1 (the host drops them)
__global__ void reasons_invalidate_kernel(int* __restrict__ rpool, unsigned int n, int rstride, const unsigned char* __restrict__ flagged) {
    for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        int* r = rpool + (size_t)k * rstride;
        if (r[0] >= 0 && flagged[r[0]]) r[0] = -1;
    }
}

// One launch of the filter plus, when windows exceeded a capacity, a second launch over just those windows with capacities sized from what they
// reported (need[]) and their number of candidate matures.  scope: the launch covers the windows wsel[0..n_sel) (slots wslot or k) or, with
// wsel == nullptr, every window except skip[w] >= 0.  Returns 0, or a negative code with *err set (LDS budget exceeded: -5).
int run_predict_launch(hipStream_t stream, int n_cu, const MirpWindow* windows, int n_windows, const MirpMature* matures, const MirpAln* alns, long long n_alns,
                       const MirpFoldLine* lines, const char* ss, int ss_stride, int max_lines, const int* n_lines, MirpPredictParams pp, MirpMirna* out, int* n_out,
                       int* status, unsigned int* rcount, int* rpool, unsigned int rcap, int rstride, const int* wsel, int n_sel, const int* skip, std::string* err,
                       int* need_buf) {
    const int n_iter = wsel ? n_sel : n_windows;
    if (n_iter <= 0) return 0;
    // need_buf: 12 * n_windows bytes of the caller's (the resident pipeline keeps one; an allocation and a release per call cost more than the
    // filter's own device work on a small batch), else allocated here
    int* need = need_buf;
    if (!need && hipMalloc((void**)&need, 12 * (size_t)n_windows) != hipSuccess) { *err = "device allocation failed (predict)"; return -6; }
    struct Free { int* p; int* a = nullptr; int* b = nullptr; unsigned char* f = nullptr; ~Free() { if (p) (void)hipFree(p); if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (f) (void)hipFree(f); } } guard{need_buf ? nullptr : need};
    if (hipMemsetAsync(need, 0, 12 * (size_t)n_windows, stream) != hipSuccess) { *err = "memset failed"; return -2; }
#ifndef MIRP_PRED_GRID
#define MIRP_PRED_GRID 64
#endif
    const int grid = std::min(n_iter, n_cu * MIRP_PRED_GRID);
    if (launch_predict(stream, grid, windows, n_windows, matures, alns, n_alns, lines, ss, ss_stride, max_lines, n_lines, pp, out, n_out, status, rcount, rpool, rcap, rstride,
                       wsel, n_sel, skip, nullptr, nullptr, need) != hipSuccess) { *err = "predict kernel launch failed"; return -2; }
    std::vector<int> h_status((size_t)n_windows), h_sel, h_skip;
    unsigned int n1 = 0;
    if (hipMemcpyAsync(h_status.data(), status, 4 * (size_t)n_windows, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        (rcount && hipMemcpyAsync(&n1, rcount, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) || hipStreamSynchronize(stream) != hipSuccess) {
        *err = "predict kernel execution failed"; return -2;
    }
    if (wsel) { h_sel.resize((size_t)n_sel); if (hipMemcpy(h_sel.data(), wsel, 4 * (size_t)n_sel, hipMemcpyDeviceToHost) != hipSuccess) { *err = "D2H failed"; return -2; } }
    else if (skip) { h_skip.resize((size_t)n_windows); if (hipMemcpy(h_skip.data(), skip, 4 * (size_t)n_windows, hipMemcpyDeviceToHost) != hipSuccess) { *err = "D2H failed"; return -2; } }
    std::vector<int> rw, rs;          // windows to run again, their slots
    for (int it = 0; it < n_iter; it++) {
        const int w = wsel ? h_sel[(size_t)it] : it;
        if (!wsel && skip && h_skip[(size_t)it] >= 0) continue;
        if (h_status[(size_t)w] == 2 || h_status[(size_t)w] == 3) { rw.push_back(w); rs.push_back(it); }
    }
    if (rw.empty()) return 0;
    std::vector<MirpWindow> h_w((size_t)n_windows);
    if (hipMemcpy(h_w.data(), windows, sizeof(MirpWindow) * (size_t)n_windows, hipMemcpyDeviceToHost) != hipSuccess) { *err = "D2H failed"; return -2; }
    if (hipMalloc((void**)&guard.a, 4 * rw.size()) != hipSuccess || hipMalloc((void**)&guard.b, 4 * rw.size()) != hipSuccess) { *err = "device allocation failed (predict)"; return -6; }
    // (not gated on the first pass's record count: a re-run round can write the first records of a window that the next round voids)
    if (rpool && hipMalloc((void**)&guard.f, (size_t)n_windows) != hipSuccess) { *err = "device allocation failed (predict)"; return -6; }
    PredictCaps caps = predict_default_caps(max_lines, ss_stride);
    std::vector<int> h_need(3 * (size_t)n_windows);
    // A flagged window says what it needs (need[]), but what it says can be short of the truth: a line that was not staged contributed neither its
    // structures nor its pieces.  So the windows that are still flagged after a re-run go round again with what they report then; every round
    // stages at least the lines the one before asked for, so the third has seen everything.
    for (int round = 0; round < 3 && !rw.empty(); round++) {
        if (hipMemcpy(h_need.data(), need, 12 * (size_t)n_windows, hipMemcpyDeviceToHost) != hipSuccess) { *err = "D2H failed"; return -2; }
        for (int w : rw) {
            caps.s_cap = std::max(caps.s_cap, h_need[3 * (size_t)w]);
            caps.p_cap = std::max(caps.p_cap, h_need[3 * (size_t)w + 1]);
            caps.l_cap = std::max(caps.l_cap, std::min(h_need[3 * (size_t)w + 2], max_lines));
            caps.m_cap = std::max(caps.m_cap, h_w[(size_t)w].n_matures);
        }
        if (predict_lds_bytes_min(max_lines, ss_stride, caps) > 160 * 1024) {
            *err = "a window has more structures / candidate matures than the filter kernel can hold in LDS (" + std::to_string(caps.s_cap) + " structures, " +
                   std::to_string(caps.m_cap) + " matures)";
            return -5;
        }
        if (hipMemcpy(guard.a, rw.data(), 4 * rw.size(), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(guard.b, rs.data(), 4 * rs.size(), hipMemcpyHostToDevice) != hipSuccess) {
            *err = "H2D failed"; return -2;
        }
        if (rpool) {          // records written so far for the windows of THIS round are void (truncated by a capacity)
            std::vector<unsigned char> fl((size_t)n_windows, 0);
            for (int w : rw) fl[(size_t)w] = 1;
            if (hipMemcpy(guard.f, fl.data(), (size_t)n_windows, hipMemcpyHostToDevice) != hipSuccess) { *err = "H2D failed"; return -2; }
            unsigned int n2 = 0;
            if (hipMemcpy(&n2, rcount, 4, hipMemcpyDeviceToHost) != hipSuccess) { *err = "D2H failed"; return -2; }
            const unsigned int nn = std::min(n2, rcap);
            if (nn > 0)
                hipLaunchKernelGGL(reasons_invalidate_kernel, dim3((nn + 255) / 256 > 4096 ? 4096 : (nn + 255) / 256), dim3(256), 0, stream, rpool, nn, rstride, (const unsigned char*)guard.f);
        }
        if (launch_predict(stream, std::min((int)rw.size(), n_cu * 4), windows, n_windows, matures, alns, n_alns, lines, ss, ss_stride, max_lines, n_lines, pp, out, n_out, status,
                           rcount, rpool, rcap, rstride, guard.a, (int)rw.size(), nullptr, guard.b, &caps, need) != hipSuccess) { *err = "predict kernel launch failed (re-run)"; return -2; }
        if (hipMemcpyAsync(h_status.data(), status, 4 * (size_t)n_windows, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            *err = "predict kernel execution failed (re-run)"; return -2;
        }
        std::vector<int> rw2, rs2;
        for (size_t k = 0; k < rw.size(); k++)
            if (h_status[(size_t)rw[k]] == 2 || h_status[(size_t)rw[k]] == 3) { rw2.push_back(rw[k]); rs2.push_back(rs[k]); }
        rw.swap(rw2); rs.swap(rs2);
    }
    return 0;
}

}  // namespace mirp
