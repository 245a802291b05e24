/*
 * oracle/candidate.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's candidate stage (rows a1-a6 of SURVEY.md section 8):
 *   a1  read weights            /root/reference/miR_PREFeR.py:716-746 (expand_bamfile), :759-769, :861-873
 *   a2  coverage -> peaks       :877-962 (gen_contig_typeA / get_next_non_zero_region), samtools depth | awk :937-941
 *   a3  merge + windows         :1246-1371 (next_region_typeA :1256-1270, extend_region :1272-1300)
 *   a4  window sequences        :1070-1198 (dump_piece), :232-239 (get_reverse_complement)
 *   a5  per-position read table :1374-1392, :1395-1468
 *   a6  candidate matures       :1471-1510
 * Pinned against fixtures generated from the real reference stack (tests/golden/*\/expected.json.gz).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ---- a1 + a2 ------------------------------------------------------------------------------ */
/* Alignments are ungapped (`<len>M`), so depth = sum of min(depth, cutoff) over covering reads
 * (each SAM line is physically repeated min(xN, READS_DEPTH_CUTOFF) times, MP:734-738), split
 * by flag 16 (MP:870-873).  `samtools depth A B | awk '$3+$4>CUT'` (MP:937-938) keeps positions
 * with d+ + d- > cutoff; the scan at MP:879-933 forms maximal runs of consecutive positions. */
int oracle_coverage_peaks(const OracleAln *a, size_t n, const int64_t *contig_len, int n_contigs, int cutoff, int min_len,
                          OracleDepthPos **depth_out, size_t *n_depth, OraclePeak **peaks_out, size_t *n_peaks) {
    size_t dcap = 1024, pcap = 256, nd = 0, np = 0;
    OracleDepthPos *D = (OracleDepthPos *)malloc(dcap * sizeof(*D));
    OraclePeak *P = (OraclePeak *)malloc(pcap * sizeof(*P));
    int have_prev_contig = 0; /* some earlier contig already produced a line of the depth file */
    size_t ai = 0;
    for (int t = 0; t < n_contigs; t++) {
        int64_t L = contig_len[t];
        int32_t *dp = (int32_t *)calloc((size_t)L + 2, sizeof(int32_t));
        int32_t *dm = (int32_t *)calloc((size_t)L + 2, sizeof(int32_t));
        /* records are sorted by tid in @SQ order */
        while (ai < n && a[ai].tid < t) ai++;
        for (; ai < n && a[ai].tid == t; ai++) {
            int w = (int)(a[ai].depth > (uint32_t)cutoff ? (uint32_t)cutoff : a[ai].depth);
            int64_t s = a[ai].pos, e = (int64_t)a[ai].pos + a[ai].len; /* [s,e) 1-based */
            if (s < 1) s = 1;
            if (e > L + 1) e = L + 1;
            if (s >= e) continue;
            int32_t *d = a[ai].strand ? dm : dp;
            d[s] += w;
            d[e] -= w;
        }
        int64_t run_start = 0, sum_p = 0, sum_m = 0, prev = 0;
        int in_run = 0, first_line_of_contig = 1;
        int32_t cp = 0, cm = 0;
        for (int64_t x = 1; x <= L + 1; x++) {
            if (x <= L) { cp += dp[x]; cm += dm[x]; }
            int above = (x <= L) && (cp + cm > cutoff);
            if (above) {
                if (nd == dcap) { dcap *= 2; D = (OracleDepthPos *)realloc(D, dcap * sizeof(*D)); }
                D[nd].tid = t; D[nd].pos = (int32_t)x; D[nd].dp = cp; D[nd].dm = cm; nd++;
                if (!in_run) {
                    in_run = 1; run_start = x; sum_p = cp; sum_m = cm;
                    /* MP:905-906 then :926-929: on a contig change the first position is added twice */
                    if (first_line_of_contig && have_prev_contig) { sum_p += cp; sum_m += cm; }
                } else { sum_p += cp; sum_m += cm; }
                first_line_of_contig = 0;
                prev = x;
            } else if (in_run) {
                in_run = 0;
                if (prev + 1 - run_start >= min_len) { /* MP:956 */
                    if (np == pcap) { pcap *= 2; P = (OraclePeak *)realloc(P, pcap * sizeof(*P)); }
                    P[np].tid = t; P[np].start = (int32_t)run_start; P[np].end = (int32_t)(prev + 1);
                    P[np].strand = (sum_p > sum_m) ? 0 : 1; /* ties -> '-' (MP:901-904) */
                    np++;
                }
            }
        }
        if (!first_line_of_contig) have_prev_contig = 1;
        free(dp); free(dm);
    }
    *depth_out = D; *n_depth = nd; *peaks_out = P; *n_peaks = np;
    return 0;
}

/* ---- a3 ------------------------------------------------------------------------------------ */
/* extend_region (MP:1272-1300); returns number of windows (0,1,2) */
static int extend_region(int s, int e, int L, int64_t seqlen, int out[2][2]) {
    int length = e - s;
    if (length > L + 50) return 0;
    if (length > L) { out[0][0] = s; out[0][1] = e; return 1; }
    if (length < 60) {
        int64_t ls = (int64_t)s - (L - length - 25) - 25, le = (int64_t)e + 25;
        int64_t rs = (int64_t)s - 25, re = (int64_t)e + (L - length - 25) + 25;
        if (ls < 0) ls = 0;
        if (le > seqlen) le = seqlen;
        if (rs < 0) rs = 0;
        if (re > seqlen) re = seqlen;
        out[0][0] = (int)ls; out[0][1] = (int)le; out[1][0] = (int)rs; out[1][1] = (int)re;
        return 2;
    }
    int ext = (L - length) / 2; /* py2 floor division of non-negative ints */
    int64_t left = (int64_t)s - ext, right = (int64_t)e + ext;
    if (left < 0) left = 1;
    if (right > seqlen) right = seqlen + 1;
    out[0][0] = (int)left; out[0][1] = (int)right;
    return 1;
}

static char rc_char(char c) { /* get_complement MP:232-235: upper-case ATGCU only */
    switch (c) { case 'A': return 'U'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; case 'U': return 'A'; default: return c; }
}

typedef struct { int len_max, depth_max, total; int present; } PosInfo;

/* a5: reads with ws <= pos <= we (MP:1439) of contig tid; per (pos,strand) most abundant read, first-seen max (MP:1457) */
static void build_pos_table(const OracleAln *a, size_t lo, size_t hi, int ws, int we, PosInfo *tab /* [2][we-ws+1] */) {
    int W = we - ws + 1;
    memset(tab, 0, sizeof(PosInfo) * 2 * (size_t)W);
    for (size_t k = lo; k < hi; k++) {
        if (a[k].pos < ws || a[k].pos > we) continue;
        PosInfo *p = &tab[(size_t)a[k].strand * W + (a[k].pos - ws)];
        if (!p->present) { p->present = 1; p->len_max = a[k].len; p->depth_max = (int)a[k].depth; p->total = (int)a[k].depth; }
        else {
            p->total += (int)a[k].depth;
            if ((int)a[k].depth > p->depth_max) { p->len_max = a[k].len; p->depth_max = (int)a[k].depth; }
        }
    }
}

/* a6: gen_matures_one_peak (MP:1472-1490) */
static int matures_one_peak(const PosInfo *tab, int ws, int we, int strand, double min_depth, int ps, int pe, OracleMature *out) {
    int W = we - ws + 1, n = 0;
    OracleMature hi = {0, 0, 0, 0};
    int hi_set = 0;
    for (int pos = ps - 20; pos < pe; pos++) {
        if (pos < ws || pos > we) continue;
        const PosInfo *p = &tab[(size_t)strand * W + (pos - ws)];
        if (!p->present) continue;
        if (p->depth_max > hi.depth) { hi.start = pos; hi.end = pos + p->len_max; hi.strand = strand; hi.depth = p->depth_max; hi_set = 1; }
        if ((double)p->depth_max > min_depth) {
            OracleMature m = {pos, pos + p->len_max, strand, p->depth_max};
            if (n < 2) out[n++] = m;
            else if (p->depth_max > out[n - 1].depth) out[n - 1] = m;
        }
    }
    if (n == 0) {
        if (!hi_set) { hi.start = 0; hi.end = 0; hi.strand = 0; hi.depth = 0; hi.strand = -1; /* (0,0,0,0): strand field is int 0 */ }
        out[n++] = hi;
    }
    return n;
}

static size_t lower_bound_aln(const OracleAln *a, size_t n, int tid, int64_t pos) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (a[mid].tid < tid || (a[mid].tid == tid && a[mid].pos < pos)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

typedef struct {
    OracleWindow *w; size_t nw, capw;
    OraclePeak *pk; size_t npk, cappk;
    OracleMature *mt; size_t nmt, capmt;
    char *seq; size_t nseq, capseq;
} Builder;

static void push_window(Builder *B, const OracleAln *a, size_t na, const char *const *genome, const int64_t *contig_len,
                        int tid, int ws, int we, int strand, int loc_s, int loc_e, int tag, const OraclePeak *peaks, int npeaks,
                        int only_strand /* -1 = all peaks listed */, double min_mature_depth) {
    if (B->nw == B->capw) { B->capw = B->capw ? B->capw * 2 : 256; B->w = (OracleWindow *)realloc(B->w, B->capw * sizeof(OracleWindow)); }
    OracleWindow *w = &B->w[B->nw++];
    w->tid = tid; w->ws = ws; w->we = we; w->strand = strand; w->loc_s = loc_s; w->loc_e = loc_e; w->tag = tag;
    /* header peak list: all peaks (single-strand case, MP:1111-1114) or only this strand's (MP:1157-1175) */
    w->peak_off = (int64_t)B->npk; w->n_peaks = 0;
    for (int k = 0; k < npeaks; k++) {
        if (only_strand >= 0 && peaks[k].strand != only_strand) continue;
        if (B->npk == B->cappk) { B->cappk = B->cappk ? B->cappk * 2 : 256; B->pk = (OraclePeak *)realloc(B->pk, B->cappk * sizeof(OraclePeak)); }
        B->pk[B->npk++] = peaks[k]; w->n_peaks++;
    }
    /* a5 + a6 */
    int W = we - ws + 1;
    PosInfo *tab = (PosInfo *)malloc(sizeof(PosInfo) * 2 * (size_t)(W > 0 ? W : 1));
    size_t lo = lower_bound_aln(a, na, tid, ws), hi = lower_bound_aln(a, na, tid, (int64_t)we + 1);
    build_pos_table(a, lo, hi, ws, we, tab);
    w->mature_off = (int64_t)B->nmt; w->n_matures = 0;
    for (int k = 0; k < npeaks; k++) {
        if (peaks[k].strand != strand) continue; /* MP:1494 */
        OracleMature tmp[2];
        int nm = matures_one_peak(tab, ws, we, strand, min_mature_depth, peaks[k].start, peaks[k].end, tmp);
        for (int x = 0; x < nm; x++) {
            if (B->nmt == B->capmt) { B->capmt = B->capmt ? B->capmt * 2 : 256; B->mt = (OracleMature *)realloc(B->mt, B->capmt * sizeof(OracleMature)); }
            B->mt[B->nmt++] = tmp[x]; w->n_matures++;
        }
    }
    free(tab);
    /* a4: samtools faidx chr:ws-(we-1) (MP:1098-1105): start 0 is treated as 1, end clamped to the contig */
    int64_t s = ws < 1 ? 1 : ws, e = (int64_t)we - 1;
    if (e > contig_len[tid]) e = contig_len[tid];
    int64_t len = e >= s ? e - s + 1 : 0;
    if (B->nseq + (size_t)len + 1 > B->capseq) {
        while (B->nseq + (size_t)len + 1 > B->capseq) B->capseq = B->capseq ? B->capseq * 2 : 65536;
        B->seq = (char *)realloc(B->seq, B->capseq);
    }
    w->seq_off = (int64_t)B->nseq; w->seq_len = (int32_t)len;
    const char *g = genome[tid];
    if (strand == 0) memcpy(B->seq + B->nseq, g + (s - 1), (size_t)len);
    else for (int64_t x = 0; x < len; x++) B->seq[B->nseq + x] = rc_char(g[(e - 1) - x]);
    B->nseq += (size_t)len;
}

/* a3-a6 driver.  peaks: per contig in position order (as a2 emits them); contig_order: the order
 * `sorted(dict_contigs)` visits contigs (MP:1309).  Emits one OracleWindow per FASTA entry in the
 * order dump_piece writes them, including the both-strand L/R duplication (MP:1184-1192). */
int oracle_make_windows(const OraclePeak *peaks, size_t n_peaks, const OracleAln *a, size_t na, const char *const *genome,
                        const int64_t *contig_len, int n_contigs, const int *contig_order, int max_gap, int precursor_len,
                        double min_mature_depth, OracleWindow **w_out, size_t *nw_out, OraclePeak **wpeaks_out, size_t *nwpeaks,
                        OracleMature **mat_out, size_t *nmat, char **seq_out, size_t *nseq, OracleLocus **loci_out, size_t *nloci_out) {
    Builder B; memset(&B, 0, sizeof(B));
    OracleLocus *loci = NULL; size_t nloci = 0, caploci = 0;
    for (int oi = 0; oi < n_contigs; oi++) {
        int t = contig_order[oi];
        size_t lo = 0;
        while (lo < n_peaks && peaks[lo].tid != t) lo++;
        size_t hi = lo;
        while (hi < n_peaks && peaks[hi].tid == t) hi++;
        size_t k = lo;
        while (k < hi) {
            /* next_region_typeA (MP:1256-1270) */
            size_t first = k;
            int rs = peaks[k].start, re = peaks[k].end;
            k++;
            while (k < hi && peaks[k].start - re < max_gap) { re = peaks[k].end; k++; }
            int npk = (int)(k - first);
            int ext[2][2];
            int nwin = extend_region(rs, re, precursor_len, contig_len[t], ext);
            if (nwin == 0) continue;
            if (nloci == caploci) { caploci = caploci ? caploci * 2 : 256; loci = (OracleLocus *)realloc(loci, caploci * sizeof(OracleLocus)); }
            OracleLocus *lc = &loci[nloci++];
            lc->tid = t; lc->start = rs; lc->end = re; lc->n_windows = nwin;
            lc->w[0][0] = ext[0][0]; lc->w[0][1] = ext[0][1]; lc->w[1][0] = nwin > 1 ? ext[1][0] : 0; lc->w[1][1] = nwin > 1 ? ext[1][1] : 0;
            lc->peak_first = (int64_t)first; lc->n_peaks = npk;
            int has_p = 0, has_m = 0;
            for (size_t x = first; x < k; x++) { if (peaks[x].strand == 0) has_p = 1; else has_m = 1; }
            int nacc = 0; /* length of plus_tag / minus_tag lists (MP:1167,1180) */
            for (int idx = 0; idx < nwin; idx++) {
                int tag = nwin == 2 ? (idx == 0 ? 1 : 2) : 0; /* 0, L, R (MP:1116-1120) */
                if (has_p != has_m) {
                    push_window(&B, a, na, genome, contig_len, t, ext[idx][0], ext[idx][1], has_p ? 0 : 1, rs, re, tag, peaks + first, npk, -1, min_mature_depth);
                } else {
                    nacc++;
                    /* write-out block inside the loop (MP:1184-1192): all accumulated '+' then all '-' */
                    for (int s = 0; s < 2; s++)
                        for (int j = 0; j < nacc; j++) {
                            int tg = nwin == 2 ? (j == 0 ? 1 : 2) : 0;
                            push_window(&B, a, na, genome, contig_len, t, ext[j][0], ext[j][1], s, rs, re, tg, peaks + first, npk, s, min_mature_depth);
                        }
                }
            }
        }
    }
    *w_out = B.w; *nw_out = B.nw; *wpeaks_out = B.pk; *nwpeaks = B.npk; *mat_out = B.mt; *nmat = B.nmt; *seq_out = B.seq; *nseq = B.nseq;
    *loci_out = loci; *nloci_out = nloci;
    return 0;
}

void oracle_free(void *p) { free(p); }
