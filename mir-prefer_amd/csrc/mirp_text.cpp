// Stage artefacts in the reference's own text formats, written by native worker threads straight from the device-resident state: the
// thresholded depth file `bam.depth.cut<CUT>` (MP:946-949), the folder's input FASTA `<prefix>.rnalfold.in_<i>.fa` (MP:1124-1142) and the
// fold stage's RNALfold-format output (MP:3085-3098, consumed by MP:1541-1599).  These files are what the reference's stages exchange; on
// the host they were the bulk of the end-to-end wall-clock (Python string formatting of 10^5..10^6 lines), here a stage formats its chunk of
// lines per thread into memory and the file is written in one go.
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "mirp_ctx.h"

namespace {

int n_workers(size_t items) {
    unsigned hw = std::thread::hardware_concurrency();
    int n = (int)std::min<size_t>(hw ? hw : 4, 16);
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)n, items / 256 + 1));
}

// fn(first, last, std::string& out) formats items [first, last) into out; the pieces are written to path in order
template <class F>
int write_parallel(mirp_ctx* c, const char* path, size_t items, size_t bytes_per_item_hint, F fn, const char* who) {
    const int nt = n_workers(items);
    std::vector<std::string> parts((size_t)nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            const size_t a = items * t / nt, b = items * (t + 1) / nt;
            parts[t].reserve((b - a) * bytes_per_item_hint + 64);
            fn(a, b, parts[t]);
        });
    for (auto& t : th) t.join();
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(c, -8, std::string(who) + ": cannot open " + path);
    bool ok = true;
    for (auto& p : parts)
        if (!p.empty() && std::fwrite(p.data(), 1, p.size(), f) != p.size()) ok = false;
    if (std::fclose(f) != 0) ok = false;
    return ok ? 0 : fail(c, -8, std::string(who) + ": I/O error on " + path);
}

inline void put_int(std::string& o, long long v) {
    char b[24];
    int n = 0;
    bool neg = v < 0;
    unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) o.push_back('-');
    while (n) o.push_back(b[--n]);
}

std::vector<std::string> split_names(const char* blob, int n) {
    std::vector<std::string> v;
    const char* p = blob;
    for (int k = 0; k < n; k++) { v.emplace_back(p); p += v.back().size() + 1; }
    return v;
}

template <class T>
T* host_copy2(mirp_ctx* c, const void* dev, size_t n) {
    T* h = (T*)std::malloc(std::max<size_t>(n, 1) * sizeof(T));
    if (!h) return nullptr;
    if (n && hipMemcpy(h, dev, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) { std::free(h); return nullptr; }
    return h;
}

}  // namespace

namespace mirp {
struct PackedLine { int32_t len, energy, start; };
// per window: bytes and number of its printed structure lines (the window's own lines, or its slot in the side buffers when it was folded again)
__global__ void fold_text_size_kernel(const MirpFoldLine* __restrict__ lines, const int* __restrict__ nlines, int max_lines, const int* __restrict__ side_idx,
                                      const MirpFoldLine* __restrict__ lines2, const int* __restrict__ nlines2, int max_lines2, long long n, int* __restrict__ sz,
                                      int* __restrict__ cnt) {
    for (long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x; w < n; w += (long long)gridDim.x * blockDim.x) {
        const int sk = side_idx ? side_idx[w] : -1;
        const MirpFoldLine* wl = sk >= 0 ? lines2 + (size_t)sk * max_lines2 : lines + (size_t)w * max_lines;
        const int nl = sk >= 0 ? min(nlines2[sk], max_lines2) : min(nlines[w], max_lines);
        int s = 0, k2 = 0;
        for (int k = 0; k < nl; k++) if (wl[k].printed) { s += wl[k].len; k2++; }
        sz[w] = s; cnt[w] = k2;
    }
}
// one wavefront per window: its printed structure texts back to back at packed[base[w]..], their {len, energy, start} at plines[lbase[w]..]
__global__ void fold_text_pack_kernel(const MirpFoldLine* __restrict__ lines, const char* __restrict__ ss, const int* __restrict__ nlines, int max_lines, int stride,
                                      const int* __restrict__ side_idx, const MirpFoldLine* __restrict__ lines2, const char* __restrict__ ss2,
                                      const int* __restrict__ nlines2, int max_lines2, long long n, const long long* __restrict__ base,
                                      const long long* __restrict__ lbase, char* __restrict__ packed, PackedLine* __restrict__ plines) {
    const int lane = threadIdx.x & 63;
    for (long long w = blockIdx.x * (long long)(blockDim.x / 64) + (threadIdx.x >> 6); w < n; w += (long long)gridDim.x * (blockDim.x / 64)) {
        const int sk = side_idx ? side_idx[w] : -1;
        const MirpFoldLine* wl = sk >= 0 ? lines2 + (size_t)sk * max_lines2 : lines + (size_t)w * max_lines;
        const char* wt = sk >= 0 ? ss2 + (size_t)sk * max_lines2 * stride : ss + (size_t)w * max_lines * stride;
        const int nl = sk >= 0 ? min(nlines2[sk], max_lines2) : min(nlines[w], max_lines);
        long long o = base[w], lo = lbase[w];
        for (int k = 0; k < nl; k++) {
            const MirpFoldLine ln = wl[k];
            if (!ln.printed) continue;
            for (int x = lane; x < ln.len; x += 64) packed[o + x] = wt[(size_t)k * stride + x];
            if (lane == 0) { PackedLine p; p.len = ln.len; p.energy = ln.energy; p.start = ln.start; plines[lo] = p; }
            o += ln.len; lo++;
        }
    }
}
}  // namespace mirp

extern "C" int mirp_write_depth_text(mirp_ctx* c, const char* path, const char* contig_names, int32_t n_names) {
    if (!c) return -1;
    if (!path || !contig_names || n_names < c->n_contigs) return fail(c, -1, "mirp_write_depth_text: bad argument");
    MirpDepthPos* d = nullptr;
    int64_t n = 0;
    if (int rc = mirp_get_depth(c, &d, &n)) return rc;
    const std::vector<std::string> names = split_names(contig_names, n_names);
    const int rc = write_parallel(c, path, (size_t)n, 24, [&](size_t a, size_t b, std::string& o) {
        for (size_t k = a; k < b; k++) {
            o += names[(size_t)d[k].tid]; o.push_back('\t');
            put_int(o, d[k].pos); o.push_back('\t'); put_int(o, d[k].dp); o.push_back('\t'); put_int(o, d[k].dm); o.push_back('\n');
        }
    }, "mirp_write_depth_text");
    std::free(d);
    return rc;
}

extern "C" int mirp_write_window_fasta(mirp_ctx* c, const char* path, const char* contig_names, int32_t n_names) {
    if (!c) return -1;
    if (!path || !contig_names || n_names < c->n_contigs) return fail(c, -1, "mirp_write_window_fasta: bad argument");
    MirpWindow* W = nullptr; MirpPeak* P = nullptr; MirpMature* M = nullptr; char* S = nullptr;
    int64_t nw = 0, np = 0, nm = 0, nb = 0;
    if (int rc = mirp_get_windows(c, &W, &nw, &P, &np, &M, &nm, &S, &nb)) return rc;
    const std::vector<std::string> names = split_names(contig_names, n_names);
    static const char STRAND[2] = {'+', '-'};
    static const char TAG[3] = {'0', 'L', 'R'};
    const int rc = write_parallel(c, path, (size_t)nw, 512, [&](size_t a, size_t b, std::string& o) {
        for (size_t k = a; k < b; k++) {
            const MirpWindow& w = W[k];
            // `>chr:ws-we strand locS-locE tag s,e,strand;... M:s-e/strand/depth ...` (MP:1124-1140, 1162-1178)
            o.push_back('>'); o += names[(size_t)w.tid]; o.push_back(':'); put_int(o, w.ws); o.push_back('-'); put_int(o, w.we); o.push_back(' ');
            o.push_back(STRAND[w.strand & 1]); o.push_back(' '); put_int(o, w.loc_s); o.push_back('-'); put_int(o, w.loc_e); o.push_back(' ');
            o.push_back(TAG[w.tag]); o.push_back(' ');
            for (int x = 0; x < w.n_peaks; x++) {
                const MirpPeak& p = P[w.peak_off + x];
                if (x) o.push_back(';');
                put_int(o, p.start); o.push_back(','); put_int(o, p.end); o.push_back(','); o.push_back(STRAND[p.strand & 1]);
            }
            for (int x = 0; x < w.n_matures; x++) {
                const MirpMature& m = M[w.mature_off + x];
                o += " M:";
                if (m.strand < 0) o += "0-0/0/0";      // the (0,0,0,0) fallback of gen_possible_matures_loci (MP:1476)
                else { put_int(o, m.start); o.push_back('-'); put_int(o, m.end); o.push_back('/'); o.push_back(STRAND[m.strand & 1]); o.push_back('/'); put_int(o, m.depth); }
            }
            o.push_back('\n');
            o.append(S + w.seq_off, (size_t)w.seq_len);
            o.push_back('\n');
        }
    }, "mirp_write_window_fasta");
    std::free(W); std::free(P); std::free(M); std::free(S);
    return rc;
}

// `<prefix>_rnalfoldoutput_<i>`: for every resident window the `>` header line of the candidate stage's FASTA, the printed structure lines
// "%s (%6.2f) %4d", the upper-cased T->U sequence and " (%6.2f)" (MP:3085-3098).  Phase 1 (on the context's stream, a few milliseconds): the
// printed structure texts and their {length, energy, start} are packed on the device into buffers this call owns -- only the printed
// characters cross the bus, not the padded line slots.  Phase 2 (host, optionally behind the caller's next stage): the copies to the host,
// the formatting by worker threads and the write.  Phase 2 touches nothing of the context.
static int write_fold_text_impl(mirp_ctx* c, const char* fasta_path, const char* out_path, bool async) {
    if (!c) return -1;
    if (!fasta_path || !out_path) return fail(c, -1, "mirp_write_fold_text: null argument");
    if (!c->have_fold) return fail(c, -1, "mirp_write_fold_text: run mirp_fold first");
    HIPCHK(c, hipSetDevice(c->device));
    const long long nw = c->n_windows;
    const size_t ml = (size_t)c->fold_max_lines, stride = (size_t)c->fold_stride;
    const long long ns = c->n_side;
    const size_t ml2 = (size_t)c->side_max_lines;
    struct State {
        std::vector<char> fa;
        std::vector<size_t> line_start;          // two lines per window: header, sequence
        long long nw = 0, total = 0, nlines = 0;
        int device = 0;
        void* d_packed = nullptr; void* d_plines = nullptr; void* d_base = nullptr; void* d_lbase = nullptr; void* d_mfe = nullptr;      // owned
        ~State() { for (void* p : {d_packed, d_plines, d_base, d_lbase, d_mfe}) if (p) (void)hipFree(p); }
    };
    auto S = std::make_shared<State>();
    S->nw = nw; S->device = c->device;
    {
        FILE* fin = std::fopen(fasta_path, "rb");
        if (!fin) return fail(c, -8, "mirp_write_fold_text: cannot open the FASTA file");
        std::fseek(fin, 0, SEEK_END);
        const long sz = std::ftell(fin);
        std::fseek(fin, 0, SEEK_SET);
        S->fa.resize((size_t)std::max<long>(sz, 0));
        if (sz > 0 && std::fread(S->fa.data(), 1, (size_t)sz, fin) != (size_t)sz) { std::fclose(fin); return fail(c, -8, "mirp_write_fold_text: cannot read the FASTA file"); }
        std::fclose(fin);
    }
    S->line_start.reserve(2 * (size_t)nw + 2);
    for (size_t p = 0; p < S->fa.size() && S->line_start.size() < 2 * (size_t)nw;) {
        S->line_start.push_back(p);
        const char* nl = (const char*)memchr(S->fa.data() + p, '\n', S->fa.size() - p);
        p = nl ? (size_t)(nl - S->fa.data()) + 1 : S->fa.size();
    }
    if (S->line_start.size() < 2 * (size_t)nw) return fail(c, -8, "mirp_write_fold_text: I/O error or FASTA shorter than the window list");
    S->line_start.push_back(S->fa.size());
    if (nw > 0) {
        TmpDevice T;
        int* d_sz = (int*)T.get(4 * (size_t)nw);
        int* d_cnt = (int*)T.get(4 * (size_t)nw);
        if (!d_sz || !d_cnt || hipMalloc(&S->d_base, 8 * (size_t)(nw + 1)) != hipSuccess || hipMalloc(&S->d_lbase, 8 * (size_t)(nw + 1)) != hipSuccess ||
            hipMalloc(&S->d_mfe, 4 * (size_t)nw) != hipSuccess)
            return fail(c, -6, "device allocation failed (fold text)");
        const int* d_side = ns > 0 ? (const int*)c->side_idx.p : nullptr;
        hipLaunchKernelGGL(mirp::fold_text_size_kernel, dim3((unsigned)std::min<long long>((nw + 255) / 256, 4096)), dim3(256), 0, c->stream, (const MirpFoldLine*)c->lines.p,
                           (const int*)c->nlines.p, (int)ml, d_side, (const MirpFoldLine*)c->lines2.p, (const int*)c->nlines2.p, (int)ml2, nw, d_sz, d_cnt);
        mirp::launch_excl_scan(c->stream, d_sz, (long long*)S->d_base, nw);
        mirp::launch_excl_scan(c->stream, d_cnt, (long long*)S->d_lbase, nw);
        HIPCHK(c, hipMemcpyAsync(&S->total, (long long*)S->d_base + nw, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&S->nlines, (long long*)S->d_lbase + nw, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(S->d_mfe, c->mfe.p, 4 * (size_t)nw, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (hipMalloc(&S->d_packed, (size_t)S->total + 16) != hipSuccess || hipMalloc(&S->d_plines, sizeof(mirp::PackedLine) * (size_t)std::max<long long>(S->nlines, 1)) != hipSuccess)
            return fail(c, -6, "device allocation failed (fold text)");
        hipLaunchKernelGGL(mirp::fold_text_pack_kernel, dim3((unsigned)std::min<long long>((nw + 3) / 4, 8192)), dim3(256), 0, c->stream, (const MirpFoldLine*)c->lines.p,
                           (const char*)c->ss.p, (const int*)c->nlines.p, (int)ml, (int)stride, d_side, (const MirpFoldLine*)c->lines2.p, (const char*)c->ss2.p,
                           (const int*)c->nlines2.p, (int)ml2, nw, (const long long*)S->d_base, (const long long*)S->d_lbase, (char*)S->d_packed,
                           (mirp::PackedLine*)S->d_plines);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const std::string out(out_path);
    auto work = [S, out](std::string* err) -> int {
        State& s = *S;
        const long long nw = s.nw;
        (void)hipSetDevice(s.device);
        std::vector<char> ht((size_t)s.total + 1);
        std::vector<mirp::PackedLine> hl((size_t)std::max<long long>(s.nlines, 1));
        std::vector<long long> hb((size_t)nw + 1), hlb((size_t)nw + 1);
        std::vector<int32_t> hm((size_t)std::max<long long>(nw, 1));
        if (nw > 0) {
            if ((s.total && hipMemcpy(ht.data(), s.d_packed, (size_t)s.total, hipMemcpyDeviceToHost) != hipSuccess) ||
                (s.nlines && hipMemcpy(hl.data(), s.d_plines, sizeof(mirp::PackedLine) * (size_t)s.nlines, hipMemcpyDeviceToHost) != hipSuccess) ||
                hipMemcpy(hb.data(), s.d_base, 8 * (size_t)(nw + 1), hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(hlb.data(), s.d_lbase, 8 * (size_t)(nw + 1), hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(hm.data(), s.d_mfe, 4 * (size_t)nw, hipMemcpyDeviceToHost) != hipSuccess) {
                *err = "mirp_write_fold_text: D2H failed";
                return -2;
            }
        }
        const int nt = n_workers((size_t)nw);
        std::vector<std::string> parts((size_t)nt);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                const size_t a = (size_t)nw * t / nt, b = (size_t)nw * (t + 1) / nt;
                std::string& o = parts[t];
                o.reserve((b - a) * 6200 + 64);
                char num[48];
                for (size_t w = a; w < b; w++) {
                    const char* head = s.fa.data() + s.line_start[2 * w];
                    const size_t head_len = s.line_start[2 * w + 1] - s.line_start[2 * w];
                    o.append(head, head_len);
                    if (head_len == 0 || head[head_len - 1] != '\n') o.push_back('\n');
                    const char* t2 = ht.data() + hb[w];
                    for (long long k = hlb[w]; k < hlb[w + 1]; k++) {
                        const mirp::PackedLine& ln = hl[(size_t)k];
                        o.append(t2, (size_t)ln.len);
                        t2 += ln.len;
                        const int m = std::snprintf(num, sizeof(num), " (%6.2f) %4d\n", ln.energy / 100., ln.start);
                        o.append(num, (size_t)m);
                    }
                    const char* sq = s.fa.data() + s.line_start[2 * w + 1];
                    const char* se = s.fa.data() + s.line_start[2 * w + 2];
                    for (const char* p = sq; p < se && *p != '\n' && *p != '\r'; p++) {
                        char ch = *p;
                        if (ch >= 'a' && ch <= 'z') ch -= 32;
                        if (ch == 'T') ch = 'U';
                        o.push_back(ch);
                    }
                    const int m = std::snprintf(num, sizeof(num), "\n (%6.2f)\n", hm[w] / 100.);
                    o.append(num, (size_t)m);
                }
            });
        for (auto& t : th) t.join();
        FILE* f = std::fopen(out.c_str(), "wb");
        if (!f) { *err = "mirp_write_fold_text: cannot open " + out; return -8; }
        bool ok = true;
        for (auto& p : parts)
            if (!p.empty() && std::fwrite(p.data(), 1, p.size(), f) != p.size()) ok = false;
        if (std::fclose(f) != 0) ok = false;
        if (!ok) { *err = "mirp_write_fold_text: I/O error on " + out; return -8; }
        return 0;
    };
    if (!async) {
        std::string err;
        const int rc = work(&err);
        return rc ? fail(c, rc, err) : 0;
    }
    auto job = std::make_shared<mirp_ctx::TextJob>();
    job->th = std::thread([job, work] { job->rc = work(&job->err); });
    c->text_jobs.push_back(job);
    return 0;
}

extern "C" int mirp_write_fold_text(mirp_ctx* c, const char* fasta_path, const char* out_path) { return write_fold_text_impl(c, fasta_path, out_path, false); }
extern "C" int mirp_write_fold_text_async(mirp_ctx* c, const char* fasta_path, const char* out_path) { return write_fold_text_impl(c, fasta_path, out_path, true); }

extern "C" int mirp_wait_text(mirp_ctx* c) {
    if (!c) return -1;
    int rc = 0;
    for (auto& j : c->text_jobs) {
        if (j->th.joinable()) j->th.join();
        if (j->rc && !rc) { rc = j->rc; c->err = j->err; }
    }
    c->text_jobs.clear();
    return rc;
}

// Report side: the bodies of the per-locus read-mapping files of gen_map_result (MP:2907-2959) for a list of loci, from the
// position-sorted alignment records and the genome instead of one `samtools view` + `samtools faidx` per locus.  Host-only (no device):
// loci[n][8] = {tid, fold_s, fold_e, mat_s, mat_e, star_s, star_e, strand}; ss = the n structure strings back to back, NUL-terminated;
// contig_seq[t] / contig_len[t] = the bases of contig t (NULL for contigs this process does not hold); counts[n][n_samples] = reads on the
// precursor per sample (the `total_mapped_reads=` figure).  Out: text = the bodies back to back (everything after the `>name chr:s-e strand` header
// line, '\n'-terminated lines), offsets[n+1].  The read text is the reference sequence under the alignment, upper case (exact for the
// perfect-match alignments the reference's aligner script produces).
extern "C" int mirp_report_readmapping(const int32_t* loci, int64_t n_loci, const char* ss, const MirpAln* alns, int64_t n_alns, const uint8_t* const* contig_seq,
                                       const int64_t* contig_len, int32_t n_contigs, const char* sample_names, int32_t n_samples, const int64_t* counts,
                                       char** text, int64_t** offsets) {
    if (!loci || n_loci < 0 || !ss || (n_alns > 0 && !alns) || !contig_seq || !contig_len || !sample_names || n_samples < 1 || !counts || !text || !offsets) return -1;
    const std::vector<std::string> samples = split_names(sample_names, n_samples);
    std::vector<const char*> ssp((size_t)n_loci);
    { const char* p = ss; for (int64_t k = 0; k < n_loci; k++) { ssp[(size_t)k] = p; p += std::strlen(p) + 1; } }
    const int nt = n_workers((size_t)n_loci);
    std::vector<std::string> parts((size_t)nt);
    std::vector<std::vector<int64_t>> sizes((size_t)nt);
    std::vector<int> bad((size_t)nt, 0);
    std::vector<std::thread> th;
    auto up = [](uint8_t ch) -> char { char c = (char)ch; if (c >= 'a' && c <= 'z') c -= 32; return c; };
    auto comp = [](char c) -> char { switch (c) { case 'A': return 'U'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; case 'U': return 'A'; default: return c; } };
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            const size_t a = (size_t)n_loci * t / nt, b = (size_t)n_loci * (t + 1) / nt;
            std::string& o = parts[t];
            o.reserve((b - a) * 4096 + 64);
            std::string pre, line;
            std::vector<int64_t> sel;
            for (size_t k = a; k < b; k++) {
                const size_t o0 = o.size();
                const int32_t* m = loci + 8 * k;
                const int tid = m[0], fs = m[1], fe = m[2], ms = m[3], me = m[4], ss0 = m[5], se = m[6];
                const bool minus = m[7] != 0;
                if (tid < 0 || tid >= n_contigs || !contig_seq[tid] || fs < 1 || fe - 1 > contig_len[tid] || fe <= fs) { bad[t] = 1; sizes[t].push_back(0); continue; }
                const uint8_t* g = contig_seq[tid];
                // precursor chr:fs-(fe-1), upper case, T -> U; reverse complement on the minus strand (get_reverse_complement, MP:232-239)
                pre.clear();
                for (int x = fs; x < fe; x++) { char c = up(g[x - 1]); if (c == 'T') c = 'U'; pre.push_back(c); }
                if (minus) { std::string r(pre.rbegin(), pre.rend()); for (char& c : r) c = comp(c); pre.swap(r); }
                // reads that start in [fs, fe) and end inside the precursor, on the locus' strand
                const MirpAln* lo = std::lower_bound(alns, alns + n_alns, std::make_pair(tid, fs), [](const MirpAln& r, const std::pair<int, int>& key) {
                    return r.tid < key.first || (r.tid == key.first && r.pos < key.second); });
                sel.clear();
                for (const MirpAln* r = lo; r < alns + n_alns && r->tid == tid && r->pos < fe; r++)
                    if ((long long)r->pos + r->len <= fe && (r->strand != 0) == minus) sel.push_back(r - alns);
                const int mlen = me - ms, slen = se - ss0;
                for (int s = 0; s < n_samples; s++) {
                    o += ">> Read mappings for sample: "; o += samples[(size_t)s]; o += "\n5'->3'\n";
                    o += pre; o += "\ttotal_mapped_reads="; put_int(o, counts[k * (size_t)n_samples + s]); o.push_back('\n');
                    o += ssp[k]; o.push_back('\n');
                    // start positions ascending (descending on the minus strand); within one start position by read length, stable
                    std::vector<int64_t> rs;
                    for (int64_t x : sel) if (alns[x].sample == s) rs.push_back(x);
                    std::stable_sort(rs.begin(), rs.end(), [&](int64_t x, int64_t y) {
                        if (alns[x].pos != alns[y].pos) return minus ? alns[x].pos > alns[y].pos : alns[x].pos < alns[y].pos;
                        return alns[x].len < alns[y].len; });
                    for (int64_t x : rs) {
                        const MirpAln& r = alns[x];
                        const int rl = r.len, sp = r.pos;
                        const char pad = (sp == ms && rl == mlen) ? 'm' : (sp == ss0 && rl == slen) ? 's' : '.';
                        line.assign((size_t)(sp - fs), pad);
                        for (int y = 0; y < rl; y++) line.push_back(sp - 1 + y < contig_len[tid] ? up(g[sp - 1 + y]) : pad);
                        if (line.size() < pre.size()) line.append(pre.size() - line.size(), pad);
                        if (minus) {      // get_reverse_complement, then U -> T (MP:2946-2947): pads are not in the table and stay
                            std::string r2(line.rbegin(), line.rend());
                            for (char& c : r2) { c = comp(c); if (c == 'U') c = 'T'; }
                            line.swap(r2);
                        }
                        o += line; o += "\tdepth="; put_int(o, r.depth); o += ", length="; put_int(o, rl);
                        if (pad == 'm') o += " [mature]";
                        if (pad == 's') o += " [star]";
                        o.push_back('\n');
                    }
                }
                sizes[t].push_back((int64_t)(o.size() - o0));
            }
        });
    for (auto& x : th) x.join();
    for (int b2 : bad) if (b2) return -2;
    size_t total = 0;
    for (auto& p : parts) total += p.size();
    char* out = (char*)std::malloc(std::max<size_t>(total, 1));
    int64_t* off = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)(n_loci + 1));
    if (!out || !off) { std::free(out); std::free(off); return -6; }
    size_t w = 0, k = 0;
    off[0] = 0;
    for (int t = 0; t < nt; t++) {
        if (!parts[t].empty()) std::memcpy(out + w, parts[t].data(), parts[t].size());
        w += parts[t].size();
        for (int64_t sz : sizes[t]) { off[k + 1] = off[k] + sz; k++; }
    }
    *text = out; *offsets = off;
    return 0;
}
