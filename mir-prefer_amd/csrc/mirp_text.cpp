// Stage artefacts in the reference's own text formats, written by native worker threads straight from the device-resident state: the
// thresholded depth file `bam.depth.cut<CUT>` (MP:946-949), the folder's input FASTA `<prefix>.rnalfold.in_<i>.fa` (MP:1124-1142) and the
// fold stage's RNALfold-format output (MP:3085-3098, consumed by MP:1541-1599).  These files are what the reference's stages exchange; on
// the host they were the bulk of the end-to-end wall-clock (Python string formatting of 10^5..10^6 lines), here a stage formats its chunk of
// lines per thread into memory and the file is written in one go.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
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

// The pieces, in order, into one file; -8 with a message on any failure.  One writer: buffered writes to ONE file serialise on the inode in the
// kernel -- a thread per piece (pwrite at its offset) made the 135 MB fold text slower, not faster (tmpfs and overlay, round 4).
int write_parts(const char* path, const std::vector<std::string>& parts, std::string* err) {
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { *err = std::string("cannot open ") + path; return -8; }
    std::vector<size_t> off(parts.size() + 1, 0);
    for (size_t k = 0; k < parts.size(); k++) off[k + 1] = off[k] + parts[k].size();
    std::vector<char> bad(parts.size(), 0);
    auto put = [&](size_t k) {
        const char* p = parts[k].data();
        size_t left = parts[k].size(), at = off[k];
        while (left) {
            const ssize_t w = ::pwrite(fd, p, left, (off_t)at);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) { bad[k] = 1; return; }
            p += w; at += (size_t)w; left -= (size_t)w;
        }
    };
    for (size_t k = 0; k < parts.size(); k++) put(k);
    bool ok = true;
    for (char b : bad) if (b) ok = false;
    if (::close(fd) != 0) ok = false;
    if (!ok) { *err = std::string("I/O error on ") + path; return -8; }
    return 0;
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
    std::string err;
    const int rc = write_parts(path, parts, &err);
    return rc ? fail(c, rc, std::string(who) + ": " + err) : 0;
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
        std::string werr;
        if (const int rc = write_parts(out.c_str(), parts, &werr)) { *err = "mirp_write_fold_text: " + werr; return rc; }
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


// ------------------------------------------------------------------------------------------------------------------------------------------------
// Report files of the predict stage: the reference formats them locus by locus in Python (MP:2619-2641, 2644-2779, 2793-2904,
// 2963-3019, 3585-3593); here four threads format the seven files from flat arrays.  The texts are the reference's, byte for byte.
// ------------------------------------------------------------------------------------------------------------------------------------------------
namespace {
// Python's s[a:b] on a string of length n (negative indices count from the end, everything clamps)
inline void py_slice(const char* s, long long n, long long a, long long b, const char** out, size_t* len) {
    if (a < 0) a += n;
    if (b < 0) b += n;
    a = std::max<long long>(0, std::min(a, n));
    b = std::max<long long>(0, std::min(b, n));
    *out = s + a; *len = b > a ? (size_t)(b - a) : 0;
}
inline void append_revcomp(std::string& o, const char* s, size_t n) {      // get_reverse_complement (MP:232-239): upper-case ATGCU only
    for (size_t k = n; k-- > 0;) {
        char c = s[k];
        switch (c) { case 'A': c = 'U'; break; case 'T': c = 'A'; break; case 'G': c = 'C'; break; case 'C': c = 'G'; break; case 'U': c = 'A'; break; default: break; }
        o.push_back(c);
    }
}
int write_whole(const char* path, const std::string& text, std::string* err) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { *err = std::string("cannot open ") + path; return -8; }
    bool ok = text.empty() || std::fwrite(text.data(), 1, text.size(), f) == text.size();
    if (std::fclose(f) != 0) ok = false;
    if (!ok) { *err = std::string("I/O error on ") + path; return -8; }
    return 0;
}
}  // namespace

extern "C" int mirp_write_reports(int64_t n, const int32_t* loci, const char* contig_names, int32_t n_contigs, const char* ss_blob, const char* pre_blob,
                                  const char* sample_names, int32_t n_samples, const int64_t* counts, const char* mirbase_form, const char* gff_path,
                                  const char* mature_fa_path, const char* precursor_fa_path, const char* ss_path, const char* csv_path, const char* html_path,
                                  const char* stat_path, char* errbuf, size_t errbuf_len) {
    auto bail = [&](int code, const std::string& m) { if (errbuf && errbuf_len) { std::snprintf(errbuf, errbuf_len, "%s", m.c_str()); } return code; };
    if (n < 0 || (n > 0 && (!loci || !contig_names || !ss_blob || !pre_blob || !counts)) || n_samples < 0 || (n_samples > 0 && !sample_names) ||
        (html_path && !mirbase_form))
        return bail(-1, "mirp_write_reports: bad argument");
    const std::vector<std::string> names = split_names(contig_names ? contig_names : "", n > 0 ? n_contigs : 0);
    const std::vector<std::string> samples = split_names(sample_names ? sample_names : "", n_samples);
    std::vector<const char*> ssp((size_t)n), prep((size_t)n);
    std::vector<size_t> ssl((size_t)n), prel((size_t)n);
    {
        const char* p = ss_blob; const char* q = pre_blob;
        for (int64_t k = 0; k < n; k++) {
            ssp[k] = p; ssl[k] = std::strlen(p); p += ssl[k] + 1;
            prep[k] = q; prel[k] = std::strlen(q); q += prel[k] + 1;
            if (loci[10 * k] < 0 || loci[10 * k] >= n_contigs) return bail(-1, "mirp_write_reports: contig index out of range");
        }
    }
    std::string form[4];
    if (mirbase_form) { const char* p = mirbase_form; for (int k = 0; k < 4; k++) { form[k] = p; p += form[k].size() + 1; } }
    // per locus: precursor / mature / star as the detail tables print them, length and first base of the forward-strand mature (locus_texts)
    struct Txt { std::string pre, mat, star; long long flen; char ffirst; };
    std::vector<Txt> T((size_t)n);
    auto texts = [&](size_t a, size_t b) {
        for (size_t k = a; k < b; k++) {
            const int32_t* m = loci + 10 * k;
            const char* s; size_t l;
            Txt& t = T[k];
            const char *ms, *ss2; size_t ml, sl;
            py_slice(prep[k], (long long)prel[k], m[3] - m[1], m[4] - 1 - m[1] + 1, &ms, &ml);
            py_slice(prep[k], (long long)prel[k], m[5] - m[1], m[6] - 1 - m[1] + 1, &ss2, &sl);
            py_slice(prep[k], (long long)prel[k], 0, m[2] - 1 - m[1] + 1, &s, &l);
            t.flen = (long long)ml; t.ffirst = ml ? ms[0] : '?';
            if (m[7]) { append_revcomp(t.pre, s, l); append_revcomp(t.mat, ms, ml); append_revcomp(t.star, ss2, sl); }
            else { t.pre.assign(s, l); t.mat.assign(ms, ml); t.star.assign(ss2, sl); }
        }
    };
    {
        const int nt = n_workers((size_t)n);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] { texts((size_t)n * t / nt, (size_t)n * (t + 1) / nt); });
        for (auto& t : th) t.join();
    }
    // the two distribution tables of the stat file and the html page: length (numeric order) and first base (character order) of the forward-strand matures
    std::vector<std::pair<long long, long long>> dlen;
    std::vector<std::pair<char, long long>> dfirst;
    for (int64_t k = 0; k < n; k++) {
        auto a = std::find_if(dlen.begin(), dlen.end(), [&](const std::pair<long long, long long>& x) { return x.first == T[k].flen; });
        if (a == dlen.end()) dlen.push_back({T[k].flen, 1}); else a->second++;
        auto b = std::find_if(dfirst.begin(), dfirst.end(), [&](const std::pair<char, long long>& x) { return x.first == T[k].ffirst; });
        if (b == dfirst.end()) dfirst.push_back({T[k].ffirst, 1}); else b->second++;
    }
    std::sort(dlen.begin(), dlen.end());
    std::sort(dfirst.begin(), dfirst.end());
    auto name_of = [&](std::string& o, const char* what, int64_t k) { o += what; put_int(o, k); };
    std::string errs[4];
    int rcs[4] = {0, 0, 0, 0};
    std::vector<std::thread> th;
    // ---- gff3
    th.emplace_back([&] {
        if (!gff_path) return;
        std::string o;
        o.reserve((size_t)n * 260 + 16);
        for (int64_t k = 0; k < n; k++) {
            const int32_t* m = loci + 10 * k;
            const std::string& chr = names[(size_t)m[0]];
            const char strand = m[7] ? '-' : '+';
            o += chr; o += "\tmiR-PREFeR\tmiRNA-precursor\t"; put_int(o, m[1]); o.push_back('\t'); put_int(o, m[2] - 1); o += "\t.\t"; o.push_back(strand); o += "\t.\tID=";
            name_of(o, "miRNA-precursor_", k); o += ";NAME="; name_of(o, "miRNA-precursor_", k);
            o += ";Other=mature_expressed=y;star_expressed="; o.push_back(m[8] ? 'y' : 'n'); o += ";overhangsize="; o += m[9] == 2 ? "3:3" : m[9] == 1 ? "2:3" : "2:2"; o.push_back('\n');
            o += chr; o += "\tmiR-PREFeR\tmiRNA\t"; put_int(o, m[3]); o.push_back('\t'); put_int(o, m[4] - 1); o += "\t.\t"; o.push_back(strand); o += "\t.\tID=";
            name_of(o, "miRNA_", k); o += ";NAME="; name_of(o, "miRNA_", k); o += ";Other=\n";
        }
        rcs[0] = write_whole(gff_path, o, &errs[0]);
    });
    // ---- mature / precursor FASTA and the structure file with its M / S annotation line
    th.emplace_back([&] {
        if (!mature_fa_path && !precursor_fa_path && !ss_path) return;
        std::string fm, fp, fs;
        fm.reserve((size_t)n * 80); fp.reserve((size_t)n * 200); fs.reserve((size_t)n * 520);
        std::string dot, id;
        for (int64_t k = 0; k < n; k++) {
            const int32_t* m = loci + 10 * k;
            const std::string& chr = names[(size_t)m[0]];
            const char strand = m[7] ? '-' : '+';
            const Txt& t = T[k];
            id.clear(); id.push_back('>'); id += chr; id.push_back(':'); put_int(id, m[3]); id.push_back('-'); put_int(id, m[4] - 1); id.push_back(' '); id.push_back(strand);
            id.push_back(' '); name_of(id, "miRNA-precursor_", k);
            fm += id; fm.push_back('\n'); fm += t.mat; fm.push_back('\n');
            id.clear(); id.push_back('>'); id += chr; id.push_back(':'); put_int(id, m[1]); id.push_back('-'); put_int(id, m[2] - 1); id.push_back(' '); id.push_back(strand);
            id.push_back(' '); name_of(id, "miRNA-precursor_", k);
            fp += id; fp.push_back('\n'); fp += t.pre; fp.push_back('\n');
            // "." / "M" / "S" over the forward-strand precursor, reversed on the minus strand (MP:2994-3006); Python's str * negative = ""
            const long long ms = m[3] - m[1], me = m[4] - m[1], s0 = m[5] - m[1], s1 = m[6] - m[1], len = (long long)t.pre.size();
            dot.clear();
            auto rep = [&](char c, long long cnt) { if (cnt > 0) dot.append((size_t)cnt, c); };
            if (ms < s0) { rep('.', ms); rep('M', me - ms); rep('.', s0 - me); rep('S', s1 - s0); rep('.', len - s1); }
            else { rep('.', s0); rep('S', s1 - s0); rep('.', ms - s1); rep('M', me - ms); rep('.', len - me); }
            if (m[7]) std::reverse(dot.begin(), dot.end());
            fs += id; fs.push_back('\n'); fs += t.pre; fs.push_back('\n'); fs.append(ssp[k], ssl[k]); fs.push_back('\n'); fs += dot; fs.push_back('\n');
        }
        if (mature_fa_path) rcs[1] = write_whole(mature_fa_path, fm, &errs[1]);
        if (!rcs[1] && precursor_fa_path) rcs[1] = write_whole(precursor_fa_path, fp, &errs[1]);
        if (!rcs[1] && ss_path) rcs[1] = write_whole(ss_path, fs, &errs[1]);
    });
    // ---- detail csv and miRNA.stat.txt
    th.emplace_back([&] {
        if (csv_path) {
            std::string o;
            o.reserve((size_t)n * 700 + 1024);
            const char* head = "miRNAID, Seqid(chromosome), start position, end position, strand, precursor sequence, secondary structure, mature sequence, star sequence, ";
            o += head;
            for (const std::string& s : samples) { o += s; o.push_back(','); o += s; o.push_back(','); o += s; o.push_back(','); o += s; o.push_back(','); }
            o.push_back('\n');
            o += head;
            for (int s = 0; s < n_samples; s++) o += "reads mapped to precursor, reads mapped to mature, reads mapped to star, reads mapped to antisense region,";
            o.push_back('\n');
            for (int64_t k = 0; k < n; k++) {
                const int32_t* m = loci + 10 * k;
                const Txt& t = T[k];
                if (k) o.push_back('\n');
                name_of(o, "miRNA-precursor_", k); o += ", "; o += names[(size_t)m[0]]; o += ", "; put_int(o, m[1]); o += ", "; put_int(o, m[2]); o += ", "; o.push_back(m[7] ? '-' : '+');
                o += ", "; o += t.pre; o += ", "; o.append(ssp[k], ssl[k]); o += ", "; o += t.mat; o += ", "; o += t.star;
                for (int s = 0; s < n_samples; s++) for (int c = 0; c < 4; c++) { o += ", "; put_int(o, counts[((size_t)k * n_samples + s) * 4 + c]); }
            }
            if (n > 0) o.push_back('\n');
            rcs[2] = write_whole(csv_path, o, &errs[2]);
        }
        if (!rcs[2] && stat_path) {
            std::string o = "Total number of predicted miRNAs: ";
            put_int(o, n); o += "\nDistribution of the length of the mature miRNAs:\n";
            for (auto& x : dlen) { put_int(o, x.first); o += ": "; put_int(o, x.second); o.push_back('\n'); }
            o += "Distribution of the nucleotide of the first base of the mature miRNAs:\n";
            for (auto& x : dfirst) { o.push_back(x.first); o += ": "; put_int(o, x.second); o.push_back('\n'); }
            rcs[2] = write_whole(stat_path, o, &errs[2]);
        }
    });
    // ---- detail html
    th.emplace_back([&] {
        if (!html_path) return;
        std::string o;
        o.reserve((size_t)n * 3300 + 4096);
        o += "<h1 > microRNAs predicted by miR-PREFeR </h1>\n<div>\n<h2 > Total number of prediction:"; put_int(o, n); o += "  </h2>\n";
        auto table_head = [&](const char* title, const char* head) {
            o += "<h3>"; o += title; o += "</h3>\n<table border=\"1\">\n\t<thead>\n\t\t<tr>\n\t\t<th>"; o += head; o += "</th>\n\t\t<th>Count</th>\n\t\t</tr>\n\t</thead>\n\t<tbody>\n";
        };
        table_head("Distribution of the lengths of the mature sequences", "Length");
        for (auto& x : dlen) { o += "\t\t<tr>\n\t\t\t<td>"; put_int(o, x.first); o += " </td>\n\t\t\t<td>"; put_int(o, x.second); o += " </td>\n\t\t</tr>\n"; }
        o += "\t</tbody>\n</table>\n</div>\n";
        table_head("Distribution of the nucleotide of the first base of the mature sequences", "Nucleotide");
        for (auto& x : dfirst) { o += "\t\t<tr>\n\t\t\t<td>"; o.push_back(x.first); o += " </td>\n\t\t\t<td>"; put_int(o, x.second); o += " </td>\n\t\t</tr>\n"; }
        o += "\t</tbody>\n</table>\n</div>";
        o += "<div><h3>Detailed infomation </h3>\n<table border=\"1\">\n<colgroup>\n\t<col span=8 style=\"background-color:#CECEF6\">\n";
        static const char* colors[3] = {"#A9E2F3", "#ACFA58", "#F5A9BC"};
        for (int s = 0; s < n_samples; s++) { o += "\t<col span=\"4\" style=\"background-color:"; o += colors[s % 3]; o += "\">"; }
        o += "</colgroup>\n\t<thead>\n\t\t<tr>\n";
        static const char* head1[8] = {"miRNA precursor ID", "Chromosome", "start position", "end position", "strand", "precursor sequence and secondary structure",
                                       "mature sequence", "star sequence"};
        for (const char* h : head1) { o += "\t\t\t<th rowspan=\"2\">"; o += h; o += "</th>\n"; }
        for (const std::string& s : samples) { o += "\t\t\t<th colspan=\"4\">"; o += s; o += "</th>\n"; }
        o += "\t\t\t<th colspan=\"2\">read mappings</th>\n\t\t</tr>\n\t\t<tr>\n";
        static const char* per_sample[4] = {"precursor", "mature", "star", "antisense region"};
        for (int s = 0; s < n_samples; s++) for (const char* w : per_sample) { o += "\t\t\t<th> reads mapped to "; o += w; o += " </th>\n"; }
        o += "\t\t</tr>\n\t</thead>\n\t<tbody>\n";
        auto td = [&](const std::string& c) { o += "\t\t\t<td nowrap>"; o += c; o += "</td>\n"; };
        std::string tmp;
        for (int64_t k = 0; k < n; k++) {
            const int32_t* m = loci + 10 * k;
            const Txt& t = T[k];
            o += "\t\t<tr>\n";
            tmp.clear(); name_of(tmp, "miRNA-precursor_", k); td(tmp);
            td(names[(size_t)m[0]]);
            tmp.clear(); put_int(tmp, m[1]); td(tmp);
            tmp.clear(); put_int(tmp, m[2]); td(tmp);
            tmp.assign(1, m[7] ? '-' : '+'); td(tmp);
            o += "\t\t\t<td nowrap> <code>"; o += t.pre; o += "<BR>"; o.append(ssp[k], ssl[k]); o += " </code></td>";
            o += "\t\t\t<td nowrap>"; o += t.mat; o += form[0]; o += t.mat; o += form[1]; o += form[2]; o += t.mat; o += form[3]; o += "</td>\n";
            td(t.star);
            for (int s = 0; s < n_samples; s++) for (int c = 0; c < 4; c++) { tmp.clear(); put_int(tmp, counts[((size_t)k * n_samples + s) * 4 + c]); td(tmp); }
            o += "\t\t\t<td><a href=\"readmapping/"; name_of(o, "miRNA-precursor_", k); o += ".map.txt\" target=\"_blank\">Click to see detailed mapping.</a></td>\t\t</tr>\n";
        }
        o += "\t</tbody>\n</table>\n</div>\n";
        rcs[3] = write_whole(html_path, o, &errs[3]);
    });
    for (auto& t : th) t.join();
    for (int k = 0; k < 4; k++) if (rcs[k]) return bail(rcs[k], "mirp_write_reports: " + errs[k]);
    return 0;
}

extern "C" int mirp_write_files(int64_t n, const char* paths, const char* text, const int64_t* offs, int32_t n_threads, char* errbuf, size_t errbuf_len) {
    if (n < 0 || (n > 0 && (!paths || !text || !offs))) { if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "mirp_write_files: bad argument"); return -1; }
    std::vector<const char*> name((size_t)n);
    { const char* p = paths; for (int64_t k = 0; k < n; k++) { name[k] = p; p += std::strlen(p) + 1; } }
    // one writer unless told otherwise: creating files in ONE directory serialises on the directory in the kernel -- 1 / 2 / 4 / 8 threads measured
    // the same on tmpfs (13-16 ms for 4,002 files) and no better on an overlay file system (profiles/tools/smallfiles.py)
    const int cap = n_threads > 1 ? (n_threads < 64 ? n_threads : 64) : 1;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(cap, n / 64));
    std::vector<int64_t> bad((size_t)nt, -1);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            for (int64_t k = n * t / nt; k < n * (t + 1) / nt; k++) {
                FILE* f = std::fopen(name[k], "wb");
                const size_t len = (size_t)(offs[k + 1] - offs[k]);
                bool ok = f && (len == 0 || std::fwrite(text + offs[k], 1, len, f) == len);
                if (f && std::fclose(f) != 0) ok = false;
                if (!ok) { bad[t] = k; return; }
            }
        });
    for (auto& t : th) t.join();
    for (int t = 0; t < nt; t++)
        if (bad[t] >= 0) { if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "mirp_write_files: cannot write %s", name[bad[t]]); return -8; }
    return 0;
}
