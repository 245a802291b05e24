// The tail of the predict stage in native code (host side): from the result list of mirp_predict / mirp_gather_loci to every report file the
// reference writes after its queue is drained -- the mature/star swap of gen_miRNA_loci_nopredict's caller (MP:2611-2617), resultlist.sort() of
// gen_gff_from_result (MP:2622), the per-sample read counts of gen_mirna_info (MP:2644-2728), the read-mapping file of every locus (gen_map_result,
// MP:2907-2959) and the seven report files (MP:2619-2641, 2744-2779, 2793-2904, 2963-3019, 3585-3593).  The Python host used to do the list handling
// between these (a Python object per locus and field; 0.05 s at 4,002 loci, 0.25 s at 16,016); here the flat record array goes in and files come
// out.  Formatting is shared with the single-purpose entry points (mirp_report_readmapping, mirp_write_reports, mirp_write_files), so the bytes are
// the same by construction; tests/test_host_cpu.py holds this against the reference's files as well.
//
// Two entry points: mirp_write_result_reports (the whole list at once) and mirp_fold_predict_report_stream, which runs the fold and the filter over
// the window list in a few chunks cut where the list order of the results is decided by the windows alone, and writes a chunk's read-mapping files on a
// host thread WHILE THE DEVICE FOLDS THE NEXT CHUNK: one file per locus is the reference's output format, creating a file costs 10 us on tmpfs and
// 10 - 100 us on an overlay file system (profiles/tools/fs_regime.py), i.e. 0.04 - 0.4 s at config[1] and 0.16 - 1.5 s at config[2] -- the largest
// host item of the end-to-end wall-clock once everything else is native.
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include "mirp_ctx.h"

namespace {
struct ReportInputs {
    const char* contig_names; int32_t n_contigs; const uint8_t* const* contig_seq; const int64_t* contig_len; const MirpAln* alns; int64_t n_alns;
    const char* sample_names; int32_t n_samples; const char* mirbase_form; const char* outdir; const char* prefix;
};

// Accumulates the report inputs chunk by chunk; a chunk's read-mapping files are written when the chunk is added (their names need only the number of
// loci before the chunk), the seven report files at the end (the html and stat files open with totals).
struct ReportAccum {
    ReportInputs in;
    std::vector<const char*> cname;
    std::vector<int32_t> loci10;
    std::string ss_blob, pre_blob;
    std::vector<int64_t> counts;          // [n][n_samples][4]
    std::vector<MirpMirna> sorted;        // records in list order, after the swap
    std::vector<int32_t> order;           // input index (within its chunk) of list position i
    int64_t n = 0;
    bool dirs = false;
    std::string err;

    explicit ReportAccum(const ReportInputs& i) : in(i) {
        cname.resize((size_t)in.n_contigs);
        const char* p = in.contig_names;
        for (int t = 0; t < in.n_contigs; t++) { cname[(size_t)t] = p; p += std::strlen(p) + 1; }
    }
    int fail(int code, const std::string& m) { err = m; return code; }

    // list order of resultlist.sort() (MP:2622): Python compares [chr, fold_s, fold_e, mat_s, mat_e, star_s, star_e, ss, strand, has_star] element by
    // element; strings by code point = bytes for ASCII
    bool less(const MirpMirna& x, const char* sx, const MirpMirna& y, const char* sy) const {
        if (x.tid != y.tid) { const int c = std::strcmp(cname[(size_t)x.tid], cname[(size_t)y.tid]); if (c) return c < 0; }
        if (x.fold_s != y.fold_s) return x.fold_s < y.fold_s;
        if (x.fold_e != y.fold_e) return x.fold_e < y.fold_e;
        if (x.mat_s != y.mat_s) return x.mat_s < y.mat_s;
        if (x.mat_e != y.mat_e) return x.mat_e < y.mat_e;
        if (x.star_s != y.star_s) return x.star_s < y.star_s;
        if (x.star_e != y.star_e) return x.star_e < y.star_e;
        const int la = x.ss_len, lb = y.ss_len;
        const int c = std::memcmp(sx, sy, (size_t)std::min(la, lb));
        if (c) return c < 0;
        if (la != lb) return la < lb;
        if (x.strand != y.strand) return x.strand < y.strand;          // '+' (0) sorts before '-' (1), as the characters do
        return (x.has_star != 0) < (y.has_star != 0);
    }

    int add_chunk(const MirpMirna* result, int64_t m, const char* ss_text, int32_t ss_stride) {
        if (m == 0) return 0;
        const int ns = in.n_samples;
        // ---- the more abundant arm becomes the mature (MP:2611-2617), then the list order within the chunk
        std::vector<MirpMirna> rec(result, result + m);
        for (int64_t k = 0; k < m; k++) {
            MirpMirna& r = rec[(size_t)k];
            if (r.tid < 0 || r.tid >= in.n_contigs) return fail(-1, "mirp_write_result_reports: record with a contig index outside the contig table");
            if (r.total_depth_mature < r.total_depth_star) {
                std::swap(r.total_depth_mature, r.total_depth_star);
                std::swap(r.mat_s, r.star_s);
                std::swap(r.mat_e, r.star_e);
            }
        }
        auto ss_of = [&](int32_t k) { return ss_text + (size_t)k * (size_t)ss_stride; };
        std::vector<int32_t> ord((size_t)m);
        for (int64_t k = 0; k < m; k++) ord[(size_t)k] = (int32_t)k;
        std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return less(rec[(size_t)a], ss_of(a), rec[(size_t)b], ss_of(b)); });
        // a later chunk must sort behind everything before it (the chunk boundaries are chosen so that it does; checked, never assumed)
        if (n > 0) {
            const MirpMirna& last = sorted.back();
            const char* last_ss = ss_blob.c_str() + last_ss_off;
            if (less(rec[(size_t)ord[0]], ss_of(ord[0]), last, last_ss))
                return fail(-9, "mirp_fold_predict_report_stream: a chunk's first locus sorts before the previous chunk's last one (window order does not decide the list order here)");
        }
        // ---- flat inputs of the formatters, in list order
        const int64_t base = n;
        std::vector<int32_t> loci8((size_t)m * 8);
        std::vector<int64_t> counts0((size_t)m * ns, 0);
        loci10.resize((size_t)(base + m) * 10);
        counts.resize((size_t)(base + m) * ns * 4, 0);
        const size_t ss_base = ss_blob.size();
        for (int64_t i = 0; i < m; i++) {
            const MirpMirna& r = rec[(size_t)ord[(size_t)i]];
            int32_t* a = &loci8[(size_t)i * 8];
            a[0] = r.tid; a[1] = r.fold_s; a[2] = r.fold_e; a[3] = r.mat_s; a[4] = r.mat_e; a[5] = r.star_s; a[6] = r.star_e; a[7] = r.strand ? 1 : 0;
            int32_t* b = &loci10[(size_t)(base + i) * 10];
            std::memcpy(b, a, 8 * sizeof(int32_t));
            b[8] = r.total_depth_star == 0 ? 0 : 1;
            const int f = r.reserved;
            b[9] = ((f & 1) && (f & 8)) ? ((((f >> 1) & 3) - 1) == 2 ? 2 : 1) : 0;          // overhangsize 2:2 / 2:3 / 3:3 (MP:2631-2637)
            if (i == m - 1) last_ss_off = ss_blob.size();
            ss_blob.append(ss_of(ord[(size_t)i]), (size_t)r.ss_len); ss_blob.push_back('\0');
            // `samtools faidx chr:fold_s-(fold_e-1)`, upper case, T -> U (MP:2570-2590)
            const uint8_t* g = in.contig_seq[(size_t)r.tid];
            const int64_t gl = in.contig_len[(size_t)r.tid];
            if (!g) return fail(-1, std::string("mirp_write_result_reports: a locus lies on contig ") + cname[(size_t)r.tid] + ", which this process does not hold");
            const int64_t s0 = std::max<int64_t>(r.fold_s - 1, 0), s1 = std::min<int64_t>((int64_t)r.fold_e - 1, gl);
            for (int64_t p = s0; p < s1; p++) { char c = (char)g[p]; if (c >= 'a' && c <= 'z') c -= 32; if (c == 'T') c = 'U'; pre_blob.push_back(c); }
            pre_blob.push_back('\0');
            sorted.push_back(r);
            order.push_back(ord[(size_t)i]);
        }
        // ---- reads per locus and sample: on the precursor / exactly the mature / exactly the star / antisense (gen_mirna_info, MP:2644-2728), from the
        // (tid, pos)-sorted records instead of one `samtools view` per locus
        {
            const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, m / 256));
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++)
                th.emplace_back([&, t] {
                    for (int64_t i = m * t / nt; i < m * (t + 1) / nt; i++) {
                        const int32_t* a = &loci8[(size_t)i * 8];
                        auto lower = [&](int64_t tid, int64_t pos) {
                            int64_t lo = 0, hi = in.n_alns;
                            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; const MirpAln& r = in.alns[mid]; if (r.tid < tid || (r.tid == tid && r.pos < pos)) lo = mid + 1; else hi = mid; }
                            return lo;
                        };
                        const int64_t lo = lower(a[0], a[1]), hi = lower(a[0], a[2]);
                        int64_t* cl = &counts[(size_t)(base + i) * ns * 4];
                        for (int64_t r = lo; r < hi; r++) {
                            const MirpAln& x = in.alns[r];
                            if ((int64_t)x.pos + x.len > a[2]) continue;
                            if (x.sample >= ns) continue;
                            int64_t* c = cl + (size_t)x.sample * 4;
                            const bool sense = (x.strand ? 1 : 0) == a[7];
                            if (sense) {
                                c[0] += x.depth;
                                if (x.pos == a[3] && x.len == a[4] - a[3]) c[1] += x.depth;
                                if (x.pos == a[5] && x.len == a[6] - a[5]) c[2] += x.depth;
                            } else {
                                c[3] += x.depth;
                            }
                        }
                        for (int s = 0; s < ns; s++) counts0[(size_t)i * ns + s] = cl[(size_t)s * 4];
                    }
                });
            for (auto& t : th) t.join();
        }
        // ---- read-mapping bodies and files of this chunk
        char* body = nullptr; int64_t* boffs = nullptr;
        int rc = mirp_report_readmapping(loci8.data(), m, ss_blob.c_str() + ss_base, in.alns, in.n_alns, in.contig_seq, in.contig_len, in.n_contigs, in.sample_names, ns,
                                         counts0.data(), &body, &boffs);
        if (rc) return fail(rc, "mirp_write_result_reports: read-mapping bodies failed (a locus lies on a contig this process does not hold)");
        const std::string out(in.outdir), rmdir = out + "/readmapping";
        if (!dirs) { ::mkdir(out.c_str(), 0777); ::mkdir(rmdir.c_str(), 0777); dirs = true; }
        std::string paths, text;
        std::vector<int64_t> toffs((size_t)m + 1, 0);
        text.reserve((size_t)boffs[m] + (size_t)m * 64);
        for (int64_t i = 0; i < m; i++) {
            const int32_t* a = &loci8[(size_t)i * 8];
            char name[64];
            std::snprintf(name, sizeof name, "miRNA-precursor_%lld", (long long)(base + i));
            paths += rmdir; paths += "/"; paths += name; paths += ".map.txt"; paths.push_back('\0');
            char head[96];
            std::snprintf(head, sizeof head, ":%d-%d %c\n", a[1], a[2], a[7] ? '-' : '+');
            text += ">"; text += name; text += " "; text += cname[(size_t)a[0]]; text += head;
            text.append(body + boffs[i], (size_t)(boffs[i + 1] - boffs[i]));
            toffs[(size_t)i + 1] = (int64_t)text.size();
        }
        mirp_free(body); mirp_free(boffs);
        // the files go out on a thread of their own, one chunk after the other (creating files in one directory serialises in the kernel anyway), while
        // this thread formats the next chunk -- or, at the end, the seven report files
        if (int frc = join_files()) return frc;
        files_job = std::make_shared<FilesJob>();
        files_job->paths.swap(paths); files_job->text.swap(text); files_job->offs.swap(toffs); files_job->m = m;
        std::shared_ptr<FilesJob> job = files_job;
        files_thread = std::thread([job] {
            job->rc = mirp_write_files(job->m, job->paths.c_str(), job->text.data(), job->offs.data(), 1, job->err, sizeof job->err);
        });
        n = base + m;
        return 0;
    }

    struct FilesJob { std::string paths, text; std::vector<int64_t> offs; int64_t m = 0; int rc = 0; char err[512] = ""; };
    std::shared_ptr<FilesJob> files_job;
    std::thread files_thread;
    int join_files() {
        if (files_thread.joinable()) {
            files_thread.join();
            if (files_job && files_job->rc) return fail(files_job->rc, files_job->err);
        }
        return 0;
    }
    ~ReportAccum() { if (files_thread.joinable()) files_thread.join(); }

    int finish() {
        if (n == 0) return join_files();          // "0 miRNA identified. No result files generated." (MP:3547-3550)
        const std::string out(in.outdir), pre(in.prefix);
        const std::string gff = out + "/" + pre + "_miRNA.gff3", mat = out + "/" + pre + "_miRNA.mature.fa", stem = out + "/" + pre + "_miRNA.precursor.fa",
                          ssf = out + "/" + pre + "_miRNA.precursor.ss", csv = out + "/" + pre + "_miRNA.detail.csv", html = out + "/" + pre + "_miRNA.detail.html",
                          stat = out + "/miRNA.stat.txt";
        char e2[512] = "";
        const int rc = mirp_write_reports(n, loci10.data(), in.contig_names, in.n_contigs, ss_blob.c_str(), pre_blob.c_str(), in.sample_names, in.n_samples, counts.data(),
                                          in.mirbase_form, gff.c_str(), mat.c_str(), stem.c_str(), ssf.c_str(), csv.c_str(), html.c_str(), stat.c_str(), e2, sizeof e2);
        const int frc = join_files();
        if (rc) return fail(rc, e2);
        return frc;
    }

    size_t last_ss_off = 0;
};

bool bad_inputs(const ReportInputs& i) {
    return !i.contig_names || i.n_contigs < 0 || !i.contig_seq || !i.contig_len || (i.n_alns > 0 && !i.alns) || i.n_samples < 1 || !i.sample_names || !i.mirbase_form ||
           !i.outdir || !i.prefix;
}
}  // namespace

extern "C" int mirp_write_result_reports(const MirpMirna* result, int64_t n, const char* ss_text, int32_t ss_stride, const char* contig_names, int32_t n_contigs,
                                         const uint8_t* const* contig_seq, const int64_t* contig_len, const MirpAln* alns, int64_t n_alns,
                                         const char* sample_names, int32_t n_samples, const char* mirbase_form, const char* outdir, const char* prefix,
                                         int32_t* order_out, MirpMirna* sorted_out, int64_t* counts_out, char* errbuf, size_t errbuf_len) {
    auto bail = [&](int code, const std::string& m) { if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", m.c_str()); return code; };
    const ReportInputs in = {contig_names, n_contigs, contig_seq, contig_len, alns, n_alns, sample_names, n_samples, mirbase_form, outdir, prefix};
    if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!result || !ss_text || ss_stride < 1)) || bad_inputs(in)) return bail(-1, "mirp_write_result_reports: bad argument");
    ReportAccum A(in);
    // the seven report files and the small files do not depend on each other: the small files of the (one) chunk go out on a thread of their own
    if (int rc = A.add_chunk(result, n, ss_text, ss_stride)) return bail(rc, A.err);
    if (order_out && n) std::memcpy(order_out, A.order.data(), sizeof(int32_t) * (size_t)n);
    if (sorted_out && n) std::memcpy(sorted_out, A.sorted.data(), sizeof(MirpMirna) * (size_t)n);
    if (counts_out && n) std::memcpy(counts_out, A.counts.data(), sizeof(int64_t) * A.counts.size());
    if (int rc = A.finish()) return bail(rc, A.err);
    return 0;
}

// Chunk boundaries of the window list at which the list order of the RESULTS is decided by the windows: a locus found in a window lies inside it
// (fold_s >= ws, fold_e <= we), so if every window before the cut ends before every window behind it starts -- or the contig changes; windows come
// in sorted contig-name order (MP:1309) -- every locus before the cut sorts before every locus behind it.  An (L, R) pair of windows overlaps and is
// never split.  -> first window of every chunk plus the end.
static std::vector<long long> plan_chunks(const std::vector<MirpWindow>& W, int n_chunks) {
    const long long nw = (long long)W.size();
    std::vector<long long> cuts{0};
    if (nw == 0) { cuts.push_back(0); return cuts; }
    std::vector<char> safe((size_t)nw + 1, 0);
    int max_we = -1, cur_tid = -1;
    for (long long k = 0; k < nw; k++) {
        if (W[(size_t)k].tid != cur_tid) { safe[(size_t)k] = 1; cur_tid = W[(size_t)k].tid; max_we = -1; }
        else if (W[(size_t)k].ws > max_we) safe[(size_t)k] = 1;
        max_we = std::max(max_we, W[(size_t)k].we);
    }
    for (int c = 1; c < n_chunks; c++) {
        long long k = nw * c / n_chunks;
        while (k < nw && !safe[(size_t)k]) k++;
        if (k < nw && k > cuts.back()) cuts.push_back(k);
    }
    cuts.push_back(nw);
    return cuts;
}

extern "C" int mirp_fold_predict_report_stream(mirp_ctx* c, int32_t span, int32_t max_lines, const MirpPredictParams* pp, int32_t n_chunks, const char* contig_names,
                                               int32_t n_contigs, const uint8_t* const* contig_seq, const int64_t* contig_len, const MirpAln* alns, int64_t n_alns,
                                               const char* sample_names, int32_t n_samples, const char* mirbase_form, const char* outdir, const char* prefix,
                                               int64_t* n_loci, int32_t* n_chunks_used, double device_ms[2]) {
    if (!c) return -1;
    const ReportInputs in = {contig_names, n_contigs, contig_seq, contig_len, alns, n_alns, sample_names, n_samples, mirbase_form, outdir, prefix};
    if (!pp || bad_inputs(in) || !n_loci || n_chunks < 1) return fail(c, -1, "mirp_fold_predict_report_stream: bad argument");
    if (!c->have_candidate) return fail(c, -1, "mirp_fold_predict_report_stream: run mirp_candidate first");
    if (c->sel_total >= 0) return fail(c, -1, "mirp_fold_predict_report_stream: a window view is active");
    HIPCHK(c, hipSetDevice(c->device));
    const long long nw = c->n_windows;
    std::vector<MirpWindow> W((size_t)nw);
    if (nw) HIPCHK(c, hipMemcpy(W.data(), c->windows.p, sizeof(MirpWindow) * (size_t)nw, hipMemcpyDeviceToHost));
    const std::vector<long long> cuts = plan_chunks(W, n_chunks);
    W.clear(); W.shrink_to_fit();
    // ---- the report side: a host thread takes a chunk's result as soon as the filter has produced it
    struct Item { MirpMirna* rec; int64_t n; char* text; int32_t stride; };
    std::deque<Item> q;
    std::mutex mu; std::condition_variable cv;
    bool done = false;
    int wrc = 0;
    ReportAccum A(in);
    std::thread worker([&] {
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done || !q.empty(); });
                if (q.empty()) break;
                it = q.front(); q.pop_front();
            }
            if (!wrc) wrc = A.add_chunk(it.rec, it.n, it.text, it.stride);
            std::free(it.rec); std::free(it.text);
        }
        if (!wrc) wrc = A.finish();
    });
    auto stop_worker = [&] { { std::lock_guard<std::mutex> lk(mu); done = true; } cv.notify_all(); worker.join(); };
    double ms_fold = 0, ms_pred = 0, km0 = 0, km1 = 0;
    long long fallbacks = 0, overflow = 0; unsigned int dense = 0;
    int rc = 0;
    for (size_t k = 0; k + 1 < cuts.size() && !rc; k++) {
        const long long first = cuts[k], cnt = cuts[k + 1] - cuts[k];
        if ((rc = mirp_select_windows(c, first, cnt))) break;
        if ((rc = mirp_fold(c, span, max_lines))) break;
        ms_fold += c->ms[2]; km0 += c->fold_kernel_ms[0]; km1 += c->fold_kernel_ms[1];
        fallbacks += c->last_fallback; overflow += c->n_side; dense += c->last_dense;
        {   // a window the fold could not finish is an error of the run (MP:3103-3106), never a silently missing locus
            std::vector<int> st((size_t)cnt);
            if (cnt && hipMemcpy(st.data(), c->status.p, 4 * (size_t)cnt, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(c, -2, "D2H failed (fold status)"); break; }
            for (long long w = 0; w < cnt; w++)
                if (st[(size_t)w] != 0) { rc = fail(c, -5, "Error occurred when folding sequences (window " + std::to_string(first + w) + ", status " + std::to_string(st[(size_t)w]) + ")."); break; }
            if (rc) break;
        }
        MirpMirna* res = nullptr; int64_t nres = 0; char* text = nullptr; int32_t stride = 0; int32_t* npass = nullptr; int32_t* pst = nullptr; int64_t nwin = 0;
        if ((rc = mirp_predict(c, pp, &res, &nres, &text, &stride, &npass, &pst, &nwin))) break;
        ms_pred += c->ms[3];
        for (int64_t w = 0; w < nwin && !rc; w++)
            if (pst[w] != 0) rc = fail(c, -5, "Error occurred when predicting miRNAs: window " + std::to_string(first + w) + " exceeds the capacity of the filter kernel (status " + std::to_string(pst[w]) + ").");
        std::free(npass); std::free(pst);
        if (rc) { std::free(res); std::free(text); break; }
        { std::lock_guard<std::mutex> lk(mu); q.push_back({res, nres, text, stride}); }
        cv.notify_all();
    }
    stop_worker();
    (void)mirp_select_windows(c, 0, -1);
    c->ms[2] = ms_fold; c->ms[3] = ms_pred; c->fold_kernel_ms[0] = km0; c->fold_kernel_ms[1] = km1;
    c->last_fallback = fallbacks; c->n_side = 0; c->last_dense = dense;
    (void)overflow;
    if (rc) return rc;
    if (wrc) return fail(c, wrc, A.err);
    *n_loci = A.n;
    if (n_chunks_used) *n_chunks_used = (int32_t)cuts.size() - 1;
    if (device_ms) { device_ms[0] = ms_fold; device_ms[1] = ms_pred; }
    return 0;
}
