// The tail of the predict stage in one native call (host only): from the result list of mirp_predict / mirp_gather_loci to every report file the
// reference writes after its queue is drained -- the mature/star swap of gen_miRNA_loci_nopredict's caller (MP:2611-2617), resultlist.sort() of
// gen_gff_from_result (MP:2622), the per-sample read counts of gen_mirna_info (MP:2644-2728), the read-mapping file of every locus (gen_map_result,
// MP:2907-2959) and the seven report files (MP:2619-2641, 2744-2779, 2793-2904, 2963-3019, 3585-3593).  The Python host used to do the list handling
// between these (a Python object per locus and field; 0.05 s at 4,002 loci, 0.25 s at 16,016); here the flat record array goes in and files come
// out.  Formatting is shared with the single-purpose entry points (mirp_report_readmapping, mirp_write_reports, mirp_write_files), so the bytes are
// the same by construction; tests/test_host_cpu.py holds this entry point against the reference's files as well.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include "../../include/mirprefer.h"

namespace {
struct Key {
    int32_t idx;
};
}  // namespace

extern "C" int mirp_write_result_reports(const MirpMirna* result, int64_t n, const char* ss_text, int32_t ss_stride, const char* contig_names, int32_t n_contigs,
                                         const uint8_t* const* contig_seq, const int64_t* contig_len, const MirpAln* alns, int64_t n_alns,
                                         const char* sample_names, int32_t n_samples, const char* mirbase_form, const char* outdir, const char* prefix,
                                         int32_t* order_out, MirpMirna* sorted_out, int64_t* counts_out, char* errbuf, size_t errbuf_len) {
    auto bail = [&](int code, const std::string& m) { if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", m.c_str()); return code; };
    if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!result || !ss_text || ss_stride < 1)) || !contig_names || n_contigs < 0 || !contig_seq || !contig_len ||
        (n_alns > 0 && !alns) || n_samples < 1 || !sample_names || !mirbase_form || !outdir || !prefix)
        return bail(-1, "mirp_write_result_reports: bad argument");
    std::vector<const char*> cname((size_t)n_contigs);
    { const char* p = contig_names; for (int t = 0; t < n_contigs; t++) { cname[(size_t)t] = p; p += std::strlen(p) + 1; } }
    // ---- records with the more abundant arm as the mature (MP:2611-2617), then the list order of resultlist.sort() (MP:2622): Python compares
    // [chr, fold_s, fold_e, mat_s, mat_e, star_s, star_e, ss, strand, has_star] element by element; strings by code point = bytes for ASCII
    std::vector<MirpMirna> rec(result, result + n);
    for (int64_t k = 0; k < n; k++) {
        MirpMirna& m = rec[(size_t)k];
        if (m.tid < 0 || m.tid >= n_contigs) return bail(-1, "mirp_write_result_reports: record with a contig index outside the contig table");
        if (m.total_depth_mature < m.total_depth_star) {
            std::swap(m.total_depth_mature, m.total_depth_star);
            std::swap(m.mat_s, m.star_s);
            std::swap(m.mat_e, m.star_e);
        }
    }
    std::vector<int32_t> order((size_t)n);
    for (int64_t k = 0; k < n; k++) order[(size_t)k] = (int32_t)k;
    auto ss_of = [&](int32_t k) { return ss_text + (size_t)k * (size_t)ss_stride; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const MirpMirna& x = rec[(size_t)a]; const MirpMirna& y = rec[(size_t)b];
        if (x.tid != y.tid) { const int c = std::strcmp(cname[(size_t)x.tid], cname[(size_t)y.tid]); if (c) return c < 0; }
        if (x.fold_s != y.fold_s) return x.fold_s < y.fold_s;
        if (x.fold_e != y.fold_e) return x.fold_e < y.fold_e;
        if (x.mat_s != y.mat_s) return x.mat_s < y.mat_s;
        if (x.mat_e != y.mat_e) return x.mat_e < y.mat_e;
        if (x.star_s != y.star_s) return x.star_s < y.star_s;
        if (x.star_e != y.star_e) return x.star_e < y.star_e;
        const int la = x.ss_len, lb = y.ss_len;
        const int c = std::memcmp(ss_of(a), ss_of(b), (size_t)std::min(la, lb));
        if (c) return c < 0;
        if (la != lb) return la < lb;
        if (x.strand != y.strand) return x.strand < y.strand;          // '+' (0) sorts before '-' (1), as the characters do
        return (x.has_star != 0) < (y.has_star != 0);
    });
    // ---- flat inputs of the formatters, in list order
    std::vector<int32_t> loci8((size_t)n * 8), loci10((size_t)n * 10);
    std::string ss_blob, pre_blob;
    std::vector<int64_t> counts((size_t)n * n_samples * 4, 0), counts0((size_t)n * n_samples, 0);
    for (int64_t i = 0; i < n; i++) {
        const MirpMirna& m = rec[(size_t)order[(size_t)i]];
        int32_t* a = &loci8[(size_t)i * 8];
        a[0] = m.tid; a[1] = m.fold_s; a[2] = m.fold_e; a[3] = m.mat_s; a[4] = m.mat_e; a[5] = m.star_s; a[6] = m.star_e; a[7] = m.strand ? 1 : 0;
        int32_t* b = &loci10[(size_t)i * 10];
        std::memcpy(b, a, 8 * sizeof(int32_t));
        b[8] = m.total_depth_star == 0 ? 0 : 1;
        const int f = m.reserved;
        b[9] = ((f & 1) && (f & 8)) ? ((((f >> 1) & 3) - 1) == 2 ? 2 : 1) : 0;          // overhangsize 2:2 / 2:3 / 3:3 (MP:2631-2637)
        ss_blob.append(ss_of(order[(size_t)i]), (size_t)m.ss_len); ss_blob.push_back('\0');
        // `samtools faidx chr:fold_s-(fold_e-1)`, upper case, T -> U (MP:2570-2590)
        const uint8_t* g = contig_seq[(size_t)m.tid];
        const int64_t gl = contig_len[(size_t)m.tid];
        if (!g) return bail(-1, std::string("mirp_write_result_reports: a locus lies on contig ") + cname[(size_t)m.tid] + ", which this process does not hold");
        const int64_t s0 = std::max<int64_t>(m.fold_s - 1, 0), s1 = std::min<int64_t>((int64_t)m.fold_e - 1, gl);
        for (int64_t p = s0; p < s1; p++) { char c = (char)g[p]; if (c >= 'a' && c <= 'z') c -= 32; if (c == 'T') c = 'U'; pre_blob.push_back(c); }
        pre_blob.push_back('\0');
    }
    // ---- reads per locus and sample: on the precursor / exactly the mature / exactly the star / antisense (gen_mirna_info, MP:2644-2728), from the
    // (tid, pos)-sorted records instead of one `samtools view` per locus
    {
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, n / 256));
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                for (int64_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
                    const int32_t* a = &loci8[(size_t)i * 8];
                    auto lower = [&](int64_t tid, int64_t pos) {
                        int64_t lo = 0, hi = n_alns;
                        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; const MirpAln& r = alns[mid]; if (r.tid < tid || (r.tid == tid && r.pos < pos)) lo = mid + 1; else hi = mid; }
                        return lo;
                    };
                    const int64_t lo = lower(a[0], a[1]), hi = lower(a[0], a[2]);
                    for (int64_t r = lo; r < hi; r++) {
                        const MirpAln& x = alns[r];
                        if ((int64_t)x.pos + x.len > a[2]) continue;
                        if (x.sample >= n_samples) continue;
                        int64_t* c = &counts[((size_t)i * n_samples + x.sample) * 4];
                        const bool sense = (x.strand ? 1 : 0) == a[7];
                        if (sense) {
                            c[0] += x.depth;
                            if (x.pos == a[3] && x.len == a[4] - a[3]) c[1] += x.depth;
                            if (x.pos == a[5] && x.len == a[6] - a[5]) c[2] += x.depth;
                        } else {
                            c[3] += x.depth;
                        }
                    }
                    for (int s = 0; s < n_samples; s++) counts0[(size_t)i * n_samples + s] = counts[((size_t)i * n_samples + s) * 4];
                }
            });
        for (auto& t : th) t.join();
    }
    if (order_out) std::memcpy(order_out, order.data(), sizeof(int32_t) * (size_t)n);
    if (sorted_out) for (int64_t i = 0; i < n; i++) sorted_out[i] = rec[(size_t)order[(size_t)i]];
    if (counts_out) std::memcpy(counts_out, counts.data(), sizeof(int64_t) * counts.size());
    if (n == 0) return 0;          // "0 miRNA identified. No result files generated." (MP:3547-3550)
    // ---- read-mapping bodies, then the small files behind the seven report files
    char* body = nullptr; int64_t* boffs = nullptr;
    int rc = mirp_report_readmapping(loci8.data(), n, ss_blob.c_str(), alns, n_alns, contig_seq, contig_len, n_contigs, sample_names, n_samples, counts0.data(), &body, &boffs);
    if (rc) return bail(rc, "mirp_write_result_reports: read-mapping bodies failed (a locus lies on a contig this process does not hold)");
    const std::string out(outdir), pre(prefix), rmdir = out + "/readmapping";
    ::mkdir(out.c_str(), 0777);
    ::mkdir(rmdir.c_str(), 0777);
    std::string paths, text;
    std::vector<int64_t> toffs((size_t)n + 1, 0);
    text.reserve((size_t)boffs[n] + (size_t)n * 64);
    for (int64_t i = 0; i < n; i++) {
        const int32_t* a = &loci8[(size_t)i * 8];
        char name[64];
        std::snprintf(name, sizeof name, "miRNA-precursor_%lld", (long long)i);
        paths += rmdir; paths += "/"; paths += name; paths += ".map.txt"; paths.push_back('\0');
        char head[96];
        std::snprintf(head, sizeof head, ":%d-%d %c\n", a[1], a[2], a[7] ? '-' : '+');
        text += ">"; text += name; text += " "; text += cname[(size_t)a[0]]; text += head;
        text.append(body + boffs[i], (size_t)(boffs[i + 1] - boffs[i]));
        toffs[(size_t)i + 1] = (int64_t)text.size();
    }
    mirp_free(body); mirp_free(boffs);
    int rc_files = 0; char err_files[512] = "";
    std::thread small([&] { rc_files = mirp_write_files(n, paths.c_str(), text.data(), toffs.data(), 1, err_files, sizeof err_files); });
    const std::string gff = out + "/" + pre + "_miRNA.gff3", mat = out + "/" + pre + "_miRNA.mature.fa", stem = out + "/" + pre + "_miRNA.precursor.fa",
                      ssf = out + "/" + pre + "_miRNA.precursor.ss", csv = out + "/" + pre + "_miRNA.detail.csv", html = out + "/" + pre + "_miRNA.detail.html",
                      stat = out + "/miRNA.stat.txt";
    char err_rep[512] = "";
    const int rc_rep = mirp_write_reports(n, loci10.data(), contig_names, n_contigs, ss_blob.c_str(), pre_blob.c_str(), sample_names, n_samples, counts.data(), mirbase_form,
                                          gff.c_str(), mat.c_str(), stem.c_str(), ssf.c_str(), csv.c_str(), html.c_str(), stat.c_str(), err_rep, sizeof err_rep);
    small.join();
    if (rc_rep) return bail(rc_rep, err_rep);
    if (rc_files) return bail(rc_files, err_files);
    return 0;
}
