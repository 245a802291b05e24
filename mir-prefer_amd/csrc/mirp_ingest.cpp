// Native SAM ingest: multi-threaded tokenizer + stable (tid, pos) sort producing the packed 16-byte
// alignment records the kernels consume.  Replaces the reference's prepare-stage plumbing -- sam2bam, `samtools cat`,
// `samtools sort`, expand_bamfile, strand split (/root/reference/miR_PREFeR.py:656-746, 772-874) -- with an in-memory
// equivalent; record order = sample order, then file order, stably sorted by (tid, pos), which is what the reference's
// combined sorted BAM presents to gen_loci_alignment_info (first-seen maximum at MP:1457).
// Host-only code: no device is touched.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/mirprefer.h"

namespace {

struct Mapped {
    const char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char*)m;
        return true;
    }
    ~Mapped() {
        if (p && n) munmap((void*)p, n);
        if (fd >= 0) ::close(fd);
    }
};

inline const char* field_end(const char* s, const char* e) { while (s < e && *s != '\t' && *s != '\n') s++; return s; }

struct ChunkOut {
    std::vector<MirpAln> recs;
    std::string err;
};

// parse the alignment lines in [b, e) (b at a line start)
void parse_chunk(const char* b, const char* e, const std::unordered_map<std::string, int>& tid_of, int sample, ChunkOut* out) {
    const char* s = b;
    std::string key;
    while (s < e) {
        const char* le = (const char*)memchr(s, '\n', (size_t)(e - s));
        if (!le) le = e;
        if (le > s && *s != '@') {
            const char* f[11];
            const char* fe[11];
            int nf = 0;
            const char* q = s;
            while (nf < 11 && q <= le) {
                f[nf] = q;
                const char* t = field_end(q, le);
                fe[nf] = t;
                nf++;
                if (t >= le) break;
                q = t + 1;
            }
            if (nf >= 10) {
                long flag = strtol(f[1], nullptr, 10);
                if (!(flag & 0x704)) {
                    // depth: ^\S+_x([0-9]+)  (get_read_depth_fromID_as_string, MP:242-253): greedy \S+ -> last "_x<digits>" with digits following
                    const char* id = f[0];
                    const char* ide = fe[0];
                    long depth = -1;
                    for (const char* t = ide - 1; t > id + 2; t--) {
                        if (*t >= '0' && *t <= '9' && t[-1] == 'x' && t[-2] == '_') {   // candidate start of the digit run
                            const char* d0 = t;
                            // the regex takes the LAST position where "_x" is followed by >= 1 digit; scanning from the right finds it first
                            depth = strtol(d0, nullptr, 10);
                            break;
                        }
                    }
                    if (depth < 0) { out->err = "Read Id format is not right. Read id must be in \"samplename_rA_xN\" format."; return; }
                    key.assign(f[2], fe[2]);
                    auto it = tid_of.find(key);
                    if (it == tid_of.end()) { out->err = "alignment refers to a sequence that is not in the @SQ header: " + key; return; }
                    long pos = strtol(f[3], nullptr, 10);
                    long rl = (long)(fe[9] - f[9]);
                    // ungapped alignments only: CIGAR must be "<len>M"
                    char* cend = nullptr;
                    long cl = strtol(f[5], &cend, 10);
                    if (!(cend && cend + 1 == fe[5] && *cend == 'M' && cl == rl)) {
                        out->err = "only ungapped alignments (<len>M) are supported, got CIGAR " + std::string(f[5], fe[5]);
                        return;
                    }
                    if (rl > 65535) { out->err = "read longer than 65535"; return; }
                    MirpAln r;
                    r.tid = it->second; r.pos = (int32_t)pos; r.depth = (uint32_t)depth; r.len = (uint16_t)rl;
                    r.strand = (flag & 16) ? 1 : 0; r.sample = (uint8_t)sample;
                    out->recs.push_back(r);
                }
            }
        }
        s = le + 1;
    }
}

inline uint64_t key_of(const MirpAln& r) { return ((uint64_t)(uint32_t)r.tid << 32) | (uint32_t)r.pos; }

}  // namespace

extern "C" void mirp_free_sam_data(MirpSamData* d) {
    if (!d) return;
    std::free(d->contig_names); std::free(d->contig_len); std::free(d->sample_names); std::free(d->alns);
    std::memset(d, 0, sizeof(*d));
}

extern "C" int mirp_ingest_sams(const char* const* paths, int32_t n_paths, int32_t n_threads, MirpSamData* out, char* errbuf, size_t errbuf_len) {
    auto fail = [&](const std::string& m) {
        if (errbuf && errbuf_len) { std::snprintf(errbuf, errbuf_len, "%s", m.c_str()); }
        return -1;
    };
    if (!paths || n_paths < 1 || !out) return fail("mirp_ingest_sams: bad argument");
    if (n_paths > MIRP_MAX_SAMPLES) return fail("mirp_ingest_sams: too many samples");
    std::memset(out, 0, sizeof(*out));
    if (n_threads < 1) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::string> names;
    std::vector<int64_t> lens;
    std::unordered_map<std::string, int> tid_of;
    std::vector<std::string> samples;
    std::vector<std::vector<ChunkOut>> per_file(n_paths);
    for (int fi = 0; fi < n_paths; fi++) {
        Mapped m;
        if (!m.open(paths[fi])) return fail(std::string("cannot open ") + paths[fi]);
        const char* b = m.p;
        const char* e = m.p + m.n;
        // header (get_length_from_sam, MP:500-509: @SQ order of the FIRST file defines the contig indices)
        const char* s = b;
        while (s < e && *s == '@') {
            const char* le = (const char*)memchr(s, '\n', (size_t)(e - s));
            if (!le) le = e;
            if (fi == 0 && le - s > 3 && s[1] == 'S' && s[2] == 'Q') {
                std::string line(s, le), sn;
                long ln = -1;
                size_t p = 0;
                while (p < line.size()) {
                    size_t t = line.find('\t', p);
                    if (t == std::string::npos) t = line.size();
                    if (line.compare(p, 3, "SN:") == 0) sn = line.substr(p + 3, t - p - 3);
                    if (line.compare(p, 3, "LN:") == 0) ln = strtol(line.c_str() + p + 3, nullptr, 10);
                    p = t + 1;
                }
                if (!sn.empty() && ln >= 0) { tid_of[sn] = (int)names.size(); names.push_back(sn); lens.push_back(ln); }
            }
            s = le < e ? le + 1 : e;
        }
        if (fi == 0 && names.empty()) return fail("Can not get the sequence length from the input SAM files. Make sure SAM files have headers.");
        // sample name from the first alignment line (get_samplename_from_sam, MP:3300-3308)
        {
            const char* t = field_end(s, e);
            std::string id(s, t), sname;
            std::vector<size_t> us;
            for (size_t k = 0; k < id.size(); k++) if (id[k] == '_') us.push_back(k);
            if (us.size() >= 2) sname = id.substr(0, us[us.size() - 2]);
            samples.push_back(sname);
        }
        // split the body at line starts
        const size_t body = (size_t)(e - s);
        int nt = (int)std::min<size_t>((size_t)n_threads, std::max<size_t>(1, body / (1 << 20)));
        std::vector<const char*> cuts(nt + 1);
        cuts[0] = s; cuts[nt] = e;
        for (int k = 1; k < nt; k++) {
            const char* c = s + body * k / nt;
            const char* le = (const char*)memchr(c, '\n', (size_t)(e - c));
            cuts[k] = le ? le + 1 : e;
        }
        per_file[fi].resize(nt);
        std::vector<std::thread> th;
        for (int k = 0; k < nt; k++) th.emplace_back(parse_chunk, cuts[k], cuts[k + 1], std::cref(tid_of), fi, &per_file[fi][k]);
        for (auto& t : th) t.join();
        for (auto& c : per_file[fi]) if (!c.err.empty()) return fail(c.err);
    }
    // concatenate in (sample, file) order
    size_t total = 0;
    for (auto& f : per_file) for (auto& c : f) total += c.recs.size();
    MirpAln* all = (MirpAln*)std::malloc(std::max<size_t>(total, 1) * sizeof(MirpAln));
    if (!all) return fail("out of memory");
    size_t o = 0;
    for (auto& f : per_file) for (auto& c : f) { std::memcpy(all + o, c.recs.data(), c.recs.size() * sizeof(MirpAln)); o += c.recs.size(); std::vector<MirpAln>().swap(c.recs); }
    // stable sort by (tid, pos): sort equal slices in parallel (stable), then merge neighbours (std::inplace_merge is stable)
    {
        int nt = (int)std::min<size_t>((size_t)n_threads, std::max<size_t>(1, total / (1 << 16)));
        std::vector<size_t> cut(nt + 1);
        for (int k = 0; k <= nt; k++) cut[k] = total * k / nt;
        auto cmp = [](const MirpAln& a, const MirpAln& b) { return key_of(a) < key_of(b); };
        std::vector<std::thread> th;
        for (int k = 0; k < nt; k++) th.emplace_back([&, k] { std::stable_sort(all + cut[k], all + cut[k + 1], cmp); });
        for (auto& t : th) t.join();
        for (int step = 1; step < nt; step *= 2) {
            std::vector<std::thread> mt;
            for (int k = 0; k + step < nt; k += 2 * step) {
                int hi = std::min(k + 2 * step, nt);
                mt.emplace_back([&, k, step, hi] { std::inplace_merge(all + cut[k], all + cut[k + step], all + cut[hi], cmp); });
            }
            for (auto& t : mt) t.join();
        }
    }
    // outputs
    size_t nb = 0;
    for (auto& s : names) nb += s.size() + 1;
    out->contig_names = (char*)std::malloc(std::max<size_t>(nb, 1));
    out->contig_len = (int64_t*)std::malloc(std::max<size_t>(names.size(), 1) * sizeof(int64_t));
    size_t sb = 0;
    for (auto& s : samples) sb += s.size() + 1;
    out->sample_names = (char*)std::malloc(std::max<size_t>(sb, 1));
    if (!out->contig_names || !out->contig_len || !out->sample_names) { std::free(all); mirp_free_sam_data(out); return fail("out of memory"); }
    char* w = out->contig_names;
    for (size_t k = 0; k < names.size(); k++) { std::memcpy(w, names[k].c_str(), names[k].size() + 1); w += names[k].size() + 1; out->contig_len[k] = lens[k]; }
    w = out->sample_names;
    for (auto& s : samples) { std::memcpy(w, s.c_str(), s.size() + 1); w += s.size() + 1; }
    out->n_contigs = (int32_t)names.size(); out->n_samples = (int32_t)samples.size(); out->alns = all; out->n_alns = (int64_t)total;
    return 0;
}
