// Native SAM ingest: multi-threaded tokenizer producing the packed 16-byte alignment records the kernels consume, then the
// stable (tid, pos) sort -- on the device (mirp_ingest_sams_gpu: radix sort + optional GFF region filter, sort_kernels.hip) or on the host
// (mirp_ingest_sams, no device needed).  Replaces the reference's prepare-stage plumbing -- sam2bam, `samtools cat`, `samtools sort`,
// `samtools view -L`, expand_bamfile, strand split (/root/reference/miR_PREFeR.py:656-746, 772-874) -- with an in-memory equivalent; record
// order = sample order, then file order, stably sorted by (tid, pos), which is what the reference's combined sorted BAM presents to
// gen_loci_alignment_info (first-seen maximum at MP:1457).
//
// Gapped alignments (CIGAR with I / D / N / S / H / P / = / X): the record keeps POS and the SEQ length, which is all the reference's read
// bookkeeping looks at (`samtools view` fields 3 and 9, MP:1439-1457, 2021); the per-base coverage of `samtools depth` counts the M / = / X
// blocks only, so such a read also emits coverage segments: one that takes its [POS, POS + len(SEQ)) interval back out
// and one per M / = / X block (mirp_load_coverage_segments).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sched.h>
#include <thread>
#include <vector>
#include "mirp_ctx.h"

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
        (void)madvise(m, n, MADV_SEQUENTIAL);
        p = (const char*)m;
        return true;
    }
    ~Mapped() {
        if (p && n) munmap((void*)p, n);
        if (fd >= 0) ::close(fd);
    }
};

// contig name -> index: open addressing over FNV-1a of the name bytes (no allocation per line)
struct NameTable {
    std::vector<std::string> names;
    std::vector<int> slot;      // -1 = empty
    unsigned mask = 0;
    static unsigned hash(const char* s, size_t n) { unsigned h = 2166136261u; for (size_t k = 0; k < n; k++) { h ^= (unsigned char)s[k]; h *= 16777619u; } return h; }
    void build() {
        unsigned cap = 16;
        while (cap < 4 * names.size() + 8) cap <<= 1;
        slot.assign(cap, -1); mask = cap - 1;
        for (size_t t = 0; t < names.size(); t++) {
            unsigned h = hash(names[t].data(), names[t].size()) & mask;
            while (slot[h] >= 0) { if (names[slot[h]] == names[t]) break; h = (h + 1) & mask; }
            if (slot[h] < 0) slot[h] = (int)t;      // a repeated @SQ name keeps its first index
        }
    }
    int find(const char* s, size_t n) const {
        unsigned h = hash(s, n) & mask;
        while (slot[h] >= 0) {
            const std::string& c = names[slot[h]];
            if (c.size() == n && std::memcmp(c.data(), s, n) == 0) return slot[h];
            h = (h + 1) & mask;
        }
        return -1;
    }
};

struct ChunkOut {
    std::vector<MirpAln> recs, segs;
    std::vector<int32_t> seg_owner;      // index of the owning record inside this chunk
    std::vector<int32_t> seg_span;       // for the subtract segment of a gapped record: its reference span + 1 (bam_calend - POS: what `samtools view -L` tests); else 0
    std::string err;
};

inline bool parse_uint(const char* s, const char* e, long* out) {
    if (s >= e) return false;
    long v = 0;
    for (; s < e; s++) { const unsigned d = (unsigned)(*s - '0'); if (d > 9) return false; v = v * 10 + d; if (v > 0x7fffffffffffL / 16) return false; }
    *out = v;
    return true;
}

// parse the alignment lines in [b, e) (b at a line start)
void parse_chunk(const char* b, const char* e, const NameTable* tab, int sample, ChunkOut* out) {
    const char* s = b;
    out->recs.reserve((size_t)(e - b) / 64 + 16);
    int last_tid = -1;
    const char* last_name = nullptr;
    size_t last_len = 0;
    while (s < e) {
        const char* le = (const char*)memchr(s, '\n', (size_t)(e - s));
        if (!le) le = e;
        const char* lend = (le > s && le[-1] == '\r') ? le - 1 : le;
        if (lend > s && *s != '@') {
            const char* f[11];
            const char* fe[11];
            int nf = 0;
            const char* q = s;
            while (nf < 11) {
                f[nf] = q;
                const char* t = (const char*)memchr(q, '\t', (size_t)(lend - q));
                if (!t) t = lend;
                fe[nf++] = t;
                if (t >= lend) break;
                q = t + 1;
            }
            if (nf >= 10) {
                long flag = 0;
                if (!parse_uint(f[1], fe[1], &flag)) { out->err = "SAM flag is not a number"; return; }
                if (!(flag & 0x704)) {
                    // depth: ^\S+_x([0-9]+)  (get_read_depth_fromID_as_string, MP:242-253): greedy \S+ -> the LAST "_x" that digits follow
                    const char* id = f[0];
                    const char* ide = fe[0];
                    long depth = -1;
                    for (const char* t = ide - 1; t > id + 2; t--) {
                        if (*t >= '0' && *t <= '9' && t[-1] == 'x' && t[-2] == '_') {
                            long v = 0;
                            for (const char* d = t; d < ide && *d >= '0' && *d <= '9'; d++) { v = v * 10 + (*d - '0'); if (v > 0xffffffffL) { v = 0xffffffffL; break; } }
                            depth = v;
                            break;
                        }
                    }
                    if (depth < 0) { out->err = "Read Id format is not right. Read id must be in \"samplename_rA_xN\" format."; return; }
                    const size_t nl = (size_t)(fe[2] - f[2]);
                    int tid;
                    if (last_name && nl == last_len && std::memcmp(last_name, f[2], nl) == 0) tid = last_tid;
                    else {
                        tid = tab->find(f[2], nl);
                        if (tid < 0) { out->err = "alignment refers to a sequence that is not in the @SQ header: " + std::string(f[2], fe[2]); return; }
                        last_tid = tid; last_name = f[2]; last_len = nl;
                    }
                    long pos = 0;
                    if (!parse_uint(f[3], fe[3], &pos) || pos > 0x7ffffff0L) { out->err = "SAM position is not a number"; return; }
                    const long rl = (long)(fe[9] - f[9]);
                    if (rl > 65535) { out->err = "read longer than 65535"; return; }
                    MirpAln r;
                    r.tid = tid; r.pos = (int32_t)pos; r.depth = (uint32_t)depth; r.len = (uint16_t)rl;
                    r.strand = (flag & 16) ? 1 : 0; r.sample = (uint8_t)sample;
                    // CIGAR: "<len>M" with len == len(SEQ) is the plain case; anything else gets coverage segments
                    const char* c = f[5];
                    const char* ce = fe[5];
                    long cl = 0;
                    const char* p = c;
                    while (p < ce && *p >= '0' && *p <= '9') { cl = cl * 10 + (*p - '0'); p++; }
                    const bool plain = p > c && p + 1 == ce && *p == 'M' && cl == rl;
                    if (!plain) {
                        if (ce - c == 1 && *c == '*') { out->err = "alignment without a CIGAR string"; return; }
                        const int32_t owner = (int32_t)out->recs.size();
                        MirpAln sg = r;
                        sg.strand = (uint8_t)(r.strand | 2);                 // bit 1: subtract (takes the record's own interval back out)
                        out->segs.push_back(sg); out->seg_owner.push_back(owner); out->seg_span.push_back(0);
                        const size_t sub_idx = out->seg_span.size() - 1;
                        // Two passes over the CIGAR.  The bundled samtools 0.1.18 piles a read up only below its bam_calend(), which adds up the
                        // M, D and N lengths but not = and X: blocks are cut at pos + sum(M, D, N) (probed against the binary with
                        // tests/golden/tools/gen_gapped_golden.py; an alignment written with = / X loses its tail there, as in the reference run).
                        long calend = pos;
                        p = c;
                        while (p < ce) {
                            long len = 0;
                            const char* d0 = p;
                            while (p < ce && *p >= '0' && *p <= '9') { len = len * 10 + (*p - '0'); p++; if (len > 0x7ffffff0L) break; }
                            if (p == d0 || p >= ce) { out->err = "malformed CIGAR " + std::string(c, ce); return; }
                            const char op = *p++;
                            if (op == 'M' || op == 'D' || op == 'N') calend += len;
                            else if (!(op == 'I' || op == 'S' || op == 'H' || op == 'P' || op == '=' || op == 'X')) { out->err = "malformed CIGAR " + std::string(c, ce); return; }
                        }
                        out->seg_span[sub_idx] = (int32_t)std::min<long>(calend - pos, 0x7ffffff0L) + 1;
                        long ref = pos;
                        p = c;
                        while (p < ce) {
                            long len = 0;
                            while (p < ce && *p >= '0' && *p <= '9') { len = len * 10 + (*p - '0'); p++; }
                            const char op = *p++;
                            if (op == 'M' || op == '=' || op == 'X') {
                                const long stop = std::min(ref + len, calend);
                                long off = ref;
                                while (off < stop) {                         // blocks longer than 65535 are cut
                                    const long part = std::min<long>(stop - off, 65535);
                                    MirpAln bsg = r;
                                    bsg.pos = (int32_t)off; bsg.len = (uint16_t)part;
                                    out->segs.push_back(bsg); out->seg_owner.push_back(owner); out->seg_span.push_back(0);
                                    off += part;
                                }
                                ref += len;
                            } else if (op == 'D' || op == 'N') ref += len;
                        }
                    }
                    out->recs.push_back(r);
                }
            }
        }
        s = le + 1;
    }
}

inline uint64_t key_of(const MirpAln& r) { return ((uint64_t)(uint32_t)r.tid << 32) | (uint32_t)r.pos; }

struct Parsed {
    std::vector<std::string> names, samples;
    std::vector<int64_t> lens;
    std::vector<std::vector<ChunkOut>> per_file;
    size_t n_recs = 0, n_segs = 0, bytes = 0;
};

// header + threaded tokenizer over every file; records stay in per-chunk vectors (sample, then file order)
// (part, n_parts): this caller tokenizes the part-th of n_parts equal byte ranges of every file's body (cut at line starts): the sharded ingest
// of one process per GPU; (0, 1) = the whole file.

// CPUs this process may actually use: the affinity mask and the cgroup quota, not the box's core count (a container on a 256-thread host is often granted
// 8 or 16; 256 tokenizer threads on 16 CPUs only add switches).
static int usable_cpus() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0 && k < n) n = k; }
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = ""; long long per = 0;
        if (std::fscanf(f, "%31s %lld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) {
            const int k = (int)((std::atoll(q) + per - 1) / per);
            if (k > 0 && k < n) n = k;
        }
        std::fclose(f);
    }
    return n;
}

int parse_all(const char* const* paths, int32_t n_paths, int32_t n_threads, Parsed* P, std::string* err, int part = 0, int n_parts = 1) {
    if (n_threads < 1) n_threads = usable_cpus();
    NameTable tab;
    P->per_file.resize(n_paths);
    for (int fi = 0; fi < n_paths; fi++) {
        Mapped m;
        if (!m.open(paths[fi])) { *err = std::string("cannot open ") + paths[fi]; return -1; }
        P->bytes += m.n;
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
                while (!sn.empty() && (sn.back() == '\r' || sn.back() == '\n')) sn.pop_back();
                if (!sn.empty() && ln >= 0) { tab.names.push_back(sn); P->lens.push_back(ln); }
            }
            s = le < e ? le + 1 : e;
        }
        if (fi == 0) {
            if (tab.names.empty()) { *err = "Can not get the sequence length from the input SAM files. Make sure SAM files have headers."; return -1; }
            tab.build();
        }
        // sample name from the first alignment line (get_samplename_from_sam, MP:3300-3308)
        {
            const char* t = s;
            while (t < e && *t != '\t' && *t != '\n') t++;
            std::string id(s, t), sname;
            std::vector<size_t> us;
            for (size_t k = 0; k < id.size(); k++) if (id[k] == '_') us.push_back(k);
            if (us.size() >= 2) sname = id.substr(0, us[us.size() - 2]);
            P->samples.push_back(sname);
        }
        if (n_parts > 1) {      // this rank's byte range of the body, cut at line starts
            const size_t whole = (size_t)(e - s);
            auto cut_at = [&](int k) -> const char* {
                if (k <= 0) return s;
                if (k >= n_parts) return e;
                const char* c = s + whole / n_parts * k;
                if (c <= s) return s;
                const char* le = (const char*)memchr(c - 1, '\n', (size_t)(e - (c - 1)));      // c itself is a line start if c[-1] is the newline
                return le ? le + 1 : e;
            };
            const char* s2 = cut_at(part);
            e = cut_at(part + 1);
            s = s2;
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
        P->per_file[fi].resize(nt);
        std::vector<std::thread> th;
        for (int k = 0; k < nt; k++) th.emplace_back(parse_chunk, cuts[k], cuts[k + 1], &tab, fi, &P->per_file[fi][k]);
        for (auto& t : th) t.join();
        for (auto& c : P->per_file[fi]) if (!c.err.empty()) { *err = c.err; return -1; }
    }
    for (auto& f : P->per_file) for (auto& c : f) { P->n_recs += c.recs.size(); P->n_segs += c.segs.size(); }
    P->names = tab.names;
    return 0;
}

// concatenation in (sample, file) order; segment owners become indices into the concatenated record array
void concat(Parsed& P, MirpAln* all, MirpAln* segs, int32_t* owner, int32_t* span = nullptr) {
    size_t o = 0, so = 0;
    for (auto& f : P.per_file)
        for (auto& c : f) {
            if (!c.recs.empty()) std::memcpy(all + o, c.recs.data(), c.recs.size() * sizeof(MirpAln));
            if (!c.segs.empty()) {
                std::memcpy(segs + so, c.segs.data(), c.segs.size() * sizeof(MirpAln));
                for (size_t k = 0; k < c.segs.size(); k++) owner[so + k] = (int32_t)(o + (size_t)c.seg_owner[k]);
                if (span) std::memcpy(span + so, c.seg_span.data(), c.segs.size() * sizeof(int32_t));
                so += c.segs.size();
            }
            o += c.recs.size();
            std::vector<MirpAln>().swap(c.recs); std::vector<MirpAln>().swap(c.segs); std::vector<int32_t>().swap(c.seg_owner); std::vector<int32_t>().swap(c.seg_span);
        }
}

int fill_meta(const Parsed& P, MirpSamData* out) {
    size_t nb = 0;
    for (auto& s : P.names) nb += s.size() + 1;
    out->contig_names = (char*)std::malloc(std::max<size_t>(nb, 1));
    out->contig_len = (int64_t*)std::malloc(std::max<size_t>(P.names.size(), 1) * sizeof(int64_t));
    size_t sb = 0;
    for (auto& s : P.samples) sb += s.size() + 1;
    out->sample_names = (char*)std::malloc(std::max<size_t>(sb, 1));
    if (!out->contig_names || !out->contig_len || !out->sample_names) return -1;
    char* w = out->contig_names;
    for (size_t k = 0; k < P.names.size(); k++) { std::memcpy(w, P.names[k].c_str(), P.names[k].size() + 1); w += P.names[k].size() + 1; out->contig_len[k] = P.lens[k]; }
    w = out->sample_names;
    for (auto& s : P.samples) { std::memcpy(w, s.c_str(), s.size() + 1); w += s.size() + 1; }
    out->n_contigs = (int32_t)P.names.size(); out->n_samples = (int32_t)P.samples.size();
    return 0;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" void mirp_free_sam_data(MirpSamData* d) {
    if (!d) return;
    std::free(d->contig_names); std::free(d->contig_len); std::free(d->sample_names); std::free(d->alns); std::free(d->segs);
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
    if (n_threads < 1) n_threads = usable_cpus();
    Parsed P;
    std::string err;
    if (parse_all(paths, n_paths, n_threads, &P, &err)) return fail(err);
    const size_t total = P.n_recs;
    MirpAln* all = (MirpAln*)std::malloc(std::max<size_t>(total, 1) * sizeof(MirpAln));
    MirpAln* segs = (MirpAln*)std::malloc(std::max<size_t>(P.n_segs, 1) * sizeof(MirpAln));
    std::vector<int32_t> owner(std::max<size_t>(P.n_segs, 1));
    if (!all || !segs) { std::free(all); std::free(segs); return fail("out of memory"); }
    concat(P, all, segs, owner.data());
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
    if (fill_meta(P, out)) { std::free(all); std::free(segs); mirp_free_sam_data(out); return fail("out of memory"); }
    out->alns = all; out->n_alns = (int64_t)total; out->segs = segs; out->n_segs = (int64_t)P.n_segs;
    return 0;
}

// Device path: host threads tokenize, the records go to the GPU in (sample, file) order, the optional keep-region filter
// (`samtools view -L`, MP:817-859) and the stable radix sort run there, and the sorted records stay resident as the context's alignments
// (as after mirp_load_alignments + mirp_load_coverage_segments) besides being returned to the host.
//
// Sharded form (one process per GPU, mirp_dist_init done): every rank tokenizes its own byte range of every file, routes each record to the
// rank that owns its contig (owner_of_tid) with one all-to-all over RCCL, and filters / sorts what it receives.  The receiver lays the blocks
// out by (file, source rank), i.e. in (sample, file offset) order, so the stable sort sees exactly the sequence the single-process ingest sees for
// those contigs and ties resolve identically (first-seen maximum, MP:1457).
static int ingest_impl(mirp_ctx* c, const char* const* paths, int32_t n_paths, int32_t n_threads, const MirpRegion* keep_regions, int64_t n_regions,
                       bool shard, const int32_t* owner_of_tid, MirpSamData* out, double seconds[4], const char* who, Parsed* pre = nullptr) {
    if (!c) return -1;
    if ((!paths && !pre) || n_paths < 1 || !out) return fail(c, -1, std::string(who) + ": bad argument");
    if (n_paths > MIRP_MAX_SAMPLES) return fail(c, -1, std::string(who) + ": too many samples");
    std::memset(out, 0, sizeof(*out));
    HIPCHK(c, hipSetDevice(c->device));
    const int W = shard ? c->dist_world : 1, me = shard ? c->dist_rank : 0;
    if (shard && W > 1 && !owner_of_tid) return fail(c, -1, std::string(who) + ": owner_of_tid is required with more than one rank");
    const double t0 = now_s();
    Parsed P_own;
    Parsed& P = pre ? *pre : P_own;          // pre: the files were tokenized before the device was open (mirp_tokenize_sams)
    std::string err;
    int bad = pre ? 0 : (parse_all(paths, n_paths, n_threads, &P, &err, me, W) ? 1 : 0);
    const int nc = (int)P.names.size();
    if (!bad && W > 1)
        for (int t = 0; t < nc && !bad; t++)
            if (owner_of_tid[t] < 0 || owner_of_tid[t] >= W) { bad = 1; err = std::string(who) + ": owner_of_tid out of range"; }
    // A record past the end of its contig would overflow the position bits of the sort key.  Refused on both ingest paths (here and in
    // mir-prefer_amd/ingest.py) -- a documented deviation: probed in round 4, the bundled samtools 0.1.18 ACCEPTS such a record (view -bS, sort, index,
    // depth all succeed and print positions beyond LN), so the reference runs on with peaks outside the contig; this build stops with a message
    // instead of reproducing that (waived behaviours).
    if (!bad)
        for (auto& f : P.per_file)
            for (auto& ch : f)
                for (const MirpAln& r : ch.recs)
                    if (!bad && (int64_t)r.pos > P.lens[r.tid] + 1) {
                        bad = 1;
                        err = "alignment position " + std::to_string(r.pos) + " is beyond the end of sequence " + P.names[r.tid] + " (LN:" + std::to_string(P.lens[r.tid]) + ")";
                    }
    if (W > 1) {          // a rank that found a bad line in ITS byte range must not leave the others inside the exchange: agree first
        long long mine = bad;
        std::vector<long long> all;
        if (int rc = mirp::dist_allgather_ll(c, &mine, 1, all)) return rc;
        for (int r = 0; r < W; r++)
            if (all[(size_t)r] && !bad) { bad = 1; err = std::string(who) + ": rank " + std::to_string(r) + " failed to parse its part of the SAM files"; }
    }
    if (bad) return fail(c, -1, err);
    const double t1 = now_s();
    MirpAln* all = nullptr;
    MirpAln* segs = nullptr;
    std::vector<int32_t> owner, span;
    long long n = 0, ns = 0;
    auto bail = [&](int code, const std::string& m) { std::free(all); std::free(segs); all = segs = nullptr; mirp_free_sam_data(out); return fail(c, code, m); };
    if (W == 1) {
        n = (long long)P.n_recs; ns = (long long)P.n_segs;
        if (n > 0x7fffffffLL) return fail(c, -5, std::string(who) + ": more than 2^31 records");
        all = (MirpAln*)std::malloc(std::max<size_t>((size_t)n, 1) * sizeof(MirpAln));
        segs = (MirpAln*)std::malloc(std::max<size_t>((size_t)ns, 1) * sizeof(MirpAln));
        owner.resize(std::max<size_t>((size_t)ns, 1)); span.resize(std::max<size_t>((size_t)ns, 1));
        if (!all || !segs) return bail(-6, "out of memory");
        concat(P, all, segs, owner.data(), span.data());
        if (c->alns.ensure(sizeof(MirpAln) * (size_t)std::max<long long>(n, 1)) || c->sort_tmp.ensure(sizeof(MirpAln) * (size_t)std::max<long long>(n, 1)) ||
            c->segs.ensure(sizeof(MirpAln) * (size_t)std::max<long long>(ns, 1)))
            return bail(-6, "device allocation failed (ingest)");
        if (n && hipMemcpyAsync(c->alns.p, all, sizeof(MirpAln) * (size_t)n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return bail(-2, "H2D failed");
        if (ns && hipMemcpyAsync(c->segs.p, segs, sizeof(MirpAln) * (size_t)ns, hipMemcpyHostToDevice, c->stream) != hipSuccess) return bail(-2, "H2D failed");
    } else {
        // ---- bucket by destination rank, per file: records, segments, and for every segment the index of its record inside the (file, destination) block
        const int F = n_paths;
        std::vector<std::vector<MirpAln>> brec((size_t)F * W), bseg((size_t)F * W);
        std::vector<std::vector<int32_t>> bown((size_t)F * W), bspan((size_t)F * W);
        std::vector<int32_t> newidx;
        for (int f = 0; f < F; f++)
            for (auto& ch : P.per_file[f]) {
                newidx.resize(ch.recs.size());
                for (size_t k = 0; k < ch.recs.size(); k++) {
                    auto& b = brec[(size_t)f * W + owner_of_tid[ch.recs[k].tid]];
                    newidx[k] = (int32_t)b.size();
                    b.push_back(ch.recs[k]);
                }
                for (size_t k = 0; k < ch.segs.size(); k++) {
                    const size_t q = (size_t)f * W + owner_of_tid[ch.segs[k].tid];
                    bseg[q].push_back(ch.segs[k]);
                    bown[q].push_back(newidx[(size_t)ch.seg_owner[k]]);
                    bspan[q].push_back(ch.seg_span[k]);
                }
                std::vector<MirpAln>().swap(ch.recs); std::vector<MirpAln>().swap(ch.segs); std::vector<int32_t>().swap(ch.seg_owner);
            }
        // counts of every (source, file, destination): cnt[(s * F + f) * W + q] records, then the same for segments
        std::vector<long long> mine((size_t)2 * F * W), cnt;
        for (int f = 0; f < F; f++)
            for (int q = 0; q < W; q++) { mine[(size_t)f * W + q] = (long long)brec[(size_t)f * W + q].size(); mine[(size_t)F * W + (size_t)f * W + q] = (long long)bseg[(size_t)f * W + q].size(); }
        if (int rc = mirp::dist_allgather_ll(c, mine.data(), 2 * F * W, cnt)) return rc;
        auto nrec = [&](int s, int f, int q) { return cnt[(size_t)s * 2 * F * W + (size_t)f * W + q]; };
        auto nseg = [&](int s, int f, int q) { return cnt[(size_t)s * 2 * F * W + (size_t)F * W + (size_t)f * W + q]; };
        // one blob per destination: [records of file 0..F-1][segments of file 0..F-1][owner indices of file 0..F-1][reference spans of file 0..F-1]
        auto blob_bytes = [&](int s, int q) { long long r = 0, g = 0; for (int f = 0; f < F; f++) { r += nrec(s, f, q); g += nseg(s, f, q); } return r * 16 + g * 16 + g * 8; };
        std::vector<long long> soff(W), scnt(W), roff(W), rcnt(W);
        long long stot = 0, rtot = 0;
        for (int q = 0; q < W; q++) { soff[q] = stot; scnt[q] = blob_bytes(me, q); stot += (scnt[q] + 15) & ~15LL; }
        for (int s2 = 0; s2 < W; s2++) { roff[s2] = rtot; rcnt[s2] = blob_bytes(s2, me); rtot += (rcnt[s2] + 15) & ~15LL; }
        for (int s2 = 0; s2 < W; s2++) for (int f = 0; f < F; f++) { n += nrec(s2, f, me); ns += nseg(s2, f, me); }
        // Everything that can fail on ONE rank between the count exchange above and the all-to-all below (record limit, allocations, the upload) is
        // collected in prep_rc instead of returning: the ranks agree on the outcome first, so that no rank leaves while its peers sit in the grouped
        // ncclSend / ncclRecv of the exchange (which has no timeout).
        int prep_rc = 0;
        if (n > 0x7fffffffLL) prep_rc = fail(c, -5, std::string(who) + ": more than 2^31 records on one rank");
        std::vector<char> sendbuf((size_t)std::max<long long>(stot, 16));
        for (int q = 0; q < W; q++) {
            char* w = sendbuf.data() + soff[q];
            for (int f = 0; f < F; f++) { auto& b = brec[(size_t)f * W + q]; if (!b.empty()) std::memcpy(w, b.data(), b.size() * 16); w += b.size() * 16; }
            for (int f = 0; f < F; f++) { auto& b = bseg[(size_t)f * W + q]; if (!b.empty()) std::memcpy(w, b.data(), b.size() * 16); w += b.size() * 16; }
            for (int f = 0; f < F; f++) { auto& b = bown[(size_t)f * W + q]; if (!b.empty()) std::memcpy(w, b.data(), b.size() * 4); w += b.size() * 4; }
            for (int f = 0; f < F; f++) { auto& b = bspan[(size_t)f * W + q]; if (!b.empty()) std::memcpy(w, b.data(), b.size() * 4); w += b.size() * 4; }
        }
        brec.clear(); bseg.clear(); bown.clear(); bspan.clear();
        TmpDevice T;
        char* d_send = nullptr; char* d_recv = nullptr;
        if (!prep_rc) {
            d_send = (char*)T.get((size_t)stot + 16);
            d_recv = (char*)T.get((size_t)rtot + 16);
            if (!d_send || !d_recv) prep_rc = fail(c, -6, "device allocation failed (ingest exchange)");
        }
        if (!prep_rc && (c->alns.ensure(sizeof(MirpAln) * (size_t)std::max<long long>(n, 1)) || c->sort_tmp.ensure(sizeof(MirpAln) * (size_t)std::max<long long>(n, 1)) ||
                         c->segs.ensure(sizeof(MirpAln) * (size_t)std::max<long long>(ns, 1))))
            prep_rc = fail(c, -6, "device allocation failed (ingest)");
        if (!prep_rc && stot) {
            hipError_t he = hipMemcpyAsync(d_send, sendbuf.data(), (size_t)stot, hipMemcpyHostToDevice, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) prep_rc = fail(c, -2, std::string(who) + ": upload of the exchange buffer failed: " + hipGetErrorString(he));
        }
        // the outcome is collective: one flag per rank, every rank returns an error if any rank has one
        if (int rc = mirp::dist_agree(c, prep_rc, who)) return rc;
        if (int rc = mirp::dist_alltoallv_bytes(c, d_send, soff, scnt, d_recv, roff, rcnt)) return rc;
        // unpack into (file, source) order; segment owners are re-based to the record's index in this rank's array
        owner.resize(std::max<size_t>((size_t)ns, 1)); span.resize(std::max<size_t>((size_t)ns, 1));
        std::vector<char> hrecv;        // owner indices come back to the host (tiny unless the input is mostly gapped)
        std::vector<long long> rbase((size_t)F * W), sbase((size_t)F * W);
        { long long a = 0, b = 0; for (int f = 0; f < F; f++) for (int s2 = 0; s2 < W; s2++) { rbase[(size_t)f * W + s2] = a; sbase[(size_t)f * W + s2] = b; a += nrec(s2, f, me); b += nseg(s2, f, me); } }
        std::vector<int32_t> own_tmp;
        for (int s2 = 0; s2 < W; s2++) {
            long long ro = roff[s2];
            long long rsum = 0, gsum = 0;
            for (int f = 0; f < F; f++) { rsum += nrec(s2, f, me); gsum += nseg(s2, f, me); }
            long long o_rec = ro, o_seg = ro + rsum * 16, o_own = ro + rsum * 16 + gsum * 16;
            if (gsum) {      // owner indices and spans of this source's segments (both int32, back to back)
                own_tmp.resize((size_t)gsum * 2);
                HIPCHK(c, hipMemcpyAsync(own_tmp.data(), d_recv + o_own, (size_t)gsum * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
            long long go = 0;
            for (int f = 0; f < F; f++) {
                const long long r = nrec(s2, f, me), g = nseg(s2, f, me);
                if (r) HIPCHK(c, hipMemcpyAsync((char*)c->alns.p + rbase[(size_t)f * W + s2] * 16, d_recv + o_rec, (size_t)r * 16, hipMemcpyDeviceToDevice, c->stream));
                if (g) HIPCHK(c, hipMemcpyAsync((char*)c->segs.p + sbase[(size_t)f * W + s2] * 16, d_recv + o_seg, (size_t)g * 16, hipMemcpyDeviceToDevice, c->stream));
                for (long long k = 0; k < g; k++) {
                    owner[(size_t)(sbase[(size_t)f * W + s2] + k)] = (int32_t)(rbase[(size_t)f * W + s2] + own_tmp[(size_t)(go + k)]);
                    span[(size_t)(sbase[(size_t)f * W + s2] + k)] = own_tmp[(size_t)(gsum + go + k)];
                }
                o_rec += r * 16; o_seg += g * 16; go += g;
            }
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        all = (MirpAln*)std::malloc(std::max<size_t>((size_t)n, 1) * sizeof(MirpAln));
        segs = (MirpAln*)std::malloc(std::max<size_t>((size_t)ns, 1) * sizeof(MirpAln));
        if (!all || !segs) return bail(-6, "out of memory");
    }
    // contig count / lengths decide the key width
    int64_t maxlen = 1;
    for (int64_t l : P.lens) maxlen = std::max(maxlen, l);
    int posbits = 1, tidbits = 1;
    while ((1LL << posbits) <= maxlen + 65536 && posbits < 31) posbits++;
    while ((1LL << tidbits) < (long long)P.names.size() && tidbits < 31) tidbits++;
    if (n_regions > 0 && keep_regions && n > 0) {
        // per-contig slices of the regions, sorted by start, with the running maximum of the ends
        std::vector<std::vector<std::pair<int, int>>> by(nc);
        for (int64_t k = 0; k < n_regions; k++)
            if (keep_regions[k].tid >= 0 && keep_regions[k].tid < nc && keep_regions[k].end > keep_regions[k].start)
                by[keep_regions[k].tid].push_back({keep_regions[k].start, keep_regions[k].end});
        std::vector<long long> rfirst(nc + 1, 0);
        std::vector<int> rs, rm;
        for (int t = 0; t < nc; t++) {
            std::stable_sort(by[t].begin(), by[t].end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
            rfirst[t] = (long long)rs.size();
            int mx = 0;
            for (auto& iv : by[t]) { mx = std::max(mx, iv.second); rs.push_back(iv.first); rm.push_back(mx); }
        }
        rfirst[nc] = (long long)rs.size();
        TmpDevice T;
        long long* d_rf = (long long*)T.get(8 * rfirst.size());
        int* d_rs = (int*)T.get(4 * std::max<size_t>(rs.size(), 1));
        int* d_rm = (int*)T.get(4 * std::max<size_t>(rm.size(), 1));
        int* d_own = (int*)T.get(4 * (size_t)std::max<long long>(ns, 1));
        int* d_span = (int*)T.get(4 * (size_t)std::max<long long>(ns, 1));
        MirpAln* d_segtmp = (MirpAln*)T.get(sizeof(MirpAln) * (size_t)std::max<long long>(ns, 1));
        if (!d_rf || !d_rs || !d_rm || !d_own || !d_span || !d_segtmp) return bail(-6, "device allocation failed (regions)");
        if (hipMemcpy(d_rf, rfirst.data(), 8 * rfirst.size(), hipMemcpyHostToDevice) != hipSuccess ||
            (!rs.empty() && (hipMemcpy(d_rs, rs.data(), 4 * rs.size(), hipMemcpyHostToDevice) != hipSuccess ||
                             hipMemcpy(d_rm, rm.data(), 4 * rm.size(), hipMemcpyHostToDevice) != hipSuccess)) ||
            (ns && (hipMemcpy(d_own, owner.data(), 4 * (size_t)ns, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(d_span, span.data(), 4 * (size_t)ns, hipMemcpyHostToDevice) != hipSuccess)))
            return bail(-2, "H2D failed");
        if (int rc = mirp_device_mask_alns(c, (MirpAln*)c->alns.p, (MirpAln*)c->sort_tmp.p, &n, (MirpAln*)c->segs.p, d_segtmp, d_own, d_span, &ns, d_rf, d_rs, d_rm)) {
            std::free(all); std::free(segs); mirp_free_sam_data(out); return rc;
        }
    }
    const double t2 = now_s();
    if (int rc = mirp_device_sort_alns(c, (MirpAln*)c->alns.p, (MirpAln*)c->sort_tmp.p, n, posbits, tidbits)) { std::free(all); std::free(segs); mirp_free_sam_data(out); return rc; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(-2, "device sort failed");
    const double t3 = now_s();
    if (n && hipMemcpy(all, c->alns.p, sizeof(MirpAln) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return bail(-2, "D2H failed");
    if (ns && hipMemcpy(segs, c->segs.p, sizeof(MirpAln) * (size_t)ns, hipMemcpyDeviceToHost) != hipSuccess) return bail(-2, "D2H failed");
    const double t4 = now_s();
    if (fill_meta(P, out)) return bail(-6, "out of memory");
    out->alns = all; out->n_alns = n; out->segs = segs; out->n_segs = ns;
    c->n_alns = n; c->n_segs = ns; c->max_aln_len = -1;
    c->have_candidate = c->have_fold = c->have_result = false;
    c->ingest_resident = true; c->ingest_n_contigs = (int)P.names.size();
    if (seconds) { seconds[0] = t1 - t0; seconds[1] = t2 - t1; seconds[2] = t3 - t2; seconds[3] = t4 - t3; }
    return 0;
}

extern "C" int mirp_ingest_sams_gpu(mirp_ctx* c, const char* const* paths, int32_t n_paths, int32_t n_threads, const MirpRegion* keep_regions,
                                    int64_t n_regions, MirpSamData* out, double seconds[4]) {
    return ingest_impl(c, paths, n_paths, n_threads, keep_regions, n_regions, false, nullptr, out, seconds, "mirp_ingest_sams_gpu");
}

// The host half of mirp_ingest_sams_gpu on its own, without a context: header + threaded tokenizer over every file.  A caller that starts it while the device is
// still being opened (the CLI: early.py) hands the result to mirp_ingest_tokenized_gpu, which does the device half (keep-region filter, stable radix sort) and
// releases it -- for a config[4] rank shard (2.65 GB of SAM text) that takes the 25-million-record host sort of mirp_ingest_sams off the path to the first stage.
extern "C" int mirp_tokenize_sams(const char* const* paths, int32_t n_paths, int32_t n_threads, void** tokenized, char* errbuf, size_t errbuf_len) {
    auto bail = [&](const std::string& m) { if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", m.c_str()); return -1; };
    if (!paths || n_paths < 1 || !tokenized) return bail("mirp_tokenize_sams: bad argument");
    if (n_paths > MIRP_MAX_SAMPLES) return bail("mirp_tokenize_sams: too many samples");
    *tokenized = nullptr;
    Parsed* P = new Parsed();
    std::string err;
    if (parse_all(paths, n_paths, n_threads, P, &err)) { delete P; return bail(err); }
    *tokenized = P;
    return 0;
}

extern "C" void mirp_free_tokenized(void* tokenized) { delete (Parsed*)tokenized; }

extern "C" int mirp_ingest_tokenized_gpu(mirp_ctx* c, void* tokenized, const MirpRegion* keep_regions, int64_t n_regions, MirpSamData* out, double seconds[4]) {
    if (!c) return -1;
    if (!tokenized) return fail(c, -1, "mirp_ingest_tokenized_gpu: bad argument");
    Parsed* P = (Parsed*)tokenized;
    const int rc = ingest_impl(c, nullptr, (int32_t)std::max<size_t>(P->per_file.size(), 1), 0, keep_regions, n_regions, false, nullptr, out, seconds, "mirp_ingest_tokenized_gpu", P);
    delete P;
    return rc;
}

extern "C" int mirp_ingest_sams_shard(mirp_ctx* c, const char* const* paths, int32_t n_paths, int32_t n_threads, const MirpRegion* keep_regions,
                                      int64_t n_regions, const int32_t* owner_of_tid, MirpSamData* out, double seconds[4]) {
    return ingest_impl(c, paths, n_paths, n_threads, keep_regions, n_regions, true, owner_of_tid, out, seconds, "mirp_ingest_sams_shard");
}

// ---- FASTA reader (the host side of `samtools faidx`, MP:1100-1105): name = first word of the header line, sequence lines joined with their
// surrounding white space stripped, bytes kept as they are (case preserved).  With a `want` list only those sequences are materialised (a rank of
// a sharded run keeps the contigs it owns); the others are skipped at memchr speed.
extern "C" void mirp_free_fasta_data(MirpFastaData* d) {
    if (!d) return;
    std::free(d->names); std::free(d->len); std::free(d->seq);
    std::memset(d, 0, sizeof(*d));
}

extern "C" int mirp_read_fasta(const char* path, const char* const* want, int32_t n_want, MirpFastaData* out, char* errbuf, size_t errbuf_len) {
    auto failf = [&](const std::string& m) {
        if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", m.c_str());
        return -1;
    };
    if (!path || !out || n_want < 0 || (n_want > 0 && !want)) return failf("mirp_read_fasta: bad argument");
    std::memset(out, 0, sizeof(*out));
    Mapped m;
    if (!m.open(path)) return failf(std::string("cannot open ") + path);
    NameTable wt;
    for (int k = 0; k < n_want; k++) wt.names.push_back(want[k]);
    if (n_want) wt.build();
    const char* b = m.p;
    const char* e = m.p + m.n;
    // pass 1: record boundaries ('>' at a line start)
    struct Rec { const char* name; size_t name_len; const char* body; const char* end; bool keep; };
    std::vector<Rec> recs;
    const char* s = b;
    while (s < e) {
        if (*s == '>') {
            const char* le = (const char*)memchr(s, '\n', (size_t)(e - s));
            if (!le) le = e;
            const char* nb = s + 1;
            const char* ne = nb;
            while (ne < le && *ne != ' ' && *ne != '\t' && *ne != '\r' && *ne != '\v' && *ne != '\f') ne++;
            if (!recs.empty()) recs.back().end = s;
            Rec r; r.name = nb; r.name_len = (size_t)(ne - nb); r.body = le < e ? le + 1 : e; r.end = e;
            r.keep = n_want == 0 || wt.find(nb, r.name_len) >= 0;
            recs.push_back(r);
            s = r.body;
        } else {      // next line start that begins with '>': memchr over '>' and check the byte before
            const char* g = (const char*)memchr(s, '>', (size_t)(e - s));
            while (g && g > b && g[-1] != '\n') g = (const char*)memchr(g + 1, '>', (size_t)(e - (g + 1)));
            s = g ? g : e;
        }
    }
    size_t cap = 0, nb = 0;
    for (auto& r : recs) { nb += r.name_len + 1; if (r.keep) cap += (size_t)(r.end - r.body); }
    out->names = (char*)std::malloc(std::max<size_t>(nb, 1));
    out->len = (int64_t*)std::malloc(std::max<size_t>(recs.size(), 1) * sizeof(int64_t));
    out->seq = (uint8_t*)std::malloc(std::max<size_t>(cap, 1));
    if (!out->names || !out->len || !out->seq) { mirp_free_fasta_data(out); return failf("out of memory"); }
    char* w = out->names;
    uint8_t* q = out->seq;
    for (size_t k = 0; k < recs.size(); k++) {
        const Rec& r = recs[k];
        std::memcpy(w, r.name, r.name_len); w[r.name_len] = 0; w += r.name_len + 1;
        int64_t L = -1;      // -1: skipped (not wanted)
        if (r.keep) {
            const uint8_t* q0 = q;
            const char* p = r.body;
            while (p < r.end) {
                const char* le = (const char*)memchr(p, '\n', (size_t)(r.end - p));
                if (!le) le = r.end;
                const char* a = p;
                const char* z = le;
                while (a < z && (*a == ' ' || *a == '\t' || *a == '\r' || *a == '\v' || *a == '\f')) a++;
                while (z > a && (z[-1] == ' ' || z[-1] == '\t' || z[-1] == '\r' || z[-1] == '\v' || z[-1] == '\f')) z--;
                if (z > a) { std::memcpy(q, a, (size_t)(z - a)); q += z - a; }
                p = le + 1;
            }
            L = (int64_t)(q - q0);
        }
        out->len[k] = L;
    }
    out->n_contigs = (int32_t)recs.size();
    out->n_bytes = (int64_t)(q - out->seq);
    return 0;
}
