"""Seeded synthetic inputs for the candidate -> fold -> predict path:
a uniform-random genome with planted imperfect hairpins (miRNA-like loci with mature / star /
isoform reads) and siRNA-like read clusters, plus writers for the reference's input formats
(FASTA + one SAM per sample, read ids `sample_rA_xN`, /root/reference/README.md "Prepare input
data").  Real genomes / sRNA-seq SAMs are unavailable offline, so every benchmark and fixture in
this repository is built from this generator.
"""
import os

import numpy as np

ALN_DTYPE = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("depth", "<u4"), ("len", "<u2"), ("strand", "u1"), ("sample", "u1")])
assert ALN_DTYPE.itemsize == 16

_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    _COMP[_a] = _b
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _revcomp(a):
    return _COMP[a[::-1]]


class Dataset:
    """genome: list of (name, uint8 ASCII array); alns: ALN_DTYPE array in generation (unsorted) order,
    `sample` indexes sample_names; read ids are assigned per sample in array order."""

    def __init__(self, contigs, sample_names, alns, planted):
        self.contigs = contigs
        self.sample_names = sample_names
        self.alns = alns
        self.planted = planted

    @property
    def contig_names(self):
        return [n for n, _ in self.contigs]

    @property
    def contig_lens(self):
        return np.array([len(s) for _, s in self.contigs], dtype=np.int64)

    def sorted_alns(self):
        """Stable sort by (tid, pos) of the sample-ordered concatenation: the order `samtools cat` +
        `samtools sort` gives the reference's combined BAM (miR_PREFeR.py:667-713, 807-859)."""
        a = self.alns
        order = np.lexsort((np.arange(len(a)), a["sample"]))  # sample-major, file order inside
        a = a[order]
        key = a["tid"].astype(np.int64) << 32 | a["pos"].astype(np.int64)
        return a[np.argsort(key, kind="stable")]

    def write_fasta(self, path, width=60):
        with open(path, "wb") as f:
            for name, seq in self.contigs:
                f.write(b">" + name.encode() + b"\n")
                n = len(seq)
                full = (n // width) * width
                if full:
                    body = seq[:full].reshape(-1, width)
                    out = np.empty((body.shape[0], width + 1), dtype=np.uint8)
                    out[:, :width] = body
                    out[:, width] = 10
                    f.write(out.tobytes())
                if n > full:
                    f.write(seq[full:].tobytes() + b"\n")

    def write_sams(self, outdir, sq_order=None):
        """One SAM per sample (unsorted, generation order), header @SQ lines in `sq_order`."""
        names = self.contig_names
        lens = self.contig_lens
        order = list(range(len(names))) if sq_order is None else list(sq_order)
        paths = []
        for si, sname in enumerate(self.sample_names):
            p = os.path.join(outdir, sname + ".sam")
            sel = self.alns[self.alns["sample"] == si]
            with open(p, "w") as f:
                f.write("@HD\tVN:1.0\tSO:unsorted\n")
                for t in order:
                    f.write("@SQ\tSN:%s\tLN:%d\n" % (names[t], lens[t]))
                for k, r in enumerate(sel):
                    seq = self.contigs[r["tid"]][1][r["pos"] - 1:r["pos"] - 1 + r["len"]].tobytes().decode().upper()
                    f.write("%s_r%d_x%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:0\n" % (
                        sname, k, r["depth"], 16 if r["strand"] else 0, names[r["tid"]], r["pos"], r["len"], seq, "I" * int(r["len"])))
            paths.append(p)
        return paths


def make_dataset(contig_lens, n_loci, n_samples=1, seed=2, contig_names=None, hairpin_frac=0.4, edge_cases=False):
    rng = np.random.RandomState(seed)
    ncont = len(contig_lens)
    names = contig_names or ["Chr%d" % (i + 1) for i in range(ncont)]
    contigs = [rng.randint(0, 4, size=int(L)).astype(np.uint8) for L in contig_lens]
    contigs = [_BASES[c] for c in contigs]
    sample_names = ["S%d" % (i + 1) for i in range(n_samples)]
    recs = []  # (tid,pos,depth,len,strand,sample)
    planted = []

    def add_read(tid, gstart0, length, strand, depth):
        # gstart0: 0-based genome start of the read's forward-strand footprint
        if gstart0 < 0 or gstart0 + length > len(contigs[tid]) or length < 1:
            return
        if n_samples == 1:
            recs.append((tid, gstart0 + 1, int(depth), length, strand, 0))
            return
        for s in range(n_samples):
            d = int(round(depth * rng.uniform(0.3, 1.5)))
            if d > 0:
                recs.append((tid, gstart0 + 1, d, length, strand, s))

    tot = float(sum(contig_lens))
    per = [max(0, int(round(n_loci * L / tot))) for L in contig_lens]
    for tid in range(ncont):
        L = int(contig_lens[tid])
        k = per[tid]
        if k == 0:
            continue
        slot = L // k
        if slot < 800:
            k = max(1, L // 800)
            slot = L // k
        for li in range(k):
            g0 = li * slot + int(rng.randint(0, max(1, slot - 700))) + 150
            if g0 + 450 > L:
                continue
            if rng.rand() < hairpin_frac:
                arm = int(rng.randint(24, 35))
                loop = int(rng.randint(6, 41))
                a = _BASES[rng.randint(0, 4, size=arm)]
                b = _revcomp(a).copy()
                for _ in range(int(rng.randint(0, 5))):
                    b[rng.randint(0, arm)] = _BASES[rng.randint(0, 4)]
                hp = np.concatenate([a, _BASES[rng.randint(0, 4, size=loop)], b])
                H = len(hp)
                strand = int(rng.randint(0, 2))
                contigs[tid][g0:g0 + H] = hp if strand == 0 else _revcomp(hp)
                ml = int(rng.randint(20, 25))
                ms = int(rng.randint(2, 6))
                if rng.rand() < 0.5:  # mature on the 5' arm
                    m0, m1 = ms, ms + ml
                    s0, s1 = H - ms - ml + 2, H - ms + 2
                else:                  # mature on the 3' arm
                    m1 = H - ms + 2
                    m0 = m1 - ml
                    s0, s1 = ms, ms + ml
                    if m1 > H:
                        m0, m1 = m0 - (m1 - H), H

                def tr(x0, x1):  # transcript [x0,x1) -> 0-based genome start
                    return (g0 + x0) if strand == 0 else (g0 + H - x1)

                md = int(rng.randint(20, 501))
                add_read(tid, tr(m0, m1), m1 - m0, strand, md)
                for _ in range(2):
                    dx = int(rng.randint(-1, 2)); dl = int(rng.randint(-1, 2))
                    if dx == 0 and dl == 0:
                        dl = 1
                    add_read(tid, tr(m0 + dx, m1 + dx + dl), m1 - m0 + dl, strand, max(1, md // int(rng.randint(5, 20))))
                sd = int(rng.randint(0, 41))
                if sd > 0 and s0 >= 0 and s1 <= H:
                    add_read(tid, tr(s0, s1), s1 - s0, strand, sd)
                planted.append((tid, g0 + 1, g0 + H + 1, "+-"[strand]))
            else:
                nr = int(rng.randint(2, 13))
                width = int(rng.randint(30, 200))
                for _ in range(nr):
                    rl = int(rng.randint(18, 26))
                    add_read(tid, g0 + int(rng.randint(0, width)), rl, int(rng.randint(0, 2)), int(rng.randint(1, 61)))
    if edge_cases:
        _plant_edge_cases(rng, contigs, add_read)
    alns = np.array(recs, dtype=[("tid", "<i4"), ("pos", "<i4"), ("depth", "<u4"), ("len", "<u2"), ("strand", "u1"), ("sample", "u1")])
    alns = alns.astype(ALN_DTYPE)
    return Dataset(list(zip(names, contigs)), sample_names, alns, planted)


def _plant_edge_cases(rng, contigs, add_read):
    """Cases the reference handles specially: contig-edge loci, a < 60-nt locus,
    a both-strand locus, a > PRECURSOR_LEN locus, a > PRECURSOR_LEN+50 locus, soft-masked and N stretches."""
    tid = 0
    L = len(contigs[tid])
    # locus hard at the contig start and at the contig end
    add_read(tid, 2, 21, 0, 60); add_read(tid, 5, 22, 0, 30)
    add_read(tid, L - 25, 22, 1, 80); add_read(tid, L - 30, 21, 1, 15)
    t2 = len(contigs) - 1
    L2 = len(contigs[t2])
    base = L2 // 2
    # short (< 60 nt) locus with peaks on both strands -> L/R windows on both strands (Appendix A-4)
    add_read(t2, base, 21, 0, 50); add_read(t2, base + 30, 21, 1, 50)
    # 320-nt locus (PRECURSOR_LEN < len <= PRECURSOR_LEN+50): folded as is
    for x in range(0, 300, 15):
        add_read(t2, base + 3000 + x, 24, 0, 25)
    # 420-nt locus: dropped
    for x in range(0, 400, 15):
        add_read(t2, base + 6000 + x, 24, 1, 25)
    # soft-masked stretch and an N run under reads on the minus strand
    seg = contigs[t2][base + 9000:base + 9100]
    contigs[t2][base + 9000:base + 9100] = np.where(seg < 97, seg + 32, seg)
    contigs[t2][base + 9150:base + 9156] = ord("N")
    add_read(t2, base + 9020, 22, 1, 90); add_read(t2, base + 9060, 21, 1, 40)
    add_read(t2, base + 9140, 24, 0, 70)


def write_sam_fast(path, sample_name, recs, contig_names, contig_lens, cigars=None):
    """Vectorised SAM writer for large record arrays (bench / scale tests): one line per record of `recs` (ALN_DTYPE, any order), read ids
    `<sample>_r<k>_x<depth>` with zero-padded numbers (fixed-width lines per read length, so a whole group is one numpy byte matrix),
    sequence = poly-A (the ingest keeps coordinates, not bases).  Returns the number of bytes written."""
    names = [n.encode() for n in contig_names]
    total = 0
    with open(path, "wb") as f:
        f.write(b"@HD\tVN:1.0\tSO:unsorted\n")
        for n, l in zip(names, contig_lens):
            f.write(b"@SQ\tSN:" + n + b"\tLN:%d\n" % int(l))
        order = np.argsort(recs["len"], kind="stable")
        r = recs[order]
        idx = order.astype(np.int64)
        bounds = np.flatnonzero(np.diff(r["len"].astype(np.int64))) + 1
        for a, b in zip(np.concatenate([[0], bounds]), np.concatenate([bounds, [len(r)]])):
            g = r[a:b]
            if len(g) == 0:
                continue
            rl = int(g["len"][0])
            for t in np.unique(g["tid"]):
                gt = g[g["tid"] == t]
                it = idx[a:b][g["tid"] == t]
                tpl = b"%s_r%010d_x%010d\t%02d\t%s\t%010d\t255\t%dM\t*\t0\t0\t%s\t%s\n" % (sample_name.encode(), 0, 0, 0, names[int(t)], 0, rl, b"A" * rl, b"I" * rl)
                m = np.tile(np.frombuffer(tpl, dtype=np.uint8), (len(gt), 1))
                o_id = len(sample_name) + 2
                o_dp = o_id + 10 + 2
                o_fl = o_dp + 10 + 1
                o_pos = o_fl + 2 + 1 + len(names[int(t)]) + 1

                def digits(col, vals, width):
                    v = vals.astype(np.int64)
                    for k in range(width):
                        m[:, col + width - 1 - k] = 48 + (v % 10)
                        v //= 10
                digits(o_id, it, 10)
                digits(o_dp, gt["depth"], 10)
                digits(o_fl, np.where(gt["strand"] != 0, 16, 0), 2)
                digits(o_pos, gt["pos"], 10)
                f.write(m.tobytes())
                total += m.size
    return total


def packed_records_shard(n_contigs, contig_len, n_loci, reads_per_locus, seed):
    """config[4]: alignment records generated directly as packed 16-byte records: clusters of distinct isomiR-like reads."""
    rng = np.random.RandomState(seed)
    per = n_loci // n_contigs
    tid = np.repeat(np.arange(n_contigs, dtype=np.int32), per)
    slot = contig_len // per
    start = (np.tile(np.arange(per, dtype=np.int64), n_contigs) * slot + rng.randint(100, slot - 500, size=n_contigs * per)).astype(np.int64)
    strand = rng.randint(0, 2, size=len(start)).astype(np.uint8)
    width = rng.randint(8, 90, size=len(start))
    nl = len(start)
    a = np.zeros(nl * reads_per_locus, dtype=ALN_DTYPE)
    a["tid"] = np.repeat(tid, reads_per_locus)
    a["pos"] = (np.repeat(start, reads_per_locus) + (rng.randint(0, 1 << 30, size=len(a)) % np.repeat(width, reads_per_locus))).astype(np.int32) + 1
    a["len"] = rng.randint(18, 26, size=len(a)).astype(np.uint16)
    a["depth"] = rng.randint(1, 40, size=len(a)).astype(np.uint32)
    flip = rng.rand(len(a)) < 0.04
    a["strand"] = np.repeat(strand, reads_per_locus) ^ flip.astype(np.uint8)
    a["sample"] = 0
    key = a["tid"].astype(np.int64) << 32 | a["pos"].astype(np.int64)
    return a[np.argsort(key, kind="stable")]
