"""Host-side ingest: FASTA genome and per-sample SAM files -> the packed, position-sorted alignment
record array the device kernels consume.  Replaces the reference's prepare-stage BAM plumbing
(sam2bam / cat / sort / index / expand / strand split, /root/reference/miR_PREFeR.py:656-746,
772-874) with an in-memory equivalent: the physical x-min(N,CUT) read expansion becomes a per-record
weight applied inside the coverage kernel.
"""
import gzip
import re

import numpy as np

from .synth import ALN_DTYPE

_DEPTH_RE = re.compile(r"^\S+_x([0-9]+)")


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path)


def read_fasta(path):
    """-> list of (name, uint8 ASCII array); name = first word of the header (as samtools faidx)."""
    contigs = []
    name, chunks = None, []
    with _open(path) as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    contigs.append((name, np.frombuffer("".join(chunks).encode(), dtype=np.uint8).copy()))
                name, chunks = line[1:].split()[0], []
            else:
                chunks.append(line.strip())
    if name is not None:
        contigs.append((name, np.frombuffer("".join(chunks).encode(), dtype=np.uint8).copy()))
    return contigs


def read_sam_header(path):
    """@SQ names and lengths in header order (get_length_from_sam, miR_PREFeR.py:500-509)."""
    names, lens = [], []
    with _open(path) as f:
        for line in f:
            if not line.startswith("@"):
                break
            if line.startswith("@SQ"):
                sp = line.split()
                names.append(sp[1].split(":", 1)[1])
                lens.append(int(sp[2].split(":", 1)[1]))
    return names, lens


_CIGAR_RE = re.compile(r"([0-9]+)([MIDNSHP=X])")


def cigar_segments(rec, cigar):
    """Coverage segments of a gapped alignment (see mirp_load_coverage_segments in include/mirprefer.h): `samtools depth` counts the
    M / = / X bases only.  rec = (tid, pos, depth, len(SEQ), strand, sample) -> list of such tuples, the first one
    with strand bit 1 set (takes the record's own [pos, pos + len(SEQ)) back out), then one per M / = / X block."""
    tid, pos, depth, rl, strand, sample = rec
    if cigar == "*" or "".join(a + b for a, b in _CIGAR_RE.findall(cigar)) != cigar:
        raise ValueError("malformed CIGAR %s" % cigar)
    out = [(tid, pos, depth, rl, strand | 2, sample)]
    # the bundled samtools 0.1.18 piles a read up only below its bam_calend() = pos + sum of the M, D and N lengths (= and X are not added):
    # blocks are cut there (probed against the binary, tests/golden/tools/gen_gapped_golden.py)
    calend = pos + sum(int(ln) for ln, op in _CIGAR_RE.findall(cigar) if op in "MDN")
    ref = pos
    for ln, op in _CIGAR_RE.findall(cigar):
        ln = int(ln)
        if op in "M=X":
            off, stop = ref, min(ref + ln, calend)
            while off < stop:
                part = min(stop - off, 65535)
                out.append((tid, off, depth, part, strand, sample))
                off += part
            ref += ln
        elif op in "DN":
            ref += ln
    return out


def read_sams(paths, native=True, regions=None, with_segments=False):
    """Uses the native multi-threaded parser of libmirprefer.so for plain-text SAM files; gzip-compressed files (test fixtures)
    go through the Python parser below, which implements the same rules.

    -> (contig_names, contig_lens, sample_names, alns[, segs]) with alns stably sorted by (tid, pos) over
    the sample-ordered concatenation (what `samtools cat` + `samtools sort` produce for the reference).  A record keeps POS and len(SEQ);
    alignments whose CIGAR is not `<len(SEQ)>M` also yield coverage segments (cigar_segments).  regions: keep regions
    [(tid, start0, end0)] applied like `samtools view -L` before the sort.  Unmapped reads are skipped."""
    if native and not any(str(p).endswith(".gz") for p in paths) and regions is None:
        from . import capi
        out = capi.ingest_sams(paths, with_segments=True)
        return out if with_segments else out[:4]
    names, lens = read_sam_header(paths[0])
    tid_of = {n: i for i, n in enumerate(names)}
    sample_names = []
    recs, segs, owner, span = [], [], [], []
    for si, p in enumerate(paths):
        sname = None
        with _open(p) as f:
            for line in f:
                if line.startswith("@"):
                    continue
                sp = line.rstrip("\r\n").split("\t")
                if sname is None:
                    sname = "_".join(sp[0].split("_")[0:-2])  # get_samplename_from_sam, miR_PREFeR.py:3300-3308
                flag = int(sp[1])
                if flag & 0x704:
                    continue
                m = _DEPTH_RE.match(sp[0])
                if not m:
                    raise ValueError('Read Id format is not right. Read id must be in "samplename_rA_xN" format.')
                cigar = sp[5]
                rl = len(sp[9])
                rec = (tid_of[sp[2]], int(sp[3]), int(m.group(1)), rl, 1 if flag & 16 else 0, si)
                if rec[1] > lens[rec[0]] + 1:      # same rule and message as the native tokenizer (mirp_ingest.cpp): refused on both ingest paths
                    raise ValueError("alignment position %d is beyond the end of sequence %s (LN:%d)" % (rec[1], sp[2], lens[rec[0]]))
                sp_ref = rl
                if cigar != "%dM" % rl:
                    for sg in cigar_segments(rec, cigar):
                        segs.append(sg); owner.append(len(recs))
                    sp_ref = sum(int(ln) for ln, op in _CIGAR_RE.findall(cigar) if op in "MDN")      # bam_calend - POS: what `samtools view -L` tests
                recs.append(rec); span.append(sp_ref)
        sample_names.append(sname)
    dt = [("tid", "<i4"), ("pos", "<i4"), ("depth", "<u4"), ("len", "<u2"), ("strand", "u1"), ("sample", "u1")]
    a = np.array(recs, dtype=dt).astype(ALN_DTYPE)
    sg = np.array(segs, dtype=dt).astype(ALN_DTYPE) if segs else np.zeros(0, dtype=ALN_DTYPE)
    if regions is not None:
        from . import gffmask
        keep = gffmask.keep_mask(a, regions, span=np.array(span, dtype=np.int64))
        if len(sg):
            sg = sg[keep[np.array(owner, dtype=np.int64)]]
        a = a[keep]
    key = a["tid"].astype(np.int64) << 32 | a["pos"].astype(np.int64)
    a = a[np.argsort(key, kind="stable")]
    out = (names, np.array(lens, dtype=np.int64), sample_names, a)
    return out + (sg,) if with_segments else out
