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


def read_sams(paths, native=True):
    """Uses the native multi-threaded parser of libmirprefer.so for plain-text SAM files; gzip-compressed files (test fixtures)
    go through the Python parser below, which implements the same rules.

    -> (contig_names, contig_lens, sample_names, alns) with alns stably sorted by (tid, pos) over
    the sample-ordered concatenation (what `samtools cat` + `samtools sort` produce for the reference).
    Only ungapped alignments (`<len>M`, as bowtie -v 0 emits) are accepted; unmapped reads are skipped."""
    if native and not any(str(p).endswith(".gz") for p in paths):
        from . import capi
        return capi.ingest_sams(paths)
    names, lens = read_sam_header(paths[0])
    tid_of = {n: i for i, n in enumerate(names)}
    sample_names = []
    recs = []
    for si, p in enumerate(paths):
        sname = None
        with _open(p) as f:
            for line in f:
                if line.startswith("@"):
                    continue
                sp = line.split("\t")
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
                if cigar != "%dM" % rl:
                    raise ValueError("only ungapped alignments (<len>M) are supported, got CIGAR %s" % cigar)
                recs.append((tid_of[sp[2]], int(sp[3]), int(m.group(1)), rl, 1 if flag & 16 else 0, si))
        sample_names.append(sname)
    a = np.array(recs, dtype=[("tid", "<i4"), ("pos", "<i4"), ("depth", "<u4"), ("len", "<u2"), ("strand", "u1"), ("sample", "u1")]).astype(ALN_DTYPE)
    key = a["tid"].astype(np.int64) << 32 | a["pos"].astype(np.int64)
    a = a[np.argsort(key, kind="stable")]
    return names, np.array(lens, dtype=np.int64), sample_names, a
