"""GFF include / exclude masking of the alignment records: the reference turns the GFF file into a BED file of regions to
keep and filters the combined BAM with `samtools view -L` before anything else runs (/root/reference/miR_PREFeR.py: gen_temp_gff :524-540,
gen_keep_regions_from_exclude_gff :543-610, gen_keep_regions_from_include_gff :613-652, prepare_data :817-859).  Here the same keep
regions are computed in memory and applied as an interval mask on the packed record array.

The BED construction keeps the reference's quirks: contigs without any feature contribute no keep region (all their reads are dropped),
the region in front of a contig's first feature ends at the feature's 1-based start (so it includes the feature's first base), gaps use
`end - 1` of the previous merged region as 0-based start, and only gaps / features of at least `minlen` (55) are kept."""
import numpy as np


def _gff_lines(path):
    """gen_temp_gff: data lines up to a ##FASTA directive, then `sort -k1,1 -k4,4n` (byte order, whole line as the last-resort key)."""
    lines = []
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                if line.startswith("##FASTA"):
                    break
                continue
            if not line.strip():
                continue
            lines.append(line if line.endswith("\n") else line + "\n")

    def key(l):
        sp = l.split()
        try:
            k4 = float(sp[3])
        except (IndexError, ValueError):
            k4 = 0.0
        return (sp[0].encode(), k4, l.encode())
    return sorted(lines, key=key)


def keep_regions_exclude(path, dict_len, minlen=55):
    """-> list of (seqid, start0, end0) BED intervals: everything not covered by the (merged) features. dict_len: name -> length."""
    out = []
    seqid, region = "NOTKNOW", None
    for line in _gff_lines(path):
        sp = line.split()
        if sp[0] not in dict_len:
            continue
        cur = sp[0]
        s, e = int(sp[3]), int(sp[4])
        if cur != seqid:
            if seqid != "NOTKNOW" and dict_len[seqid] - region[1] >= minlen:
                out.append((seqid, region[1] - 1, int(dict_len[seqid])))
            if s >= minlen:
                out.append((cur, 0, s))
            seqid, region = cur, (s, e)
        else:
            if region[1] < s or region[0] > e:          # no overlap with the running region
                if s - region[1] >= minlen:
                    out.append((cur, region[1] - 1, s))
                region = (s, e)
            else:
                region = (min(region[0], s), max(region[1], e))
    if seqid == "NOTKNOW":
        return out
    if dict_len[seqid] - region[1] >= minlen:
        out.append((seqid, region[1] - 1, int(dict_len[seqid])))
    return out


def keep_regions_include(path, minlen=55):
    """-> list of (seqid, start0, end0): the features themselves (end - start >= minlen)."""
    out = []
    for line in _gff_lines(path):
        sp = line.split()
        if int(sp[4]) - int(sp[3]) >= minlen:
            out.append((sp[0], int(sp[3]) - 1, int(sp[4])))
    return out


def keep_mask(alns, regions, span=None):
    """`samtools view -L bed` as a boolean mask: True for the records that overlap any keep region.  regions: [(tid, start0, end0)] 0-based
    half-open; alns: ALN_DTYPE in any order, pos 1-based.  The bundled samtools 0.1.18 tests [POS - 1, bam_calend): span = the reference span
    (sum of the M / D / N lengths) per record; None = len(SEQ), which is the same thing for ungapped alignments."""
    keep = np.zeros(len(alns), dtype=bool)
    a0 = alns["pos"].astype(np.int64) - 1
    a1 = a0 + (alns["len"].astype(np.int64) if span is None else np.asarray(span, dtype=np.int64))
    by_tid = {}
    for t, s, e in regions:
        if e > s:
            by_tid.setdefault(int(t), []).append((s, e))
    for t, ivs in by_tid.items():
        idx = np.nonzero(alns["tid"] == t)[0]
        if not len(idx):
            continue
        s = np.array([x[0] for x in ivs], dtype=np.int64)
        e = np.array([x[1] for x in ivs], dtype=np.int64)
        o = np.argsort(s, kind="stable")
        s, e = s[o], e[o]
        emax = np.maximum.accumulate(e)                 # intervals may overlap: running maximum of the ends
        # a record [b0, b1) overlaps some interval iff among the intervals with start < b1 the largest end exceeds b0
        k = np.searchsorted(s, a1[idx], side="left")    # number of intervals with start < b1
        hit = k > 0
        hit[hit] = emax[k[hit] - 1] > a0[idx][hit]
        keep[idx] = hit
    return keep


def regions_by_tid(regions, names):
    """(seqid, start0, end0) BED regions -> (tid, start0, end0) for the contigs of the SAM header (unknown sequence ids drop out)."""
    tid_of = {n: t for t, n in enumerate(names)}
    return [(tid_of[seqid], int(s), int(e)) for seqid, s, e in regions if seqid in tid_of]


def apply_keep(alns, names, regions):
    """`samtools view -L bed`: keep the records that overlap any BED interval (0-based, half-open).  alns: ALN_DTYPE, pos 1-based."""
    return alns[keep_mask(alns, regions_by_tid(regions, names))]
