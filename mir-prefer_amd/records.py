"""Record layouts shared by the ctypes C-ABI (include/mirprefer.h) and the host stage drivers, plus the
text renderings the reference writes into its stage artefacts (depth file, FASTA headers)."""
import numpy as np

DEPTH_DTYPE = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("dp", "<i4"), ("dm", "<i4")])
PEAK_DTYPE = np.dtype([("tid", "<i4"), ("start", "<i4"), ("end", "<i4"), ("strand", "<i4")])
MATURE_DTYPE = np.dtype([("start", "<i4"), ("end", "<i4"), ("strand", "<i4"), ("depth", "<i4")])
WINDOW_DTYPE = np.dtype([("tid", "<i4"), ("ws", "<i4"), ("we", "<i4"), ("strand", "<i4"), ("loc_s", "<i4"), ("loc_e", "<i4"),
                         ("tag", "<i4"), ("n_peaks", "<i4"), ("peak_off", "<i8"), ("n_matures", "<i4"), ("_pad0", "<i4"),
                         ("mature_off", "<i8"), ("seq_off", "<i8"), ("seq_len", "<i4"), ("_pad1", "<i4")], align=False)
LOCUS_DTYPE = np.dtype([("tid", "<i4"), ("start", "<i4"), ("end", "<i4"), ("n_windows", "<i4"), ("w", "<i4", (2, 2)),
                        ("peak_first", "<i8"), ("n_peaks", "<i4"), ("_pad", "<i4")])

MAX_MIRNA_PER_WINDOW = 8
MIRNA_DTYPE = np.dtype([(n, "<i4") for n in ("window", "tid", "fold_s", "fold_e", "mat_s", "mat_e", "star_s", "star_e", "strand", "has_star",
                                             "line", "ss_off", "ss_len", "reserved", "total_depth_mature", "total_depth_star")])

STRAND = "+-"
TAG = "0LR"


def depth_text(depth, contig_names):
    """bam.depth.cut<CUT> text: `chr\\tpos\\td+\\td-` (miR_PREFeR.py:937-949)."""
    if len(depth) == 0:
        return ""
    names = np.array(contig_names, dtype=object)[depth["tid"]]
    cols = [names, depth["pos"].astype(str).astype(object), depth["dp"].astype(str).astype(object), depth["dm"].astype(str).astype(object)]
    lines = cols[0] + "\t" + cols[1] + "\t" + cols[2] + "\t" + cols[3]
    return "\n".join(lines.tolist()) + "\n"


def peaks_to_dict(peaks, contig_names):
    """dict_contigs of gen_contig_typeA (miR_PREFeR.py:952-962)."""
    d = {}
    for p in peaks:
        d.setdefault(contig_names[p["tid"]], []).append((int(p["start"]), int(p["end"]), STRAND[p["strand"]]))
    return d


def loci_to_dict(loci, peaks, contig_names, precursor_len=300):
    """dict_loci of gen_candidate_region_typeA (miR_PREFeR.py:1302-1316).  The record arrays are turned into Python lists once (field access on a
    numpy record costs about a microsecond, and there are a dozen per locus)."""
    d = {}
    if len(loci) == 0:
        return d
    allpk = list(zip(peaks["start"].tolist(), peaks["end"].tolist(), [STRAND[x] for x in peaks["strand"].tolist()]))
    tid, start, end = loci["tid"].tolist(), loci["start"].tolist(), loci["end"].tolist()
    nwin, wins = loci["n_windows"].tolist(), loci["w"].tolist()
    first, npk = loci["peak_first"].tolist(), loci["n_peaks"].tolist()
    for k in range(len(tid)):
        pk = allpk[first[k]:first[k] + npk[k]]
        # a single-peak region is the peak tuple itself, strand included (r_now = contiglist[0], miR_PREFeR.py:1259)
        region = pk[0] if len(pk) == 1 else (start[k], end[k])
        info = [region]
        long_region = end[k] - start[k] > precursor_len          # extend_region returns [region] itself (miR_PREFeR.py:1276-1277)
        w = wins[k]
        for x in range(nwin[k]):
            info.append((region if long_region else (w[x][0], w[x][1]), pk))
        name = contig_names[tid[k]]
        if name in d:
            d[name].append(info)
        else:
            d[name] = [info]
    return d


def exregion_gff_text(dict_loci):
    """`<prefix>_ExRegionA.gff3` (miR_PREFeR.py:1357-1369) from dict_loci: one line per extended region, `Other=` = the peaks of the locus
    as `start:end:strand|`, names numbered per contig in sorted contig order."""
    out = []
    for seqid in sorted(dict_loci):
        cnt = 0
        head = seqid + "\tmiR-PREFeR\tExRegionA\t"
        for info in dict_loci[seqid]:
            other = None
            for win, peaks in info[1:]:          # a single-peak region longer than PRECURSOR_LEN is the peak tuple itself (start, end, strand)
                if other is None:
                    other = "".join(["%d:%d:%s|" % p for p in peaks])
                out.append("%s%d\t%d\t.\t+\t.\tID=ExRegionA_%d;NAME=ExRegionA_%d;Other=%s" % (head, win[0], win[1], cnt, cnt, other))
                cnt += 1
    return "\n".join(out) + "\n" if out else ""


def mature_tuple(m):
    if m["strand"] < 0:
        return (0, 0, 0, 0)
    return (int(m["start"]), int(m["end"]), STRAND[m["strand"]], int(m["depth"]))


def fasta_header(w, wpeaks, matures, contig_names):
    """`>chr:ws-we strand locS-locE tag peaks M:..` (miR_PREFeR.py:1124-1140, 1162-1178)."""
    pk = wpeaks[w["peak_off"]:w["peak_off"] + w["n_peaks"]]
    other = ";".join("%d,%d,%s" % (p["start"], p["end"], STRAND[p["strand"]]) for p in pk)
    h = ">%s:%d-%d %s %d-%d %s %s" % (contig_names[w["tid"]], w["ws"], w["we"], STRAND[w["strand"]], w["loc_s"], w["loc_e"], TAG[w["tag"]], other)
    for m in matures[w["mature_off"]:w["mature_off"] + w["n_matures"]]:
        t = mature_tuple(m)
        h += " M:%d-%d/%s/%d" % (t[0], t[1], t[2], t[3])
    return h
