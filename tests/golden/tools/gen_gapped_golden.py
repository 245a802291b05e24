#!/usr/bin/env python3
"""Dev-only (build container): golden vector for gapped alignments.

A seeded SAM with CIGARs that contain I / D / N / S / H / = / X next to plain `<len>M` reads goes through the steps of the reference's
prepare + candidate stages that touch per-base depth, with the REAL bundled samtools 0.1.18: SAM -> BAM -> sort -> expand (every line
repeated min(N, CUT) times, expand_bamfile MP:716-746) -> strand split by flag 16 (MP:759-769, 861-873) ->
`samtools depth plus.bam minus.bam | awk '$3+$4>CUT'` (MP:937-941).  The text it prints is what bam.depth.cut<CUT> holds.
Output: tests/golden/gapped.json.gz = {sam, contigs, cutoff, depth_cut}"""
import gzip, json, os, random, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ORA_BIN = os.environ.get("MIRP_ORACLE_BIN", "/tmp/ora/bin")
GOLD = os.path.dirname(HERE)
ST = os.path.join(ORA_BIN, "samtools")


def cigar_and_seqlen(r):
    """random CIGAR: optional clips, M / = / X blocks separated by I / D / N"""
    ops = []
    if r.random() < 0.3:
        ops.append((r.randint(1, 4), r.choice("SH")))
    nblk = r.choice([1, 2, 2, 3])
    for b in range(nblk):
        ops.append((r.randint(3, 14), r.choice("MMM=X")))
        if b + 1 < nblk:
            ops.append((r.randint(1, 6), r.choice("IDDN")))
    if r.random() < 0.3:
        ops.append((r.randint(1, 4), "S"))
    seqlen = sum(n for n, o in ops if o in "MIS=X")
    return "".join("%d%s" % x for x in ops), seqlen


def main():
    r = random.Random(23)
    contigs = [("ctgB", 3000), ("ctgA", 2000)]
    cut = 10
    sam = ["@HD\tVN:1.0\tSO:unsorted"] + ["@SQ\tSN:%s\tLN:%d" % c for c in contigs]
    k = 0
    for c, L in contigs:
        for centre in range(150, L - 150, 260):            # read stacks deep enough to pass the threshold, gapped and plain mixed
            for _ in range(r.randint(4, 9)):
                pos = centre + r.randint(-12, 12)
                flag = r.choice([0, 0, 16])
                if r.random() < 0.55:
                    cg, sl = cigar_and_seqlen(r)
                else:
                    sl = r.randint(18, 25); cg = "%dM" % sl
                sam.append("\t".join(["S1_r%d_x%d" % (k, r.choice([1, 3, 8, 14, 40])), str(flag), c, str(pos), "255", cg, "*", "0", "0", "A" * sl, "I" * sl]))
                k += 1
    text = "\n".join(sam) + "\n"
    with tempfile.TemporaryDirectory() as tmp:
        p = lambda n: os.path.join(tmp, n)
        open(p("in.sam"), "w").write(text)
        subprocess.check_call([ST, "view", "-bS", "-o", p("in.bam"), p("in.sam")], stderr=subprocess.DEVNULL)
        subprocess.check_call([ST, "sort", p("in.bam"), p("sorted")], stderr=subprocess.DEVNULL)
        body = subprocess.run([ST, "view", "-h", p("sorted.bam")], capture_output=True, text=True, check=True).stdout
        with open(p("exp.sam"), "w") as f:                   # expand_bamfile
            for line in body.splitlines(True):
                if line.startswith("@"):
                    f.write(line); continue
                n = min(int(line.split()[0].split("_")[-1].lstrip("x")), cut)
                f.write(line * n)
        subprocess.check_call([ST, "view", "-bS", "-o", p("exp.bam"), p("exp.sam")], stderr=subprocess.DEVNULL)
        subprocess.check_call([ST, "view", "-b", "-F", "16", "-o", p("plus.bam"), p("exp.bam")])
        subprocess.check_call([ST, "view", "-b", "-f", "16", "-o", p("minus.bam"), p("exp.bam")])
        depth = subprocess.run("%s depth %s %s | awk '$3+$4>%d'" % (ST, p("plus.bam"), p("minus.bam"), cut), shell=True, capture_output=True, text=True, check=True).stdout
        # `samtools view -L <bed>` on the gapped reads (the GFF masking of prepare_data, MP:817-859): 0.1.18 tests [POS-1, bam_calend) -- the M / D / N
        # span, not len(SEQ) -- so a region inside an intron keeps the read and a soft-clipped read just before a region does not reach it.
        # Regions: seeded ones plus, for a sample of gapped reads, a region just past POS-1+len(SEQ) or inside the first D / N gap.
        import re
        regions = []
        recs = [l.split("\t") for l in text.splitlines() if not l.startswith("@")]
        for f in recs[::7]:
            pos0, cg, sl = int(f[3]) - 1, f[5], len(f[9])
            span = sum(int(n) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", cg) if o in "MDN")
            if cg != "%dM" % sl and span != sl:
                lo, hi = sorted((pos0 + span, pos0 + sl))
                regions.append((f[2], lo, hi))                               # between the reference span and the SEQ length: only one of the two rules sees it
        for _ in range(12):
            c, L = r.choice(contigs)
            a = r.randint(0, L - 40)
            regions.append((c, a, a + r.randint(1, 30)))
        open(p("k.bed"), "w").write("".join("%s\t%d\t%d\n" % x for x in regions))
        kept = subprocess.run([ST, "view", "-L", p("k.bed"), p("in.bam")], capture_output=True, text=True, check=True).stdout
        kept_ids = sorted(l.split("\t")[0] for l in kept.splitlines())
    path = os.path.join(GOLD, "gapped.json.gz")
    with gzip.open(path, "wt", compresslevel=9) as f:
        json.dump({"generator": "bundled samtools 0.1.18: view -bS, sort, expand (x min(N, CUT)), strand split, depth | awk (miR_PREFeR.py:716-746, 759-769, 937-941)",
                   "sam": text, "contigs": [list(c) for c in contigs], "cutoff": cut, "depth_cut": depth,
                   "view_L": {"bed": [list(x) for x in regions], "kept_ids": kept_ids}}, f)
    print("wrote", path, os.path.getsize(path), len(sam), "SAM lines,", len(depth.splitlines()), "depth lines,", len(kept_ids), "of", len(recs), "reads kept by view -L")


if __name__ == "__main__":
    main()
