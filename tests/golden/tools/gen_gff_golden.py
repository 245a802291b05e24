#!/usr/bin/env python3
"""Dev-only (build container): golden vectors for the GFF include / exclude masking.

The reference's own GFF path cannot run end to end here (its `samtools sort in.bam -o out.bam` call, MP:656-665, is not valid for the
bundled samtools 0.1.18), so the two halves are pinned separately against the real thing:
  * the keep-region BED text of the reference's own functions (gen_keep_regions_from_exclude_gff / _include_gff, imported through the
    py3 shim and run on seeded synthetic GFF files);
  * which alignments `samtools view -L <bed>` of the bundled samtools keeps, on a seeded SAM.
Output: tests/golden/gffmask.json.gz"""
import gzip, json, os, random, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import ref_shim  # noqa: E402

ORA_BIN = os.environ.get("MIRP_ORACLE_BIN", "/tmp/ora/bin")
GOLD = os.path.dirname(HERE)


def synth_gff(r, contigs, n_feat, with_header=True):
    lines = []
    if with_header:
        lines.append("##gff-version 3\n")
    for _ in range(n_feat):
        c, L = r.choice(contigs)
        s = r.randint(1, L - 10)
        e = min(L, s + r.choice([5, 30, 54, 55, 56, 120, 400, 1500]))
        lines.append("\t".join([c, "src", r.choice(["gene", "CDS", "exon"]), str(s), str(e), ".", r.choice("+-"), ".", "ID=f%d" % len(lines)]) + "\n")
    if r.random() < 0.5:
        lines.insert(r.randrange(len(lines)), "# a comment\n")
        lines.insert(r.randrange(len(lines)), "\n")
    if r.random() < 0.5:
        lines += ["##FASTA\n", ">x\n", "ACGT\n", "ctgA\tsrc\tgene\t1\t900\t.\t+\t.\tID=after_fasta\n"]
    return "".join(lines)


def main():
    g = ref_shim.load_reference()
    r = random.Random(17)
    cases = []
    contigs = [("ctgB", 5000), ("ctgA", 3000), ("ctg10", 8000), ("unused", 2000)]
    for k in range(12):
        with tempfile.TemporaryDirectory() as tmp:
            use = contigs[:3] if k % 3 else contigs[:2]
            text = synth_gff(r, use, r.choice([1, 3, 8, 25]), with_header=k % 2 == 0)
            p = os.path.join(tmp, "in.gff")
            open(p, "w").write(text)
            dict_len = dict(contigs)
            bed_ex = open(g["gen_keep_regions_from_exclude_gff"](p, tmp, dict_len, 55)).read()
            bed_in = open(g["gen_keep_regions_from_include_gff"](p, tmp, 55)).read()
            cases.append({"gff": text, "dict_len": [[n, l] for n, l in contigs], "bed_exclude": bed_ex, "bed_include": bed_in})
    # samtools view -L on a seeded SAM
    views = []
    for k in range(4):
        with tempfile.TemporaryDirectory() as tmp:
            sam = ["@HD\tVN:1.0\tSO:unsorted"] + ["@SQ\tSN:%s\tLN:%d" % (n, l) for n, l in contigs]
            recs = []
            for i in range(400):
                c, L = r.choice(contigs[:3])
                ln = r.randint(18, 25)
                pos = r.randint(1, L - ln)
                recs.append((c, pos, ln))
                sam.append("\t".join(["S_r%d_x%d" % (i, r.randint(1, 50)), r.choice(["0", "16"]), c, str(pos), "255", "%dM" % ln, "*", "0", "0", "A" * ln, "I" * ln]))
            sp = os.path.join(tmp, "x.sam"); bp = os.path.join(tmp, "x.bam"); bed = os.path.join(tmp, "k.bed")
            open(sp, "w").write("\n".join(sam) + "\n")
            subprocess.check_call([os.path.join(ORA_BIN, "samtools"), "view", "-bS", "-o", bp, sp], stderr=subprocess.DEVNULL)
            regions = []
            for _ in range(r.choice([1, 4, 12])):
                c, L = r.choice(contigs[:3])
                s = r.randint(0, L - 2)
                regions.append((c, s, min(L, s + r.choice([1, 20, 55, 300, 2000]))))
            open(bed, "w").write("".join("%s\t%d\t%d\n" % x for x in regions))
            out = subprocess.run([os.path.join(ORA_BIN, "samtools"), "view", "-L", bed, bp], capture_output=True, text=True).stdout
            kept = [l.split("\t")[0] for l in out.splitlines()]
            views.append({"contigs": [[n, l] for n, l in contigs], "records": [[c, p, l] for c, p, l in recs], "bed": [list(x) for x in regions], "kept_names": kept})
    path = os.path.join(GOLD, "gffmask.json.gz")
    with gzip.open(path, "wt", compresslevel=9) as f:
        json.dump({"generator": "reference gen_keep_regions_from_{exclude,include}_gff (py3 shim) + bundled samtools 0.1.18 view -L", "bed_cases": cases, "view_cases": views}, f)
    print("wrote", path, os.path.getsize(path), [len(c["bed_exclude"].splitlines()) for c in cases], [len(v["kept_names"]) for v in views])


if __name__ == "__main__":
    main()
