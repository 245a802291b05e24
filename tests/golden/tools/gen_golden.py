#!/usr/bin/env python3
"""Dev-only (build container): generate the committed golden fixtures under tests/golden/ by running
the REAL reference stack on seeded synthetic inputs:
  * /root/reference/miR_PREFeR.py under the py3 shim (ref_shim.py),
  * bundled samtools 0.1.18,
  * bundled RNALfold-2.1.2 (Mach-O) through tests/golden/tools/mloader.c.
Prerequisite: /tmp/ora/bin holds `mloader`, `samtools`, `RNALfold212` wrappers (see tools/setup_oracle_bin.sh).
Fixtures are data only: inputs and the reference's outputs.
"""
import gzip
import json
import os
import pickle
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from mir_prefer_amd import synth  # noqa: E402
from tests import seqgen  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ORA_BIN = ref_shim.ORA_BIN


def jsonable(o):
    if isinstance(o, dict):
        return {"__dict__": [[jsonable(k), jsonable(v)] for k, v in o.items()]}
    if isinstance(o, tuple):
        return {"__tuple__": [jsonable(x) for x in o]}
    if isinstance(o, list):
        return [jsonable(x) for x in o]
    if isinstance(o, (str, int, float, bool)) or o is None:
        return o
    raise TypeError(type(o))


def dump_gz(obj, path):
    with gzip.open(path, "wt", compresslevel=9) as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


def _run_rnalfold(binary, seqs, span):
    fa = "".join(">w%d\n%s\n" % (i, s) for i, s in enumerate(seqs))
    out = subprocess.run([binary, "-L", str(span)], input=fa, capture_output=True, text=True, cwd="/tmp").stdout
    res = []
    cur = None
    for line in out.splitlines():
        if line.startswith(">"):
            cur = {"lines": [], "mfe": None}
            res.append(cur)
            continue
        sp = line.split()
        if line.startswith(" ("):
            cur["mfe"] = int(round(float(line.strip().strip("()")) * 100))
        elif len(sp) >= 3 and line[:1] in ".(":
            e = line[line.index(" (") + 2:line.rindex(")")]
            cur["lines"].append([sp[0], int(round(float(e) * 100)), int(sp[-1])])
    assert len(res) == len(seqs)
    return res


def gen_fold_golden():
    """RNALfold 2.1.2 -L outputs on seeded windows (structure lines + final MFE)."""
    cases = []
    for seed, cnt, lo, hi, span in [(101, 160, 5, 120, 300), (102, 60, 60, 200, 40), (103, 40, 300, 350, 300),
                                    (104, 40, 120, 350, 100), (105, 30, 30, 90, 20)]:
        seqs = seqgen.windows(seed, cnt, lo, hi)
        cases.append({"seed": seed, "span": span, "seqs": seqs, "expected": _run_rnalfold(os.path.join(ORA_BIN, "RNALfold212"), seqs, span)})
    dump_gz({"generator": "RNALfold 2.1.2 (reference dependency/Mac/osx-10.9/RNALfold-2.1.2) -L span, default dangles",
             "cases": cases}, os.path.join(GOLD, "fold_rnalfold212.json.gz"))


def gen_fold_golden_185():
    """RNALfold 1.8.5 -L outputs (the Linux binary the reference bundles: Turner-1999, default dangles = 1) on seeded windows."""
    cases = []
    for seed, cnt, lo, hi, span in [(201, 160, 5, 120, 300), (202, 60, 60, 200, 40), (203, 40, 300, 350, 300),
                                    (204, 40, 120, 350, 100), (205, 30, 30, 90, 20)]:
        seqs = seqgen.windows(seed, cnt, lo, hi)
        cases.append({"seed": seed, "span": span, "seqs": seqs, "expected": _run_rnalfold(os.path.join(ORA_BIN, "RNALfold185"), seqs, span)})
    extra = ["AUCACUUCUUCGUAAUGUUGUUCGUAUAUAU", "AAAAAAAAAA", "GGGAAAACCC", "ACGU", "A", "GUGG" * 59, "GC" * 100, "N" * 30]
    cases.append({"seed": 0, "span": 300, "seqs": extra, "expected": _run_rnalfold(os.path.join(ORA_BIN, "RNALfold185"), extra, 300)})
    dump_gz({"generator": "RNALfold 1.8.5 (reference dependency/Linux/x64/RNALfold) -L span, default dangles (d1)",
             "cases": cases}, os.path.join(GOLD, "fold_rnalfold185.json.gz"))


def gen_pipeline_golden(name, contig_lens, names, n_loci, n_samples, seed, sq_order, config_extra, rnalfold="RNALfold212"):
    work = os.path.join("/tmp", "golden_" + name)
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    ds = synth.make_dataset(contig_lens, n_loci, n_samples=n_samples, seed=seed, contig_names=names, edge_cases=True)
    fa = os.path.join(work, "genome.fa")
    ds.write_fasta(fa)
    sams = ds.write_sams(work, sq_order=sq_order)
    cfg = os.path.join(work, "config")
    opts = {"PIPELINE_PATH": "/root/reference", "FASTA_FILE": fa, "ALIGNMENT_FILE": ", ".join(sams), "PRECURSOR_LEN": 300,
            "READS_DEPTH_CUTOFF": 10, "NUM_OF_CORE": 2, "OUTFOLDER": os.path.join(work, "out"), "NAME_PREFIX": name,
            "MAX_GAP": 100, "MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 23, "ALLOW_NO_STAR_EXPRESSION": "Y",
            "ALLOW_3NT_OVERHANG": "N", "CHECKPOINT_SIZE": 300}
    opts.update(config_extra)
    with open(cfg, "w") as f:
        for k, v in opts.items():
            f.write("%s = %s\n" % (k, v))
    ref_shim.run_pipeline(["-L", "-k", "-d", "pipeline", cfg], rnalfold)
    g = ref_shim.load_reference(rnalfold)
    out = opts["OUTFOLDER"]
    tmp = os.path.join(out, name + "_tmp")
    exp = {"config": {k: v for k, v in opts.items() if k not in ("PIPELINE_PATH", "FASTA_FILE", "ALIGNMENT_FILE", "OUTFOLDER")},
           "sample_names": ds.sample_names, "sq_order": [names[t] for t in (sq_order or range(len(names)))],
           "fold_model": "vienna-1.8.5" if rnalfold == "RNALfold185" else "vienna-2.1.2"}
    exp["depth_cut"] = open(os.path.join(tmp, "bam.depth.cut%d" % opts["READS_DEPTH_CUTOFF"])).read()
    dict_option = {"READS_DEPTH_CUTOFF": opts["READS_DEPTH_CUTOFF"]}
    _, dict_contigs = g["gen_contig_typeA"](os.path.join(tmp, "expanded.plus.bam"), os.path.join(tmp, "expanded.minus.bam"),
                                            dict_option, 19, None, work)
    exp["dict_contigs"] = jsonable(dict_contigs)
    with open(os.path.join(tmp, name + "_loci_dump.dump"), "rb") as f:
        exp["dict_loci"] = jsonable(pickle.load(f))
    exp["exregion_gff"] = open(os.path.join(tmp, name + "_ExRegionA.gff3")).read()
    pieces = []
    samplenames = ds.sample_names
    bam = os.path.join(tmp, "combined.filtered.sort.bam")
    allow3 = opts["ALLOW_3NT_OVERHANG"] == "Y"
    allow_no_star = opts["ALLOW_NO_STAR_EXPRESSION"] == "Y"
    raw_result = []
    i = 0
    while os.path.exists(os.path.join(tmp, "%s.rnalfold.in_%d.fa" % (name, i))):
        fasta = os.path.join(tmp, "%s.rnalfold.in_%d.fa" % (name, i))
        dumpname = os.path.join(tmp, "%s.alndump_%d.dump" % (name, i))
        foldname = os.path.join(tmp, "%s_rnalfoldoutput_%d" % (name, i))
        lines = open(fasta).read().splitlines()
        entries = [[lines[k], lines[k + 1]] for k in range(0, len(lines), 2)]
        dumps = []
        with open(dumpname) as f:
            while True:
                try:
                    dumps.append(ref_shim.sys.modules["cPickle"].load(f))
                except EOFError:
                    break
        structs = list(g["get_structures_next_extendregion"](foldname, 55))
        # per (window, mature, structure) outputs of get_maturestar_info / check_expression_new
        ms_info = []
        for w, (dump, st) in enumerate(zip(dumps, structs)):
            region, which, dict_aln, matures = dump
            mapinfo = g["gen_mapinfo_each_sample"](bam, samplenames, region[0], region[1][0], region[1][1])
            for (m0, m1, strand, mdepth) in matures:
                for (energy, foldstart, ss, sstype) in st[2]:
                    r = g["get_maturestar_info"](ss, (m0, m1), foldstart, foldstart + len(ss), region[1][0], region[1][1], strand)
                    ex = None
                    if not isinstance(r, str):
                        ex = g["check_expression_new"](mapinfo, samplenames, r[2], r[3], (m0, m1), mdepth, (r[0], r[1]), strand, allow3)
                        ex = {k: v for k, v in ex.items() if k not in samplenames and k != "samplenames"} | {
                            "per_sample": {s: {k: v for k, v in ex[s].items() if k != "reads_maps"} for s in samplenames}}
                    ms_info.append([w, [m0, m1, strand, mdepth], foldstart, ss, jsonable(r), jsonable(ex)])
        decisions = []
        for mir in g["filter_next_loci"](dumpname, foldname, bam, samplenames, allow3, allow_no_star, True,
                                         opts["MIN_MATURE_LEN"], opts["MAX_MATURE_LEN"], opts["READS_DEPTH_CUTOFF"], minlen=55):
            if isinstance(mir, list):
                recs = []
                for m in mir:
                    e = m[-1]
                    recs.append(m[:-1] + [{k: e[k] for k in ("total_depth_mature", "total_depth_star", "total_depth_isoform",
                                                             "total_depth_just_this_strand", "total_depth_anti", "mature_star_distance")}])
                decisions.append({"pass": True, "mirnas": jsonable(recs)})
                raw_result.append(mir[0])
            else:
                key = [k for k in mir if isinstance(k, tuple)][0]
                decisions.append({"pass": False, "region": jsonable(list(key)), "which": mir["which"]})
        pieces.append({"fasta": entries, "alndump": jsonable(dumps), "rnalfold_out": open(foldname).read(),
                       "structures": jsonable(structs), "maturestar_expr": ms_info, "decisions": decisions})
        i += 1
    exp["pieces"] = pieces

    def strip(m):
        e = m[-1]
        return m[:-1] + [{"total_depth_mature": e["total_depth_mature"], "total_depth_star": e["total_depth_star"]}]

    exp["result_raw"] = jsonable([strip(m) for m in raw_result])
    exp["gff3"] = open(os.path.join(out, name + "_miRNA.gff3")).read()
    exp["reports"] = {"mature_fa": open(os.path.join(out, name + "_miRNA.mature.fa")).read(),
                      "precursor_fa": open(os.path.join(out, name + "_miRNA.precursor.fa")).read(),
                      "precursor_ss": open(os.path.join(out, name + "_miRNA.precursor.ss")).read(),
                      "detail_csv": open(os.path.join(out, name + "_miRNA.detail.csv")).read(),
                      "stat_txt": open(os.path.join(out, "miRNA.stat.txt")).read(),
                      "detail_html": open(os.path.join(out, name + "_miRNA.detail.html")).read()}
    # -d artefact (convert_failure_reasons_list MP:2505-2529, write_dict_reasons MP:2532-2567); dict order = the shim's insertion order
    rm = os.path.join(out, "readmapping")
    exp["readmapping"] = {fn: open(os.path.join(rm, fn)).read() for fn in sorted(os.listdir(rm))}   # gen_map_result, MP:2907-2959
    exp["reasons_txt"] = open(os.path.join(out, name + "_reason_why_not_miRNA.txt")).read()
    frm = os.path.join(out, "failed_readmapping")             # write_dict_reasons' gen_map_result over the expression failures (MP:2561-2567)
    exp["failed_readmapping"] = {fn: open(os.path.join(frm, fn)).read() for fn in sorted(os.listdir(frm))} if os.path.isdir(frm) else None
    d = os.path.join(GOLD, name)
    os.makedirs(d, exist_ok=True)
    for p in [fa] + sams:
        with open(p, "rb") as fi, gzip.open(os.path.join(d, os.path.basename(p) + ".gz"), "wb", compresslevel=9) as fo:
            fo.write(fi.read())
    dump_gz(exp, os.path.join(d, "expected.json.gz"))
    print(name, "loci:", sum(len(v) for v in pickle.load(open(os.path.join(tmp, name + "_loci_dump.dump"), "rb")).values()),
          "windows:", sum(len(p["fasta"]) for p in pieces), "miRNAs:", len(raw_result))


if __name__ == "__main__":
    what = sys.argv[1:] or ["fold", "mini"]
    if "fold" in what:
        gen_fold_golden()
    if "fold185" in what:
        gen_fold_golden_185()
    if "mini" in what:
        # 3 contigs, @SQ order deliberately non-lexicographic, 2 samples, edge cases planted
        gen_pipeline_golden("mini", [120000, 60000, 90000], ["Chr2", "Chr10", "Chr1"], 130, 2, 5, None, {})
    if "mini185" in what:
        # the "mini" dataset with the RNALfold the reference bundles for Linux (1.8.5) on PATH instead of 2.1.2
        gen_pipeline_golden("mini185", [120000, 60000, 90000], ["Chr2", "Chr10", "Chr1"], 130, 2, 5, None, {}, rnalfold="RNALfold185")
    if "mini400" in what:
        # PRECURSOR_LEN = 400: window extension at another length, RNALfold -L 400 on windows of 400 / 425 nt (the generic fold kernels' domain)
        gen_pipeline_golden("mini400", [90000, 50000], ["cB", "cA"], 60, 2, 17, None, {"PRECURSOR_LEN": 400})
    if "mini24" in what:
        # 24 ALIGNMENT_FILEs: more samples than one 16-entry register array held in rounds 1-4 (the reference has no limit, MP:3300-3308); both
        # no-star rules that look at every sample (MP:2316-2330) are in play with ALLOW_NO_STAR_EXPRESSION = Y
        gen_pipeline_golden("mini24", [70000, 40000], ["ctgY", "ctgX"], 70, 24, 11, None, {})
    if "mini3" in what:
        gen_pipeline_golden("mini3", [80000, 50000], ["ctgB", "ctgA"], 70, 3, 9, [1, 0],
                            {"ALLOW_3NT_OVERHANG": "Y", "ALLOW_NO_STAR_EXPRESSION": "N", "MAX_MATURE_LEN": 24})
