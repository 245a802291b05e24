#!/usr/bin/env python3
"""Dev-only (build container): the WIDER pin of the oracle on the real reference stack (task 2).

(i)  tests/golden/headline_folds.json.gz -- the REAL bundled RNALfold binaries (2.1.2 through mloader.c, 1.8.5 native) at spans 300 and 150 on
     * every 9th window of the headline workload (bench.py config1, 19,686 windows: 2,188 of them, 86 % are the 325-nt L/R windows with planted
       hairpins the benchmark is made of),
     * 600 windows of the five stress families (tests/seqgen.stress_family, seed 20261004) and the 125 microsatellite windows
       (tests/seqgen.microsatellites),
     stored as one 6-byte digest per (window, model, span) of the printed lines + final MFE, the MFE itself, and a digest of every input sequence
     (the windows are regenerated from seeds by the test; the digest pins that they are the same ones).
(ii) tests/golden/mid/expected.json.gz -- the WHOLE py3-shimmed reference pipeline (miR_PREFeR.py + bundled samtools + RNALfold) on a mid-size
     dataset the other fixtures never saw (2 contigs named so that @SQ order is not lexicographic, 650 kb, 420 planted loci, 2 samples, seed 4242)
     under BOTH folders: per-window decisions, the raw result list and the gff3.

Prerequisite: tests/golden/tools/setup_oracle_bin.sh (wrappers under /tmp/ora/bin).  Fixtures are data: inputs' digests and the reference's outputs.
    python tests/golden/tools/gen_headline_golden.py [folds] [mid]
"""
import base64
import concurrent.futures as cf
import hashlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402
import ref_shim  # noqa: E402
from mir_prefer_amd import synth  # noqa: E402
from tests import seqgen  # noqa: E402

GOLD = gen_golden.GOLD
ORA_BIN = ref_shim.ORA_BIN
MODELS = [("vienna-2.1.2", "RNALfold212"), ("vienna-1.8.5", "RNALfold185")]
SPANS = [300, 150]
HEADLINE_STEP = 9
STRESS_SEED, STRESS_COUNT = 20261004, 600
MID = {"lens": [400000, 250000], "names": ["Chr2", "Chr10"], "loci": 420, "samples": 2, "seed": 4242}


from tests.golden.tools_digest import array_digest, fold_digest, seq_digest  # noqa: E402


def headline_windows():
    """The window sequences of bench.py's config1 workload as the oracle's candidate stage cuts them (the GPU tests hold the device's windows equal
    to these field by field), every HEADLINE_STEP-th."""
    import bench
    from tests import oracle_binding
    specs, ns, bg, _, _ = bench.workload_specs("config1", 1)
    contigs, alns, _ = bench.build_shard(specs, {0}, ns, bg)
    o = oracle_binding.load()
    lens = np.array([len(s) for _, s in contigs], dtype=np.int64)
    _, peaks = o.coverage_peaks(alns, lens, bench.CUT)
    win = o.make_windows(peaks, alns, contigs, np.arange(len(contigs), dtype=np.int32), bench.GAP, bench.L, bench.CUT * 0.5)
    W = win["windows"]
    return len(W), [win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes().decode() for b in W[::HEADLINE_STEP]]


def stress_windows():
    import random
    r = random.Random(STRESS_SEED)
    return [seqgen.stress_family(r, i % 5) for i in range(STRESS_COUNT)] + seqgen.microsatellites()


def _fold_chunk(args):
    binary, seqs, span = args
    return gen_golden._run_rnalfold(binary, seqs, span)


def real_folds(binary, seqs, span):
    step = 40
    tasks = [(binary, seqs[k:k + step], span) for k in range(0, len(seqs), step)]
    out = []
    with cf.ProcessPoolExecutor(os.cpu_count() or 4) as ex:
        for part in ex.map(_fold_chunk, tasks):
            out.extend(part)
    return out


def gen_folds():
    n_all, head = headline_windows()
    groups = {"headline": head, "stress": stress_windows()}
    fix = {"generator": "RNALfold 2.1.2 (dependency/Mac/osx-10.9/RNALfold-2.1.2 through mloader.c) and RNALfold 1.8.5 (dependency/Linux/x64/RNALfold), "
                        "-L span, default dangles; digest = blake2b-6 over 'ss energy start\\n' per printed line + '|mfe' (energies in 0.01 kcal/mol)",
           "headline_windows_total": n_all, "headline_step": HEADLINE_STEP, "stress_seed": STRESS_SEED, "stress_count": STRESS_COUNT, "groups": {}}
    for gname, seqs in groups.items():
        g = {"n": len(seqs), "seq_digests": base64.b64encode(b"".join(seq_digest(s) for s in seqs)).decode(), "folds": {}}
        for model, binary in MODELS:
            for span in SPANS:
                res = real_folds(os.path.join(ORA_BIN, binary), seqs, span)
                g["folds"]["%s/%d" % (model, span)] = {
                    "digests": base64.b64encode(b"".join(fold_digest(r["lines"], r["mfe"]) for r in res)).decode(),
                    "mfe": [r["mfe"] for r in res], "n_lines": [len(r["lines"]) for r in res]}
                print(gname, model, span, len(res), "windows,", sum(len(r["lines"]) for r in res), "lines")
        fix["groups"][gname] = g
    gen_golden.dump_gz(fix, os.path.join(GOLD, "headline_folds.json.gz"))


def gen_mid():
    ds = synth.make_dataset(MID["lens"], MID["loci"], n_samples=MID["samples"], seed=MID["seed"], contig_names=MID["names"], edge_cases=True)
    exp = {"dataset": dict(MID, genome_sha256=[array_digest(s) for _, s in ds.contigs], alns_sha256=array_digest(ds.sorted_alns())),
           "sample_names": ds.sample_names, "runs": {}}
    for model, binary in MODELS:
        name = "mid"
        work = os.path.join("/tmp", "golden_mid_" + binary)
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(work)
        fa = os.path.join(work, "genome.fa")
        ds.write_fasta(fa)
        sams = ds.write_sams(work)
        opts = {"PIPELINE_PATH": "/root/reference", "FASTA_FILE": fa, "ALIGNMENT_FILE": ", ".join(sams), "PRECURSOR_LEN": 300, "READS_DEPTH_CUTOFF": 10,
                "NUM_OF_CORE": 4, "OUTFOLDER": os.path.join(work, "out"), "NAME_PREFIX": name, "MAX_GAP": 100, "MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 23,
                "ALLOW_NO_STAR_EXPRESSION": "Y", "ALLOW_3NT_OVERHANG": "N", "CHECKPOINT_SIZE": 300}
        cfg = os.path.join(work, "config")
        with open(cfg, "w") as f:
            for k, v in opts.items():
                f.write("%s = %s\n" % (k, v))
        ref_shim.run_pipeline(["-k", "pipeline", cfg], binary)
        g = ref_shim.load_reference(binary)
        out = opts["OUTFOLDER"]
        tmp = os.path.join(out, name + "_tmp")
        bam = os.path.join(tmp, "combined.filtered.sort.bam")
        headers, decisions, raw_result = [], [], []
        i = 0
        while os.path.exists(os.path.join(tmp, "%s.rnalfold.in_%d.fa" % (name, i))):
            fasta = os.path.join(tmp, "%s.rnalfold.in_%d.fa" % (name, i))
            headers += [ln for ln in open(fasta).read().splitlines() if ln.startswith(">")]
            for mir in g["filter_next_loci"](os.path.join(tmp, "%s.alndump_%d.dump" % (name, i)), os.path.join(tmp, "%s_rnalfoldoutput_%d" % (name, i)), bam,
                                             ds.sample_names, False, True, True, 18, 23, 10, minlen=55):
                if isinstance(mir, list):
                    decisions.append(len(mir))          # number of (mature, structure) pairs that passed in this region
                    raw_result.append(mir[0])
                else:
                    decisions.append(0)
            i += 1

        def strip(m):
            e = m[-1]
            return m[:-1] + [{"total_depth_mature": e["total_depth_mature"], "total_depth_star": e["total_depth_star"]}]
        exp["runs"][model] = {"config": {k: v for k, v in opts.items() if k not in ("PIPELINE_PATH", "FASTA_FILE", "ALIGNMENT_FILE", "OUTFOLDER")},
                              "n_windows": len(headers), "fasta_headers_sha256": hashlib.sha256("\n".join(headers).encode()).hexdigest(),
                              "decisions": decisions, "result_raw": gen_golden.jsonable([strip(m) for m in raw_result]),
                              "gff3": open(os.path.join(out, name + "_miRNA.gff3")).read()}
        print("mid", model, "windows:", len(headers), "regions decided:", len(decisions), "miRNAs:", len(raw_result))
    os.makedirs(os.path.join(GOLD, "mid"), exist_ok=True)
    gen_golden.dump_gz(exp, os.path.join(GOLD, "mid", "expected.json.gz"))


def gen_long():
    """tests/golden/long_folds.json.gz: both RNALfold binaries at spans 400 and 330 on 160 windows of 360 .. 480 nt (tests/seqgen.windows, seed 4001) and 40
    windows of the stress families stretched to that length -- PRECURSOR_LEN beyond what the LDS-resident kernels hold: the generic kernels' domain
    (fold_generic_kernel / fold185_kernel, rewritten in round 5)."""
    seqs = seqgen.long_windows()
    fix = {"generator": "RNALfold 2.1.2 / 1.8.5 (bundled binaries), -L span, default dangles; digests as headline_folds.json.gz", "seed": 4001, "n": len(seqs),
           "seq_digests": base64.b64encode(b"".join(seq_digest(s) for s in seqs)).decode(), "folds": {}}
    for model, binary in MODELS:
        for span in (400, 330):
            res = real_folds(os.path.join(ORA_BIN, binary), seqs, span)
            fix["folds"]["%s/%d" % (model, span)] = {"digests": base64.b64encode(b"".join(fold_digest(x["lines"], x["mfe"]) for x in res)).decode(),
                                                     "mfe": [x["mfe"] for x in res], "n_lines": [len(x["lines"]) for x in res]}
            print("long", model, span, len(res), "windows,", sum(len(x["lines"]) for x in res), "lines")
    gen_golden.dump_gz(fix, os.path.join(GOLD, "long_folds.json.gz"))


def gen_xl():
    """tests/golden/xl_folds.json.gz: both RNALfold binaries at span 3000 -- PRECURSOR_LEN's upper limit in the reference (MP:167-184) -- on a window of
    3,020 nt and one of 1,500 nt (tests/seqgen.xl_windows)."""
    seqs = seqgen.xl_windows()
    fix = {"generator": "RNALfold 2.1.2 / 1.8.5 (bundled binaries), -L 3000, default dangles; digests as headline_folds.json.gz", "n": len(seqs),
           "seq_digests": base64.b64encode(b"".join(seq_digest(s) for s in seqs)).decode(), "folds": {}}
    for model, binary in MODELS:
        res = real_folds(os.path.join(ORA_BIN, binary), seqs, 3000)
        fix["folds"]["%s/%d" % (model, 3000)] = {"digests": base64.b64encode(b"".join(fold_digest(x["lines"], x["mfe"]) for x in res)).decode(),
                                                 "mfe": [x["mfe"] for x in res], "n_lines": [len(x["lines"]) for x in res]}
        print("xl", model, len(res), "windows,", [len(x["lines"]) for x in res], "lines")
    gen_golden.dump_gz(fix, os.path.join(GOLD, "xl_folds.json.gz"))


if __name__ == "__main__":
    what = sys.argv[1:] or ["folds", "mid"]
    if "xl" in what:
        gen_xl()
    if "long" in what:
        gen_long()
    if "folds" in what:
        gen_folds()
    if "mid" in what:
        gen_mid()
