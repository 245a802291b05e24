"""Dev tool: differential stress of the lean `pipeline` run (early device open, no stage artefacts, streamed fold / filter / reports with a random number of chunks)
against the stage-by-stage `-k pipeline` run of the same inputs: every output file byte for byte, on random datasets (contig counts / name orders, 1 .. 40 samples,
both fold models, the overhang / no-star options).  usage (GPU box): python profiles/tools/stress_cli.py [n_datasets] [seed]"""
import filecmp, os, random, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mir_prefer_amd import synth
n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for k in range(n_sets):
    nc = r.randint(1, 6)
    names = r.sample(["Chr1", "Chr2", "Chr10", "chrM", "scaffold_9", "scaffold_10", "A", "b", "ctg.7", "Z_1"], nc)
    lens = [r.randint(30000, 400000) for _ in range(nc)]
    ns = r.choice([1, 2, 3, 5, 17, 40])
    loci = r.randint(20, 500)
    ds = synth.make_dataset(lens, loci, n_samples=ns, seed=r.randint(1, 10 ** 6), contig_names=names, edge_cases=True)
    tmp = tempfile.mkdtemp(prefix="stresscli_")
    try:
        sams = ds.write_sams(tmp, sq_order=r.sample(range(nc), nc))
        ds.write_fasta(os.path.join(tmp, "g.fa"))
        model = r.choice(["vienna-2.1.2", "vienna-1.8.5"])
        extra = "ALLOW_3NT_OVERHANG = %s\nALLOW_NO_STAR_EXPRESSION = %s\nPRECURSOR_LEN = %d\n" % (r.choice("YN"), r.choice("YN"), r.choice([300, 300, 250, 320, 420, 900]))
        outs = {}
        for mode, flags, env in (("lean", [], {"MIRP_STREAM_CHUNKS": str(r.choice([0, 1, 2, 5, 9]))}), ("keep", ["-k"], {})):
            cfg = os.path.join(tmp, "cfg_" + mode)
            open(cfg, "w").write("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %s\nNAME_PREFIX = s\n%s" % (os.path.join(tmp, "g.fa"), ", ".join(sams), os.path.join(tmp, mode), extra))
            p = subprocess.run([sys.executable, "-m", "mir_prefer_amd.cli", "--fold-model", model] + flags + ["pipeline", cfg], cwd=tmp,
                               env=dict(os.environ, PYTHONPATH=ROOT, **env), capture_output=True, text=True)
            if p.returncode != 0:
                print("dataset %d: %s run failed (%d): %s" % (k, mode, p.returncode, p.stderr[-400:])); bad += 1
            outs[mode] = os.path.join(tmp, mode)
        files = sorted(f for f in os.listdir(outs["keep"]) if os.path.isfile(os.path.join(outs["keep"], f)))
        lean_files = sorted(f for f in os.listdir(outs["lean"]) if os.path.isfile(os.path.join(outs["lean"], f))) if os.path.isdir(outs["lean"]) else []
        _, mism, errs = filecmp.cmpfiles(outs["keep"], outs["lean"], files, shallow=False)
        rm_k = sorted(os.listdir(os.path.join(outs["keep"], "readmapping"))) if os.path.isdir(os.path.join(outs["keep"], "readmapping")) else []
        rm_l = sorted(os.listdir(os.path.join(outs["lean"], "readmapping"))) if os.path.isdir(os.path.join(outs["lean"], "readmapping")) else []
        _, mism2, errs2 = filecmp.cmpfiles(os.path.join(outs["keep"], "readmapping"), os.path.join(outs["lean"], "readmapping"), rm_k, shallow=False) if rm_k else ([], [], [])
        ok = not mism and not errs and not mism2 and not errs2 and rm_k == rm_l and files == lean_files
        bad += 0 if ok else 1
        print("dataset %d: %d contigs %s, %d samples, %d loci planted, %s, %s -> %d result files + %d read-mapping files: %s" % (
            k, nc, names, ns, loci, model, extra.replace("\n", " "), len(files), len(rm_k), "identical" if ok else "DIFFER %s %s %s %s" % (mism, errs, mism2[:3], errs2[:3])), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
print("mismatching datasets: %d" % bad)
