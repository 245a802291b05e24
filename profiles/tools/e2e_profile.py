"""Dev tool: cProfile of the CLI `pipeline` verb on files of the bench workload (config1), to see where the host spends the end-to-end time."""
import cProfile, io, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mir_prefer_amd import config, pipeline, synth
specs, ns, bg, _, _ = bench.workload_specs("config1", 1)
contigs, alns, samples = bench.build_shard(specs, {0}, ns, bg)
ds = synth.Dataset(contigs, samples, alns, [])
tmp = tempfile.mkdtemp(prefix="mirp_e2e_")
sams = ds.write_sams(tmp)
fa = os.path.join(tmp, "genome.fa"); ds.write_fasta(fa)
cfg = os.path.join(tmp, "config")
open(cfg, "w").write("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %s\nNAME_PREFIX = bench\nPRECURSOR_LEN = 300\nREADS_DEPTH_CUTOFF = 10\nMAX_GAP = 100\n" % (fa, ", ".join(sams), os.path.join(tmp, "out")))
so = sys.stdout; sys.stdout = open(os.devnull, "w")
for rep in range(2):
    shutil.rmtree(os.path.join(tmp, "out"), ignore_errors=True)
    opt = config.parse_configfile(cfg); opt["OUTPUT_DETAILS_FOR_DEBUG"] = False
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    p = pipeline.Pipeline(opt, 0)
    st = {}
    for s, kw in (("prepare", {}), ("candidate", {"defer": True}), ("fold", {"defer": True}), ("predict", {})):      # the `pipeline` verb's sequence
        t = time.time(); getattr(p, "run_" + s)(**kw); st[s] = time.time() - t
    pr.disable()
    wall = time.time() - t0
    p.ctx.close()
sys.stdout = so
print("wall %.3f" % wall, st)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
shutil.rmtree(tmp, ignore_errors=True)
