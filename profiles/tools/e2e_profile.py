"""Where the host time of the CLI `pipeline` verb goes: bench.e2e_cli on the config[1] workload, five runs' wall-clock, then the predict stage of one more
run under cProfile.  usage (GPU box): python profiles/tools/e2e_profile.py > gpurun_out/e2e_profile.txt"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mir_prefer_amd import pipeline
from mir_prefer_amd import synth
specs, ns, bg, _, _ = bench.workload_specs("config1", 1)
contigs, alns, samples = bench.build_shard(specs, {0}, ns, bg)
ds = synth.Dataset(contigs, samples, alns, [])
for k in range(5):
    r = bench.e2e_cli(ds, "vienna-2.1.2")
    print("run", k, "wall %.3f first %.3f" % (r["wall_s"], r["wall_s_first_run"]), {a: round(b, 4) for a, b in r["stage_s"].items()})
orig = pipeline.Pipeline.run_predict
pr = cProfile.Profile()
def wrapped(self, *a, **kw):
    pr.enable()
    try:
        return orig(self, *a, **kw)
    finally:
        pr.disable()
pipeline.Pipeline.run_predict = wrapped
bench.e2e_cli(ds, "vienna-2.1.2")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
