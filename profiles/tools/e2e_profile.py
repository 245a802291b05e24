"""Where the host time of the CLI `pipeline` verb goes: bench.e2e_cli on the config[1] workload, five runs' wall-clock, then every stage of two more
runs under cProfile.  usage (GPU box): python profiles/tools/e2e_profile.py > gpurun_out/e2e_profile.txt"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mir_prefer_amd import pipeline
from mir_prefer_amd import synth
specs, ns, bg, _, _ = bench.workload_specs("config1", 1)
contigs, alns, samples = bench.build_shard(specs, {0}, ns, bg)
ds = synth.Dataset(contigs, samples, alns, [])
BASE = "/dev/shm" if os.path.isdir("/dev/shm") else None
for k in range(5):
    r = bench.e2e_cli(ds, "vienna-2.1.2", base=BASE)
    print("run", k, "wall %.3f first %.3f" % (r["wall_s"], r["wall_s_first_run"]), {a: round(b, 4) for a, b in r["stage_s"].items()})
for stage in ("run_prepare", "run_candidate", "run_fold", "run_predict"):
    orig = getattr(pipeline.Pipeline, stage)
    pr = cProfile.Profile()
    def wrapped(self, *a, _orig=orig, _pr=pr, **kw):
        _pr.enable()
        try:
            return _orig(self, *a, **kw)
        finally:
            _pr.disable()
    setattr(pipeline.Pipeline, stage, wrapped)
    bench.e2e_cli(ds, "vienna-2.1.2", base=BASE)
    setattr(pipeline.Pipeline, stage, orig)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("=====", stage, "(two runs)")
    print(s.getvalue())
