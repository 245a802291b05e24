"""Dev tool: is the slow FIRST fold of a fresh process (0.07 - 0.4 s for the same 19,686 windows) first-touch cost or an idle GPU's clock ramp?  Folds the
config[1] windows 4 times back to back, sleeps, folds again; prints the fill / epilogue kernel times of every call.
usage (GPU box): python profiles/tools/first_fold.py [sleep seconds ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from mir_prefer_amd import capi
specs, ns, bg, _, _ = bench.workload_specs("config1", 1)
contigs, alns, samples = bench.build_shard(specs, {0}, ns, bg)
t0 = time.time()
ctx = capi.Context(0)
print("context %.3f s" % (time.time() - t0))
ctx.load_genome(contigs); ctx.load_alignments(alns)
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
def fold(tag):
    t = time.time(); ctx.fold(300); w = time.time() - t
    km = ctx.last_fold_kernel_ms()
    print("%-28s wall %.4f s  fill %.2f ms  epilogue %.2f ms" % (tag, w, km[0], km[1]), flush=True)
for k in range(4):
    fold("fold %d" % k)
for s in [float(x) for x in sys.argv[1:]] or [0.05, 0.2, 1.0, 3.0]:
    time.sleep(s)
    fold("after %.2f s idle" % s)
    fold("  again")
