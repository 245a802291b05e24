"""Dev tool: the generic fold path at PRECURSOR_LEN = 400 on the config[1] generator (what bench.py's configs.L400 times).
usage (GPU box): python profiles/tools/l400_time.py [L] [vienna-2.1.2|vienna-1.8.5]      (MIRP_LIB=<variant> for A/B builds)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from mir_prefer_amd import capi
L = int(sys.argv[1]) if len(sys.argv) > 1 else 400
specs, ns, bg, _, _ = bench.workload_specs("config1", 1)
contigs, alns, samples = bench.build_shard(specs, {0}, ns, bg)
ctx = capi.Context(0)
model = sys.argv[2] if len(sys.argv) > 2 else "vienna-2.1.2"
ctx.set_fold_model(model)
ctx.load_genome(contigs); ctx.load_alignments(alns)
_, _, nw = ctx.candidate(10, 100, L, np.zeros(1, dtype=np.int32))
for k in range(2):
    t = time.time(); ctx.fold(L); w = time.time() - t
    print("%s L = %d: %d windows, fold %.3f s = %.0f windows/s, generic fallbacks %d" % (model, L, nw, w, nw / w, ctx.last_fold_fallbacks()), flush=True)
out = ctx.predict(ns, 18, 23, False, True)
print("loci", len(out["result"]))
