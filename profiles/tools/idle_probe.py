"""Dev tool: cost of the first fold of a second context in a process that already folded (first context / second context / again).
usage (GPU box): python profiles/tools/idle_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
for pause in (0, 0, 0.2, 1, 3, 0, 6, 0, 0):
    time.sleep(pause)
    ctx.fold(300)
    print("after %.1f s idle: fold %.1f ms device" % (pause, ctx.last_timings()["fold_ms"]), flush=True)
ctx2 = capi.Context(0)
ctx2.load_genome(ds.contigs); ctx2.load_alignments(ds.sorted_alns())
ctx2.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
for k in range(3):
    ctx2.fold(300); print("second context, fold %d: %.1f ms device" % (k, ctx2.last_timings()["fold_ms"]), flush=True)
