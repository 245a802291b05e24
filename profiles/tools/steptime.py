"""Dev tool: wall time of the three C-ABI calls of one bench step against the device time the library reports for them (host overhead per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
o = np.zeros(1, dtype=np.int32)
for rep in range(6):
    t0 = time.time(); ctx.candidate(10, 100, 300, o); t1 = time.time(); ctx.fold(300); t2 = time.time(); out = ctx.predict(1, 18, 23, False, True); t3 = time.time()
    tm = ctx.last_timings()
    print("wall ms: candidate %.2f fold %.2f predict %.2f total %.2f | device ms: cov %.2f rest %.2f fold %.2f predict %.2f" %
          (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), tm["coverage_ms"], tm["candidate_rest_ms"], tm["fold_ms"], tm["predict_ms"]), flush=True)
