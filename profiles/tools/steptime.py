import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
order = np.zeros(1, dtype=np.int32)
for it in range(4):
    t0 = time.time(); ctx.candidate(10, 100, 300, order); t1 = time.time(); ctx.fold(300); t2 = time.time(); out = ctx.predict(1, 18, 23, False, True); t3 = time.time()
    tm = ctx.last_timings()
    print("wall ms: candidate %.2f fold %.2f predict %.2f | kernels: %s" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, {k: round(v,2) for k,v in tm.items()}))
