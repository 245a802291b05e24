"""Dev tool: predict (filter) kernel time of the bench workload with the library given by MIRP_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
ctx.fold(300)
ms = []
for _ in range(6):
    out = ctx.predict(1, 18, 23, False, True); ms.append(ctx.last_timings()["predict_ms"])
print("%-32s predict %.3f ms (min %.3f), %d loci" % (os.path.basename(capi.LIB_PATH), float(np.mean(ms[1:])), min(ms), len(out["result"])), flush=True)
