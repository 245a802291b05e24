"""Dev tool: like ab_time.py, for the vienna-1.8.5 fold model."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.set_fold_model("vienna-1.8.5")
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
ms = []
for _ in range(4):
    ctx.fold(300); ms.append(ctx.last_fold_kernel_ms())
print("%-40s vienna-1.8.5: fill %.2f ms  epilogue %.2f ms" % (os.path.basename(capi.LIB_PATH), float(np.mean([m[0] for m in ms[1:]])), float(np.mean([m[1] for m in ms[1:]]))), flush=True)
print("windows handed to the dense pass: %d, generic fallbacks: %d" % (ctx.last_fold_dense(), ctx.last_fold_fallbacks()), flush=True)
