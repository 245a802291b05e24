"""Dev tool: fill / epilogue kernel times of the bench workload with the library given by MIRP_LIB (product or a compile-time ablation build,
make -C mir-prefer_amd/csrc ABLATE=<flags>).  Prints one line; run several libraries back to back on one box to compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
ms = []
for _ in range(5):
    ctx.fold(300); ms.append(ctx.last_fold_kernel_ms())
print("%-40s fill %.2f ms (min %.2f)  epilogue %.2f ms" % (os.path.basename(capi.LIB_PATH), float(np.mean([m[0] for m in ms[1:]])), min(m[0] for m in ms), float(np.mean([m[1] for m in ms[1:]]))), flush=True)
