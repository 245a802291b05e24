"""Dev tool: fill-kernel time of the bench windows at several spans (-L): how the time is spread over the diagonals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
for span in [int(x) for x in (sys.argv[1:] or ["8", "20", "36", "68", "100", "150", "200", "250", "300"])]:
    ms = []
    for _ in range(3):
        ctx.fold(span); ms.append(ctx.last_fold_kernel_ms())
    print("span %3d: fill %.2f ms  epilogue %.2f ms" % (span, min(m[0] for m in ms), min(m[1] for m in ms)), flush=True)
