"""Dev tool: fold-phase ablation timings in one process (MIRP_FOLD_DEBUG is re-read on every fold call).
needs the diagnostics build: make -C mir-prefer_amd/csrc DIAG=1, then MIRP_LIB=mir-prefer_amd/libmirprefer_diag.so python <this file> 0 1 2 ...
(results of ablated runs are wrong by construction); variants are interleaved over two rounds in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
for rnd in range(2):
    for f in (sys.argv[1:] or ["0"]):
        os.environ["MIRP_FOLD_DEBUG"] = f
        ms = []
        for _ in range(3):
            ctx.fold(300); ms.append(ctx.last_fold_kernel_ms()[0])
        print("round %d flags %-5s fill kernel %.2f ms (min %.2f)" % (rnd, f, float(np.mean(ms[1:])), min(ms)), flush=True)
