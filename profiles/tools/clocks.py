"""Dev tool: per-wave phase clocks of the fold fill kernel (MIRP_FOLD_CLOCKS=1) on the benchmark workload."""
import os, sys
# needs the diagnostics build: make -C mir-prefer_amd/csrc DIAG=1, then MIRP_LIB=mir-prefer_amd/libmirprefer_diag.so python <this file>
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
ctx.fold(300)
os.environ["MIRP_FOLD_CLOCKS"] = os.environ.get("CLOCKS_MODE", "1")   # 2: light mode (busy + barrier wait per wave only)
for f in (sys.argv[1:] or ["0"]):
    os.environ["MIRP_FOLD_DEBUG"] = f
    print("==== MIRP_FOLD_DEBUG=%s" % f, flush=True)
    sys.stderr.write("==== MIRP_FOLD_DEBUG=%s\n" % f); sys.stderr.flush()
    ctx.fold(300)
    print(ctx.last_timings(), flush=True)
