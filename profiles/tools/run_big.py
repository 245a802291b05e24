"""Dev tool: ~0.5 M-window scale check (10 contigs x 30 Mb, 300k loci) through the device pipeline: buffer sizing, sub-batching, throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
t = time.time()
ds = synth.make_dataset([30000000] * 10, 300000, n_samples=1, seed=11)
alns = ds.sorted_alns()
print("dataset: %d records, %.1f s" % (len(alns), time.time() - t), flush=True)
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(alns)
order = np.argsort(np.array(ds.contig_names, dtype=object), kind="stable").astype(np.int32)
for rep in range(2):
    t = time.time()
    npk, nloci, nwin = ctx.candidate(10, 100, 300, order)
    t1 = time.time(); ctx.fold(300); t2 = time.time()
    out = ctx.predict(1, 18, 23, False, True); t3 = time.time()
    print("rep %d: windows %d -> %d loci; candidate %.1f ms fold %.1f ms predict %.1f ms; %.0f windows/s; fallbacks %d; %s" % (
        rep, nwin, len(out["result"]), 1e3 * (t1 - t), 1e3 * (t2 - t1), 1e3 * (t3 - t2), nwin / (t3 - t), ctx.last_fold_fallbacks(), ctx.last_timings()), flush=True)
print("status nonzero:", int((ctx.fold_status() != 0).sum()))
