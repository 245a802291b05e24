"""Dev tool: the resident path (candidate -> fold -> filter) at PRECURSOR_LEN = 1000 and 3000 on a small synthetic dataset: window count, fold time, statuses, the
largest number of structure lines, loci.  usage (GPU box): python profiles/tools/l3000_probe.py      (MIRP_LIB=<variant> for A/B builds)"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from mir_prefer_amd import capi, synth
ds = synth.make_dataset([60000, 40000], 12, n_samples=2, seed=77, contig_names=["k2", "k1"], edge_cases=True)
names, alns = ds.contig_names, ds.sorted_alns()
order = np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(alns)
for L in (1000, 3000):
    _, _, nwin = ctx.candidate(8, 80, L, order)
    t = time.time(); ctx.fold(L); print("L", L, "windows", nwin, "fold s", round(time.time() - t, 2), flush=True)
    raw = ctx.get_fold()
    print(" status", np.unique(raw["status"], return_counts=True), "lines max", raw["n_lines"].max())
    out = ctx.predict(2, 18, 24, True, True)
    print(" loci", len(out["result"]))
