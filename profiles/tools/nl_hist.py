"""Dev tool: distribution of the number of structure lines per window of the bench workload (what the filter kernel stages per window)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
ctx.fold(300)
f = ctx.get_fold()
nl = f["n_lines"]
ln = f["lines"]
used = np.array([int(((ln[w]["printed"] != 0) & (ln[w]["len"] >= 55))[:min(nl[w], f["max_lines"])].sum()) for w in range(len(nl))])
print("windows", len(nl), "lines: mean %.1f, quantiles 50/90/99/max" % nl.mean(), np.percentile(nl, [50, 90, 99]).tolist(), nl.max())
print("lines printed and >= 55 long: mean %.1f, quantiles 50/90/99/max" % used.mean(), np.percentile(used, [50, 90, 99]).tolist(), used.max())
for c in (24, 32, 40, 48, 64):
    print("  windows with more than %d lines: %.2f %%   with more than %d used lines: %.2f %%" % (c, 100.0 * (nl > c).mean(), c, 100.0 * (used > c).mean()))
