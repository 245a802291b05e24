"""Dev tool: BASELINE config[2]-like scale check (TAIR10-sized genome: 5 contigs, 119 Mb; 3 samples; ~80k windows) through the device pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
t = time.time()
lens = [30427671, 19698289, 23459830, 18585056, 26975502]
ds = synth.make_dataset(lens, 48000, n_samples=3, seed=3, contig_names=["Chr1", "Chr2", "Chr3", "Chr4", "Chr5"])
alns = ds.sorted_alns()
print("dataset: %d records, %.1f s" % (len(alns), time.time() - t), flush=True)
ctx = capi.Context(0)
t = time.time(); ctx.load_genome(ds.contigs); ctx.load_alignments(alns); print("upload %.3f s" % (time.time() - t))
order = np.arange(5, dtype=np.int32)
for rep in range(2):
    t = time.time()
    npk, nloci, nwin = ctx.candidate(10, 100, 300, order)
    t1 = time.time(); ctx.fold(300); t2 = time.time()
    out = ctx.predict(3, 18, 23, False, True); t3 = time.time()
    tm = ctx.last_timings()
    print("rep %d: peaks %d loci %d windows %d -> %d miRNA loci; candidate %.1f ms fold %.1f ms predict %.1f ms; total %.3f s -> %.0f windows/s; kernel ms %s" % (
        rep, npk, nloci, nwin, len(out["result"]), 1e3 * (t1 - t), 1e3 * (t2 - t1), 1e3 * (t3 - t2), t3 - t, nwin / (t3 - t), tm))
st = ctx.fold_status()
print("fold status nonzero:", int((st != 0).sum()))
