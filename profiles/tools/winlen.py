import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from mir_prefer_amd import synth, capi
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
ctx = capi.Context(0)
ctx.load_genome(ds.contigs); ctx.load_alignments(ds.sorted_alns())
ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
L = ctx.get_windows()["windows"]["seq_len"]
print("n windows", len(L), "len min/mean/max", L.min(), L.mean(), L.max(), "hist by 10 from 250:", np.bincount(L // 10)[25:40])
