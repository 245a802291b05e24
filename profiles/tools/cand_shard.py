"""Dev tool: the candidate stage alone at the size of a config[4] rank shard (8 x 31.25 Mb, 2.5e7 packed records), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ctx = capi.Context(0)
rng = np.random.RandomState(4)
contigs = [("ctg%02d" % t, synth._BASES[rng.randint(0, 4, size=31250000, dtype=np.uint8)]) for t in range(8)]
alns = synth.packed_records_shard(8, 31250000, 150000, 167, seed=44)
ctx.load_genome(contigs); ctx.load_alignments(alns)
o = np.arange(8, dtype=np.int32)
for _ in range(6):
    print(ctx.candidate(10, 100, 300, o), ctx.last_timings(), flush=True)
