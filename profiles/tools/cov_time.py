"""Dev tool: coverage stage (scatter + scan + clean-up) time and HBM rate at config[1] and at a cfg[4]-shard size (8 x 31.25 Mb, 2.5e7 records)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mir_prefer_amd import synth, capi
ctx = capi.Context(0)
if os.environ.get("MIRP_COV_FUSED") in ("0", "1"):          # tools only: pin the coverage path
    ctx.set_coverage_path(int(os.environ["MIRP_COV_FUSED"]))
def run(tag, contigs, alns, order):
    ctx.load_genome(contigs); ctx.load_alignments(alns)
    ms = []
    for _ in range(6):
        ctx.candidate(10, 100, 300, order); ms.append(ctx.last_timings()["coverage_ms"])
    g = sum(len(s) + 1 for _, s in contigs)
    b = 16.0 * len(alns) + 16.0 * g
    print("%s: coverage %.3f ms (min %.3f) -> %.2f TB/s of B_cov = 16A + 16G = %.2f GB (%.3f of 8 TB/s)" % (tag, np.mean(ms[1:]), min(ms), b / min(ms) / 1e9, b / 1e9, b / min(ms) / 1e9 / 8), flush=True)
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
run("config[1]", ds.contigs, ds.sorted_alns(), np.zeros(1, np.int32))
from tests.test_configs_gpu import _packed_records_shard
rng = np.random.RandomState(4)
contigs = [("ctg%02d" % t, synth._BASES[rng.randint(0, 4, size=31250000, dtype=np.uint8)]) for t in range(8)]
run("cfg[4] shard", contigs, _packed_records_shard(8, 31250000, 150000, 167, seed=44), np.arange(8, dtype=np.int32))
