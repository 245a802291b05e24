#!/usr/bin/env python3
"""Dev tool (test infrastructure): write the first N window sequences of the BASELINE config[1] synthetic workload (seed 2), one per line,
as the CPU oracle's candidate stage produces them.  Input of tests/tools/splitcand_gate.c."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tests import oracle_binding

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/bench_windows.txt"
    specs, ns, bg, _, _ = bench.workload_specs("config1", 1)
    contigs, alns, names = bench.build_shard(specs, {0}, ns, bg)
    from mir_prefer_amd import synth
    o = oracle_binding.load()
    lens = np.array([len(s) for _, s in contigs], dtype=np.int64)
    _, peaks = o.coverage_peaks(alns, lens, bench.CUT)
    win = o.make_windows(peaks, alns, contigs, list(range(len(contigs))), bench.GAP, bench.L, bench.CUT * 0.5)
    W = win["windows"]
    step = max(1, len(W) // n)
    with open(out, "w") as f:
        for k in range(0, len(W), step):
            b = W[k]
            f.write(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes().decode() + "\n")
    print(len(W), "windows; wrote", len(range(0, len(W), step)))
main()
