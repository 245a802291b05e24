"""Dev tool: end-to-end differential stress (candidate -> fold -> predict) of the device pipeline against the CPU oracle on a larger
synthetic dataset than the test-suite uses; the oracle folds run in a process pool.
usage: python profiles/tools/stress_pipeline.py [n_loci] [seed] [n_samples] [vienna-2.1.2|vienna-1.8.5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import concurrent.futures as cf
import numpy as np
from mir_prefer_amd import capi, records, synth
from tests import oracle_binding
from tests.test_oracle_golden import mirna_record, run_predict

def fold_chunk(args):
    seqs, L, model = args
    o = oracle_binding.load()
    out = []
    for s in seqs:
        r = o.lfold(s, L, model=model)
        out.append(o.structures_from_lines(r["lines"], 55))
    return out

def main():
    n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    model = sys.argv[4] if len(sys.argv) > 4 else "vienna-2.1.2"
    ds = synth.make_dataset([2500000, 1500000, 2000000], n_loci, n_samples=ns, seed=seed, contig_names=["c9", "c10", "c1"], edge_cases=True)
    names, alns = ds.contig_names, ds.sorted_alns()
    cut, gap, L = 10, 100, 300
    order = np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)
    o = oracle_binding.load()
    depth, peaks = o.coverage_peaks(alns, ds.contig_lens, cut)
    win = o.make_windows(peaks, alns, ds.contigs, order, gap, L, cut * 0.5)
    ctx = capi.Context(0)
    ctx.set_fold_model(model)
    ctx.load_genome(ds.contigs); ctx.load_alignments(alns)
    npk, nloci, nwin = ctx.candidate(cut, gap, L, order)
    assert np.array_equal(ctx.get_depth(), depth) and np.array_equal(ctx.get_peaks(), peaks) and nwin == len(win["windows"])
    t0 = time.time()
    ctx.fold(L)
    print("fold %s: %d windows in %.2f s" % (model, nwin, time.time() - t0), flush=True)
    st = ctx.fold_status()
    if (st == 1).any():
        ctx.fold(L, max_lines=L + 52)
    assert (ctx.fold_status() == 0).all()
    for allow3, nostar in ((False, True), (True, False)):
        out = ctx.predict(ns, 18, 23, allow3, nostar)
        t = time.time()
        seqs = [win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes() for b in win["windows"]]
        ncpu = min(64, os.cpu_count() or 1)
        with cf.ProcessPoolExecutor(ncpu) as ex:
            res = list(ex.map(fold_chunk, [(seqs[i::ncpu], L, model) for i in range(ncpu)]))
        structs = [None] * len(seqs)
        for ci, c in enumerate(res):
            for k, s in enumerate(c): structs[ci + k * ncpu] = s
        case = {"cfg": {"MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 23, "ALLOW_3NT_OVERHANG": "Y" if allow3 else "N", "ALLOW_NO_STAR_EXPRESSION": "Y" if nostar else "N"},
                "win": win, "sample_names": ds.sample_names, "alns": alns}
        _, result = run_predict(case, o, structs)
        want = [mirna_record(m, names) for _, m in result]
        got = [[names[m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]), int(m["star_e"]), ss,
                records.STRAND[m["strand"]], bool(m["has_star"])] for m, ss in zip(out["result"], out["ss"])]
        ok = got == want
        print("allow_3nt=%s no_star=%s: windows %d, loci gpu %d / oracle %d, identical %s (oracle %.0f s)" % (allow3, nostar, nwin, len(got), len(want), ok, time.time() - t), flush=True)
        if not ok:
            for a, b in zip(got, want):
                if a != b: print("first difference:\n ", a, "\n ", b); break
            sys.exit(1)

if __name__ == "__main__":
    main()
