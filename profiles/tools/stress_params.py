"""Dev tool: differential stress of the whole device path (candidate -> fold -> predict) against the CPU oracle chain under RANDOM SETTINGS --
READS_DEPTH_CUTOFF, MAX_GAP, PRECURSOR_LEN, MIN/MAX_MATURE_LEN, ALLOW_3NT_OVERHANG, ALLOW_NO_STAR_EXPRESSION, sample count, fold model -- on
small random datasets (the other stress tools keep the reference's defaults: MP:84-105).  Depth records, peaks, windows, the result list.
usage (GPU box): python profiles/tools/stress_params.py [n_trials] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import concurrent.futures as cf
import numpy as np
from mir_prefer_amd import capi, records, synth
from tests import oracle_binding
from tests.test_oracle_golden import mirna_record, run_predict


def fold_chunk(args):
    seqs, L, model, minlen = args
    o = oracle_binding.load()
    return [o.structures_from_lines(o.lfold(s, L, model=model)["lines"], minlen) for s in seqs]


def main():
    n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    o = oracle_binding.load()
    ncpu = min(32, len(os.sched_getaffinity(0)))
    bad = 0
    with cf.ProcessPoolExecutor(ncpu) as ex:
        for t in range(n_trials):
            cut = r.choice([2, 3, 5, 10, 10, 20, 40])
            gap = r.choice([20, 60, 100, 100, 200, 300])
            L = r.choice([100, 160, 200, 250, 300, 300, 320, 400])
            mn, mx = r.choice([16, 18, 18, 19, 20]), r.choice([21, 22, 23, 23, 24, 26])
            allow3, nostar = r.random() < 0.5, r.random() < 0.5
            ns = r.choice([1, 2, 3, 5])
            model = r.choice(["vienna-2.1.2", "vienna-2.1.2", "vienna-1.8.5"])
            nc = r.randint(1, 4)
            names = r.sample(["Chr1", "Chr2", "Chr10", "chrM", "scaffold_9", "b", "ctg.7"], nc)
            lens = [r.randint(60000, 500000) for _ in range(nc)]
            ds = synth.make_dataset(lens, r.randint(60, 500), n_samples=ns, seed=r.randint(1, 10 ** 6), contig_names=names, edge_cases=True)
            alns = ds.sorted_alns()
            order = np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)
            depth, peaks = o.coverage_peaks(alns, ds.contig_lens, cut)
            win = o.make_windows(peaks, alns, ds.contigs, order, gap, L, cut * 0.5)
            ctx = capi.Context(0)
            ctx.set_fold_model(model)
            ctx.load_genome(ds.contigs); ctx.load_alignments(alns)
            npk, nloci, nwin = ctx.candidate(cut, gap, L, order)
            ok = np.array_equal(ctx.get_depth(), depth) and np.array_equal(ctx.get_peaks(), peaks) and nwin == len(win["windows"])
            if ok and nwin:
                gw = ctx.get_windows()
                GW, OW = gw["windows"], win["windows"]
                ok = all(np.array_equal(GW[f], OW[f]) for f in ("tid", "ws", "we", "strand", "loc_s", "loc_e", "tag", "n_peaks", "n_matures", "seq_len"))
                for k in range(nwin if ok else 0):
                    a, b = GW[k], OW[k]
                    ok = ok and np.array_equal(gw["matures"][a["mature_off"]:a["mature_off"] + a["n_matures"]], win["matures"][b["mature_off"]:b["mature_off"] + b["n_matures"]])
            got = want = []
            if ok and nwin:
                ctx.fold(L)
                if (ctx.fold_status() == 1).any():
                    ctx.fold(L, max_lines=L + 52)
                ok = bool((ctx.fold_status() == 0).all())
                out = ctx.predict(ns, mn, mx, allow3, nostar)
                seqs = [win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes() for b in win["windows"]]
                res = list(ex.map(fold_chunk, [(seqs[i::ncpu], L, model, 55) for i in range(ncpu)]))
                structs = [None] * len(seqs)
                for ci, c in enumerate(res):
                    for k, s in enumerate(c): structs[ci + k * ncpu] = s
                case = {"cfg": {"MIN_MATURE_LEN": mn, "MAX_MATURE_LEN": mx, "ALLOW_3NT_OVERHANG": "Y" if allow3 else "N", "ALLOW_NO_STAR_EXPRESSION": "Y" if nostar else "N"},
                        "win": win, "sample_names": ds.sample_names, "alns": alns}
                _, result = run_predict(case, o, structs)
                want = [mirna_record(m, names) for _, m in result]
                got = [[names[m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]), int(m["star_e"]), ss,
                        records.STRAND[m["strand"]], bool(m["has_star"])] for m, ss in zip(out["result"], out["ss"])]
                ok = ok and got == want
            print("trial %2d: cut %2d gap %3d L %3d mature %d-%d 3nt %d nostar %d samples %d %s contigs %d: windows %5d, loci %4d / %4d -> %s"
                  % (t, cut, gap, L, mn, mx, allow3, nostar, ns, model, nc, nwin, len(got), len(want), "identical" if ok else "DIFFERENT"), flush=True)
            bad += 0 if ok else 1
            del ctx
    print("stress_params: %d trials, %d different" % (n_trials, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
