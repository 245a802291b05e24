"""Dev tool: differential stress of the generic (global-workspace) fold kernels -- windows longer than 350 nt and spans above 300, where
the LDS-resident path does not apply -- against the oracles, both models.  usage: python profiles/tools/stress_generic.py [n] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import concurrent.futures as cf
import stress_fold

def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    base = [stress_fold.family(r, i % 5) for i in range(nw)]
    seqs = []
    for s in base:                       # stretch a third of them beyond 350 nt
        if r.random() < 0.35:
            s = (s + "".join(r.choice("ACGU") for _ in range(r.randint(10, 120))) + s[::-1])[:r.randint(351, 480)]
        seqs.append(s)
    from mir_prefer_amd import capi
    from tests import oracle_binding
    oracle_binding.load()
    ctx = capi.Context(0)
    bad = 0
    for model in ("vienna-2.1.2", "vienna-1.8.5"):
        ctx.set_fold_model(model)
        for span in (300, 420):
            t = time.time()
            got = ctx.fold_batch(seqs, span, max_lines=500)
            tg = time.time() - t
            ncpu = min(64, os.cpu_count() or 1)
            with cf.ProcessPoolExecutor(ncpu) as ex:
                res = list(ex.map(stress_fold.oracle_chunk, [(seqs[i::ncpu], span, model) for i in range(ncpu)]))
            want = [None] * len(seqs)
            for ci, c in enumerate(res):
                for k, w in enumerate(c): want[ci + k * ncpu] = w
            for s, g, w in zip(seqs, got, want):
                if g["status"] != 0 or g["mfe"] != w["mfe"] or g["lines"] != w["lines"]:
                    bad += 1
                    if bad <= 3: print("MISMATCH", model, span, g["status"], s[:100], flush=True)
            print("%s span %d: %d windows (%d longer than 350), gpu %.2f s, fallbacks %d, mismatches so far %d" % (
                model, span, len(seqs), sum(len(s) > 350 for s in seqs), tg, ctx.last_fold_fallbacks(), bad), flush=True)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
