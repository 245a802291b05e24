"""Dev tool: large differential stress of the HIP fold against the CPU oracle (structure lines bit-exact), with sequence families that
provoke energy ties (repeats, low complexity, GC-only, long N runs).  Oracle runs in a process pool.
usage: python profiles/tools/stress_fold.py [n_windows] [seed] [vienna-2.1.2|vienna-1.8.5]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import concurrent.futures as cf
from tests import seqgen

def family(r, k):
    n = r.randint(40, 350)
    if k == 0:   # plain mixed generator
        return seqgen.window(r, 60, 350)
    if k == 1:   # short tandem repeats
        unit = "".join(r.choice("ACGU") for _ in range(r.randint(1, 7)))
        s = (unit * (n // len(unit) + 1))[:n]
    elif k == 2: # two-letter alphabets
        ab = r.choice(["GC", "AU", "GU", "ACG", "AGU"])
        s = "".join(r.choice(ab) for _ in range(n))
    elif k == 3: # perfect / near-perfect long hairpins
        arm = r.randint(20, 160)
        a = "".join(r.choice("ACGU") for _ in range(arm))
        rc = {"A": "U", "C": "G", "G": "C", "U": "A"}
        b = [rc[c] for c in reversed(a)]
        for _ in range(r.randint(0, 6)): b[r.randrange(arm)] = r.choice("ACGU")
        s = a + "".join(r.choice("ACGU") for _ in range(r.randint(3, 12))) + "".join(b)
        s = s[:350]
    else:        # N-rich
        s = "".join(r.choice("ACGUN") if r.random() < 0.3 else r.choice("ACGU") for _ in range(n))
    return s

def oracle_chunk(args):
    seqs, span, model = args
    from tests import oracle_binding
    o = oracle_binding.load()
    return [o.lfold(s, span, model=model) for s in seqs]

def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    model = sys.argv[3] if len(sys.argv) > 3 else "vienna-2.1.2"
    r = random.Random(seed)
    seqs = [family(r, i % 5) for i in range(nw)]
    from tests import oracle_binding
    oracle_binding.load()
    from mir_prefer_amd import capi
    ctx = capi.Context(0)
    ctx.set_fold_model(model)
    bad = 0
    for span in (300, 150):
        t = time.time()
        got = ctx.fold_batch(seqs, span)
        tg = time.time() - t
        ncpu = min(64, os.cpu_count() or 1)
        chunks = [seqs[i::ncpu] for i in range(ncpu)]
        with cf.ProcessPoolExecutor(ncpu) as ex:
            res = list(ex.map(oracle_chunk, [(c, span, model) for c in chunks]))
        want = [None] * nw
        for ci, c in enumerate(res):
            for k, w in enumerate(c): want[ci + k * ncpu] = w
        capped = [k for k, g in enumerate(got) if g["status"] == 1]      # > 96 lines: fold those again with the full capacity (as the host does)
        if capped:
            again = ctx.fold_batch([seqs[k] for k in capped], span, max_lines=352)
            for k, g in zip(capped, again): got[k] = g
            print("span %d: %d windows exceeded 96 lines and were folded again with capacity 352" % (span, len(capped)), flush=True)
        for s, g, w in zip(seqs, got, want):
            if g["status"] != 0 or g["mfe"] != w["mfe"] or g["lines"] != w["lines"]:
                bad += 1
                if bad <= 5: print("MISMATCH span", span, s, g["mfe"], w["mfe"], flush=True)
        print("span %d: %d windows, gpu %.2f s, oracle %.1f s, mismatches so far %d" % (span, nw, tg, time.time() - t - tg, bad), flush=True)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
