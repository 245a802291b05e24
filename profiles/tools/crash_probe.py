import sys, subprocess, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
if len(sys.argv) > 1:
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    import seqgen
    from mir_prefer_amd import capi
    ws = seqgen.windows(11, 200, 5, 120)[lo:hi]
    ctx = capi.Context(0)
    r = ctx.fold_batch(ws, 300)
    print("ok", lo, hi, [len(w) for w in ws][:10])
else:
    def run(lo, hi):
        p = subprocess.run([sys.executable, __file__, str(lo), str(hi)], capture_output=True, text=True)
        return p.returncode == 0
    lo, hi = 0, 200
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if not run(lo, mid): hi = mid
        elif not run(mid, hi): lo = mid
        else: print("both halves ok", lo, mid, hi); break
    print("culprit range", lo, hi)
    import seqgen
    print(seqgen.windows(11, 200, 5, 120)[lo:hi])
