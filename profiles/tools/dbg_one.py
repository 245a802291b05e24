import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mir_prefer_amd import capi
from tests import oracle_binding
o = oracle_binding.load()
ctx = capi.Context(0)
for s in sys.argv[2:]:
    span = int(sys.argv[1])
    g = ctx.fold_batch([s], span)[0]
    w = o.lfold(s, span)
    print("status", g["status"], "mfe", g["mfe"], w["mfe"], "nlines", len(g["lines"]), len(w["lines"]))
    for k in range(max(len(g["lines"]), len(w["lines"]))):
        a = g["lines"][k] if k < len(g["lines"]) else None
        b = w["lines"][k] if k < len(w["lines"]) else None
        if a != b:
            print("line", k); print("  gpu", a); print("  cpu", b)
            break
