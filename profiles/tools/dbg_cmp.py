import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
which = sys.argv[1]
from mir_prefer_amd import capi
if which == "old":
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libmirprefer_old.so")
os.makedirs("gpurun_out", exist_ok=True)
os.environ["MIRP_FOLD_DUMP"] = "gpurun_out/slab_%s.bin" % which
seqs = [sys.argv[2]]
ctx = capi.Context(0)
got = ctx.fold_batch(seqs, 300)
print(which, got[0]["mfe"])
