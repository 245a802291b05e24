import sys, time
sys.path.insert(0, '.')
from mir_prefer_amd import capi
seqs = ["A", "ACGU", "GGGGAAAACCCC", "A" * 24, "N" * 30, "GGGAAAUCCCGGGAAAUCCCAAAAGGGGGGAUUUCCCCCCUUUUGGGAUUUCCCGGAUUUCCC", "GC" * 150, "G" * 150 + "C" * 150, ""]
idx = [int(x) for x in sys.argv[1].split(",")]
ctx = capi.Context(0)
t = time.time()
r = ctx.fold_batch([seqs[k] for k in idx], 300)
print(idx, "ok", [x["mfe"] for x in r], "%.2fs" % (time.time() - t), flush=True)
