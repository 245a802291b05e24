import sys, time
sys.path.insert(0, '.')
from mir_prefer_amd import capi
seqs = ["A", "ACGU", "GGGGAAAACCCC", "A" * 24, "N" * 30, "GGGAAAUCCCGGGAAAUCCCAAAAGGGGGGAUUUCCCCCCUUUUGGGAUUUCCCGGAUUUCCC", "GC" * 150, "G" * 150 + "C" * 150, ""]
k = int(sys.argv[1])
ctx = capi.Context(0)
t = time.time()
r = ctx.fold_batch([seqs[k]], 300)
print(k, len(seqs[k]), "ok", r[0]["mfe"], len(r[0]["lines"]), "%.2fs" % (time.time() - t), flush=True)
