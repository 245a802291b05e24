import sys, time
sys.path.insert(0, '.')
from mir_prefer_amd import capi
from tests import oracle_binding, seqgen
o = oracle_binding.load(); ctx = capi.Context(0)
for (seed,cnt,lo,hi,span) in [(1,100,5,120,300),(2,50,60,200,40),(3,64,300,350,300)]:
    seqs = seqgen.windows(seed,cnt,lo,hi)
    t=time.time(); got = ctx.fold_batch(seqs, span); dt=time.time()-t
    bad=0
    for s,g in zip(seqs,got):
        w=o.lfold(s,span)
        if g['status']!=0 or g['mfe']!=w['mfe'] or g['lines']!=w['lines']:
            bad+=1
            if bad<=3: print("MISMATCH", s[:60], g['status'], g['mfe'], w['mfe'], len(g['lines']), len(w['lines'])); 
            if bad<=1:
                for a,b in zip(g['lines'],w['lines']): print(a==b, a, b)
    print("seed",seed,"n",cnt,"bad",bad,"time %.3fs"%dt)
seqs = seqgen.windows(7,2048,300,300)
for rep in range(2):
    t=time.time(); got = ctx.fold_batch(seqs, 300); dt=time.time()-t
    print("2048 windows n=300: %.3fs -> %.0f windows/s (incl. PCIe/host)"%(dt, 2048/dt))
