"""Dev tool: when does small-file creation on this file system turn slow?  Rounds of 4,000 creates (1.1 KB each) in a fresh directory, (a) keeping
everything, (b) deleting the previous round first, (c) after a pause.  usage: python profiles/tools/fs_regime.py [base dir]"""
import os, shutil, sys, tempfile, time
base = tempfile.mkdtemp(prefix="fsreg_", dir=sys.argv[1] if len(sys.argv) > 1 else None)
body = b"A" * 1100
def rnd(d):
    os.makedirs(d)
    t = time.time()
    for k in range(4000):
        fd = os.open(os.path.join(d, "miRNA-precursor_%d.map.txt" % k), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.write(fd, body); os.close(fd)
    return (time.time() - t) / 4000 * 1e6
print("keep:    ", " ".join("%.0f" % rnd(os.path.join(base, "k%d" % i)) for i in range(8)), "us/file")
out = []
for i in range(8):
    if i: t = time.time(); shutil.rmtree(os.path.join(base, "d%d" % (i - 1))); dt = time.time() - t
    out.append("%.0f" % rnd(os.path.join(base, "d%d" % i)))
print("delete:  ", " ".join(out), "us/file   (last rmtree %.3f s)" % dt)
time.sleep(6)
print("after 6 s:", " ".join("%.0f" % rnd(os.path.join(base, "p%d" % i)) for i in range(3)), "us/file")
t = time.time(); shutil.rmtree(base); print("rmtree of everything %.2f s" % (time.time() - t))
print("fresh base:", " ".join("%.0f" % rnd(os.path.join(tempfile.mkdtemp(prefix="fsreg2_", dir=sys.argv[1] if len(sys.argv) > 1 else None), "x")) for i in range(3)), "us/file")
