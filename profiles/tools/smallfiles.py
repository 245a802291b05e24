"""How long 4,002 small files (the read-mapping files of config[1], ~1.1 KB each) take to create: Python loop vs mirp_write_files, /tmp vs /dev/shm.
usage: python profiles/tools/smallfiles.py"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mir_prefer_amd import capi
texts = [(">miRNA-precursor_%d chr:1-2 +\n" % k) + "ACGU" * 280 for k in range(4002)]
for base in (tempfile.gettempdir(), "/dev/shm"):
    for how in ("python", "native", "python", "native"):
        d = tempfile.mkdtemp(prefix="sf_", dir=base)
        paths = [os.path.join(d, "miRNA-precursor_%d.map.txt" % k) for k in range(len(texts))]
        t = time.time()
        if how == "python":
            for p, x in zip(paths, texts):
                fd = os.open(p, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                os.write(fd, x.encode()); os.close(fd)
        else:
            capi.write_files(paths, texts, int(os.environ.get("MIRP_FILE_THREADS", "1")))
        w = time.time() - t
        t = time.time(); shutil.rmtree(d); r = time.time() - t
        print("%-10s %-7s threads=%s create %.3f s  rmtree %.3f s" % (base, how, os.environ.get("MIRP_FILE_THREADS", "default"), w, r))
