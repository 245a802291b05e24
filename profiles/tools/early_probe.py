"""Dev tool: how long mirp_create takes inside a Python process, alone and next to what the CLI does at the same time (numpy import, genome read).
usage: python profiles/tools/early_probe.py <mode> [fasta]   mode: alone | numpy | numpy_fasta      (one mode per process: a process opens the device once)"""
import os, sys, time
t00 = time.time()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mir_prefer_amd import early
mode = sys.argv[1]
t0 = time.time()
early.start_context(0)
if mode == "numpy_fasta" and len(sys.argv) > 2:
    early.start_fasta(sys.argv[2])
if mode != "alone":
    import numpy
t1 = time.time()
rc, h = early.take_context(0)
t2 = time.time()
print("%-12s start at %.3f s after process start, numpy import %.3f s, join wait %.3f s, mirp_create done %.3f s after its start (rc %d)" % (mode, t0 - t00, t1 - t0, t2 - t1, t2 - t0, rc))
sys.stdout.flush(); os._exit(0)
