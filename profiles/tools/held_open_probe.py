"""Dev tool: what a HIP process pays for being the ONLY user of the GPU.  profiles/tools/bin/ctx_probe (bare HIP start-up, step by step, exit clocked by the
parent) run (a) with the GPU otherwise idle and (b) while this process holds a device context open.  usage: python profiles/tools/held_open_probe.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PROBE = os.path.join(ROOT, "profiles", "tools", "bin", "ctx_probe")
def run(label):
    for rep in range(4):
        t0 = time.time()
        r = subprocess.run([PROBE, "bare"], capture_output=True, text=True, cwd=ROOT)
        wall = time.time() - t0
        vals = {}
        for ln in r.stdout.splitlines():
            sp = ln.split()
            if len(sp) >= 3 and sp[-1] == "ms":
                vals[" ".join(sp[:-2])] = float(sp[-2])
        tot = vals.get("child total before exit", 0.0)
        print("%-28s hipInit %6.1f  stream %6.1f  child total %6.1f  exit %6.1f  wall %6.1f ms" % (label, vals.get("hipInit(0)", 0), vals.get("hipStreamCreateWithFlags", 0), tot, 1e3 * wall - tot, 1e3 * wall), flush=True)
run("GPU otherwise idle")
from mir_prefer_amd import capi
ctx = capi.Context(0)
time.sleep(0.5)
run("another process holds it")
ctx.close()
time.sleep(0.5)
run("idle again (holder closed)")
