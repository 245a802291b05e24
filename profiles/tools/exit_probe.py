"""Dev tool: does where a process wrote its files change what its EXIT costs the parent?  The CLI's `exit` segment (end of main() -> process reaped) is 0.09-0.11 s
with the output folder on the overlay /tmp and 0.01 s on /dev/shm.  A child without any GPU work writes N small files + one 5 MB file under <dir>, stamps the time and
leaves through os._exit; the parent clocks stamp -> reaped.  usage: python profiles/tools/exit_probe.py"""
import os, subprocess, sys, tempfile, time, shutil
CHILD = r'''
import os, sys, time
d, n, hip = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
if hip:
    sys.path.insert(0, sys.argv[4])
    from mir_prefer_amd import capi
    ctx = capi.Context(0)
os.makedirs(os.path.join(d, "readmapping"))
body = b"A" * 1100
for k in range(n):
    fd = os.open(os.path.join(d, "readmapping", "f%d.txt" % k), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644); os.write(fd, body); os.close(fd)
with open(os.path.join(d, "big.html"), "wb") as f: f.write(b"x" * (5 << 20))
sys.stdout.write("%.6f\n" % time.time()); sys.stdout.flush()
os._exit(0)
'''
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for base in (None, "/dev/shm"):
    for hip in ("0", "1"):
        for n in (0, 4002):
            outs = []
            for rep in range(3):
                d = tempfile.mkdtemp(prefix="exitp_", dir=base)
                r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(d, "o"), str(n), hip, ROOT], capture_output=True, text=True)
                t1 = time.time()
                outs.append(t1 - float(r.stdout.strip().splitlines()[-1]))
            print("%-9s hip=%s files=%-5d exit (stamp -> reaped): %s ms" % (base or tempfile.gettempdir(), hip, n, " ".join("%.1f" % (1e3 * x) for x in outs)), flush=True)
