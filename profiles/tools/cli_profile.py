"""Where the host time of a fresh CLI `pipeline` process goes: writes the files of a bench workload, then runs the CLI under cProfile in a child process
and prints the top of the cumulative listing.  usage (GPU box): python profiles/tools/cli_profile.py [config1|config2] [extra CLI flags] > gpurun_out/cli_profile.txt"""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mir_prefer_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "config1"
specs, ns, bg, _, _ = bench.workload_specs(wl, 1)
contigs, alns, samples = bench.build_shard(specs, set(range(len(specs))), ns, bg)
ds = synth.Dataset(contigs, samples, alns, [])
tmp = tempfile.mkdtemp(prefix="mirp_prof_")
try:
    sams = ds.write_sams(tmp)
    fa = os.path.join(tmp, "genome.fa")
    ds.write_fasta(fa)
    cfg = os.path.join(tmp, "config")
    open(cfg, "w").write("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %s\nNAME_PREFIX = bench\nPRECURSOR_LEN = 300\nREADS_DEPTH_CUTOFF = 10\nMAX_GAP = 100\n"
                         % (fa, ", ".join(sams), os.path.join(tmp, "out")))
    env = dict(os.environ, PYTHONPATH=ROOT)
    for rep in range(2):
        shutil.rmtree(os.path.join(tmp, "out"), ignore_errors=True)
        r = subprocess.run([sys.executable, "-m", "cProfile", "-s", "cumtime", "-m", "mir_prefer_amd.cli"] + sys.argv[2:] + ["pipeline", cfg], env=env, cwd=tmp,
                           capture_output=True, text=True)
    lines = r.stdout.splitlines()
    k = next(i for i, l in enumerate(lines) if "function calls" in l)
    print("\n".join(lines[k:k + 90]))
    print(r.stderr[-2000:])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
