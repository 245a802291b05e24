"""Process-level wall-clock of the CLI `pipeline` verb on a bench workload (bench.e2e_process: a fresh process per run, parent-side clock), with the
child's own segment stamps.  usage (GPU box): python profiles/tools/cli_wall.py [config1|config2] [runs] [base-dir|''] [pause seconds between runs] > gpurun_out/cli_wall.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mir_prefer_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "config1"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
base = (sys.argv[3] if len(sys.argv) > 3 else None) or None
pause = float(sys.argv[4]) if len(sys.argv) > 4 else 1.2
specs, ns, bg, _, _ = bench.workload_specs(wl, 1)
contigs, alns, samples = bench.build_shard(specs, set(range(len(specs))), ns, bg)
r = bench.e2e_process(synth.Dataset(contigs, samples, alns, []), "vienna-2.1.2", base, runs, pause_s=pause)
print(json.dumps(r, indent=1))
bench.e2e_cleanup()
