"""Process-level wall-clock of the CLI `pipeline` verb at the size of ONE RANK'S SHARE of BASELINE config[4]: 8 contigs x 31.25 Mb (250 Mb FASTA), 2.5e7 alignment
records (1.9 GB of SAM text), ~250 k windows, ~51 k loci.  Fresh processes, parent-side clock, isolated runs (bench.e2e_process with the vectorised SAM writer).
usage (GPU box): python profiles/tools/cli_wall_shard.py [runs] > gpurun_out/cli_wall_shard.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from mir_prefer_amd import dist as mdist, synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
specs, ns, bg, _, desc = bench.workload_specs("config4", 8)
owned = mdist.partition_contigs([sp[1] for sp in specs], 8)[0]
t = time.time()
contigs, alns, samples = bench.build_shard(specs, set(owned), ns, bg)
keep = [k for k, (_, sq) in enumerate(contigs) if len(sq)]
remap = -np.ones(len(contigs), dtype=np.int64); remap[keep] = np.arange(len(keep))
alns = alns.copy(); alns["tid"] = remap[alns["tid"]]
contigs = [contigs[k] for k in keep]
sys.stderr.write("shard built in %.0f s: %d contigs, %d records\n" % (time.time() - t, len(contigs), len(alns)))


class FastDataset(synth.Dataset):
    def write_sams(self, outdir, sq_order=None):
        names, lens = self.contig_names, self.contig_lens
        paths = []
        for si, sname in enumerate(self.sample_names):
            p = os.path.join(outdir, sname + ".sam")
            synth.write_sam_fast(p, sname, self.alns[self.alns["sample"] == si], names, lens)
            paths.append(p)
        return paths


r = bench.e2e_process(FastDataset(contigs, samples, alns, []), "vienna-2.1.2", None, runs, back_to_back=0)
r["workload"] = "one rank's share of " + desc
print(json.dumps(r, indent=1))
bench.e2e_cleanup()
