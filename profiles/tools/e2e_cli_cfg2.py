"""Dev tool: end-to-end wall-clock of the CLI `pipeline` verb on a BASELINE config[1]-sized synthetic dataset (files in, gff3 out)."""
import os, sys, time, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mir_prefer_amd import synth, cli
t = time.time()
ds = synth.make_dataset([30427671], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
d = tempfile.mkdtemp(prefix="mirp_e2e_")
ds.write_fasta(os.path.join(d, "genome.fa")); sams = ds.write_sams(d)
print("dataset written in %.1f s: %s" % (time.time() - t, d), flush=True)
cfg = os.path.join(d, "config")
open(cfg, "w").write("FASTA_FILE = %s/genome.fa\nALIGNMENT_FILE = %s\nOUTFOLDER = %s/out\nNAME_PREFIX = cfg2\nPRECURSOR_LEN = 300\nREADS_DEPTH_CUTOFF = 10\n" % (d, sams[0], d))
t = time.time()
if len(sys.argv) > 1:
    cProfile.run('cli.main(["-k", "pipeline", cfg])', os.path.join(d, "prof"))
    pstats.Stats(os.path.join(d, "prof")).sort_stats("cumulative").print_stats(25)
else:
    cli.main(["-k", "pipeline", cfg])
print("CLI pipeline end-to-end: %.2f s" % (time.time() - t))
