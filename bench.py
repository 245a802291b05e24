#!/usr/bin/env python3
"""Benchmark of the candidate -> fold -> predict hot path on MI355X.

Metric (BASELINE.json): precursor windows folded+filtered per second at L = 300, inputs resident in HBM; end-to-end wall-clock beside it.
A step = one pass of the whole hot path (coverage scan -> peaks -> windows -> payload -> local fold -> filter -> loci list -> gather of the loci
list) over one synthetic batch.  Workloads (`--workload`, numbered like BASELINE.json `configs`):
  config1 (default, the headline) A. thaliana chr1-sized contig (30,427,671 bp), 1 sample, 12,000 synthetic loci (~20 k windows) PER GPU:
          weak scaling, rank r owns contig Chr<r+1>;
  config2 TAIR10-sized genome, 5 contigs, 3 samples (~72 k windows);
  config3 MSU7-sized genome, 12 contigs (373 Mb), 4 samples (~240 k windows), contigs dealt to the ranks longest-first;
  config4 64 contigs x 31.25 Mb = 2 Gb, 2 x 10^8 packed alignment records, ~2 M windows (meant for 8 GPUs: a rank's shard is 8 contigs).
config2-4 are fixed-size jobs sharded by contig ("strong" scaling); every rank generates and holds only its own contigs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload configX]      (N > 1: re-launches itself under torch.distributed.run)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU; host objects and the timing barrier go over a CPU-side `gloo` group, the data path's one exchange step (the
gather of the loci list, mirp_gather_loci) over RCCL on the library's own communicator.  torch never touches the GPU in this process, so it
holds a single HIP runtime; every C-ABI call returns after its stream has drained, which is the device synchronisation of the timed region.

The JSON line carries, beside the contract's keys:
  roofline       the dominant kernel (fold_lds_kernel, the dynamic program) against the roof that bounds it -- integer min-plus relaxations
                 out of LDS: `peak` is the figure of MI355X_MICROARCH.md (75 TB/s of ds_read_b32 = 9.4e12 relaxations/s at 8 B per relaxation) and
                 `frac` the fraction of it; `peak_measured` / `frac_measured` use the conflict-free ds_read_b32 / VALU rates micro-benchmarked on this
                 GPU (mirp_microbench); `pipe_busy` = LDS / VALU / SALU busy fractions from the committed PMC passes;
  roofline_hbm   the HBM view north_star asks for (algorithmic bytes / measured time / 8 TB/s), expected << 1 for an LDS-resident DP;
  roofline_coverage  the one HBM-bound stage on the headline workload (sparse input: atomic scatter + scan + clearing of the written positions);
                 configs.coverage_config4_shard = the same stage at a config[4] rank shard's size, where the fused scan runs (tiles built from the
                 sorted records in LDS; --no-cov-shard skips it);
  configs        (N = 1, default workload) config2 on the same GPU, the vienna-1.8.5 model, and the fold micro-benchmark of SURVEY.md 8d
                 (2^16 windows, n = L = 300: uniform / 50 % planted hairpins / GC = 0.65) with generic-fallback counts per family;
  cpu_baseline   the CPU oracle (the build's own restatement of the reference, "port") timed on this box AFTER the GPU timing;
  e2e            the CLI `pipeline` verb on files of the same workload (SAM + FASTA in -> gff3 and reports out), wall-clock and device time by stage.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR1_LEN = 30427671
N_LOCI = 12000
CUT, GAP, L = 10, 100, 300
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LDS_GUIDE_RELAX_PER_S = 75e12 / 8.0      # MI355X_MICROARCH.md (LDS): ~75 TB/s aggregate for ds_read_b32; one relaxation = two 4-byte reads
LOCI_PER_BP = N_LOCI / float(CHR1_LEN)   # every workload plants loci at the density of the headline one
# Result sizes of the seeded workloads, per (workload, fold model): the number of miRNA loci the whole path reports.  tests/test_whole_workload_gpu.py
# pins the config1 / vienna-2.1.2 list record by record on the CPU oracle; the bench refuses to print a throughput whose result list has another size
# (a silent filter or fold regression must not produce a headline number).
EXPECTED_LOCI = {("config1", "vienna-2.1.2"): 4002, ("config1", "vienna-1.8.5"): 3973, ("config2", "vienna-2.1.2"): 16016}
TAIR10 = [30427671, 19698289, 23459830, 18585056, 26975502]
MSU7 = [43300000, 35900000, 36400000, 35500000, 30000000, 31200000, 29700000, 28400000, 23000000, 23200000, 29000000, 27500000]


def workload_specs(name, world, genome=CHR1_LEN, loci=N_LOCI, scale=1.0):
    """-> (contig specs [(name, length, loci, seed)], n_samples, background records per contig, scaling, description).  scale (tests only): contig
    lengths, loci and background records of config2-4 multiplied by it -- the same shape at a size a test can afford."""
    if scale != 1.0 and name != "config1":
        specs, ns, bg, scaling, desc = workload_specs(name, world, genome, loci)
        return ([(n, max(20000, int(l * scale)), max(4, int(k * scale)), sd) for n, l, k, sd in specs], ns, int(bg * scale), scaling,
                desc + " -- SCALED by %g (test size, not a benchmark configuration)" % scale)
    if name == "config1":
        return ([("Chr%d" % (r + 1), genome, loci, 2 + r) for r in range(world)], 1, 0, "weak",
                "BASELINE config[1]: A. thaliana chr1-sized contig per GPU (%d bp), 1 sample, L=300, %d synthetic loci per contig" % (genome, loci))
    if name == "config2":
        return ([("Chr%d" % (t + 1), l, int(round(l * LOCI_PER_BP)), 30 + t) for t, l in enumerate(TAIR10)], 3, 0, "strong",
                "BASELINE config[2]: TAIR10-sized genome (5 contigs, 119,146,348 bp), 3 samples, L=300")
    if name == "config3":
        return ([("Chr%d" % (t + 1), l, int(round(l * LOCI_PER_BP)), 50 + t) for t, l in enumerate(MSU7)], 4, 0, "strong",
                "BASELINE config[3]: MSU7-sized genome (12 contigs, 373 Mb), 4 samples, L=300, contigs dealt to the ranks longest-first")
    if name == "config4":
        return ([("ctg%02d" % t, 31250000, 19000, 100 + t) for t in range(64)], 1, 3125000, "strong",
                "BASELINE config[4]: 64 contigs x 31.25 Mb = 2 Gb, 2e8 packed alignment records, L=300, contigs dealt to the ranks")
    raise SystemExit("unknown workload " + name)


def build_shard(specs, owned, n_samples, background):
    """The contigs `owned` of a workload as this rank holds them: contig table over ALL contigs (the others with length 0, tids are genome-wide),
    records of the owned contigs sorted by (tid, pos) in sample order.  Every contig is generated from its own seed, so the union over the ranks
    does not depend on how many ranks there are."""
    import numpy as np
    from mir_prefer_amd import synth
    contigs, parts = [], []
    for t, (name, length, loci, seed) in enumerate(specs):
        if t not in owned:
            contigs.append((name, np.zeros(0, dtype=np.uint8)))
            continue
        ds = synth.make_dataset([length], loci, n_samples=n_samples, seed=seed, contig_names=[name])
        a = ds.alns
        if background:      # config4: packed records generated directly -- single reads scattered over the contig, far below the depth cutoff
            rng = np.random.RandomState(seed + 7919)
            b = np.zeros(background - len(a), dtype=synth.ALN_DTYPE)
            b["pos"] = rng.randint(1, length - 30, size=len(b))
            b["depth"] = 1
            b["len"] = rng.randint(18, 26, size=len(b))
            b["strand"] = rng.randint(0, 2, size=len(b))
            b["sample"] = rng.randint(0, n_samples, size=len(b))
            a = np.concatenate([a, b])
        a = a[np.lexsort((np.arange(len(a)), a["sample"]))]
        a = a[np.argsort(a["pos"], kind="stable")]
        a["tid"] = t
        parts.append(a)
        contigs.append((name, ds.contigs[0][1]))
    alns = np.concatenate(parts) if parts else np.zeros(0, dtype=synth.ALN_DTYPE)
    return contigs, alns, ["S%d" % (k + 1) for k in range(n_samples)]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)          # the first two calls of a stage pay one-off costs (code objects, staging buffers): 9 ms on the filter
    ap.add_argument("--workload", default="config1", choices=["config1", "config2", "config3", "config4", "cfg1", "cfg2", "cfg3", "cfg4"],
                    help="BASELINE.json configs[k] (cfgK = configK); config1 is the headline and the default")
    ap.add_argument("--loci", type=int, default=N_LOCI, help="config1: loci per contig")
    ap.add_argument("--genome", type=int, default=CHR1_LEN, help="config1: contig length")
    ap.add_argument("--genome-scale", type=float, default=1.0, help="tests only: config2-4 at this fraction of their size (the line says so; no expected-result check)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (config2, vienna-1.8.5, fold micro-benchmark)")
    ap.add_argument("--micro-windows", type=int, default=1 << 16, help="windows per family of the fold micro-benchmark")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-runs", type=int, default=3, help="fresh CLI processes per end-to-end leg (the first one is the reported figure)")
    ap.add_argument("--no-cov-shard", action="store_true", help="skip the coverage-stage measurement at a config[4] rank shard's size (configs.coverage_config4_shard)")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of wall-clock per CPU-baseline leg")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--ingest-records", type=int, default=8000000,
                    help="records of the synthetic SAM of the ingest leg: 8,000,000 = about a third of a config[4] rank shard (25,000,000 records = 1.9 GB of "
                         "SAM text to write first; pass 25000000 for the full shard)")
    ap.add_argument("--allow-gloo", action="store_true", help="--gpus N: if the RCCL communicator does not come up, gather the loci lists as host objects over "
                    "gloo and still print a line (marked in config.exchange); without this flag the run exits non-zero instead")
    ap.add_argument("--fold-model", default="vienna-2.1.2", choices=["vienna-2.1.2", "vienna-1.8.5"],
                    help="RNALfold flavour to reproduce (the headline metric is quoted on the default, Turner-2004)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = checker, timed as the "port" baseline; never on the product path)
# ---------------------------------------------------------------------------------------------------------------------------------
_CPU = {}


def _cpu_init(win, alns, sample_names):
    from tests import oracle_binding
    _CPU["o"] = oracle_binding.load()
    _CPU["win"], _CPU["alns"], _CPU["samples"] = win, alns, sample_names


def _cpu_fold_filter(idx):
    """fold (oracle/lfold.c) + structure list + check_loci (duplex + expression + decision, oracle/predict.c) of the windows idx."""
    o, win, alns = _CPU["o"], _CPU["win"], _CPU["alns"]
    params = (len(_CPU["samples"]), 18, 23, 0, 1, 55)
    t0 = time.time()
    t_fold = t_filt = 0.0
    for k in idx:
        b = win["windows"][k]
        a = time.time()
        r = o.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L)
        c = time.time()
        st = o.structures_from_lines(r["lines"], 55)
        mats = win["matures"][b["mature_off"]:b["mature_off"] + b["n_matures"]]
        o.check_loci(st, mats, b, alns, params)
        d = time.time()
        t_fold += c - a; t_filt += d - c
    return len(idx), time.time() - t0, t_fold, t_filt


def _cpu_info():
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid)); pid = cid = None
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_phys = len(phys) if phys else max(1, usable // 2)
    return model, min(n_phys, usable), usable


def _cpu_quota():
    """CPUs the cgroup grants this process (cpu.max), or None when unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else float(q) / p
    except Exception:
        return None


def cpu_baseline(ds, alns, order, budget_s):
    import multiprocessing as mp
    from tests import oracle_binding
    o = oracle_binding.load()
    model, phys, logical = _cpu_info()
    # leg 0 (1 thread): candidate stage over the whole input (coverage -> peaks -> windows -> payload)
    t = time.time()
    _, peaks = o.coverage_peaks(alns, ds.contig_lens, CUT)
    win = o.make_windows(peaks, alns, ds.contigs, order, GAP, L, CUT * 0.5)
    t_cand = time.time() - t
    nwin = len(win["windows"])
    # leg 1 (1 thread): fold + filter on a bounded sample
    _cpu_init(win, alns, ds.sample_names)
    n1, t1, f1, p1 = _cpu_fold_filter(range(min(8, nwin)))
    per = t1 / max(n1, 1)
    m1 = max(8, min(nwin, int(budget_s / per)))
    n1, t1, f1, p1 = _cpu_fold_filter(range(m1))
    # leg 2: one worker process per usable physical core (spawned: the parent holds a GPU context), small tasks handed out dynamically and the
    # leg cut off at the time budget, so a box whose cgroup grants fewer CPUs than it shows cannot stretch the run
    quota = _cpu_quota()
    nproc = max(1, min(phys, int(quota + 0.5)) if quota else phys)
    tasks = [list(range(k, min(k + 4, nwin))) for k in range(0, nwin, 4)]
    ctx = mp.get_context("spawn")
    res = []
    with ctx.Pool(nproc, initializer=_cpu_init, initargs=(win, alns, ds.sample_names)) as pool:
        pool.map(_cpu_fold_filter, [t[:1] for t in tasks[:nproc]])          # start-up (imports, library load) outside the timing
        t = time.time()
        for r in pool.imap_unordered(_cpu_fold_filter, tasks):
            res.append(r)
            wall = time.time() - t
            if wall >= budget_s:
                break
        pool.terminate()
    chunks = list(range(nproc))
    n2 = sum(r[0] for r in res)
    cand_share = t_cand * n2 / max(nwin, 1)       # the candidate stage is serial in the port; charge the sample its share
    return {"value": n2 / (wall + cand_share), "unit": "windows/s", "cores": len(chunks), "kind": "port",
            "sample": "CPU oracle (oracle/*.c: candidate.c + lfold.c Turner-2004 d2 + predict.c) on the same workload: candidate stage over the whole input "
                      "(1 thread, %.2f s for %d windows); fold + filter on %d windows handed out 4 at a time to one process per usable physical core (%d), %.1f s wall; "
                      "1-thread leg on the first %d windows, %.1f s" % (t_cand, nwin, n2, len(chunks), wall, n1, t1),
            "cpu_model": model, "physical_cores": phys, "logical_cpus": logical, "cgroup_cpu_quota": quota,
            "one_thread": {"value": n1 / (t1 + t_cand * n1 / max(nwin, 1)), "unit": "windows/s", "windows": n1, "fold_s_per_window": f1 / n1, "filter_s_per_window": p1 / n1},
            "all_cores": {"windows": n2, "wall_s": wall, "per_process_windows_per_s": n2 / sum(r[1] for r in res),
                          "fold_s_per_window": sum(r[2] for r in res) / n2, "filter_s_per_window": sum(r[3] for r in res) / n2},
            "candidate_stage_s_1thread": t_cand,
            # the box has more physical cores than this container is granted: the measured per-process rate times ALL of them, with the serial candidate
            # stage charged once -- an upper bound on the port's rate on the whole box (perfect scaling assumed), next to the measured figure above
            "whole_box_extrapolation": {"cores": phys, "windows_per_s": (nwin / (nwin / ((n2 / sum(r[1] for r in res)) * phys) + t_cand)) if phys and n2 else None,
                                        "note": "extrapolated, not measured: per-process fold+filter rate x physical cores of the box, candidate stage serial"}}


# ---------------------------------------------------------------------------------------------------------------------------------
# exact relaxation count of a batch: R = R_ml + R_int + R_f3 from the actual pair-type counts
# ---------------------------------------------------------------------------------------------------------------------------------
def relaxation_count(seq_bytes, offs, lens, span):
    import numpy as np
    nw = len(lens)
    width = int(lens.max()) if nw else 0
    own = np.zeros(256, np.uint8); partner = np.zeros(256, np.uint8)
    for ch, b, m in ((b"A", 1, 8), (b"C", 2, 4), (b"G", 4, 2 | 8), (b"U", 8, 1 | 4), (b"T", 8, 1 | 4)):
        for c in (ch, ch.lower()):
            own[c[0]] = b; partner[c[0]] = m
    S = np.zeros((nw, width), np.uint8)
    for k in range(nw):
        S[k, :lens[k]] = seq_bytes[offs[k]:offs[k] + lens[k]]
    O, P = own[S], partner[S]
    r_int = 0
    r_ml = 0
    r_f3 = float((lens.astype(np.float64) * span / 2.0).sum())
    for d in range(4, min(span - 1, width - 1) + 1):
        cells = np.maximum(lens - d, 0).astype(np.float64).sum()
        r_ml += cells * max(d - 8, 0)
        if d >= 6:
            m = min(30, d - 6)
            paired = np.count_nonzero(O[:, :width - d] & P[:, d:])
            r_int += float(paired) * ((m + 1) * (m + 2) // 2)
    return {"total": float(r_ml + r_int + r_f3), "multiloop_splits": float(r_ml), "interior_candidates": float(r_int), "exterior": float(r_f3)}


# ---------------------------------------------------------------------------------------------------------------------------------
def e2e_process(ds, fold_model, base=None, runs=3, extra_cfg="", pause_s=1.2, back_to_back=2):
    """The product CLI the way a user runs it: a FRESH `python -m mir_prefer_amd.cli pipeline <config>` process per run, clocked by this (parent)
    process from spawn to exit -- interpreter start, imports, library load, device context, first-touch of every allocation and code object, the four
    stages, every report file, the removal of the temporary folder and process teardown are all inside.  SAM + FASTA in -> gff3 / fasta / ss / csv /
    html / readmapping out (MP:3728-3739).  base = the directory the files live under (None: the system's temporary directory).  The child's own
    stamps (MIRP_CLI_TIMINGS) give the breakdown; nothing of it enters the headline figure."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="mirp_e2e_", dir=base)
    try:
        sams = ds.write_sams(tmp)
        fa = os.path.join(tmp, "genome.fa")
        ds.write_fasta(fa)
        cfg_text = ("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %%s\nNAME_PREFIX = bench\nPRECURSOR_LEN = %d\nREADS_DEPTH_CUTOFF = %d\nMAX_GAP = %d\n%s"
                    % (fa, ", ".join(sams), L, CUT, GAP, extra_cfg))
        in_bytes = sum(os.path.getsize(p) for p in sams) + os.path.getsize(fa)
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        # the child's stamps go to the in-memory file system when there is one: written after its last stamp, on a busy overlay file system that one small
        # file stalled 0.1 s in a journal commit and showed up as "exit" time (a user's run writes no such file)
        tdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tmp
        env["MIRP_CLI_TIMINGS"] = os.path.join(tdir, "mirp_timings_%d_%s.json" % (os.getpid(), os.path.basename(tmp)))
        recs = []
        for rep in range(runs + back_to_back):
            # `runs` isolated invocations, then `back_to_back` ones started the moment the previous process is gone.  Isolated = at least pause_s after the last
            # process that used the GPU ended: a HIP process started right behind another one's exit waits 0.13 s in open("/dev/kfd") for the kernel driver's
            # deferred teardown of that process (hipInit 0.17 - 0.22 s instead of 0.06 s, profiles/r5_hip_back_to_back.txt) -- the cost of a shell loop over
            # data sets, not of one invocation, so it is reported beside the headline figure, not in it.
            # every run writes into a folder of its own and NOTHING is deleted until the whole bench is done (_E2E_TRASH): on the box's overlay file
            # system a burst of deletions makes the next seconds' file creations 6 - 9 x slower (10 -> 65 .. 97 us per file, profiles/tools/fs_regime.py),
            # which would charge the harness's own clean-up to the next run's 4,002 / 16,016 report files
            cfg = os.path.join(tmp, "config%d" % rep)
            with open(cfg, "w") as f:
                f.write(cfg_text % os.path.join(tmp, "out%d" % rep))
            if pause_s and rep < runs:
                time.sleep(pause_s)
            t0 = time.time()
            r = subprocess.run([sys.executable, "-m", "mir_prefer_amd.cli", "--fold-model", fold_model, "pipeline", cfg], env=env, cwd=tmp,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            wall = time.time() - t0
            if r.returncode != 0:
                return {"error": "CLI exited with %d: %s" % (r.returncode, r.stderr[-300:])}
            marks = json.load(open(env["MIRP_CLI_TIMINGS"]))
            seg, prev, dev = {}, t0, {}
            for m in marks:
                seg[m["name"]] = m["t"] - prev
                prev = m["t"]
                if "device_ms" in m:
                    dev[m["name"]] = m["device_ms"] / 1e3
            seg["exit"] = t0 + wall - prev
            loci = None
            for ln in r.stdout.splitlines():
                if ln.endswith(" miRNAs identified."):
                    loci = int(ln.split()[0])
            recs.append({"process_wall_s": wall, "segments_s": {k: round(v, 4) for k, v in seg.items()}, "device_s": {k: round(v, 4) for k, v in dev.items()}, "loci": loci})
        out_bytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(tmp, "out0")) for f in fs)
        n_small = len(os.listdir(os.path.join(tmp, "out0", "readmapping"))) if os.path.isdir(os.path.join(tmp, "out0", "readmapping")) else 0
        probe = os.path.join(tmp, "fsprobe")          # what this file system charges for creating a small file right now (one per miRNA locus is the reference's
        os.makedirs(probe)                             # output format: readmapping/<id>.map.txt); it varies 12 .. 100 us from one second to the next on an overlay
        t = time.time()
        for k in range(2000):
            os.close(os.open(os.path.join(probe, "f%d" % k), os.O_WRONLY | os.O_CREAT, 0o644))
        fs_us = (time.time() - t) / 2000 * 1e6
        first = recs[0]
        dev_total = sum(first["device_s"].values())
        return {"process_wall_s": first["process_wall_s"], "process_wall_s_all_runs": [round(x["process_wall_s"], 4) for x in recs[:runs]],
                "process_wall_s_back_to_back": [round(x["process_wall_s"], 4) for x in recs[runs:]],
                "device_s_all_runs": [round(sum(x["device_s"].values()), 4) for x in recs], "segments_s_all_runs": [x["segments_s"] for x in recs],
                "device_s": dev_total, "host_s": first["process_wall_s"] - dev_total, "host_over_device": (first["process_wall_s"] - dev_total) / dev_total if dev_total else None,
                "segments_s": first["segments_s"], "stage_device_s": first["device_s"], "files_under": os.path.dirname(tmp),
                "input_bytes": in_bytes, "output_bytes": out_bytes, "loci": first["loci"], "small_report_files": n_small, "fs_create_us_per_file": round(fs_us, 1),
                "note": "parent-side clock around a fresh `python -m mir_prefer_amd.cli pipeline <config>` process: interpreter start, imports, library load, "
                        "device context, first touch of allocations and code objects, stages, report files, removal of the temporary folder and exit all "
                        "included; process_wall_s = the FIRST of the isolated runs (each a new process started >= 1.2 s after the previous GPU process ended, inputs in "
                        "the page cache); process_wall_s_back_to_back = runs started the moment the previous one is reaped, which wait ~0.13 s in open(/dev/kfd) for "
                        "the driver's deferred teardown of their predecessor; segments_s = the child's own stamps: "
                        "main = spawn -> entry of main() (interpreter + package import), imports = capi / numpy / pipeline, context = config parse + "
                        "library load + device context (the join on the early device-open thread), then the four stages -- a lean run (no -k, no -d) does fold + filter + "
                        "report files as ONE pipelined call inside the predict segment, so `fold` is 0 and `predict` carries the fold's device time --, "
                        "removetmp, exit = end of main() -> process gone; stage_device_s = device time inside each stage"}
    finally:
        _E2E_TRASH.append(tmp)
        try:
            os.remove(env["MIRP_CLI_TIMINGS"])
        except Exception:
            pass


_E2E_TRASH = []          # directories of the end-to-end legs, removed when the bench is done (see e2e_process)


def e2e_cleanup():
    import shutil
    while _E2E_TRASH:
        shutil.rmtree(_E2E_TRASH.pop(), ignore_errors=True)


def ingest_leg(ctx, n_records):
    """SAM text -> sorted packed records resident in HBM: a synthetic unsorted SAM of a cfg[4]-shard-like shape (8 contigs x
    31.25 Mb, clusters of isomiR-like reads, 3 sample files), host threads tokenize, the GPU sorts stably by (tid, pos)."""
    import shutil
    import tempfile
    import numpy as np
    from mir_prefer_amd import synth
    tmp = tempfile.mkdtemp(prefix="mirp_ingest_")
    try:
        nc, clen = 8, 31250000
        names, lens = ["ctg%02d" % t for t in range(nc)], [clen] * nc
        rng = np.random.RandomState(77)
        per = n_records // 3
        paths, nbytes = [], 0
        for s in range(3):
            a = np.zeros(per, dtype=synth.ALN_DTYPE)
            centre = rng.randint(0, 150000, size=per).astype(np.int64) * 1600 + 200           # loci on a grid, reads scattered around them
            a["tid"] = (centre // clen).astype(np.int32) % nc
            a["pos"] = (centre % (clen - 2000) + rng.randint(0, 60, size=per) + 1).astype(np.int32)
            a["depth"] = rng.randint(1, 40, size=per)
            a["len"] = rng.randint(18, 26, size=per)
            a["strand"] = rng.randint(0, 2, size=per)
            p = os.path.join(tmp, "S%d.sam" % (s + 1))
            nbytes += synth.write_sam_fast(p, "S%d" % (s + 1), a, names, lens)
            paths.append(p)
        ctx.ingest_sams(paths[:1])                        # warm-up (page cache of the first file, kernels loaded)
        t = time.time()
        cn, cl, sn, alns, segs, sec = ctx.ingest_sams(paths)
        wall = time.time() - t
        key = alns["tid"].astype(np.int64) << 32 | alns["pos"].astype(np.int64)
        assert len(alns) == 3 * per and (np.diff(key) >= 0).all()
        return {"records": int(len(alns)), "fraction_of_a_config4_rank_shard": len(alns) / 25e6, "sam_bytes": int(nbytes), "wall_s": wall, "records_per_s": len(alns) / wall,
                "sam_MB_per_s": nbytes / wall / 1e6, "seconds": sec, "seconds_sum": sum(sec.values()),
                "note": "3 unsorted SAM files -> mirp_ingest_sams_gpu (host tokenizer threads, H2D, device LSD radix sort by (tid, pos), D2H copy for the host stages); "
                        "files in the page cache; seconds = the library's four phases + native_other_s (the call's time outside them) + host_copy_s (records copied "
                        "into numpy arrays); --ingest-records 25000000 runs a whole config[4] rank shard"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------------------
def micro_families(n_windows, n=300, seed=99):
    """The three sequence families of the fold micro-benchmark: uint8 ASCII matrices [n_windows, n]."""
    import numpy as np
    rng = np.random.RandomState(seed)
    acgu = np.frombuffer(b"ACGU", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGU", b"UGCA"):
        comp[a] = b
    uni = acgu[rng.randint(0, 4, size=(n_windows, n))]
    hp = acgu[rng.randint(0, 4, size=(n_windows, n))]
    for w in range(0, n_windows, 2):          # every second window carries one planted imperfect hairpin (arm 24-34, loop 6-40, 0-4 arm mismatches)
        arm, loop = int(rng.randint(24, 35)), int(rng.randint(6, 41))
        a = acgu[rng.randint(0, 4, size=arm)]
        b = comp[a[::-1]].copy()
        for _ in range(int(rng.randint(0, 5))):
            b[rng.randint(0, arm)] = acgu[rng.randint(0, 4)]
        h = np.concatenate([a, acgu[rng.randint(0, 4, size=loop)], b])
        o = int(rng.randint(0, n - len(h) + 1))
        hp[w, o:o + len(h)] = h
    gc = acgu[rng.choice(4, size=(n_windows, n), p=[0.175, 0.325, 0.325, 0.175])]
    return [("uniform", uni), ("planted_hairpins_50pct", hp), ("gc_0.65", gc)]


def fold_microbench(ctx, n_windows):
    import numpy as np
    out = {}
    for name, mat in micro_families(n_windows):
        nw, n = mat.shape
        offs = np.arange(nw + 1, dtype=np.int64) * n
        blob = np.ascontiguousarray(mat).reshape(-1)
        if not out:
            ctx.fold_batch_summary(blob, offs, L)          # warm-up at full size once: the batch's device buffers are allocated here, not inside the timed call
        else:
            ctx.fold_batch_summary(blob[:n * 512], offs[:513], L)          # warm-up
        t = time.time()
        nl, mfe, st = ctx.fold_batch_summary(blob, offs, L)
        wall = time.time() - t
        km = ctx.last_fold_kernel_ms()
        ns = min(nw, 4096)
        R = relaxation_count(blob, offs[:ns], np.full(ns, n, dtype=np.int64), L)
        r_tot = R["total"] * nw / ns
        fb = int(ctx.last_fold_fallbacks())
        out[name] = {"windows": int(nw), "windows_per_s": nw / wall, "wall_s": wall, "fill_kernel_ms": km[0], "epilogue_kernel_ms": km[1],
                     "relaxations": r_tot, "T_relaxations_per_s_fill": (r_tot / (km[0] / 1e3) / 1e12) if km[0] > 0 else None,
                     "generic_fallback_windows": fb, "line_overflow_windows": int((st == 1).sum()), "failed_windows": int((st < 0).sum()),
                     "mean_mfe_kcal": float(mfe.mean()) / 100.0, "mean_structure_lines": float(nl.mean())}
    out["note"] = ("mirp_fold_batch_summary on 2^16 windows of n = L = 300 per family, sequences uploaded from the host, structure text left on the device; wall includes "
                   "the upload and the generic re-fold of the windows the LDS-resident kernel hands back (16-bit energy range); relaxations counted on the first 4096 windows")
    return out


def main():
    a = parse_args()
    a.workload = a.workload.replace("cfg", "config")
    if a.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start one worker process per GPU under torch.distributed.run BEFORE anything touches the GPU
        # (no HIP call has happened in this process; it only waits for the child and passes its exit code on)
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    import numpy as np
    from mir_prefer_amd import capi, synth
    from mir_prefer_amd import dist as mdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    tdist = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        tdist = mdist.init_host_group()          # CPU-side gloo group only: host objects and the timing barrier (torch never touches the GPU here)
        world = tdist.get_world_size()
    rccl_error = None
    share_dir = os.environ.get("MIRP_BENCH_SHARE_GPU")      # dev: all ranks on GPU 0, the library's exchanges over its local transport (a directory)
    one_device = bool(os.environ.get("MIRP_BENCH_ONE_DEVICE"))      # dev: all ranks on GPU 0 and RCCL tried all the same (it refuses: exercises the agreement below)
    ctx = capi.Context(0 if (share_dir or one_device) else local_rank)
    if world > 1 and share_dir and os.environ.get("MIRP_BENCH_FORCE_GLOO"):      # dev: exercises the host-object exchange below on one GPU
        rccl_error = "forced (MIRP_BENCH_FORCE_GLOO)"
    elif world > 1 and share_dir:
        ctx.dist_init_local(share_dir, rank, world)
    elif world > 1:
        # the library's RCCL communicator, one rank per GPU.  Should it not come up on some node (every rank must agree, so the outcome is shared over
        # the host group), the loci lists are gathered as host objects over gloo instead -- the line then says so; the measured step is the same.
        why = None
        import threading
        # ncclCommInitRank can also HANG (a rank that never joins, a fabric that does not come up): a stuck native call cannot be cancelled, so a
        # watchdog ends the process with a message instead of leaving the launcher to its own timeout
        joined = threading.Event()

        def _watchdog():
            if not joined.wait(float(os.environ.get("MIRP_BENCH_RCCL_TIMEOUT", "300"))):
                sys.stderr.write("[bench] rank %d: the RCCL communicator did not come up within the time limit (MIRP_BENCH_RCCL_TIMEOUT) -- giving up\n" % rank)
                sys.stderr.flush()
                os._exit(5)
        threading.Thread(target=_watchdog, daemon=True).start()
        try:
            mdist.init_context(ctx, rank, world)
        except Exception as e:          # capi.MirpError: dlopen / ncclCommInitRank failed
            why = "%s" % (e,)
        joined.set()
        box = [None] * world
        tdist.all_gather_object(box, why)
        bad = [w for w in box if w]
        if bad:
            rccl_error = bad[0]
            if why is None:
                ctx.dist_finalize()
            if not a.allow_gloo:      # a line produced over gloo would not be an RCCL number: refuse unless asked for
                sys.stderr.write("[bench] RCCL communicator not available (%s); re-run with --allow-gloo to gather over the gloo host group instead\n" % rccl_error)
                tdist.barrier()
                tdist.destroy_process_group()
                sys.exit(3)
            sys.stderr.write("[bench] RCCL communicator not available (%s): loci lists go over the gloo host group (--allow-gloo)\n" % rccl_error)
    ctx.set_fold_model(a.fold_model)

    # ---- synthetic workload: every rank generates and holds only the contigs it owns
    specs, n_samples, background, scaling, desc = workload_specs(a.workload, world, a.genome, a.loci, a.genome_scale)
    if a.workload == "config1":
        owned = [rank]
    else:
        owned = mdist.partition_contigs([sp[1] for sp in specs], world)[rank]
    contigs, alns, sample_names = build_shard(specs, set(owned), n_samples, background)
    names = [n for n, _ in contigs]
    order = np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)
    ctx.load_genome(contigs)
    ctx.load_alignments(alns)

    fb = [0, 0, 0]

    from mir_prefer_amd import balance, records
    moved = [0, 0]          # windows this rank shipped / received in the last step (window-level re-balancing, mir-prefer_amd/balance.py)

    def step():
        npk, nloci, nwin = ctx.candidate(CUT, GAP, L, order)
        imported, plan_moves = [], []
        if world > 1:
            # even out the window lists before the fold: whole contigs per rank leave the ranks uneven (config3: 1.23 x the mean on the fullest
            # rank), windows are independent.  Every rank computes the same plan from the counts; the payloads go over the library's communicator.
            if rccl_error is None:          # one small all-reduce on the library's communicator (every rank fills its own slot) instead of a pickled object gather over gloo
                slot = np.zeros(world, dtype=np.int64)
                slot[rank] = int(nwin)
                counts = [int(x) for x in ctx.dist_allreduce_sum(slot)]
            else:
                counts = [None] * world
                tdist.all_gather_object(counts, int(nwin))
            if rccl_error is None:
                xchg = ctx.exchange_bytes
            else:
                def xchg(blocks):
                    box = [None] * world
                    tdist.all_gather_object(box, blocks)
                    return [b[rank] for b in box]
            keep, imported, plan_moves = balance.exchange(xchg, rank, world, ctx.get_windows, alns, counts)
            moved[0], moved[1] = int(nwin) - keep, sum(len(p["windows"]) for p in imported)
            if keep != nwin:
                ctx.limit_windows(keep)
            nwin = keep + moved[1]
        ctx.fold(L)
        fb[0], fb[1], fb[2] = ctx.last_fold_fallbacks(), ctx.last_fold_overflow(), ctx.last_fold_dense()
        tm_own, km_own = ctx.last_timings(), ctx.last_fold_kernel_ms()
        for p in imported:
            balance.fold_imported(ctx, p, L)
        out = ctx.predict_raw(n_samples, 18, 23, False, True)          # the C-ABI call with its result in flat arrays (records + structure text rows), as the native report
                                                                        # writer takes it; ctx.predict() additionally builds one Python string per locus, which is host-side
                                                                        # presentation (0.6 ms at 4,002 loci), not the path
        tm_own = dict(tm_own, predict_ms=ctx.last_timings()["predict_ms"])          # (the dict read after the fold still held the previous step's filter time)
        imp = [balance.predict_imported(ctx, p, (n_samples, 18, 23, 0, 1, 55)) for p in imported]
        if world > 1 and rccl_error is None:          # the exchange step of the path: loci lists of all ranks on rank 0, over RCCL from the device-resident result
            g = ctx.gather_loci(0)
            total = len(g["result"])
            if plan_moves:          # every rank computed the same plan: without a transfer nobody holds imported windows and the second gather has nothing to carry
                extra = np.concatenate([x["result"] for x in imp]) if imp else np.zeros(0, dtype=records.MIRNA_DTYPE)      # loci of imported windows
                ge = ctx.gather_records(extra.view(np.int32).reshape(len(extra), records.MIRNA_DTYPE.itemsize // 4))
                total += 0 if ge is None else len(ge)
        elif world > 1:
            parts = [None] * world if rank == 0 else None
            tdist.gather_object(np.concatenate([out["result"]] + [x["result"] for x in imp]), parts, dst=0)
            total = sum(len(x) for x in parts) if rank == 0 else 0
        else:
            total = len(out["result"])
        return nwin, total, tm_own, km_own

    def sync():
        # every C-ABI call above returns after its stream has drained; across ranks: a RCCL reduction on the library's communicator and the
        # CPU-side barrier
        if world > 1:
            if rccl_error is None:
                ctx.dist_barrier()
            tdist.barrier()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.time()
    fold_ms, cov_ms, rest_ms, pred_ms, fill_ms, epi_ms = [], [], [], [], [], []
    nwin = nres = 0
    for _ in range(a.steps):
        nwin, nres, tm, km = step()
        fold_ms.append(tm["fold_ms"]); cov_ms.append(tm["coverage_ms"]); rest_ms.append(tm["candidate_rest_ms"]); pred_ms.append(tm["predict_ms"])
        fill_ms.append(km[0]); epi_ms.append(km[1])
    t_own = time.time() - t0          # this rank's own K steps (before waiting for the others)
    sync()
    elapsed = time.time() - t0
    ranks = None
    if world > 1:
        box = [None] * world
        info = ctx.dist_comm_info()
        tdist.all_gather_object(box, (elapsed, float(nwin), t_own, len(alns), len(owned), moved[0], moved[1], info["comm_count"], info["comm_device"], ctx.device, os.getpid()))
        elapsed = max(x[0] for x in box)
        total_windows = sum(x[1] for x in box)
        ranks = {"own_ms_per_step": [1e3 * x[2] / a.steps for x in box], "windows": [int(x[1]) for x in box], "alignments": [int(x[3]) for x in box],
                 "contigs": [int(x[4]) for x in box], "windows_shipped": [int(x[5]) for x in box], "windows_received": [int(x[6]) for x in box],
                 # what RCCL itself reports on every rank (ncclCommCount / ncclCommCuDevice; -1 = no RCCL communicator: local transport or gloo), the HIP
                 # device of the rank's context and its process id: N processes, N devices, one communicator of N ranks
                 "rccl_comm_count": [int(x[7]) for x in box], "rccl_comm_device": [int(x[8]) for x in box], "context_device": [int(x[9]) for x in box],
                 "pid": [int(x[10]) for x in box]}
        ranks["max_over_mean_time"] = max(ranks["own_ms_per_step"]) / (sum(ranks["own_ms_per_step"]) / world)
    else:
        total_windows = float(nwin)

    if rank == 0:
        fold_s = float(np.mean(fold_ms)) / 1e3
        fill_s = float(np.mean(fill_ms)) / 1e3
        epi_s = float(np.mean(epi_ms)) / 1e3
        cov_s = float(np.mean(cov_ms)) / 1e3
        cov_fused = ctx.last_coverage_fused()
        wins = ctx.get_windows()
        W = wins["windows"]
        lens = W["seq_len"].astype(np.int64)
        # ---- dominant kernel: the fill kernel (dynamic program).  Unit of algorithmic work = one relaxation; R is counted
        # exactly for this rank's batch from its pair-type counts (a sample of 20,000 windows, scaled, when the batch is larger).  Roofs:
        # LDS = 2 reads per relaxation at the ds_read_b32 rate -- measured on this GPU now, and the microarchitecture guide's figure beside it;
        # VALU = 3 integer lane-operations per relaxation at the measured 32-bit issue rate.
        nsmp = min(len(W), 20000)
        R = relaxation_count(wins["seq"], W["seq_off"].astype(np.int64)[:nsmp], lens[:nsmp], L)
        if nsmp < len(W):
            R = {k: v * len(W) / nsmp for k, v in R.items()}
        mb = ctx.microbench()
        lds_roof = mb["ds_read_b32_per_s"] * 64.0 / 2.0
        valu_roof = mb["valu_u32_per_s"] * 64.0 / 3.0
        roof = min(lds_roof, valu_roof)
        achieved = R["total"] / fill_s if fill_s > 0 else 0.0
        D = np.minimum(L - 1, lens - 1)
        cells = np.where(D > 3, (D - 3) * lens - (D * (D + 1) // 2 - 6), 0)
        b_fold = float((lens + 64 + 10 * cells + 6000).sum())     # c, fML, trace-back triangles written once (16-bit), c + fML read once, ~6 KB of lines
        prof = {}
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "CURRENT.json")))
        except Exception:
            pass
        headline = a.workload == "config1" and a.genome == CHR1_LEN and a.loci == N_LOCI
        loci_expected = EXPECTED_LOCI.get((a.workload, a.fold_model)) if (world == 1 and a.genome_scale == 1.0 and (headline or a.workload != "config1")) else None
        if loci_expected is not None and int(nres) != loci_expected:
            sys.stderr.write("[bench] RESULT CHECK FAILED: %d miRNA loci, expected %d for (%s, %s) -- no line printed\n" % (nres, loci_expected, a.workload, a.fold_model))
            ctx.close()
            sys.exit(4)
        pipe = prof.get("fold_fill_pipe_busy") if (a.workload == "config1" and a.fold_model == "vienna-2.1.2") else None
        same_workload = headline and a.fold_model == "vienna-2.1.2"
        traffic = prof.get("fold_fill_hbm_bytes_per_launch") if same_workload else None
        g_tot = float(sum(len(sq) + 1 for _, sq in contigs))
        b_cov = 16.0 * len(alns) + 16.0 * g_tot
        b_cov_moved = 48.0 * len(alns) + 8.0 * g_tot      # what the stage moves since the clearing pass is gone: the scan reads 8 B per position; a record is
        line = {                                           # read by the scatter and by the un-scatter (16 B each) and touches 2 x 2 x 4 B of the arrays
            "metric": "precursor windows folded+filtered/sec (L=300)", "value": total_windows * a.steps / elapsed, "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "int16/int32 (energies in 0.01 kcal/mol)", "data": "synthetic",
            "config": {"workload": desc + "; candidate+fold+predict(+gather of the loci list), inputs resident in HBM",
                       "name": a.workload, "windows_total": int(total_windows), "windows_rank0": int(nwin), "loci_found": int(nres),
                       "alignments_rank0": int(len(alns)), "contigs_rank0": len(owned), "samples": n_samples,
                       "fold_flavour": "vienna-2.1.2 (Turner-2004, d2)" if a.fold_model == "vienna-2.1.2" else "vienna-1.8.5 (Turner-1999, d1)",
                       "fold_generic_fallback_windows": int(fb[0]), "fold_line_overflow_windows": int(fb[1]),
                       "exchange": ("none (1 GPU)" if world == 1 else "mirp_gather_loci over the local transport (ranks share GPU 0: dev run)" if share_dir
                                    else "loci lists as host objects over gloo (RCCL communicator not available: %s)" % rccl_error if rccl_error
                                    else "mirp_gather_loci over RCCL (library-owned communicator), host objects over gloo")},
            "roofline": {"kernel": "fold_lds_kernel<0, true>" if a.fold_model == "vienna-2.1.2" else "fold_lds_kernel<1, true>",
                         "bound": "lds", "achieved": achieved / 1e12, "peak": LDS_GUIDE_RELAX_PER_S / 1e12, "unit": "T relaxations/s",
                         "frac": achieved / LDS_GUIDE_RELAX_PER_S, "peak_measured": roof / 1e12, "frac_measured": achieved / roof if roof > 0 else None,
                         "bound_measured": "lds" if lds_roof <= valu_roof else "valu", "avg_launch_ms": fill_s * 1e3, "pipe_busy": pipe,
                         # `frac` prices work the kernel no longer issues.  frac_executed prices what it does issue -- interior candidates + exterior +
                         # the split-candidate visits (2.5 % of the dense multiloop count on this workload, tests/tools/splitcand_gate.c) -- against the same roof:
                         # the utilisation-like figure; `frac` stays the contract's algorithmic one (speed-up over a dense kernel running at the roof)
                         "executed_relaxations_estimate": R["interior_candidates"] + R["exterior"] + 0.025 * R["multiloop_splits"],
                         "frac_executed": ((R["interior_candidates"] + R["exterior"] + 0.025 * R["multiloop_splits"]) / fill_s / LDS_GUIDE_RELAX_PER_S) if fill_s > 0 else None,
                         "windows_via_dense_split_pass": int(fb[2]),
                         "relaxations_per_launch": R, "lds_roof_T": lds_roof / 1e12, "valu_roof_T": valu_roof / 1e12, "microbench_wave_insts_per_s": mb,
                         "traffic": traffic, "traffic_source": prof.get("source") if traffic is not None else None,
                         "note": "integer min-plus dynamic program out of LDS, SURVEY.md 8d: achieved = ALGORITHMIC relaxations of the batch (the dense count "
                                 "R_ml + R_int + R_f3 from its pair-type counts; since round 4 the kernel relaxes the multiloop splits over split candidates only "
                                 "and executes ~2.5 % of R_ml) / the fill kernel's time; peak = 75 TB/s of ds_read_b32 (MI355X_MICROARCH.md) / 8 B per relaxation; "
                                 "peak_measured = min(LDS, VALU) roofs micro-benchmarked on this GPU in this run; pipe_busy = busy fractions of the LDS array, the "
                                 "vector and the scalar issue ports from the committed rocprofv3 PMC passes (profiles/CURRENT.json); traffic = HBM bytes per launch from the same"},
            "roofline_hbm": {"kernel": "fold_lds_kernel + fold_lds_epilogue_kernel", "bound": "hbm", "achieved": b_fold / fold_s / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": b_fold / fold_s / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": fold_s * 1e3,
                             "note": "HBM view of the fold (algorithmic bytes / time / 8 TB/s); << 1 is expected: the tables are LDS-resident"},
            "roofline_coverage": {"kernel": "cov_tile_first_kernel + cov_scan_kernel<fused>" if cov_fused else "cov_scatter_kernel + cov_scan_kernel + cov_unscatter_kernel", "bound": "hbm", "achieved": b_cov / cov_s / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_cov / cov_s / 1e9 / HBM_PEAK_GBS, "avg_ms": cov_s * 1e3,
                                  "bytes_algorithmic": b_cov, "bytes_moved_model": b_cov_moved, "achieved_moved": b_cov_moved / cov_s / 1e9,
                                  "frac_moved": b_cov_moved / cov_s / 1e9 / HBM_PEAK_GBS,
                                  "note": "achieved / frac price SURVEY 8d's algorithmic bytes 16 A + 16 G; the stage no longer clears the whole difference arrays, so "
                                          "the bytes it actually moves are ~ 8 G + 48 A (achieved_moved / frac_moved): part of the gain is work not done, not bandwidth"},
            "stage_ms": {"coverage": cov_s * 1e3, "candidate_rest": float(np.mean(rest_ms)), "fold": fold_s * 1e3, "fold_fill_kernel": fill_s * 1e3,
                         "fold_epilogue_kernel": epi_s * 1e3, "predict": float(np.mean(pred_ms)),
                         "per_step": {"fold": [round(x, 3) for x in fold_ms], "predict": [round(x, 3) for x in pred_ms]}},
        }
        if ranks:
            line["ranks"] = ranks
        if world == 1:      # reported side lines and baselines: rank 0 at N = 1 only, AFTER the GPU timing
            if headline and not a.no_configs:
                cfgs = {}
                other = "vienna-1.8.5" if a.fold_model == "vienna-2.1.2" else "vienna-2.1.2"
                ctx.set_fold_model(other)
                step()
                t = time.time()
                k3 = [step() for _ in range(3)]
                el = time.time() - t
                cfgs["config1_" + other] = {"windows_per_s": k3[-1][0] * 3 / el, "ms_per_step": 1e3 * el / 3, "windows": int(k3[-1][0]), "loci_found": int(k3[-1][1]),
                                            "fold_fill_kernel_ms": float(np.mean([x[3][0] for x in k3])), "fold_epilogue_kernel_ms": float(np.mean([x[3][1] for x in k3])),
                                            "fold_generic_fallback_windows": int(fb[0])}
                ctx.set_fold_model(a.fold_model)
                cfgs["fold_microbench"] = fold_microbench(ctx, a.micro_windows)
                sp2, ns2, bg2, _, d2 = workload_specs("config2", 1)
                c2, a2, _ = build_shard(sp2, set(range(len(sp2))), ns2, bg2)
                o2 = np.argsort(np.array([n for n, _ in c2], dtype=object), kind="stable").astype(np.int32)
                ctx.load_genome(c2)
                ctx.load_alignments(a2)

                def step2():
                    _, _, nw2 = ctx.candidate(CUT, GAP, L, o2)
                    ctx.fold(L)
                    f2 = ctx.last_fold_fallbacks()
                    r2 = ctx.predict_raw(ns2, 18, 23, False, True)
                    return nw2, len(r2["result"]), f2, ctx.last_timings(), ctx.last_fold_kernel_ms()
                step2()
                t = time.time()
                k2 = [step2() for _ in range(3)]
                el = time.time() - t
                if a.fold_model == "vienna-2.1.2" and int(k2[-1][1]) != EXPECTED_LOCI[("config2", a.fold_model)]:
                    sys.stderr.write("[bench] RESULT CHECK FAILED on config2: %d miRNA loci, expected %d -- no line printed\n" % (k2[-1][1], EXPECTED_LOCI[("config2", a.fold_model)]))
                    sys.exit(4)
                w2 = ctx.get_windows()
                l2 = w2["windows"]["seq_len"].astype(np.int64)
                n2s = min(len(l2), 20000)
                R2 = relaxation_count(w2["seq"], w2["windows"]["seq_off"].astype(np.int64)[:n2s], l2[:n2s], L)
                r2_total = R2["total"] * len(l2) / n2s
                fill2 = float(np.mean([x[4][0] for x in k2])) / 1e3
                cfgs["config2"] = {"workload": d2, "windows_per_s": k2[-1][0] * 3 / el, "ms_per_step": 1e3 * el / 3, "windows": int(k2[-1][0]), "loci_found": int(k2[-1][1]),
                                   "loci_expected": EXPECTED_LOCI.get(("config2", a.fold_model)),
                                   "roofline": {"kernel": "fold_lds_kernel<0, true>", "bound": "lds", "achieved": r2_total / fill2 / 1e12, "peak": LDS_GUIDE_RELAX_PER_S / 1e12,
                                                "unit": "T relaxations/s", "frac": r2_total / fill2 / LDS_GUIDE_RELAX_PER_S, "avg_launch_ms": fill2 * 1e3,
                                                "note": "as the headline's roofline: algorithmic (dense) relaxations of the batch, counted on its first 20,000 windows and scaled"},
                                   "alignments": int(len(a2)), "fold_generic_fallback_windows": int(k2[-1][2]),
                                   "stage_ms": {"coverage": float(np.mean([x[3]["coverage_ms"] for x in k2])), "candidate_rest": float(np.mean([x[3]["candidate_rest_ms"] for x in k2])),
                                                "fold_fill_kernel": float(np.mean([x[4][0] for x in k2])), "fold_epilogue_kernel": float(np.mean([x[4][1] for x in k2])),
                                                "predict": float(np.mean([x[3]["predict_ms"] for x in k2]))}}
                if not a.no_cov_shard:
                    # the coverage stage at the size of one rank's share of config[4] (8 x 31.25 Mb, 2.5e7 packed records: 0.1 records per base), where
                    # the scatter's global atomics used to bound it: the candidate stage alone, no fold
                    rng4 = np.random.RandomState(4)
                    c4 = [("ctg%02d" % t, synth._BASES[rng4.randint(0, 4, size=31250000, dtype=np.uint8)]) for t in range(8)]
                    a4 = synth.packed_records_shard(8, 31250000, 150000, 167, seed=44)
                    ctx.load_genome(c4)
                    ctx.load_alignments(a4)
                    o4 = np.arange(8, dtype=np.int32)
                    ms4, rest4, nw4 = [], [], 0
                    for _ in range(6):
                        _, _, nw4 = ctx.candidate(CUT, GAP, L, o4)
                        ms4.append(ctx.last_timings()["coverage_ms"])
                        rest4.append(ctx.last_timings()["candidate_rest_ms"])
                    g4 = float(sum(len(x) + 1 for _, x in c4))
                    alg4, mov4 = 16.0 * len(a4) + 16.0 * g4, 32.0 * len(a4) + 8.0 * g4
                    t4 = float(np.mean(ms4[1:])) / 1e3
                    cfgs["coverage_config4_shard"] = {
                        "avg_ms": t4 * 1e3, "min_ms": float(min(ms4)), "candidate_rest_ms": float(np.mean(rest4[1:])),
                        "candidate_stage_ms": t4 * 1e3 + float(np.mean(rest4[1:])), "path": "fused scan (tiles built from the sorted records in LDS)" if ctx.last_coverage_fused() else "atomic scatter + scan",
                        "records": int(len(a4)), "positions": int(g4), "windows": int(nw4), "bytes_algorithmic": alg4, "bytes_moved_model": mov4,
                        "achieved": alg4 / t4 / 1e9, "frac": alg4 / t4 / 1e9 / HBM_PEAK_GBS, "achieved_moved": mov4 / t4 / 1e9, "frac_moved": mov4 / t4 / 1e9 / HBM_PEAK_GBS,
                        "unit": "GB/s", "note": "bytes_moved_model = 32 A + 8 G, an upper bound: the records are read twice (tile index; the tile they start in, plus the few that "
                                                 "reach over a tile edge), a 256-position row of the dense arrays is written only if a run walk can read it (all rows counted "
                                                 "here; the counter-based bytes are in profiles/*_coverage_shard_pmc.json)"}
                    del c4, a4
                # PRECURSOR_LEN beyond what the LDS-resident fill kernel holds (span <= 300, windows <= 350 nt): the same seeded workload at
                # PRECURSOR_LEN = 400 (the reference accepts 60 .. 3000, MP:167-184) -- every window goes through the generic kernel, tables in HBM
                ctx.load_genome(contigs)
                ctx.load_alignments(alns)

                def step400():
                    _, _, nw4 = ctx.candidate(CUT, GAP, 400, order)
                    ctx.fold(400)
                    f4 = ctx.last_fold_fallbacks()
                    r4 = ctx.predict_raw(n_samples, 18, 23, False, True)
                    return nw4, len(r4["result"]), f4, ctx.last_timings()
                step400()
                t = time.time()
                k4 = [step400() for _ in range(2)]
                el = time.time() - t
                cfgs["L400"] = {"workload": "config1 generator at PRECURSOR_LEN = 400 (windows of 400 / 425 nt): fold_generic_kernel, DP tables in HBM",
                                "windows_per_s": k4[-1][0] * 2 / el, "ms_per_step": 1e3 * el / 2, "windows": int(k4[-1][0]), "loci_found": int(k4[-1][1]),
                                "fold_generic_fallback_windows": int(k4[-1][2]), "fold_ms": float(np.mean([x[3]["fold_ms"] for x in k4]))}
                # the same work measure as the headline's roofline (algorithmic relaxations of the batch), for comparison only: this kernel's tables live in HBM,
                # its operands reach the lanes through a staged copy in LDS -- the LDS figure is the headline kernel's roof, not a bound derived for this one
                w4 = ctx.get_windows()
                l4 = w4["windows"]["seq_len"].astype(np.int64)
                n4s = min(len(l4), 4000)
                R4 = relaxation_count(w4["seq"], w4["windows"]["seq_off"].astype(np.int64)[:n4s], l4[:n4s], 400)
                r4_total = R4["total"] * len(l4) / n4s
                cfgs["L400"]["relaxations_algorithmic"] = r4_total
                cfgs["L400"]["T_relaxations_per_s_fold"] = r4_total / (cfgs["L400"]["fold_ms"] / 1e3) / 1e12
                cfgs["L400"]["frac_of_headline_kernels_lds_roof"] = r4_total / (cfgs["L400"]["fold_ms"] / 1e3) / LDS_GUIDE_RELAX_PER_S
                cfgs["note"] = ("config3 / config4 are multi-GPU workloads: `python bench.py --gpus 8 --workload config3|config4` (a rank's shard of either is covered at "
                                "full size by tests/test_configs_gpu.py); the headline above stays config1")
                line["configs"] = cfgs
                ctx.load_genome(contigs)          # back to the headline workload for the baselines below
                ctx.load_alignments(alns)
            ds = synth.Dataset(contigs, sample_names, alns, [])
            if not a.no_e2e and headline:
                # End-to-end wall-clock, the second half of BASELINE's metric: the CLI in a fresh process, files under the default temporary directory
                # (what a user gets); the same on the in-memory file system beside it (the container's overlay file system creates 4,002 small
                # report files in 0.04 .. 0.5 s from one second to the next, profiles/tools/smallfiles.py); then config[2] (3 SAM files, 119 Mb FASTA).
                ctx.close()          # a CLI user's GPU is not shared with a bench process that holds 20 GB of it
                ctx = None
                time.sleep(3.0)      # ... and the driver's deferred teardown of THIS process' context (20 GB) is not the first child's to wait for either
                line["e2e"] = e2e_process(ds, a.fold_model, None, a.e2e_runs)
                line["e2e_wall_s"] = line["e2e"].get("process_wall_s")
                shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
                if shm and "error" not in line["e2e"]:
                    other = e2e_process(ds, a.fold_model, shm, a.e2e_runs, back_to_back=0)
                    line["e2e"]["files_on_tmpfs"] = {k: other.get(k) for k in ("process_wall_s", "process_wall_s_all_runs", "segments_s", "files_under", "error") if k in other}
                if not a.no_configs:
                    sp2, ns2, bg2, _, d2 = workload_specs("config2", 1)
                    c2, a2, sn2 = build_shard(sp2, set(range(len(sp2))), ns2, bg2)
                    r2 = e2e_process(synth.Dataset(c2, sn2, a2, []), a.fold_model, None, max(1, a.e2e_runs - 1), back_to_back=1)
                    r2["workload"] = d2
                    if r2.get("loci") is not None and r2["loci"] != EXPECTED_LOCI[("config2", a.fold_model)] and a.fold_model == "vienna-2.1.2":
                        r2["error"] = "result check failed: %d loci" % r2["loci"]
                    line["e2e_config2"] = r2
                    del c2, a2
                if line["e2e"].get("loci") is not None and line["e2e"]["loci"] != EXPECTED_LOCI.get(("config1", a.fold_model), line["e2e"]["loci"]):
                    line["e2e"]["error"] = "result check failed: %d loci" % line["e2e"]["loci"]
            if not a.no_ingest and headline:          # after the end-to-end legs: this one writes 0.6 GB of SAM text, and a file system that is flushing it
                if ctx is None:                       # creates the CLI's thousands of small report files several times slower (fs_create_us_per_file)
                    ctx = capi.Context(local_rank)
                line["ingest"] = ingest_leg(ctx, a.ingest_records)
            if not a.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(ds, alns, order, a.cpu_budget)
        print(json.dumps(line))
        sys.stdout.flush()
        e2e_cleanup()
    if world > 1:
        tdist.barrier()
        if rccl_error is None:
            ctx.dist_finalize()
    if ctx is not None:
        ctx.close()
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
