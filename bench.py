#!/usr/bin/env python3
"""Benchmark of the candidate -> fold -> predict hot path on MI355X.

Metric (BASELINE.json): precursor windows folded+filtered per second at L = 300, inputs resident in HBM; end-to-end wall-clock beside it.
A step = one pass of the whole hot path (coverage scan -> peaks -> windows -> payload -> local fold -> filter -> loci list) over one
synthetic batch.  Workload at N = 1: BASELINE config[1], an A. thaliana chr1-sized contig (30,427,671 bp), 1 sample, L = 300,
12,000 synthetic loci (~20 k windows), SURVEY.md 8d.  For N > 1 every rank owns one such contig (contig sharding, weak scaling) and the
final loci lists are gathered to rank 0 over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: re-launches itself under torch.distributed.run before any GPU call)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

The JSON line carries, beside the contract's keys:
  roofline       the dominant kernel (fold_lds_kernel, the dynamic program) against the roof that bounds it -- integer min-plus relaxations
                 out of LDS, LDS-read / VALU-issue bound, both roofs micro-benchmarked on this GPU in this run (mirp_microbench);
  roofline_hbm   the HBM view north_star asks for (algorithmic bytes / measured time / 8 TB/s), expected << 1 for an LDS-resident DP;
  roofline_coverage  the one HBM-bound stage (memset + scatter + scan);
  cpu_baseline   the CPU oracle (the build's own restatement of the reference, "port") timed on this box AFTER the GPU timing: 1-thread and
                 one-process-per-physical-core legs over candidate + fold + filter;
  e2e            the CLI `pipeline` verb on files of the same workload (SAM + FASTA in -> gff3 and reports out), wall-clock by stage.
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--loci", type=int, default=N_LOCI)
    ap.add_argument("--genome", type=int, default=CHR1_LEN)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of wall-clock per CPU-baseline leg")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--ingest-records", type=int, default=8000000,
                    help="records of the synthetic SAM of the ingest leg (a cfg[4] rank shard is 25,000,000; the default keeps the run short)")
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
            "candidate_stage_s_1thread": t_cand}


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
def e2e_cli(ds, fold_model):
    """The product CLI's `pipeline` verb on files of the bench workload: SAM + FASTA in -> gff3 / fasta / ss / csv / html / readmapping out."""
    import shutil
    import tempfile
    from mir_prefer_amd import config, pipeline
    tmp = tempfile.mkdtemp(prefix="mirp_e2e_")
    try:
        sams = ds.write_sams(tmp)
        fa = os.path.join(tmp, "genome.fa")
        ds.write_fasta(fa)
        cfg = os.path.join(tmp, "config")
        with open(cfg, "w") as f:
            f.write("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %s\nNAME_PREFIX = bench\nPRECURSOR_LEN = %d\nREADS_DEPTH_CUTOFF = %d\nMAX_GAP = %d\n"
                    % (fa, ", ".join(sams), os.path.join(tmp, "out"), L, CUT, GAP))
        in_bytes = sum(os.path.getsize(p) for p in sams) + os.path.getsize(fa)
        so = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            t0 = time.time()
            opt = config.parse_configfile(cfg)
            opt["OUTPUT_DETAILS_FOR_DEBUG"] = False
            p = pipeline.Pipeline(opt, 0, fold_model=fold_model)
            stages = {}
            for st in ("prepare", "candidate", "fold", "predict"):
                t = time.time()
                res = getattr(p, "run_" + st)()
                stages[st] = time.time() - t
            wall = time.time() - t0
            p.ctx.close()
        finally:
            sys.stdout.close()
            sys.stdout = so
        out_bytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(tmp, "out")) for f in fs)
        return {"wall_s": wall, "stage_s": stages, "input_bytes": in_bytes, "output_bytes": out_bytes, "loci": len(res or []),
                "note": "in-process CLI stage drivers (config parse -> prepare -> candidate -> fold -> predict incl. every stage artefact and report file); "
                        "interpreter start-up not included"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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
        return {"records": int(len(alns)), "sam_bytes": int(nbytes), "wall_s": wall, "records_per_s": len(alns) / wall, "sam_MB_per_s": nbytes / wall / 1e6,
                "seconds": sec, "note": "3 unsorted SAM files -> mirp_ingest_sams_gpu (host tokenizer threads, H2D, device LSD radix sort by (tid, pos), D2H copy for the host stages); "
                                        "files in the page cache"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------------------
def main():
    a = parse_args()
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
    import torch
    import torch.distributed as dist
    from mir_prefer_amd import capi, synth
    from mir_prefer_amd import dist as mdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # ---- synthetic workload: one chr1-sized contig per rank (contig sharding)
    ds = synth.make_dataset([a.genome], a.loci, n_samples=1, seed=2 + rank, contig_names=["Chr%d" % (rank + 1)])
    alns = ds.sorted_alns()
    order = np.zeros(1, dtype=np.int32)
    ctx = capi.Context(local_rank)
    ctx.set_fold_model(a.fold_model)
    ctx.load_genome(ds.contigs)
    ctx.load_alignments(alns)

    def gather_loci(out):
        """Final loci list to rank 0 over RCCL (mir_prefer_amd.dist.gather_records: all_gather of counts, gather of padded 64-B records)."""
        rec = np.zeros((len(out["result"]), 16), dtype=np.int32)
        if len(out["result"]):
            rec[:] = np.frombuffer(out["result"].tobytes(), dtype=np.int32).reshape(-1, 16)
        if world == 1:
            return rec.shape[0]
        allrec = mdist.gather_records(rec, device=dev, dst=0)
        n_all = torch.tensor([0 if allrec is None else allrec.shape[0]], device=dev, dtype=torch.int64)
        dist.broadcast(n_all, src=0)
        return int(n_all.item())

    fb = [0, 0]

    def step():
        npk, nloci, nwin = ctx.candidate(CUT, GAP, L, order)
        ctx.fold(L)
        fb[0], fb[1] = ctx.last_fold_fallbacks(), ctx.last_fold_overflow()
        out = ctx.predict(1, 18, 23, False, True)
        total = gather_loci(out)
        return nwin, total, ctx.last_timings(), ctx.last_fold_kernel_ms()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

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
    sync()
    elapsed = time.time() - t0
    if world > 1:
        t = torch.tensor([elapsed, float(nwin)], device=dev, dtype=torch.float64)
        tl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(tl, t)
        elapsed = max(float(x[0].item()) for x in tl)
        total_windows = sum(float(x[1].item()) for x in tl)
    else:
        total_windows = float(nwin)

    if rank == 0:
        fold_s = float(np.mean(fold_ms)) / 1e3
        fill_s = float(np.mean(fill_ms)) / 1e3
        epi_s = float(np.mean(epi_ms)) / 1e3
        cov_s = float(np.mean(cov_ms)) / 1e3
        wins = ctx.get_windows()
        W = wins["windows"]
        lens = W["seq_len"].astype(np.int64)
        # ---- dominant kernel: the fill kernel (dynamic program).  Unit of algorithmic work = one relaxation; R is counted
        # exactly for this batch from its pair-type counts.  Roofs measured on this GPU now: LDS = 2 reads per relaxation at the measured
        # conflict-free ds_read rate, VALU = 3 integer lane-operations per relaxation at the measured 32-bit issue rate.
        R = relaxation_count(wins["seq"], W["seq_off"].astype(np.int64), lens, L)
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
        same_workload = a.genome == CHR1_LEN and a.loci == N_LOCI and a.fold_model == "vienna-2.1.2"
        traffic = prof.get("fold_fill_hbm_bytes_per_launch") if same_workload else None
        g_tot = float(a.genome + 1)
        b_cov = 16.0 * len(alns) + 16.0 * g_tot
        line = {
            "metric": "precursor windows folded+filtered/sec (L=300)", "value": total_windows * a.steps / elapsed, "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int16/int32 (energies in 0.01 kcal/mol)", "data": "synthetic",
            "config": {"workload": "BASELINE config[1]: A. thaliana chr1-sized contig per GPU (%d bp), 1 sample, L=300, %d synthetic loci -> %d windows/GPU; "
                                   "candidate+fold+predict, inputs resident in HBM" % (a.genome, a.loci, nwin),
                       "windows_per_gpu": int(nwin), "loci_found": int(nres), "alignments_per_gpu": int(len(alns)),
                       "fold_flavour": "vienna-2.1.2 (Turner-2004, d2)" if a.fold_model == "vienna-2.1.2" else "vienna-1.8.5 (Turner-1999, d1)",
                       "fold_generic_fallback_windows": int(fb[0]), "fold_line_overflow_windows": int(fb[1])},
            "roofline": {"kernel": "fold_lds_kernel<%d>" % (0 if a.fold_model == "vienna-2.1.2" else 1),
                         "bound": "lds" if lds_roof <= valu_roof else "valu", "achieved": achieved / 1e12, "peak": roof / 1e12, "unit": "T relaxations/s",
                         "frac": achieved / roof if roof > 0 else None, "avg_launch_ms": fill_s * 1e3,
                         "relaxations_per_launch": R, "lds_roof_T": lds_roof / 1e12, "valu_roof_T": valu_roof / 1e12, "microbench_wave_insts_per_s": mb,
                         "traffic": traffic, "traffic_source": prof.get("source") if traffic is not None else None,
                         "note": "integer min-plus dynamic program out of LDS: bounded by LDS reads (2 per relaxation) or VALU issue (3 lane-ops per relaxation), "
                                 "SURVEY.md 8d; both roofs micro-benchmarked on this GPU in this run; traffic = HBM bytes per launch from the committed rocprofv3 PMC passes"},
            "roofline_hbm": {"kernel": "fold_lds_kernel + fold_lds_epilogue_kernel", "bound": "hbm", "achieved": b_fold / fold_s / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": b_fold / fold_s / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": fold_s * 1e3,
                             "note": "HBM view of the fold (algorithmic bytes / time / 8 TB/s); << 1 is expected: the tables are LDS-resident"},
            "roofline_coverage": {"kernel": "memset + cov_scatter_kernel + cov_scan_kernel", "bound": "hbm", "achieved": b_cov / cov_s / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_cov / cov_s / 1e9 / HBM_PEAK_GBS, "avg_ms": cov_s * 1e3},
            "stage_ms": {"coverage": cov_s * 1e3, "candidate_rest": float(np.mean(rest_ms)), "fold": fold_s * 1e3, "fold_fill_kernel": fill_s * 1e3,
                         "fold_epilogue_kernel": epi_s * 1e3, "predict": float(np.mean(pred_ms))},
        }
        if world == 1:      # reported baselines: rank 0 at N = 1 only, AFTER the GPU timing
            if not a.no_e2e:
                try:
                    line["e2e"] = e2e_cli(ds, a.fold_model)
                    line["e2e_wall_s"] = line["e2e"]["wall_s"]
                except SystemExit as e:
                    line["e2e"] = {"error": "CLI exited with %r" % (e.code,)}
            if not a.no_ingest:
                line["ingest"] = ingest_leg(ctx, a.ingest_records)
            if not a.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(ds, alns, order, a.cpu_budget)
        print(json.dumps(line))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
