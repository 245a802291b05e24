"""Whole-workload parity on the headline configuration (BASELINE config[1], the workload bench.py's number is quoted on): EVERY window's printed
structure lines and MFE, and the complete result list, against the CPU oracle, under both fold models -- the reference's analogue is simply its
full run (/root/reference/miR_PREFeR.py:3728-3739).  The oracle folds in a process pool (one worker per CPU the box grants): 19,686 windows take
under a minute.  Also: config[2] -- the candidate stage over the whole genome and a 12,000-window slice line by line and record by record --,
6,500 folds of the five stress families (mixed, tandem repeats, two-letter alphabets, near-perfect long hairpins, N-rich;
profiles/tools/stress_fold.py) and the two split paths of the fill kernel (split candidates / dense loop) against each other."""
import concurrent.futures as cf
import os
import random

import numpy as np
import pytest

from mir_prefer_amd import records, synth
from tests import seqgen
from tests.test_oracle_golden import mirna_record, run_predict

pytestmark = pytest.mark.gpu

G, N_LOCI, SEED, CUT, GAP, L = 30427671, 12000, 2, 10, 100, 300
# expected sizes of the seeded workload (bench.py asserts the same numbers, see EXPECTED_LOCI there)
N_WINDOWS = 19686


def _ncpu():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 32))


def _oracle_chunk(args):
    seqs, span, model = args
    from tests import oracle_binding
    o = oracle_binding.load()
    out = []
    for s in seqs:
        r = o.lfold(s, span, model=model)
        out.append((r["lines"], r["mfe"]))
    return out


def oracle_fold_all(seqs, span, model="vienna-2.1.2"):
    """[(lines, mfe)] of the CPU oracle for every sequence, folded in a spawned process pool (the parent holds a GPU context)."""
    import multiprocessing as mp
    n = len(seqs)
    step = 64
    tasks = [(seqs[k:k + step], span, model) for k in range(0, n, step)]
    out = []
    with cf.ProcessPoolExecutor(_ncpu(), mp_context=mp.get_context("spawn")) as ex:
        for part in ex.map(_oracle_chunk, tasks):
            out.extend(part)
    return out


def gpu_lines(raw, k):
    from mir_prefer_amd import capi
    wl, wss = capi.fold_window_lines(raw, k)
    nl = int(raw["n_lines"][k])
    sel = np.nonzero(wl["printed"][:nl])[0]
    return [(wss[j, :wl[j]["len"]].tobytes().decode(), int(wl[j]["energy"]), int(wl[j]["start"])) for j in sel]


@pytest.fixture(scope="module")
def config1(oracle):
    ds = synth.make_dataset([G], N_LOCI, n_samples=1, seed=SEED, contig_names=["Chr1"])
    alns = ds.sorted_alns()
    _, peaks = oracle.coverage_peaks(alns, ds.contig_lens, CUT)
    win = oracle.make_windows(peaks, alns, ds.contigs, np.zeros(1, np.int32), GAP, L, CUT * 0.5)
    seqs = [win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes() for b in win["windows"]]
    return {"ds": ds, "alns": alns, "win": win, "seqs": seqs}


@pytest.mark.parametrize("model", ["vienna-2.1.2", "vienna-1.8.5"])
def test_config1_every_window_and_the_result_list(model, gpu_ctx, oracle, config1):
    try:
        gpu_ctx.set_fold_model(model)
        _config1_whole(model, gpu_ctx, oracle, config1)
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")


def _config1_whole(model, gpu_ctx, oracle, config1):
    ds, alns, win, seqs = config1["ds"], config1["alns"], config1["win"], config1["seqs"]
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    _, _, nwin = gpu_ctx.candidate(CUT, GAP, L, np.zeros(1, np.int32))
    assert nwin == len(win["windows"]) == N_WINDOWS
    gw = gpu_ctx.get_windows()
    GW, OW = gw["windows"], win["windows"]
    for f in ("tid", "ws", "we", "strand", "loc_s", "loc_e", "tag", "n_peaks", "n_matures", "seq_len"):      # (the offsets index each side's own arrays)
        assert np.array_equal(GW[f], OW[f]), f
    for k in range(nwin):
        a, b = GW[k], OW[k]
        assert np.array_equal(gw["matures"][a["mature_off"]:a["mature_off"] + a["n_matures"]], win["matures"][b["mature_off"]:b["mature_off"] + b["n_matures"]]), k
        assert np.array_equal(gw["wpeaks"][a["peak_off"]:a["peak_off"] + a["n_peaks"]], win["wpeaks"][b["peak_off"]:b["peak_off"] + b["n_peaks"]]), k
        assert gw["seq"][a["seq_off"]:a["seq_off"] + a["seq_len"]].tobytes() == seqs[k], k
    gpu_ctx.fold(L)
    assert gpu_ctx.last_fold_fallbacks() == 0 and gpu_ctx.last_fold_dense() == 0      # every window through the candidate-pool fill kernel
    raw = gpu_ctx.get_fold()
    assert (raw["status"] == 0).all()
    want = oracle_fold_all(seqs, L, model)
    bad = [k for k in range(nwin) if raw["mfe"][k] != want[k][1] or gpu_lines(raw, k) != want[k][0]]
    assert not bad, "windows whose structure lines / MFE differ from the oracle: %s" % bad[:10]
    # the complete result list (filter_next_loci + check_loci over every window, MP:2350-2502)
    out = gpu_ctx.predict(1, 18, 23, False, True)
    assert (out["status"] == 0).all()
    structs = [oracle.structures_from_lines(w[0], 55) for w in want]
    case = {"cfg": {"MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 23, "ALLOW_3NT_OVERHANG": "N", "ALLOW_NO_STAR_EXPRESSION": "Y"}, "win": win,
            "sample_names": ds.sample_names, "alns": alns}
    decisions, result = run_predict(case, oracle, structs)
    names = ds.contig_names
    got = [[names[m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]), int(m["star_e"]), ss,
            records.STRAND[m["strand"]], bool(m["has_star"])] for m, ss in zip(out["result"], out["ss"])]
    exp = [mirna_record(m, names) for _, m in result]
    assert len(got) == len(exp)
    assert got == exp
    assert [int(m["window"]) for m in out["result"]] == [k for k, _ in result]
    import bench
    assert len(got) == bench.EXPECTED_LOCI[("config1", model)]


def test_config1_slice_dense_split_path_equals_candidate_pool_pass(gpu_ctx, config1):
    """5,000 windows of the headline workload through the default model's dense split loop must reproduce the candidate-pool pass bit for bit (the
    test above pins that one on the oracle)."""
    seqs = config1["seqs"][3000:8000]
    a = gpu_ctx.fold_batch_raw(seqs, L)
    assert gpu_ctx.last_fold_dense() == 0
    try:
        gpu_ctx.set_fold_split_path(1)
        b = gpu_ctx.fold_batch_raw(seqs, L)
    finally:
        gpu_ctx.set_fold_split_path(0)
    for key in ("n_lines", "mfe", "status", "lines", "ss"):
        assert np.array_equal(a[key], b[key]), key


_family = seqgen.stress_family


@pytest.mark.parametrize("model,count", [("vienna-2.1.2", 5000), ("vienna-1.8.5", 1500)])
def test_stress_families(gpu_ctx, model, count):
    """The five sequence families of the differential stress (energy ties, tandem repeats whose split-candidate pools overflow into the dense
    kernel, long hairpins, N runs), every line and MFE against the oracle."""
    r = random.Random(20261003)
    seqs = [_family(r, i % 5) for i in range(count)]
    try:
        gpu_ctx.set_fold_model(model)
        got = gpu_ctx.fold_batch(seqs, L)
        n_dense = gpu_ctx.last_fold_dense()
        capped = [k for k, g in enumerate(got) if g["status"] == 1]      # more than 96 lines: folded again at full capacity, as the host does
        if capped:
            for k, g in zip(capped, gpu_ctx.fold_batch([seqs[k] for k in capped], L, max_lines=352)):
                got[k] = g
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")
    if model == "vienna-2.1.2":
        assert n_dense > 0      # the tandem-repeat family reaches the overflow pass
    want = oracle_fold_all(seqs, L, model)
    bad = [k for k in range(count) if got[k]["status"] != 0 or got[k]["mfe"] != want[k][1] or got[k]["lines"] != want[k][0]]
    assert not bad, [seqs[k] for k in bad[:3]]


_microsatellites = seqgen.microsatellites


@pytest.mark.parametrize("model", ["vienna-1.8.5", "vienna-2.1.2"])
@pytest.mark.parametrize("span", [300, 150])
def test_microsatellite_pools_take_the_dense_hand_off(gpu_ctx, model, span):
    """a pool larger than the compaction's register slice must be handed to the dense kernel, never truncated.  Sparse pass ==
    dense split loop == oracle on microsatellite windows, and the hand-off is seen to happen."""
    seqs = _microsatellites()
    try:
        gpu_ctx.set_fold_model(model)
        a = gpu_ctx.fold_batch(seqs, span, max_lines=352)
        n_dense = gpu_ctx.last_fold_dense()
        gpu_ctx.set_fold_split_path(1)
        b = gpu_ctx.fold_batch(seqs, span, max_lines=352)
    finally:
        gpu_ctx.set_fold_split_path(0)
        gpu_ctx.set_fold_model("vienna-2.1.2")
    assert n_dense > 0
    want = oracle_fold_all(seqs, span, model)
    for k, s in enumerate(seqs):
        assert a[k]["status"] == 0 and b[k]["status"] == 0, s
        assert (a[k]["lines"], a[k]["mfe"]) == (b[k]["lines"], b[k]["mfe"]), s
        assert (a[k]["lines"], a[k]["mfe"]) == (want[k][0], want[k][1]), s


def test_config2_slice_of_12000_windows_lines_and_records(gpu_ctx, oracle):
    """BASELINE config[2] (TAIR10-sized, 5 contigs, 3 samples, 70,244 windows): the candidate stage against the oracle over the whole genome, and a
    contiguous slice of 12,000 windows (it crosses a contig boundary) line by line and record by record -- fold text through the batch entry
    point, filter records out of the resident run's result list."""
    from mir_prefer_amd import balance
    import bench
    specs, n_samples, background, _, _ = bench.workload_specs("config2", 1)          # the workload bench.py's `configs.config2` line is measured on
    contigs, alns, sample_names = bench.build_shard(specs, set(range(len(specs))), n_samples, background)
    ds = synth.Dataset(contigs, sample_names, alns, [])
    order = np.arange(5, dtype=np.int32)
    gpu_ctx.set_fold_model("vienna-2.1.2")
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    _, _, nwin = gpu_ctx.candidate(CUT, GAP, L, order)
    gw = gpu_ctx.get_windows()
    _, peaks = oracle.coverage_peaks(alns, ds.contig_lens, CUT)
    win = oracle.make_windows(peaks, alns, ds.contigs, order, GAP, L, CUT * 0.5)
    assert nwin == len(win["windows"]) == 70244
    for f in ("tid", "ws", "we", "strand", "loc_s", "loc_e", "tag", "n_peaks", "n_matures", "seq_len"):
        assert np.array_equal(gw["windows"][f], win["windows"][f]), f
    W = win["windows"]
    units = balance.parse_units(W["tag"])
    a = int(units[np.searchsorted(units, 14000)]); b = int(units[np.searchsorted(units, a + 12000)])
    assert len(np.unique(W["tid"][a:b])) >= 2
    seqs = [win["seq"][w["seq_off"]:w["seq_off"] + w["seq_len"]].tobytes() for w in W[a:b]]
    assert seqs == [gw["seq"][w["seq_off"]:w["seq_off"] + w["seq_len"]].tobytes() for w in gw["windows"][a:b]]
    raw = gpu_ctx.fold_batch_raw(seqs, L)
    assert (raw["status"] == 0).all()
    want = oracle_fold_all(seqs, L)
    bad = [k for k in range(b - a) if raw["mfe"][k] != want[k][1] or gpu_lines(raw, k) != want[k][0]]
    assert not bad, bad[:10]
    gpu_ctx.fold(L)
    out = gpu_ctx.predict(3, 18, 23, False, True)
    assert (out["status"] == 0).all() and len(out["result"]) == bench_expected("config2")
    params = (3, 18, 23, 0, 1, 55)
    exp = []
    for u in range(int(np.searchsorted(units, a)), int(np.searchsorted(units, b))):
        for k in range(int(units[u]), int(units[u + 1])):
            w = W[k]
            r = oracle.check_loci(oracle.structures_from_lines(want[k - a][0], 55), win["matures"][w["mature_off"]:w["mature_off"] + w["n_matures"]], w, alns, params)
            assert out["n_passed"][k] == len(r), k
            if r:
                exp.append((k, mirna_record(r[0], ds.contig_names)))
                break
    sel = (out["result"]["window"] >= a) & (out["result"]["window"] < b)
    got = [(int(m["window"]), [ds.contig_names[m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]), int(m["star_e"]), ss,
                               records.STRAND[m["strand"]], bool(m["has_star"])]) for m, ss, t in zip(out["result"], out["ss"], sel) if t]
    assert len(got) > 1500 and got == exp


def bench_expected(workload, model="vienna-2.1.2"):
    import bench
    return bench.EXPECTED_LOCI[(workload, model)]
