"""BASELINE.json configs 2-4 at full (per-GPU) size through the C-ABI on MI355X.  The oracle cannot fold 10^5 windows in a test, so
the checks are size-independent properties (coverage linearity against a brute-force pile-up, sorted / disjoint peaks, windows inside
their contigs, determinism, no generic-kernel fallbacks, zero status) plus oracle spot checks of the fold lines and of the filter's
decisions on random windows.  config[1] is tests/test_edge_and_scale_gpu.py::test_full_size_config1_properties.

  config[2]  A. thaliana TAIR10 full genome (5 contigs, 119,146,348 bp), 3 samples, L = 300, one GPU
  config[3]  O. sativa MSU7-like (12 contigs, 373 Mb), 4 samples, contig-sharded over 8 GPUs: the most loaded rank's shard
  config[4]  synthetic 2 Gb / 2 x 10^8 records over 8 GPUs: one rank's shard = 8 x 31.25 Mb contigs, 2.5 x 10^7 packed records
"""
import numpy as np
import pytest

from mir_prefer_amd import capi, dist, records, synth

pytestmark = pytest.mark.gpu

TAIR10 = [30427671, 19698289, 23459830, 18585056, 26975502]
MSU7 = [43300000, 35900000, 36400000, 35500000, 30000000, 31200000, 29700000, 28400000, 23000000, 23200000, 29000000, 27500000]


def _mem_used_gb():
    """Device memory in use, straight from the HIP runtime the library itself runs on (hipMemGetInfo)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    rc = hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    assert rc == 0, "hipMemGetInfo failed (%d)" % rc
    return (total.value - free.value) / 2.0 ** 30


def _brute_depth(alns, key, tid, pos, cut):
    """Weighted pile-up (min(depth, CUT) per record, SURVEY A-2) at (tid, pos) from the (tid, pos)-sorted records (key = tid << 32 | pos)."""
    lo = np.searchsorted(key, (int(tid) << 32) | max(int(pos) - 64, 0), side="left")
    hi = np.searchsorted(key, (int(tid) << 32) | int(pos), side="right")
    a = alns[lo:hi]
    a = a[a["pos"].astype(np.int64) + a["len"] > pos]
    w = np.minimum(a["depth"], cut).astype(np.int64)
    return int(w[a["strand"] == 0].sum()), int(w[a["strand"] == 1].sum())


def _check_candidate(ctx, alns, contig_lens, cut, L, rng, n_depth=400):
    depth, peaks = ctx.get_depth(), ctx.get_peaks()
    assert len(depth) > 0
    key = alns["tid"].astype(np.int64) << 32 | alns["pos"].astype(np.int64)
    for k in rng.choice(len(depth), min(n_depth, len(depth)), replace=False):
        d = depth[k]
        assert (int(d["dp"]), int(d["dm"])) == _brute_depth(alns, key, d["tid"], d["pos"], cut) and d["dp"] + d["dm"] > cut
    dkey = depth["tid"].astype(np.int64) << 32 | depth["pos"].astype(np.int64)
    assert (np.diff(dkey) > 0).all()
    # the thresholded positions are exactly the union of the runs: every peak is >= 19 long, sorted and disjoint inside its contig, and
    # positions just outside a peak are below the threshold
    pkey = peaks["tid"].astype(np.int64) << 32 | peaks["start"].astype(np.int64)
    assert (np.diff(pkey) > 0).all() and (peaks["end"] - peaks["start"] >= 19).all()
    same = peaks["tid"][1:] == peaks["tid"][:-1]
    assert (peaks["start"][1:][same] > peaks["end"][:-1][same]).all()
    for k in rng.choice(len(peaks), min(200, len(peaks)), replace=False):
        p = peaks[k]
        for pos in (int(p["start"]) - 1, int(p["end"])):
            if 1 <= pos <= contig_lens[p["tid"]]:
                a, b = _brute_depth(alns, key, p["tid"], pos, cut)
                inside_other = np.searchsorted(dkey, (int(p["tid"]) << 32) | pos)
                hit = inside_other < len(dkey) and dkey[inside_other] == ((int(p["tid"]) << 32) | pos)
                assert (a + b > cut) == bool(hit)
        a, b = _brute_depth(alns, key, p["tid"], int(p["start"]), cut)
        assert a + b > cut
    win = ctx.get_windows()
    W = win["windows"]
    clen = np.asarray(contig_lens)[W["tid"]]
    assert (W["ws"] >= 0).all() and (W["we"] <= clen + 1).all() and (W["ws"] <= W["loc_s"]).all() and (W["we"] >= W["loc_e"]).all()
    assert (W["seq_len"] <= L + 50).all() and (W["seq_len"] > 0).all()
    return win


def _check_fold_and_predict(ctx, win, alns, names, sample_names, L, rng, oracle, n_fold=24, n_filter=160, model="vienna-2.1.2"):
    from tests.test_edge_and_scale_gpu import _window_lines
    W = win["windows"]
    nwin = len(W)
    ctx.fold(L)
    s1 = ctx.fold_summary()
    assert (s1["status"] == 0).all() and ctx.last_fold_fallbacks() == 0
    raw = ctx.get_fold()
    ctx.fold(L)
    s2 = ctx.fold_summary()
    assert np.array_equal(s1["n_lines"], s2["n_lines"]) and np.array_equal(s1["mfe"], s2["mfe"])          # deterministic across launches
    pick = rng.choice(nwin, n_filter, replace=False)
    structs = {}
    for j, k in enumerate(pick):
        b = W[k]
        ref = oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L, model=model)
        if j < n_fold:
            assert _window_lines(raw, k) == ref["lines"] and raw["mfe"][k] == ref["mfe"], k
        else:
            assert raw["mfe"][k] == ref["mfe"] and raw["n_lines"][k] >= len(ref["lines"]), k
        structs[k] = oracle.structures_from_lines(ref["lines"], 55)
    ns = len(sample_names)
    out = ctx.predict(ns, 18, 23, False, True)
    assert (out["status"] == 0).all()
    res = out["result"]
    assert (res["fold_s"] < res["fold_e"]).all() and (res["mat_e"] - res["mat_s"] >= 18).all() and (res["mat_e"] - res["mat_s"] <= 23).all()
    assert (res["mat_s"] >= res["fold_s"]).all() and (res["mat_e"] <= res["fold_e"]).all()
    # the filter's per-window decision (len(miRNAs) of check_loci, MP:2206-2347) against the oracle on the sampled windows
    params = (ns, 18, 23, 0, 1, 55)
    by_window = {int(m["window"]): (m, ss) for m, ss in zip(res, out["ss"])}
    n_pass = 0
    for k in pick:
        b = W[k]
        mats = win["matures"][b["mature_off"]:b["mature_off"] + b["n_matures"]]
        r = oracle.check_loci(structs[k], mats, b, alns, params)
        assert out["n_passed"][k] == len(r), k
        if r and int(k) in by_window:            # reported unless it is the R entry of a pair whose L entry passed
            m, ss = by_window[int(k)]
            o = r[0]
            assert [m["fold_s"], m["fold_e"], m["mat_s"], m["mat_e"], m["star_s"], m["star_e"], ss, m["strand"], bool(m["has_star"])] == \
                   [o.fold_s, o.fold_e, o.mat_s, o.mat_e, o.star_s, o.star_e, o.ss.decode(), o.strand, bool(o.has_star)], k
            n_pass += 1
    return nwin, len(res), n_pass


@pytest.mark.parametrize("model", ["vienna-2.1.2", "vienna-1.8.5"])
def test_config2_tair10_full_genome_three_samples(model, gpu_ctx, oracle):
    """model vienna-1.8.5: the same full-size run through fold_lds_kernel<1> + fold185_lds_epilogue_kernel (Turner-1999, dangles 1, multi-component
    structure lines), fold lines and filter decisions against that model's oracle."""
    rng = np.random.RandomState(2)
    gpu_ctx.set_fold_model(model)
    ds = synth.make_dataset(TAIR10, 48000, n_samples=3, seed=3, contig_names=["Chr%d" % (i + 1) for i in range(5)])
    alns = ds.sorted_alns()
    assert len(ds.sample_names) == 3 and sum(TAIR10) == 119146348
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    order = np.arange(5, dtype=np.int32)
    npk, nloci, nwin = gpu_ctx.candidate(10, 100, 300, order)
    assert 60000 < nwin < 100000
    win = _check_candidate(gpu_ctx, alns, ds.contig_lens, 10, 300, rng)
    assert len(np.unique(win["windows"]["tid"])) == 5
    try:
        n, nres, n_pass = _check_fold_and_predict(gpu_ctx, win, alns, ds.contig_names, ds.sample_names, 300, rng, oracle, model=model,
                                                  n_filter=160 if model == "vienna-2.1.2" else 80)
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")
    assert nres > 4000 and n_pass > 5
    print("config[2] (%s): %d windows, %d loci, device memory in use %.1f GB" % (model, n, nres, _mem_used_gb()))


def _genome_with_reads_on(lens, names, with_reads, loci_per_mb, n_samples, seed):
    """A multi-contig genome whose reads live on the contigs `with_reads` only (one rank's shard of a contig-sharded run: every rank holds the
    whole genome and its own alignments, pipeline.Pipeline._load_inputs)."""
    rng = np.random.RandomState(seed)
    contigs, alns = [], []
    for t, ln in enumerate(lens):
        if t in with_reads:
            d = synth.make_dataset([ln], int(loci_per_mb * ln / 1e6), n_samples=n_samples, seed=seed + 17 * t, contig_names=[names[t]])
            contigs.append(d.contigs[0])
            a = d.sorted_alns().copy()
            a["tid"] = t
            alns.append(a)
            samples = d.sample_names
        else:
            contigs.append((names[t], synth._BASES[rng.randint(0, 4, size=ln, dtype=np.uint8)]))
    return contigs, np.concatenate(alns), samples


def test_config3_msu7_rank_shard_four_samples(gpu_ctx, oracle):
    """The 12 MSU7-sized contigs are dealt to 8 ranks by the longest-processing-time rule (dist.partition_contigs); this is the most loaded
    rank: the whole genome resident, its own contigs' alignments, 4 samples."""
    rng = np.random.RandomState(3)
    parts = dist.partition_contigs(MSU7, 8)
    assert sorted(t for p in parts for t in p) == list(range(12))
    loads = [sum(MSU7[t] for t in p) for p in parts]
    mine = parts[int(np.argmax(loads))]
    assert len(mine) == 2 and max(loads) < 1.35 * sum(MSU7) / 8
    names = ["Chr%d" % (i + 1) for i in range(12)]
    contigs, alns, samples = _genome_with_reads_on(MSU7, names, set(mine), 400, 4, seed=40)
    key = alns["tid"].astype(np.int64) << 32 | alns["pos"].astype(np.int64)
    assert (np.diff(key) >= 0).all() and len(samples) == 4 and set(np.unique(alns["sample"])) == {0, 1, 2, 3}
    gpu_ctx.load_genome(contigs)
    gpu_ctx.load_alignments(alns)
    gpu_ctx.set_contig_shard(mine[0] != 0)        # a covered contig precedes this shard's first one on another rank (MP:905-906 + 926-929)
    try:
        order = np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)
        npk, nloci, nwin = gpu_ctx.candidate(10, 100, 300, order)
        assert nwin > 25000
        lens = np.array(MSU7, dtype=np.int64)
        win = _check_candidate(gpu_ctx, alns, lens, 10, 300, rng)
        assert set(np.unique(win["windows"]["tid"])) == set(mine)
        n, nres, n_pass = _check_fold_and_predict(gpu_ctx, win, alns, names, samples, 300, rng, oracle, n_fold=16, n_filter=120)
        assert nres > 1500 and n_pass > 3
        print("config[3] shard %s: %d windows, %d loci, device memory in use %.1f GB" % (mine, n, nres, _mem_used_gb()))
    finally:
        gpu_ctx.set_contig_shard(False)


_packed_records_shard = synth.packed_records_shard


def test_config4_rank_shard_packed_records(gpu_ctx, oracle):
    """One rank's share of the 2 Gb / 2 x 10^8-record stress: 8 contigs x 31.25 Mb = 250 Mb and 2.5 x 10^7 packed records in one call.
    This is where 32-bit offsets, the slab sub-batching of the fold and the capacities sized from the record count would break."""
    rng = np.random.RandomState(4)
    nc, clen = 8, 31250000
    contigs = [("ctg%02d" % t, synth._BASES[rng.randint(0, 4, size=clen, dtype=np.uint8)]) for t in range(nc)]
    alns = _packed_records_shard(nc, clen, 150000, 167, seed=44)
    assert len(alns) == 150000 // nc * nc * 167 and len(alns) > 2.4e7
    gpu_ctx.load_genome(contigs)
    gpu_ctx.load_alignments(alns)
    names = [n for n, _ in contigs]
    order = np.arange(nc, dtype=np.int32)
    npk, nloci, nwin = gpu_ctx.candidate(10, 100, 300, order)
    assert nwin > 200000 and nloci > 140000
    lens = np.full(nc, clen, dtype=np.int64)
    win = _check_candidate(gpu_ctx, alns, lens, 10, 300, rng, n_depth=300)
    # scan totals: the number of thresholded positions of a sampled stretch equals the brute-force count
    depth = gpu_ctx.get_depth()
    for t in (0, nc - 1):
        a0 = int(rng.randint(1, clen - 200000))
        sel = alns[(alns["tid"] == t) & (alns["pos"] >= a0 - 64) & (alns["pos"] < a0 + 200000)]
        cov = np.zeros(200064 + 64, dtype=np.int64)
        w = np.minimum(sel["depth"], 10).astype(np.int64)
        for r, ww in zip(sel, w):
            x = int(r["pos"]) - (a0 - 64)
            cov[x:x + int(r["len"])] += ww
        want = int((cov[64:64 + 200000] > 10).sum())
        got = int(((depth["tid"] == t) & (depth["pos"] >= a0) & (depth["pos"] < a0 + 200000)).sum())
        assert got == want
    n, nres, n_pass = _check_fold_and_predict(gpu_ctx, win, alns, names, ["S1"], 300, rng, oracle, n_fold=16, n_filter=100)
    assert n == nwin
    used = _mem_used_gb()
    print("config[4] shard: %d records, %d windows, %d loci, device memory in use %.1f GB" % (len(alns), n, nres, used))
    assert used < 200
