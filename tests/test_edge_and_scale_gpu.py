"""GPU edge cases (empty / ragged inputs, other PRECURSOR_LEN values incl. the generic-kernel path) and full-size
(BASELINE config[1]) checks through size-independent properties plus oracle spot checks."""
import numpy as np
import pytest

from mir_prefer_amd import records, synth
from tests.test_oracle_golden import mirna_record, run_predict

pytestmark = pytest.mark.gpu


def _order(names):
    return np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)


def _gpu_records(out, names):
    return [[names[m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]), int(m["star_e"]), ss,
             records.STRAND[m["strand"]], bool(m["has_star"])] for m, ss in zip(out["result"], out["ss"])]


def test_no_alignments_and_unreached_threshold(gpu_ctx):
    ds = synth.make_dataset([5000, 3000], 0, seed=1)
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(np.zeros(0, dtype=synth.ALN_DTYPE))
    assert gpu_ctx.candidate(10, 100, 300, _order(ds.contig_names)) == (0, 0, 0)
    assert len(gpu_ctx.get_depth()) == 0 and len(gpu_ctx.get_peaks()) == 0
    gpu_ctx.fold(300)
    out = gpu_ctx.predict(1, 18, 23, False, True)
    assert len(out["result"]) == 0 and len(out["n_passed"]) == 0
    # a single read can never form a peak: its weight is capped at CUT and the threshold is strict (SURVEY A-2)
    one = np.array([(0, 100, 5000, 21, 0, 0)], dtype=synth.ALN_DTYPE)
    gpu_ctx.load_alignments(one)
    assert gpu_ctx.candidate(10, 100, 300, _order(ds.contig_names)) == (0, 0, 0)


def test_unsorted_alignments_are_rejected(gpu_ctx):
    from mir_prefer_amd import capi
    ds = synth.make_dataset([5000], 0, seed=1)
    gpu_ctx.load_genome(ds.contigs)
    bad = np.array([(0, 200, 50, 21, 0, 0), (0, 100, 50, 21, 0, 0)], dtype=synth.ALN_DTYPE)
    with pytest.raises(capi.MirpError):
        gpu_ctx.load_alignments(bad)


@pytest.mark.parametrize("L", [150, 300, 400])
def test_other_precursor_lengths_match_oracle(L, gpu_ctx, oracle):
    """L = 150 and 300 run the LDS-resident fold kernel, L = 400 the generic one (span > 300)."""
    ds = synth.make_dataset([150000, 90000], 120, n_samples=2, seed=30 + L, contig_names=["k2", "k1"], edge_cases=True)
    names, alns = ds.contig_names, ds.sorted_alns()
    cut, gap = 8, 80
    _, peaks = oracle.coverage_peaks(alns, ds.contig_lens, cut)
    win = oracle.make_windows(peaks, alns, ds.contigs, _order(names), gap, L, cut * 0.5)
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    _, _, nwin = gpu_ctx.candidate(cut, gap, L, _order(names))
    assert nwin == len(win["windows"]) and nwin > 50
    gpu_ctx.fold(L)
    raw = gpu_ctx.get_fold()
    assert (raw["status"] == 0).all()
    structs = []
    for k, b in enumerate(win["windows"]):
        r = oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L)
        got = [(raw["ss"][k, j, :raw["lines"][k, j]["len"]].tobytes().decode(), int(raw["lines"][k, j]["energy"]), int(raw["lines"][k, j]["start"]))
               for j in range(raw["n_lines"][k]) if raw["lines"][k, j]["printed"]]
        assert got == r["lines"], k
        assert raw["mfe"][k] == r["mfe"]
        structs.append(oracle.structures_from_lines(r["lines"], 55))
    out = gpu_ctx.predict(2, 18, 24, True, True)
    case = {"cfg": {"MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 24, "ALLOW_3NT_OVERHANG": "Y", "ALLOW_NO_STAR_EXPRESSION": "Y"}, "win": win,
            "sample_names": ds.sample_names, "alns": alns}
    _, result = run_predict(case, oracle, structs)
    assert _gpu_records(out, names) == [mirna_record(m, names) for _, m in result]


def test_precursor_length_2500_through_the_whole_path(gpu_ctx, oracle):
    """PRECURSOR_LEN in the thousands (the reference accepts up to 3000, MP:167-184) from the candidate stage to the loci list: windows of 2,500 nt and more go
    through the generic fold kernels (over 64 KB of LDS), carry hundreds of structure lines of thousands of characters -- the filter kernel then keeps the
    staged text in global memory instead of LDS -- and the windows, every structure line and the loci equal the oracle chain's."""
    import concurrent.futures as cf
    L = 2500
    ds = synth.make_dataset([40000, 30000], 6, n_samples=2, seed=2500, contig_names=["k2", "k1"], edge_cases=True)
    names, alns = ds.contig_names, ds.sorted_alns()
    cut, gap = 8, 80
    _, peaks = oracle.coverage_peaks(alns, ds.contig_lens, cut)
    win = oracle.make_windows(peaks, alns, ds.contigs, _order(names), gap, L, cut * 0.5)
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    _, _, nwin = gpu_ctx.candidate(cut, gap, L, _order(names))
    assert nwin == len(win["windows"]) and 4 <= nwin <= 40
    gpu_ctx.fold(L)
    raw = gpu_ctx.get_fold()
    assert (raw["status"] == 0).all()
    with cf.ThreadPoolExecutor(16) as ex:      # the oracle's folds side by side (ctypes releases the GIL): seconds each at this length
        folds = list(ex.map(lambda b: oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L), win["windows"]))
    from mir_prefer_amd import capi
    structs = []
    for k, r in enumerate(folds):
        wl, ws = capi.fold_window_lines(raw, k)          # (windows with more lines than the main buffers hold live in the side buffers)
        got = [(ws[j, :wl[j]["len"]].tobytes().decode(), int(wl[j]["energy"]), int(wl[j]["start"])) for j in range(raw["n_lines"][k]) if wl[j]["printed"]]
        assert got == r["lines"], k
        assert raw["mfe"][k] == r["mfe"]
        structs.append(oracle.structures_from_lines(r["lines"], 55))
    assert max(len(r["lines"]) for r in folds) > 150
    out = gpu_ctx.predict(2, 18, 24, True, True)
    case = {"cfg": {"MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 24, "ALLOW_3NT_OVERHANG": "Y", "ALLOW_NO_STAR_EXPRESSION": "Y"}, "win": win,
            "sample_names": ds.sample_names, "alns": alns}
    _, result = run_predict(case, oracle, structs)
    assert _gpu_records(out, names) == [mirna_record(m, names) for _, m in result] and len(result) >= 1


def test_full_size_config1_properties(gpu_ctx, oracle):
    """BASELINE config[1] size: 30,427,671-bp contig, 12,000 loci (~20 k windows)."""
    G = 30427671
    ds = synth.make_dataset([G], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
    alns = ds.sorted_alns()
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    npk, nloci, nwin = gpu_ctx.candidate(10, 100, 300, np.zeros(1, np.int32))
    depth, peaks = gpu_ctx.get_depth(), gpu_ctx.get_peaks()
    # coverage linearity: every thresholded depth equals the brute-force weighted pile-up at that position (sampled)
    w = np.minimum(alns["depth"], 10).astype(np.int64)
    rng = np.random.RandomState(0)
    for k in rng.choice(len(depth), 300, replace=False):
        pos = depth[k]["pos"]
        cov = (alns["pos"] <= pos) & (alns["pos"] + alns["len"] > pos)
        assert depth[k]["dp"] == w[cov & (alns["strand"] == 0)].sum() and depth[k]["dm"] == w[cov & (alns["strand"] == 1)].sum()
        assert depth[k]["dp"] + depth[k]["dm"] > 10
    assert (np.diff(depth["pos"].astype(np.int64)) > 0).all()
    # checksum: thresholded positions are exactly the union of the runs; peaks are sorted, disjoint, >= 19 long
    assert (peaks["end"] - peaks["start"] >= 19).all() and (peaks["start"][1:] > peaks["end"][:-1]).all()
    inpk = np.zeros(G + 2, dtype=bool)
    for p in peaks:
        inpk[p["start"]:p["end"]] = True
    assert inpk[depth["pos"]].sum() == (peaks["end"] - peaks["start"]).sum()
    # windows stay inside the contig and hold their locus
    win = gpu_ctx.get_windows()
    W = win["windows"]
    assert nwin == len(W) and 15000 < nwin < 25000
    assert (W["ws"] >= 0).all() and (W["we"] <= G + 1).all() and (W["ws"] <= W["loc_s"]).all() and (W["we"] >= W["loc_e"]).all()
    assert (W["seq_len"] <= 350).all()
    # fold: deterministic across launches, and identical to the oracle on a random sample of windows
    gpu_ctx.fold(300)
    r1 = gpu_ctx.get_fold()
    gpu_ctx.fold(300)
    r2 = gpu_ctx.get_fold()
    assert (r1["status"] == 0).all()
    for key in ("n_lines", "mfe"):
        assert np.array_equal(r1[key], r2[key])
    assert np.array_equal(r1["lines"], r2["lines"])
    for k in rng.choice(nwin, 48, replace=False):
        b = W[k]
        ref = oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), 300)
        got = [(r1["ss"][k, j, :r1["lines"][k, j]["len"]].tobytes().decode(), int(r1["lines"][k, j]["energy"]), int(r1["lines"][k, j]["start"]))
               for j in range(r1["n_lines"][k]) if r1["lines"][k, j]["printed"]]
        assert got == ref["lines"] and r1["mfe"][k] == ref["mfe"]
    # every structure line is a balanced dot-bracket string that fits its window
    for k in rng.choice(nwin, 500, replace=False):
        for j in range(r1["n_lines"][k]):
            ln = r1["lines"][k, j]
            s = r1["ss"][k, j, :ln["len"]].tobytes()
            assert s.count(b"(") == s.count(b")") and ln["start"] >= 1 and ln["start"] + ln["len"] - 1 <= W[k]["seq_len"] + 1
    out = gpu_ctx.predict(1, 18, 23, False, True)
    res = out["result"]
    assert len(res) > 1000
    assert (res["fold_s"] < res["fold_e"]).all() and (res["mat_e"] - res["mat_s"] >= 18).all() and (res["mat_e"] - res["mat_s"] <= 23).all()
    assert (res["mat_s"] >= res["fold_s"]).all() and (res["mat_e"] <= res["fold_e"]).all()
    # most planted hairpins are recovered (sanity of the synthetic workload, not a parity claim)
    found = np.zeros(G + 2, dtype=bool)
    for m in res:
        found[m["fold_s"]:m["fold_e"]] = True
    hit = sum(found[(a + b) // 2] for _, a, b, _ in ds.planted)
    assert hit > 0.6 * len(ds.planted)


def _window_lines(raw, k):
    """Printed structure lines (text, energy, start) of window k of a get_fold() result, overflow windows included."""
    from mir_prefer_amd import capi
    wl, wss = capi.fold_window_lines(raw, k)
    return [(wss[j, :wl[j]["len"]].tobytes().decode(), int(wl[j]["energy"]), int(wl[j]["start"])) for j in range(raw["n_lines"][k]) if wl[j]["printed"]]


def _repeat_dataset():
    """A 60 kb contig with a 700-nt (GTGG)n tandem repeat under a read peak: its window folds into > 96 structure lines."""
    ds = synth.make_dataset([60000], 30, n_samples=1, seed=5, contig_names=["c1"])
    name, seq = ds.contigs[0]
    seq = seq.copy()
    seq[19800:20500] = np.frombuffer(("GTGG" * 175).encode(), dtype=np.uint8)
    ds.contigs[0] = (name, seq)
    extra = np.zeros(4, dtype=synth.ALN_DTYPE)      # a read contributes min(depth, cutoff) to the coverage: stack three isomiRs
    extra[0] = (0, 20101, 60, 21, 0, 0)
    extra[1] = (0, 20101, 20, 22, 0, 0)
    extra[2] = (0, 20102, 15, 21, 0, 0)
    extra[3] = (0, 20160, 8, 21, 0, 0)
    ds.alns = np.concatenate([ds.alns, extra])
    return ds


def test_fold_line_capacity_flag_and_large_capacity(gpu_ctx, oracle):
    s = "GUGG" * 59
    want = oracle.lfold(s, 300)
    assert len(want["lines"]) > 96
    assert gpu_ctx.fold_batch([s], 300)[0]["status"] == 1           # default capacity of 96 lines: flagged, never silently truncated
    got = gpu_ctx.fold_batch([s], 300, max_lines=352)[0]
    assert got["status"] == 0 and got["mfe"] == want["mfe"] and got["lines"] == want["lines"]


def test_pipeline_with_a_window_over_the_default_line_capacity(gpu_ctx, oracle):
    from tests.test_oracle_golden import mirna_record, run_predict
    ds = _repeat_dataset()
    names, alns = ds.contig_names, ds.sorted_alns()
    cut, gap, L = 10, 100, 300
    order = np.zeros(1, dtype=np.int32)
    depth, peaks = oracle.coverage_peaks(alns, ds.contig_lens, cut)
    win = oracle.make_windows(peaks, alns, ds.contigs, order, gap, L, cut * 0.5)
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    npk, nloci, nwin = gpu_ctx.candidate(cut, gap, L, order)
    assert nwin == len(win["windows"])
    gpu_ctx.fold(L)
    # the repeat window overflows the default capacity of 96 lines: mirp_fold folds it (alone) again at full capacity into the side buffers
    assert gpu_ctx.last_fold_overflow() >= 1 and (gpu_ctx.fold_status() == 0).all()
    raw = gpu_ctx.get_fold()
    assert len(raw["overflow"]) == gpu_ctx.last_fold_overflow()
    for k in raw["overflow"]:
        b = win["windows"][k]
        ref = oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L)
        assert raw["n_lines"][k] > 96 and raw["mfe"][k] == ref["mfe"]
        assert _window_lines(raw, k) == ref["lines"]
    out = gpu_ctx.predict(1, 18, 23, False, True)
    structs = []
    for b in win["windows"]:
        r = oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L)
        structs.append(oracle.structures_from_lines(r["lines"], 55))
    case = {"cfg": {"MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 23, "ALLOW_3NT_OVERHANG": "N", "ALLOW_NO_STAR_EXPRESSION": "Y"}, "win": win,
            "sample_names": ds.sample_names, "alns": alns}
    _, result = run_predict(case, oracle, structs)
    want = [mirna_record(m, names) for _, m in result]
    got = [[names[m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]), int(m["star_e"]), ss,
            records.STRAND[m["strand"]], bool(m["has_star"])] for m, ss in zip(out["result"], out["ss"])]
    assert got == want


def test_planted_repeat_windows_are_refolded_alone(gpu_ctx, oracle):
    """BASELINE config[1] size with 60 planted tandem repeats under read peaks: only those windows are folded again (RNALfold has no line
    limit, miR_PREFeR.py:3053), the stage costs about what the repeat-free run costs, and their lines equal the oracle's."""
    G = 30427671
    ds = synth.make_dataset([G], 12000, n_samples=1, seed=2, contig_names=["Chr1"])
    alns0 = ds.sorted_alns()
    order = np.zeros(1, np.int32)
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns0)
    _, _, nwin0 = gpu_ctx.candidate(10, 100, 300, order)
    base = []
    for _ in range(3):
        gpu_ctx.fold(300)
        base.append(gpu_ctx.last_timings()["fold_ms"])
    assert gpu_ctx.last_fold_overflow() == 0
    # plant (GTGG)n repeats of 700 nt in read-free stretches and put a small read stack on each
    name, seq = ds.contigs[0]
    seq = seq.copy()
    # read-free gaps of >= 1,700 nt between consecutive alignments: a repeat goes into the middle of every such gap until 60 are planted
    ends = np.maximum.accumulate(alns0["pos"].astype(np.int64) + alns0["len"])
    gap_lo, gap_hi = ends[:-1], alns0["pos"][1:].astype(np.int64)
    extra, planted = [], 0
    for k in np.nonzero(gap_hi - gap_lo >= 1700)[0][::7]:
        p = int((gap_lo[k] + gap_hi[k]) // 2) - 350
        seq[p:p + 700] = np.frombuffer(("GTGG" * 175).encode(), dtype=np.uint8)
        extra += [(0, p + 301, 60, 21, 0, 0), (0, p + 301, 20, 22, 0, 0), (0, p + 302, 15, 21, 0, 0), (0, p + 360, 8, 21, 0, 0)]
        planted += 1
        if planted == 60:
            break
    assert planted == 60
    ds.contigs[0] = (name, seq)
    ds.alns = np.concatenate([ds.alns, np.array(extra, dtype=synth.ALN_DTYPE)])
    alns = ds.sorted_alns()
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    _, _, nwin = gpu_ctx.candidate(10, 100, 300, order)
    assert nwin >= nwin0 + 60
    rep = []
    for _ in range(3):
        gpu_ctx.fold(300)
        rep.append(gpu_ctx.last_timings()["fold_ms"])
    nov = gpu_ctx.last_fold_overflow()
    assert nov >= 50 and (gpu_ctx.fold_status() == 0).all() and gpu_ctx.last_fold_fallbacks() == 0
    assert min(rep) <= 1.10 * min(base) * nwin / nwin0, (base, rep)
    raw = gpu_ctx.get_fold()
    win = gpu_ctx.get_windows()
    W = win["windows"]
    for k in sorted(raw["overflow"])[:12]:
        ref = oracle.lfold(win["seq"][W[k]["seq_off"]:W[k]["seq_off"] + W[k]["seq_len"]].tobytes(), 300)
        assert raw["n_lines"][k] > 96 and raw["mfe"][k] == ref["mfe"] and _window_lines(raw, k) == ref["lines"]
    out = gpu_ctx.predict(1, 18, 23, False, True)
    assert (out["status"] == 0).all() and len(out["result"]) > 1000


def test_dense_region_at_large_precursor_len_matches_oracle(gpu_ctx, oracle):
    """PRECURSOR_LEN = 1000 (the reference accepts 60-3000, MP:167-184): one 1,040-nt region of 40 peaks with 80+ candidate matures -- more than the
    filter kernel's default table holds -- and structure lines of up to ~1,050 characters with many pieces.  The window is flagged by the first
    launch and run again at capacities sized for it (run_predict_launch); the answer is the oracle's check_loci, not an error."""
    L = 1000
    ds = synth.make_dataset([40000], 0, n_samples=1, seed=9, contig_names=["c1"])
    rng = np.random.RandomState(5)
    # plant a long imperfect hairpin under part of the region so that some structures carry a real duplex
    arm = synth._BASES[rng.randint(0, 4, size=60)]
    hp = np.concatenate([arm, synth._BASES[rng.randint(0, 4, size=12)], synth._revcomp(arm)])
    ds.contigs[0][1][5200:5200 + len(hp)] = hp
    recs = []
    for k in range(40):          # 40 peaks 26 nt apart (each >= 19 nt above the threshold): one region of 1,040 nt, two or three matures per peak
        p = 5000 + 26 * k
        recs += [(0, p, 60 + k, 21, 0, 0), (0, p + 1, 40, 21, 0, 0), (0, p + 2, 30, 20, 0, 0)]
    recs += [(0, 5203, 900, 22, 0, 0), (0, 5203 + 74, 35, 22, 0, 0)]
    alns = np.array(recs, dtype=synth.ALN_DTYPE)
    alns = alns[np.argsort(alns["pos"], kind="stable")]
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    order = np.zeros(1, np.int32)
    npk, nloci, nwin = gpu_ctx.candidate(10, 100, L, order)
    assert nloci == 1 and nwin == 1
    win = gpu_ctx.get_windows()
    W = win["windows"]
    assert W[0]["n_matures"] > 64
    gpu_ctx.fold(L)
    assert (gpu_ctx.fold_status() == 0).all()
    out = gpu_ctx.predict(1, 18, 23, False, True)
    assert out["status"][0] == 0
    b = W[0]
    ref = oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L)
    raw = gpu_ctx.get_fold()
    assert _window_lines(raw, 0) == ref["lines"]
    structs = oracle.structures_from_lines(ref["lines"], 55)
    mats = win["matures"][b["mature_off"]:b["mature_off"] + b["n_matures"]]
    r = oracle.check_loci(structs, mats, b, alns, (1, 18, 23, 0, 1, 55))
    assert out["n_passed"][0] == len(r)
    if r:
        m, ss, o = out["result"][0], out["ss"][0], r[0]
        assert [m["fold_s"], m["fold_e"], m["mat_s"], m["mat_e"], m["star_s"], m["star_e"], ss, m["strand"], bool(m["has_star"])] == \
               [o.fold_s, o.fold_e, o.mat_s, o.mat_e, o.star_s, o.star_e, o.ss.decode(), o.strand, bool(o.has_star)]
    # the -d records of the re-run window replace the truncated ones of the first pass: one record per (mature in range, structure) pair
    rec = gpu_ctx.predict_reasons(1, 18, 23, False, True)
    perw = [x for x in rec if x[1] < 0]
    assert len(perw) == 1 and perw[0][2] == len(structs)
    n_in = sum(1 for m_ in mats if 18 <= m_["end"] - m_["start"] <= 23)
    assert len([x for x in rec if x[1] >= 0]) == n_in * len(structs)


def test_difference_arrays_are_clean_between_passes(gpu_ctx, oracle):
    """The coverage pass does not clear the difference arrays any more: cov_unscatter_kernel takes back exactly the slots cov_scatter_kernel wrote
    (one shared slot computation).  A stale count would leak into the NEXT pass -- so: a dense alignment set, then a different, sparser one
    (and coverage segments in between) on the same context and genome, each depth file against the oracle's."""
    rng = np.random.RandomState(4)
    lens = [60000, 35000]
    genome = [("c%d" % k, synth._BASES[rng.randint(0, 4, size=l)]) for k, l in enumerate(lens)]
    order = np.array([0, 1], dtype=np.int32)

    def alns_of(n, seed, edge):
        r = np.random.RandomState(seed)
        a = np.zeros(n, dtype=synth.ALN_DTYPE)
        a["tid"] = r.randint(0, 2, size=n)
        L = np.array(lens)[a["tid"]]
        a["pos"] = (r.randint(0, 400, size=n) * 37 + r.randint(0, 30, size=n)) % (L - 30) + 1
        if edge:          # records that reach over the contig end: clamped identically by both kernels
            a["pos"][:50] = L[:50] - r.randint(0, 10, size=50)
        a["len"] = r.randint(18, 30, size=n)
        a["depth"] = r.randint(1, 30, size=n)
        a["strand"] = r.randint(0, 2, size=n)
        key = a["tid"].astype(np.int64) << 32 | a["pos"].astype(np.int64)
        return a[np.argsort(key, kind="stable")]
    gpu_ctx.load_genome(genome)
    for n, seed, edge in ((20000, 1, True), (300, 2, False), (5000, 3, True), (40, 4, False)):
        a = alns_of(n, seed, edge)
        gpu_ctx.load_alignments(a)
        if seed == 3:          # coverage segments ride along in this pass and must be cleared as well
            own = a[::25][:100]                       # gapped alignments: their own interval taken back out, a shorter block put in
            sub = own.copy(); sub["strand"] |= 2
            blk = own.copy(); blk["len"] = np.maximum(own["len"] // 2, 1)
            sg = np.concatenate([sub, blk])
            gpu_ctx.load_coverage_segments(sg)
        else:
            sg = a[:0]
        gpu_ctx.candidate(10, 100, 300, order)
        for _ in range(2):          # get_depth runs the pass again
            got = gpu_ctx.get_depth()
        neg = sg[(sg["strand"] & 2) != 0].copy(); pos_sg = sg[(sg["strand"] & 2) == 0]
        # oracle: plain records + added segments, minus the subtract segments (depth with a sign: emulate by brute force)
        cov = [np.zeros((2, l + 2), dtype=np.int64) for l in lens]
        for recs, sign in ((a, 1), (pos_sg, 1), (neg, -1)):
            for r in recs:
                w = min(int(r["depth"]), 10) * sign
                s, e = int(r["pos"]), min(int(r["pos"]) + int(r["len"]), lens[int(r["tid"])] + 1)
                cov[int(r["tid"])][int(r["strand"]) & 1, s:e] += w
        want = [(t, p, int(cov[t][0, p]), int(cov[t][1, p])) for t in range(2) for p in range(1, lens[t] + 1) if cov[t][0, p] + cov[t][1, p] > 10]
        assert [(int(d["tid"]), int(d["pos"]), int(d["dp"]), int(d["dm"])) for d in got] == want, seed


def test_fused_coverage_scan_equals_atomic_path_and_brute_force(gpu_ctx):
    """Two ways to the same depth: the scan that builds each tile's difference values from the sorted records in LDS (cov_scan_kernel<true>, picked for
    dense inputs) and the atomic scatter into the dense arrays (sparse inputs, coverage segments).  mirp_set_coverage_path forces either; both run on one
    context in turn (each leaves the dense arrays in the state the other expects), on inputs that sit on the scan's tile edges (8192 positions):
    contig boundaries next to a tile boundary, records that start in one tile and end in the next, records that reach over a contig end, an
    empty contig, a record longer than a tile (which sends the pass down the atomic path by itself).  Depth list and peaks against brute force."""
    T = 8192
    lens = [T - 1, 2 * T + 3, 5, 3 * T - 2, 40000]
    rng = np.random.RandomState(11)
    genome = [("c%d" % k, synth._BASES[rng.randint(0, 4, size=l)]) for k, l in enumerate(lens)]
    order = np.arange(len(lens), dtype=np.int32)
    gpu_ctx.load_genome(genome)

    def alns_of(n, seed, long_one):
        r = np.random.RandomState(seed)
        a = np.zeros(n, dtype=synth.ALN_DTYPE)
        a["tid"] = r.choice([0, 1, 3, 4], size=n)
        L = np.array(lens)[a["tid"]]
        # clusters around the tile edges of the guarded coordinate space and uniform background
        near = (r.randint(1, 6, size=n) * T - r.randint(-40, 40, size=n)) % np.maximum(L, 1) + 1
        a["pos"] = np.where(r.rand(n) < 0.6, near, r.randint(1, 1 << 30, size=n) % L + 1)
        a["pos"][:40] = L[:40] - r.randint(0, 12, size=40)          # reach over the contig end
        a["len"] = r.randint(18, 60, size=n)
        a["depth"] = r.randint(1, 40, size=n)
        a["strand"] = r.randint(0, 2, size=n)
        if long_one:
            a["len"][n // 2] = T + 100
        key = a["tid"].astype(np.int64) << 32 | a["pos"].astype(np.int64)
        return a[np.argsort(key, kind="stable")]

    def brute(a):
        cov = [np.zeros((2, l + 2), dtype=np.int64) for l in lens]
        for r_ in a:
            t = int(r_["tid"])
            s, e = int(r_["pos"]), min(int(r_["pos"]) + int(r_["len"]), lens[t] + 1)
            cov[t][int(r_["strand"]) & 1, s:e] += min(int(r_["depth"]), 10)
        return [(t, p, int(cov[t][0, p]), int(cov[t][1, p])) for t in range(len(lens)) for p in range(1, lens[t] + 1) if cov[t][0, p] + cov[t][1, p] > 10]

    for n, seed, long_one in ((30000, 1, False), (200, 2, False), (12000, 3, True), (30000, 4, False)):
        a = alns_of(n, seed, long_one)
        gpu_ctx.load_alignments(a)
        want = brute(a)
        got = {}
        for mode in (1, 0, 1):
            gpu_ctx.set_coverage_path(mode)
            gpu_ctx.candidate(10, 100, 300, order)
            assert gpu_ctx.last_coverage_fused() == (mode == 1 and not long_one)
            d = gpu_ctx.get_depth()
            pk = gpu_ctx.get_peaks()
            got[mode] = ([(int(x["tid"]), int(x["pos"]), int(x["dp"]), int(x["dm"])) for x in d], [tuple(int(v) for v in x) for x in pk])
            assert got[mode][0] == want, (seed, mode)
        assert got[1][1] == got[0][1], seed
    gpu_ctx.set_coverage_path(-1)


@pytest.mark.gpu
def test_exclusive_scan_single_block_and_lookback_paths(gpu_ctx):
    """The prefix sum under every compaction of the candidate stage, both kernels: one workgroup (<= 16,384 elements) and the single-pass look-back
    scan over 4,096-element blocks (beyond): sizes on the block edges, negative values (the tile aggregates of a difference array are signed),
    large partial sums, and 300 back-to-back calls on one stream (the descriptors are never cleared between calls: each call has its own epoch)."""
    rng = np.random.RandomState(3)
    for n in (0, 1, 5, 1024, 16383, 16384, 16385, 16384 + 4095, 5 * 4096, 5 * 4096 + 1, 300001, 4096 * 700 + 17, 20000003):
        for lo, hi in ((0, 2), (-1000, 1001), (2 ** 30, 2 ** 31 - 1)):
            if n > 4000000 and lo != -1000:
                continue
            v = rng.randint(lo, hi, size=n).astype(np.int32)
            want = np.concatenate([[0], np.cumsum(v.astype(np.int64))])
            got = gpu_ctx.excl_scan(v)
            assert np.array_equal(got, want), (n, lo)
    v = rng.randint(0, 7, size=70001).astype(np.int32)
    want = np.concatenate([[0], np.cumsum(v.astype(np.int64))])
    for k in range(300):
        assert np.array_equal(gpu_ctx.excl_scan(v[:70001 - 97 * (k % 5)]), want[:70001 - 97 * (k % 5) + 1]), k
