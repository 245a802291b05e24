"""GPU parity of the device-resident candidate -> fold -> predict pipeline (through the C-ABI) against
(i) the golden fixtures generated from the real reference stack and (ii) the CPU oracle on a larger seeded dataset."""
import numpy as np
import pytest

from mir_prefer_amd import records, synth
from tests import golden_util as gu
from tests.test_oracle_golden import _ref_lines, mirna_record, run_predict

pytestmark = pytest.mark.gpu


def _sorted_order(names):
    return np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)


def _gpu_record(m, ss, names):
    return [names[m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]), int(m["star_e"]), ss,
            records.STRAND[m["strand"]], bool(m["has_star"])]


@pytest.mark.parametrize("name", ["mini", "mini3", "mini185", "mini24", "mini400"])
def test_pipeline_matches_reference_fixture(name, gpu_ctx):
    c = gu.load_pipeline_case(name)
    gpu_ctx.set_fold_model(c["exp"].get("fold_model", "vienna-2.1.2"))
    try:
        _check_pipeline_fixture(c, gpu_ctx)
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")


def _check_pipeline_fixture(c, gpu_ctx):
    cfg, exp, names = c["exp"]["config"], c["exp"], c["contig_names"]
    gpu_ctx.load_genome(c["contigs"])
    gpu_ctx.load_alignments(c["alns"])
    npk, nloci, nwin = gpu_ctx.candidate(cfg["READS_DEPTH_CUTOFF"], cfg["MAX_GAP"], cfg["PRECURSOR_LEN"], _sorted_order(names))
    # a1-a2: depth file and peaks
    assert records.depth_text(gpu_ctx.get_depth(), names) == exp["depth_cut"]
    assert records.peaks_to_dict(gpu_ctx.get_peaks(), names) == gu.unjson(exp["dict_contigs"])
    # a3: loci + windows
    loci, psorted = gpu_ctx.get_loci()
    want = {k: v for k, v in gu.unjson(exp["dict_loci"]).items() if v}
    assert records.loci_to_dict(loci, psorted, names, cfg["PRECURSOR_LEN"]) == want
    assert records.exregion_gff_text(records.loci_to_dict(loci, psorted, names, cfg["PRECURSOR_LEN"])) == exp["exregion_gff"]   # MP:1357-1369
    # a4-a6: FASTA entries (headers incl. matures, sequences)
    w = gpu_ctx.get_windows()
    entries = [e for p in exp["pieces"] for e in p["fasta"]]
    assert len(w["windows"]) == len(entries) == nwin
    for win, (hdr, seq) in zip(w["windows"], entries):
        assert records.fasta_header(win, w["wpeaks"], w["matures"], names) == hdr
        assert w["seq"][win["seq_off"]:win["seq_off"] + win["seq_len"]].tobytes().decode() == seq
    # a5: the per-position read table (gen_loci_alignment_info, MP:1395-1468) against the alndump's dict_loci_info, window strand:
    # [length of the most abundant read, its depth, total depth] per start position; no other start position carries a read
    rt = gpu_ctx.get_window_readtable()
    dumps = [gu.unjson(d) for p in exp["pieces"] for d in p["alndump"]]
    assert len(dumps) == nwin
    for k, (win, d) in enumerate(zip(w["windows"], dumps)):
        info, strand = d[2][0], records.STRAND[win["strand"]]
        want_rows = {int(pos): [int(x) for x in v[strand][0]] for pos, v in info.items() if strand in v}
        got_rows = {int(win["ws"]) + x: [int(v) for v in rt[k, x]] for x in np.nonzero(rt[k, :, 1])[0]}
        assert got_rows == want_rows, k
    # a7: fold output of every window == the RNALfold text of the reference run
    gpu_ctx.fold(cfg["PRECURSOR_LEN"])
    raw = gpu_ctx.get_fold()
    assert (raw["status"] == 0).all()
    ref = [e for p in exp["pieces"] for e in _ref_lines(p["rnalfold_out"])]
    for k, want_lines in enumerate(ref):
        got = [(raw["ss"][k, j, :raw["lines"][k, j]["len"]].tobytes().decode(), int(raw["lines"][k, j]["energy"]), int(raw["lines"][k, j]["start"]))
               for j in range(raw["n_lines"][k]) if raw["lines"][k, j]["printed"]]
        assert got == want_lines, k
    # a8-a11: final result list
    out = gpu_ctx.predict(len(c["sample_names"]), cfg["MIN_MATURE_LEN"], cfg["MAX_MATURE_LEN"], cfg["ALLOW_3NT_OVERHANG"] == "Y",
                          cfg["ALLOW_NO_STAR_EXPRESSION"] == "Y")
    got = [_gpu_record(m, ss, names) for m, ss in zip(out["result"], out["ss"])]
    assert got == [e[:10] for e in gu.unjson(exp["result_raw"])]


def test_pipeline_matches_oracle_on_larger_dataset(gpu_ctx, oracle):
    ds = synth.make_dataset([400000, 250000, 350000], 500, n_samples=3, seed=21, contig_names=["c9", "c10", "c1"], edge_cases=True)
    names, alns = ds.contig_names, ds.sorted_alns()
    cut, gap, L = 10, 100, 300
    order = _sorted_order(names)
    depth, peaks = oracle.coverage_peaks(alns, ds.contig_lens, cut)
    win = oracle.make_windows(peaks, alns, ds.contigs, order, gap, L, cut * 0.5)
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    npk, nloci, nwin = gpu_ctx.candidate(cut, gap, L, order)
    assert np.array_equal(gpu_ctx.get_depth(), depth)
    assert np.array_equal(gpu_ctx.get_peaks(), peaks)
    w = gpu_ctx.get_windows()
    assert nwin == len(win["windows"]) and nwin > 400
    for a, b in zip(w["windows"], win["windows"]):
        for f in ("tid", "ws", "we", "strand", "loc_s", "loc_e", "tag", "n_peaks", "n_matures", "seq_len"):
            assert a[f] == b[f]
        assert np.array_equal(w["wpeaks"][a["peak_off"]:a["peak_off"] + a["n_peaks"]], win["wpeaks"][b["peak_off"]:b["peak_off"] + b["n_peaks"]])
        assert np.array_equal(w["matures"][a["mature_off"]:a["mature_off"] + a["n_matures"]], win["matures"][b["mature_off"]:b["mature_off"] + b["n_matures"]])
        assert np.array_equal(w["seq"][a["seq_off"]:a["seq_off"] + a["seq_len"]], win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]])
    gpu_ctx.fold(L)
    out = gpu_ctx.predict(3, 18, 23, False, True)
    # oracle end-to-end
    structs = []
    for b in win["windows"]:
        r = oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L)
        structs.append(oracle.structures_from_lines(r["lines"], 55))
    case = {"cfg": {"MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 23, "ALLOW_3NT_OVERHANG": "N", "ALLOW_NO_STAR_EXPRESSION": "Y"}, "win": win,
            "sample_names": ds.sample_names, "alns": alns}
    _, result = run_predict(case, oracle, structs)
    want = [mirna_record(m, names) for _, m in result]
    got = [_gpu_record(m, ss, names) for m, ss in zip(out["result"], out["ss"])]
    assert got == want and len(got) > 50


@pytest.mark.parametrize("n_samples,allow3,no_star", [(40, False, True), (70, True, True), (255, False, True)])
def test_many_samples_match_oracle(gpu_ctx, oracle, n_samples, allow3, no_star):
    """More ALIGNMENT_FILEs than 16 / 32 / 64 (the reference has no limit, MP:3300-3308; a record's sample index is 8 bits): the per-sample rules of the
    expression test -- mature reads in EVERY sample, start positions per sample (MP:2066-2068, 2316-2330) -- through the filter kernel's sample sets
    (eight mask words), result list and the -d records' per-sample mature depths against the oracle."""
    ds = synth.make_dataset([160000, 90000], 170, n_samples=n_samples, seed=100 + n_samples, contig_names=["k2", "k1"], edge_cases=True)
    names, alns = ds.contig_names, ds.sorted_alns()
    assert int(alns["sample"].max()) == n_samples - 1
    cut, gap, L = 10, 100, 300
    order = _sorted_order(names)
    _, peaks = oracle.coverage_peaks(alns, ds.contig_lens, cut)
    win = oracle.make_windows(peaks, alns, ds.contigs, order, gap, L, cut * 0.5)
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    _, _, nwin = gpu_ctx.candidate(cut, gap, L, order)
    assert nwin == len(win["windows"]) > 100
    gpu_ctx.fold(L)
    out = gpu_ctx.predict(n_samples, 18, 23, allow3, no_star)
    structs = [oracle.structures_from_lines(oracle.lfold(win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes(), L)["lines"], 55) for b in win["windows"]]
    case = {"cfg": {"MIN_MATURE_LEN": 18, "MAX_MATURE_LEN": 23, "ALLOW_3NT_OVERHANG": "Y" if allow3 else "N", "ALLOW_NO_STAR_EXPRESSION": "Y" if no_star else "N"},
            "win": win, "sample_names": ds.sample_names, "alns": alns}
    _, result = run_predict(case, oracle, structs)
    want = [mirna_record(m, names) for _, m in result]
    got = [_gpu_record(m, ss, names) for m, ss in zip(out["result"], out["ss"])]
    assert got == want and len(got) > 10
    # -d records: the per-sample mature depths of every evaluated (mature, structure) pair sum to its total, and name only real samples
    rec = gpu_ctx.predict_reasons(n_samples, 18, 23, allow3, no_star)
    pairs = rec[(rec[:, 1] >= 0) & (rec[:, 6] == 0)]
    assert rec.shape[1] == 21 + n_samples and len(pairs) > 20
    assert (pairs[:, 21:].sum(axis=1) == pairs[:, 14]).all()
    # one pair recomputed by the oracle, sample by sample
    w = win["windows"]
    for r in pairs[:: max(1, len(pairs) // 25)]:
        b = w[int(r[0])]
        m = win["matures"][b["mature_off"] + int(r[1])]
        e = oracle.expression(alns, n_samples, int(b["tid"]), int(b["ws"]), int(b["we"]), int(r[8]), int(r[9]), int(m["start"]), int(m["end"]), int(r[10]), int(r[11]),
                              int(m["strand"]), allow3)
        assert [int(x) for x in r[21:]] == [int(e.reads_mature[s]) for s in range(n_samples)]
        assert int(r[12]) == e.total_this_strand and int(r[14]) == e.total_mature


def test_window_views_and_streamed_reports_equal_the_whole_run(gpu_ctx, oracle, tmp_path):
    """mirp_select_windows: folding and filtering the window list view by view gives the whole run's result list; mirp_fold_predict_report_stream
    (chunks cut where the windows decide the list order, files of a chunk written while the next folds) writes the same files as the one-call writer
    on the whole run's result, for 1, 3 and 9 chunks."""
    import filecmp
    import os
    from mir_prefer_amd import capi, pipeline
    ds = synth.make_dataset([400000, 250000], 420, n_samples=2, seed=4242, contig_names=["Chr2", "Chr10"], edge_cases=True)      # the `mid` dataset
    names, alns = ds.contig_names, ds.sorted_alns()
    order = _sorted_order(names)
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(alns)
    _, _, nwin = gpu_ctx.candidate(10, 100, 300, order)
    gpu_ctx.fold(300)
    whole = gpu_ctx.predict_raw(2, 18, 23, False, True)
    assert len(whole["result"]) > 100
    w = gpu_ctx.get_windows()["windows"]
    # views at L/R-pair boundaries (tag 0 = single window, 1 = L, 2 = R)
    cuts = [0]
    for target in (nwin // 3, 2 * nwin // 3):
        k = target
        while w[k]["tag"] == 2:
            k += 1
        cuts.append(k)
    cuts.append(nwin)
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        gpu_ctx.select_windows(a, b - a)
        gpu_ctx.fold(300)
        out = gpu_ctx.predict_raw(2, 18, 23, False, True)
        r = out["result"].copy()
        r["window"] += a
        parts.append((r, out["text"]))
    gpu_ctx.select_windows(0, -1)
    got = np.concatenate([p[0] for p in parts])
    assert got.tobytes() == whole["result"].tobytes()
    assert np.concatenate([p[1] for p in parts]).tobytes() == whole["text"].tobytes()
    mark = "\x00SEQ\x00"
    form = [p for taxon in ("Viridiplantae", "ALL") for p in pipeline._mirbase_form_text(mark, taxon).split(mark)]
    ref_dir = tmp_path / "whole"
    capi.write_result_reports(whole["result"], whole["text"], names, [sq for _, sq in ds.contigs], alns, ds.sample_names, form, str(ref_dir), "x")
    for chunks in (1, 3, 9):
        d = tmp_path / ("stream%d" % chunks)
        n, used, dev = gpu_ctx.fold_predict_report_stream(300, (2, 18, 23, 0, 1, 55), chunks, names, [sq for _, sq in ds.contigs], alns, ds.sample_names, form, str(d), "x")
        assert n == len(whole["result"]) and 1 <= used <= chunks and (chunks == 1 or used > 1) and dev["fold_s"] > 0
        cmp = filecmp.dircmp(str(ref_dir), str(d))
        assert not cmp.left_only and not cmp.right_only
        for sub, files in ((".", cmp.common_files), ("readmapping", os.listdir(ref_dir / "readmapping"))):
            match, mismatch, errors = filecmp.cmpfiles(str(ref_dir / sub), str(d / sub), files, shallow=False)
            assert not mismatch and not errors and len(match) == len(files)


def test_window_view_argument_checks(gpu_ctx):
    """mirp_select_windows refuses ranges outside the window list, mirp_limit_windows refuses to run under a view, a view resets the fold state (the filter
    then asks for a fold first), count < 0 restores the list."""
    from mir_prefer_amd import capi
    ds = synth.make_dataset([60000], 30, n_samples=1, seed=12, contig_names=["c1"])
    gpu_ctx.load_genome(ds.contigs)
    gpu_ctx.load_alignments(ds.sorted_alns())
    _, _, nwin = gpu_ctx.candidate(10, 100, 300, np.zeros(1, dtype=np.int32))
    assert nwin > 10
    with pytest.raises(capi.MirpError):
        gpu_ctx.select_windows(0, nwin + 1)
    with pytest.raises(capi.MirpError):
        gpu_ctx.select_windows(-1, 2)
    gpu_ctx.fold(300)
    gpu_ctx.select_windows(2, 5)
    with pytest.raises(capi.MirpError):          # the fold of the whole list is not the view's fold
        gpu_ctx.predict_raw(1, 18, 23, False, True)
    with pytest.raises(capi.MirpError):
        gpu_ctx.limit_windows(3)
    gpu_ctx.fold(300)
    assert len(gpu_ctx.predict_raw(1, 18, 23, False, True)["status"]) == 5
    gpu_ctx.select_windows(0, -1)
    gpu_ctx.fold(300)
    assert len(gpu_ctx.predict_raw(1, 18, 23, False, True)["status"]) == nwin
