"""Pins the CPU oracle (oracle/*.c) against golden vectors produced by the REAL reference stack
(miR_PREFeR.py under the py3 shim + bundled samtools 0.1.18 + bundled RNALfold 2.1.2); CPU only."""
import numpy as np
import pytest

from mir_prefer_amd import records
from tests import golden_util as gu

CASES = ["mini", "mini3", "mini185", "mini24", "mini400"]   # mini185: the "mini" dataset run with the bundled RNALfold 1.8.5 on PATH


def test_lfold_matches_rnalfold212(oracle):
    gold = gu.load_json("fold_rnalfold212.json.gz")
    n = 0
    for case in gold["cases"]:
        for seq, exp in zip(case["seqs"], case["expected"]):
            got = oracle.lfold(seq, case["span"])
            assert got["mfe"] == exp["mfe"], seq
            assert [list(l) for l in got["lines"]] == exp["lines"], seq
            n += 1
    assert n >= 300


def test_lfold185_matches_rnalfold185(oracle):
    """vienna-1.8.5 flavour (Turner-1999, dangles 1, multi-component strings) against the Linux binary the reference bundles."""
    gold = gu.load_json("fold_rnalfold185.json.gz")
    n = 0
    for case in gold["cases"]:
        for seq, exp in zip(case["seqs"], case["expected"]):
            got = oracle.lfold(seq, case["span"], model="vienna-1.8.5")
            assert got["mfe"] == exp["mfe"], seq
            assert [list(l) for l in got["lines"]] == exp["lines"], seq
            n += 1
    assert n >= 330


@pytest.fixture(scope="module", params=CASES)
def case(request, oracle):
    c = gu.load_pipeline_case(request.param)
    cfg = c["exp"]["config"]
    depth, peaks = oracle.coverage_peaks(c["alns"], c["contig_lens"], cfg["READS_DEPTH_CUTOFF"])
    order = np.argsort(np.array(c["contig_names"], dtype=object), kind="stable").astype(np.int32)  # sorted(dict_contigs)
    win = oracle.make_windows(peaks, c["alns"], c["contigs"], order, cfg["MAX_GAP"], cfg["PRECURSOR_LEN"], cfg["READS_DEPTH_CUTOFF"] * 0.5)
    c = dict(c)
    c.update(depth=depth, peaks=peaks, win=win, cfg=cfg)
    return c


def test_depth_file(case):
    assert records.depth_text(case["depth"], case["contig_names"]) == case["exp"]["depth_cut"]


def test_peaks(case):
    assert records.peaks_to_dict(case["peaks"], case["contig_names"]) == gu.unjson(case["exp"]["dict_contigs"])


def test_loci_and_windows(case):
    got = records.loci_to_dict(case["win"]["loci"], case["peaks"], case["contig_names"], case["cfg"]["PRECURSOR_LEN"])
    exp = gu.unjson(case["exp"]["dict_loci"])
    exp = {k: v for k, v in exp.items() if v}
    assert got == exp
    # the candidate stage's debug artefact <prefix>_ExRegionA.gff3 (MP:1357-1369) is a rendering of the same dict
    assert records.exregion_gff_text(got) == case["exp"]["exregion_gff"]


def _fasta_entries(case):
    return [e for p in case["exp"]["pieces"] for e in p["fasta"]]


def test_fasta_entries(case):
    w = case["win"]
    exp = _fasta_entries(case)
    assert len(w["windows"]) == len(exp)
    for k, (win, (hdr, seq)) in enumerate(zip(w["windows"], exp)):
        assert records.fasta_header(win, w["wpeaks"], w["matures"], case["contig_names"]) == hdr, k
        got = w["seq"][win["seq_off"]:win["seq_off"] + win["seq_len"]].tobytes().decode()
        assert got == seq, k


def test_matures_match_alndump(case):
    w = case["win"]
    dumps = [gu.unjson(d) for p in case["exp"]["pieces"] for d in p["alndump"]]
    assert len(dumps) == len(w["windows"])
    for win, d in zip(w["windows"], dumps):
        key, tag, _info, matures = d
        assert key == [case["contig_names"][win["tid"]], (int(win["ws"]), int(win["we"])), records.STRAND[win["strand"]]]
        assert tag == records.TAG[win["tag"]]
        got = [records.mature_tuple(m) for m in w["matures"][win["mature_off"]:win["mature_off"] + win["n_matures"]]]
        assert got == [tuple(m) for m in matures]


def _ref_lines(text):
    """RNALfold output text -> per entry list of (ss, energy_dcal, start)"""
    entries = []
    for line in text.splitlines():
        if line.startswith(">"):
            entries.append([])
            continue
        sp = line.split()
        if len(sp) >= 3:
            e = line[line.index(" (") + 2:line.rindex(")")]
            entries[-1].append((sp[0], int(round(float(e) * 100)), int(sp[-1])))
    return entries


def test_fold_of_pipeline_windows(case, oracle):
    w = case["win"]
    ref = [e for p in case["exp"]["pieces"] for e in _ref_lines(p["rnalfold_out"])]
    assert len(ref) == len(w["windows"])
    for win, exp in zip(w["windows"], ref):
        seq = w["seq"][win["seq_off"]:win["seq_off"] + win["seq_len"]].tobytes()
        got = oracle.lfold(seq, case["cfg"]["PRECURSOR_LEN"], model=case["exp"].get("fold_model", "vienna-2.1.2"))
        assert got["lines"] == exp


def test_structures(case, oracle):
    ref_lines = [e for p in case["exp"]["pieces"] for e in _ref_lines(p["rnalfold_out"])]
    ref_structs = [gu.unjson(s) for p in case["exp"]["pieces"] for s in p["structures"]]
    assert len(ref_lines) == len(ref_structs)
    nsub = 0
    for lines, (which, peak, structs) in zip(ref_lines, ref_structs):
        got = oracle.structures_from_lines(lines, 55)
        assert got == [tuple(s) for s in structs]
        nsub += len(got)
    assert nsub > 0


def test_maturestar_and_expression(case, oracle):
    from tests.oracle_binding import MS_CODES
    w = case["win"]["windows"]
    allow3 = case["cfg"]["ALLOW_3NT_OVERHANG"] == "Y"
    ns = len(case["sample_names"])
    n_ok = 0
    base = 0
    for p in case["exp"]["pieces"]:
        for (wi, mature, foldstart, ss, r, ex) in p["maturestar_expr"]:
            win = w[base + wi]
            m0, m1, strand, mdepth = mature
            if strand == 0:
                continue  # the (0,0,0,0) fallback mature; the reference passes strand=0 -> '-' branch, length 0 anyway
            sidx = records.STRAND.index(strand)
            got = oracle.maturestar(ss, m0, m1, foldstart, int(win["ws"]), int(win["we"]), sidx)
            r = gu.unjson(r)
            if isinstance(r, str):
                assert MS_CODES[got.code] == r, (ss, mature)
                continue
            assert got.code == 0, (ss, mature, MS_CODES[got.code], r)
            assert (got.star_s, got.star_e, got.fold_s, got.fold_e) == tuple(r[:4])
            assert bool(got.prime5) == r[5] and got.total_dots == r[7] and got.total_bps == r[8]
            assert ss[got.star_l0:got.star_l1] == r[4] and ss[got.mat_l0:got.mat_l1] == r[6]
            ex = gu.unjson(ex)
            e = oracle.expression(case["alns"], ns, int(win["tid"]), int(win["ws"]), int(win["we"]), r[2], r[3], m0, m1, r[0], r[1], sidx, allow3)
            assert not e.exception
            assert e.total_this_strand == ex["total_depth_just_this_strand"] and e.total_anti == ex["total_depth_anti"]
            assert e.total_mature == ex["total_depth_mature"] and e.total_isoform == ex["total_depth_isoform"]
            assert e.total_star == ex["total_depth_star"]
            assert e.mature_star_distance == ex["mature_star_distance"]
            assert e.ratio_total == ex["mature_star_ratio_total"] and e.ratio_both == ex["mature_star_ratio_total_both_strand"]
            assert e.ratio_iso == ex["mature_iso_star_ratio_total"]
            assert list(e.total_imperfect) == ex["total_depth_imperfect_star"]
            for si, sname in enumerate(case["sample_names"]):
                ps = ex["per_sample"][sname]
                assert e.reads_pre[si] == ps["reads_pre"] and e.reads_mature[si] == ps["reads_mature"] and e.reads_star[si] == ps["reads_star"]
                assert e.reads_antisense[si] == ps["reads_antisense"] and e.reads_isoform[si] == ps["reads_mature_isoform"]
                assert e.reads_inside[si] == ps["reads_mature_inside"] and e.bases_with_reads_start[si] == ps["bases_with_reads_start"]
                assert e.ratio_start[si] == ps["ratio_bases_with_reads_start"]
                if allow3:
                    assert list(e.imperfect[si]) == ps["reads_star_imperfect"]
            if "max_imperfect_star" in ex:
                assert e.has_imperfect_key and e.max_imperfect == ex["max_imperfect_star"] and e.imperfect_start == ex["imperfect_star_start"]
                if ex["max_imperfect_star"]:
                    assert e.imperfect_end == ex["imperfect_star_end"] and e.imperfect_which == ex["imperfect_star_which"]
            else:
                assert not e.has_imperfect_key
            n_ok += 1
        base += len(p["fasta"])
    assert n_ok > 10


def run_predict(case, oracle, structs_per_window):
    """filter_next_loci pairing (miR_PREFeR.py:2350-2432) over the oracle's check_loci; returns (decisions, result_raw)."""
    cfg = case["cfg"]
    w = case["win"]
    params = (len(case["sample_names"]), cfg["MIN_MATURE_LEN"], cfg["MAX_MATURE_LEN"], 1 if cfg["ALLOW_3NT_OVERHANG"] == "Y" else 0,
              1 if cfg["ALLOW_NO_STAR_EXPRESSION"] == "Y" else 0, 55)

    def check(k):
        win = w["windows"][k]
        mats = w["matures"][win["mature_off"]:win["mature_off"] + win["n_matures"]]
        return oracle.check_loci(structs_per_window[k], mats, win, case["alns"], params)

    decisions, result = [], []
    k = 0
    n = len(w["windows"])
    while k < n:
        if w["windows"][k]["tag"] == 0:
            r = check(k); decisions.append(r); k += 1
            if r:
                result.append((k - 1, r[0]))
        else:
            r = check(k); decisions.append(r)
            if r:
                result.append((k, r[0]))
            else:
                r2 = check(k + 1); decisions.append(r2)
                if r2:
                    result.append((k + 1, r2[0]))
            k += 2
    return decisions, result


def mirna_record(m, contig_names):
    return [contig_names[m.tid], m.fold_s, m.fold_e, m.mat_s, m.mat_e, m.star_s, m.star_e, m.ss.decode(), records.STRAND[m.strand], bool(m.has_star)]


def test_decisions_and_result(case, oracle):
    ref_lines = [e for p in case["exp"]["pieces"] for e in _ref_lines(p["rnalfold_out"])]
    structs = [oracle.structures_from_lines(l, 55) for l in ref_lines]
    # decisions are per piece in the fixture; piece boundaries never split an L/R pair
    decisions, result = run_predict(case, oracle, structs)
    exp_dec = [d for p in case["exp"]["pieces"] for d in p["decisions"]]
    assert len(decisions) == len(exp_dec)
    for got, exp in zip(decisions, exp_dec):
        assert bool(got) == exp["pass"]
        if got:
            em = gu.unjson(exp["mirnas"])
            assert len(got) == len(em)
            for g, e in zip(got, em):
                assert mirna_record(g, case["contig_names"]) == e[:10]
                assert g.total_depth_mature == e[10]["total_depth_mature"] and g.total_depth_star == e[10]["total_depth_star"]
    exp_res = gu.unjson(case["exp"]["result_raw"])
    assert [mirna_record(m, case["contig_names"]) for _, m in result] == [e[:10] for e in exp_res]
    assert len(exp_res) > 5


def _coverage_records(alns, segs):
    """Records whose [pos, pos + len) is the covered interval: the ungapped alignments plus the M / = / X blocks of the gapped ones
    (a negative segment names the gapped alignment it takes out)."""
    neg = segs[(segs["strand"] & 2) != 0]
    pos = segs[(segs["strand"] & 2) == 0]
    drop = {}
    for s in neg:
        k = (int(s["tid"]), int(s["pos"]), int(s["depth"]), int(s["len"]), int(s["strand"]) & 1, int(s["sample"]))
        drop[k] = drop.get(k, 0) + 1
    keep = np.ones(len(alns), dtype=bool)
    for i, a in enumerate(alns):
        k = (int(a["tid"]), int(a["pos"]), int(a["depth"]), int(a["len"]), int(a["strand"]), int(a["sample"]))
        if drop.get(k, 0) > 0:
            drop[k] -= 1
            keep[i] = False
    both = np.concatenate([alns[keep], pos])
    key = both["tid"].astype(np.int64) << 32 | both["pos"].astype(np.int64)
    return both[np.argsort(key, kind="stable")]


def test_gapped_alignments_depth_matches_samtools(oracle, tmp_path):
    """Gapped CIGARs (I / D / N / S / H / = / X): thresholded depth lines against the bundled samtools 0.1.18 run on the expanded, strand-split
    BAMs exactly as the reference does (tests/golden/tools/gen_gapped_golden.py); the host ingest turns the M / = / X blocks into segments."""
    from mir_prefer_amd import ingest
    g = gu.load_json("gapped.json.gz")
    sam = tmp_path / "S1.sam"
    sam.write_text(g["sam"])
    for native in (True, False):
        names, lens, samples, alns, segs = ingest.read_sams([str(sam)], native=native, with_segments=True)
        assert names == [c[0] for c in g["contigs"]] and len(segs) > 50
        depth, _ = oracle.coverage_peaks(_coverage_records(alns, segs), lens, g["cutoff"])
        assert records.depth_text(depth, names) == g["depth_cut"]


MS_CODE = {"FAIL_STRUCTURE_MATCHED_BASES": 1, "FAIL_STRUCTURE_MATURE_NOT_IN_FOLD_REGION": 2, "FAIL_STRUCTURE_MATURE_NOT_IN_ONE_ARM": 3,
           "FAIL_STRUCTURE_MATURE_MATCH_SMALL_THAN_14": 4, "FAIL_STRUCTURE_MATURE_STAR_OVERLAP": 5, "FAIL_STRUCTURE_STAR_OUT_OF_FOLD_REGION": 6,
           "FAIL_STRUCTURE_STAR_NOT_IN_ONE_ARM": 7, "FAIL_STRUCTURE_TOO_MANY_BULGE_OR_LOOP": 8, "FAIL_STRUCTURE_MAX_BULGE_LARGE_THAN_2": 9,
           "FAIL_STRUCTURE_TOTAL_LOOP_SIZE_LARGER_THAN_5": 10, "FAIL_STRUCTURE_NUM_BULGE_MORE_THAN_2": 11}


def test_structure_rules_match_direct_reference_calls(oracle):
    """the oracle's a8 (oracle_structures) and a9 (oracle_maturestar) against DIRECT calls of the reference's
    get_structures_next_extendregion / is_stem_loop / filter_ss / has_one_good_bifurcation and get_maturestar_info / stat_duplex /
    pass_stat_duplex on seeded dot-brackets (tests/golden/tools/gen_struct_golden.py): every failure code of MP:1848-1999 at least 20 times,
    FAIL_STRUCTURE_TOO_MANY_BULGE_OR_LOOP and FAIL_STRUCTURE_MATCHED_BASES included, which no pipeline fixture reaches."""
    g = gu.load_json("struct_rules.json.gz")
    assert len(g["structures"]) + len(g["maturestar"]) + len(g["duplex"]) >= 10000
    counts = g["maturestar_code_counts"]
    assert all(counts.get(k, 0) >= 20 for k in MS_CODE), counts
    # a8: the structure list of one RNALfold line
    n_multi = 0
    for c in g["structures"]:
        got = oracle.structures_from_lines([(c["ss"], int(round(c["energy"] * 100)), c["start"])], 55)
        want = [(e, s, x, t) for e, s, x, t in c["extend"]]
        assert [(s, x, t) for _, s, x, t in got] == [(s, x, t) for _, s, x, t in want], c["ss"]
        for (ge, _, _, _), (we, _, _, _) in zip(got, want):
            assert ge == we          # float(E) / len(ss): the same double
        n_multi += len(want) > 1
        if len(c["ss"]) >= 55:
            # a line that is a stem-loop is exactly one type-0 entry holding the whole line
            assert c["is_stem_loop"] == (len(want) == 1 and want[0][3] == 0 and want[0][2] == c["ss"]) or not c["is_stem_loop"]
    assert n_multi > 100
    # a9: get_maturestar_info
    seen = {}
    for c in g["maturestar"]:
        r = c["result"]
        o = oracle.maturestar(c["ss"], c["mature"][0], c["mature"][1], c["foldstart"], c["regionstart"], c["regionend"], 1 if c["strand"] == "-" else 0)
        if "raises" in r:
            assert o.code == 12, c
            continue
        if isinstance(r["ok"], str):
            assert o.code == MS_CODE[r["ok"]], (c, o.code)
            seen[r["ok"]] = seen.get(r["ok"], 0) + 1
        else:
            ss_, se_, fs_, fe_, star_ss, prime5, mature_ss, dots, bps = r["ok"]
            assert o.code == 0, (c, o.code)
            assert (o.star_s, o.star_e, o.fold_s, o.fold_e, bool(o.prime5), o.total_dots, o.total_bps) == (ss_, se_, fs_, fe_, prime5, dots, bps), c
            assert c["ss"][o.star_l0:o.star_l1] == star_ss and c["ss"][o.mat_l0:o.mat_l1] == mature_ss
    assert all(seen.get(k, 0) >= 20 for k in MS_CODE)
    # stat_duplex + pass_stat_duplex on duplex halves alone
    fails = 0
    for c in g["duplex"]:
        code = oracle.duplex_code(c["mature"], c["star"])
        if "raises" in c["stat"]:
            assert code == 12
            continue
        want = c["pass"]["ok"][0]
        assert code == (MS_CODE[want] if want else 0), c
        fails += want is not None
    assert fails > 200


# ---- the wider pin (tests/golden/tools/gen_headline_golden.py): real RNALfold 2.1.2 / 1.8.5 on the benchmark's own window distribution and on the
# stress families, and the whole reference pipeline on a mid-size dataset under both folders
def headline_fold_groups():
    """-> {group: [sequences]} regenerated from the seeds, checked against the fixture's input digests."""
    import base64
    import random
    from tests import seqgen
    from tests.golden.tools_digest import seq_digest
    fix = gu.load_json("headline_folds.json.gz")
    import bench
    from tests import oracle_binding
    o = oracle_binding.load()
    specs, ns, bg, _, _ = bench.workload_specs("config1", 1)
    contigs, alns, _ = bench.build_shard(specs, {0}, ns, bg)
    lens = np.array([len(s) for _, s in contigs], dtype=np.int64)
    _, peaks = o.coverage_peaks(alns, lens, bench.CUT)
    win = o.make_windows(peaks, alns, contigs, np.arange(len(contigs), dtype=np.int32), bench.GAP, bench.L, bench.CUT * 0.5)
    W = win["windows"]
    assert len(W) == fix["headline_windows_total"]
    r = random.Random(fix["stress_seed"])
    groups = {"headline": [win["seq"][b["seq_off"]:b["seq_off"] + b["seq_len"]].tobytes().decode() for b in W[::fix["headline_step"]]],
              "stress": [seqgen.stress_family(r, i % 5) for i in range(fix["stress_count"])] + seqgen.microsatellites()}
    for name, seqs in groups.items():
        g = fix["groups"][name]
        assert len(seqs) == g["n"]
        assert b"".join(seq_digest(s) for s in seqs) == base64.b64decode(g["seq_digests"]), name          # the same windows the real binaries folded
    return fix, groups


def check_headline_folds(fix, groups, fold_many):
    """fold_many(seqs, span, model) -> [(lines, mfe)]; every (window, model, span) digest of the real binaries' output must be reproduced."""
    import base64
    from tests.golden.tools_digest import fold_digest
    n = 0
    for name, seqs in groups.items():
        for key, exp in fix["groups"][name]["folds"].items():
            model, span = key.split("/")
            got = fold_many(seqs, int(span), model)
            want = base64.b64decode(exp["digests"])
            for k, (lines, mfe) in enumerate(got):
                assert mfe == exp["mfe"][k], (name, key, k, seqs[k])
                assert fold_digest(lines, mfe) == want[6 * k:6 * k + 6], (name, key, k, seqs[k])
                n += 1
    return n


def test_headline_and_stress_windows_match_real_rnalfold(oracle):
    """2,188 windows of the headline workload + 725 stress windows, both models, spans 300 and 150: the oracle against digests of the REAL binaries' output."""
    from tests.test_whole_workload_gpu import oracle_fold_all
    fix, groups = headline_fold_groups()
    n = check_headline_folds(fix, groups, lambda seqs, span, model: oracle_fold_all(seqs, span, model))
    assert n >= 4 * (2000 + 500)


def mid_case(oracle):
    from mir_prefer_amd import synth
    from tests.golden.tools_digest import array_digest
    exp = gu.load_json("mid/expected.json.gz")
    d = exp["dataset"]
    ds = synth.make_dataset(d["lens"], d["loci"], n_samples=d["samples"], seed=d["seed"], contig_names=d["names"], edge_cases=True)
    alns = ds.sorted_alns()
    assert [array_digest(s) for _, s in ds.contigs] == d["genome_sha256"] and array_digest(alns) == d["alns_sha256"]
    return exp, ds, alns


@pytest.mark.parametrize("model", ["vienna-2.1.2", "vienna-1.8.5"])
def test_mid_dataset_whole_reference_pipeline(oracle, model):
    """The whole py3-shimmed reference (bundled samtools + RNALfold 2.1.2 / 1.8.5) on a dataset no other fixture holds: FASTA headers of all windows,
    the pass / fail decision of every region and the raw result list against the oracle chain."""
    import hashlib
    from tests.test_whole_workload_gpu import oracle_fold_all
    exp, ds, alns = mid_case(oracle)
    run = exp["runs"][model]
    cfg = run["config"]
    names = ds.contig_names
    _, peaks = oracle.coverage_peaks(alns, ds.contig_lens, cfg["READS_DEPTH_CUTOFF"])
    order = np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)
    win = oracle.make_windows(peaks, alns, ds.contigs, order, cfg["MAX_GAP"], cfg["PRECURSOR_LEN"], cfg["READS_DEPTH_CUTOFF"] * 0.5)
    W = win["windows"]
    assert len(W) == run["n_windows"] >= 600
    heads = [records.fasta_header(w, win["wpeaks"], win["matures"], names) for w in W]
    assert hashlib.sha256("\n".join(heads).encode()).hexdigest() == run["fasta_headers_sha256"]
    seqs = [win["seq"][w["seq_off"]:w["seq_off"] + w["seq_len"]].tobytes() for w in W]
    structs = [oracle.structures_from_lines(l, 55) for l, _ in oracle_fold_all(seqs, cfg["PRECURSOR_LEN"], model)]
    case = {"cfg": cfg, "win": win, "sample_names": ds.sample_names, "alns": alns}
    decisions, result = run_predict(case, oracle, structs)
    assert [len(d) for d in decisions] == run["decisions"]
    exp_res = gu.unjson(run["result_raw"])
    assert len(exp_res) >= 100
    assert [mirna_record(m, names) for _, m in result] == [e[:10] for e in exp_res]
    assert [(m.total_depth_mature, m.total_depth_star) for _, m in result] == [(e[10]["total_depth_mature"], e[10]["total_depth_star"]) for e in exp_res]


def long_fold_fixture():
    import base64
    from tests import seqgen
    from tests.golden.tools_digest import seq_digest
    fix = gu.load_json("long_folds.json.gz")
    seqs = seqgen.long_windows()
    assert len(seqs) == fix["n"] and b"".join(seq_digest(s) for s in seqs) == base64.b64decode(fix["seq_digests"])
    return fix, seqs


def check_long_folds(fix, seqs, fold_many):
    import base64
    from tests.golden.tools_digest import fold_digest
    n = 0
    for key, exp in fix["folds"].items():
        model, span = key.split("/")
        want = base64.b64decode(exp["digests"])
        for k, (lines, mfe) in enumerate(fold_many(seqs, int(span), model)):
            assert mfe == exp["mfe"][k] and fold_digest(lines, mfe) == want[6 * k:6 * k + 6], (key, k, seqs[k])
            n += 1
    return n


def test_long_windows_match_real_rnalfold(oracle):
    """PRECURSOR_LEN beyond 300 (the reference accepts 60 .. 3000, MP:167-184): 200 windows of 360 .. 480 nt at spans 400 and 330, both models, the oracle
    against digests of the real binaries' output."""
    from tests.test_whole_workload_gpu import oracle_fold_all
    fix, seqs = long_fold_fixture()
    assert check_long_folds(fix, seqs, lambda s, span, model: oracle_fold_all(s, span, model)) == 800


def xl_fold_fixture():
    import base64
    from tests import seqgen
    from tests.golden.tools_digest import seq_digest
    fix = gu.load_json("xl_folds.json.gz")
    seqs = seqgen.xl_windows()
    assert len(seqs) == fix["n"] and b"".join(seq_digest(s) for s in seqs) == base64.b64decode(fix["seq_digests"])
    return fix, seqs


def test_largest_precursor_length_matches_real_rnalfold(oracle):
    """PRECURSOR_LEN = 3000, the reference's upper limit (MP:167-184): the oracle against digests of the real binaries' output at span 3000
    (tests/golden/xl_folds.json.gz) -- the 1,500-nt window under both models and the 3,020-nt window under vienna-2.1.2, side by side (its vienna-1.8.5 fold
    alone takes the oracle a minute: the GPU twin, tests/test_fold_gpu.py, checks all four against the digests)."""
    import base64
    import concurrent.futures as cf
    from tests.golden.tools_digest import fold_digest
    fix, seqs = xl_fold_fixture()
    jobs = [("vienna-2.1.2", 0), ("vienna-2.1.2", 1), ("vienna-1.8.5", 1)]
    with cf.ThreadPoolExecutor(len(jobs)) as ex:      # ctypes releases the GIL
        res = list(ex.map(lambda mk: oracle.lfold(seqs[mk[1]].encode(), 3000, model=mk[0]), jobs))
    for (model, k), r in zip(jobs, res):
        exp = fix["folds"]["%s/3000" % model]
        assert r["mfe"] == exp["mfe"][k] and len(r["lines"]) == exp["n_lines"][k]
        assert fold_digest(r["lines"], r["mfe"]) == base64.b64decode(exp["digests"])[6 * k:6 * k + 6], (model, k)
