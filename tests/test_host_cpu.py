"""CPU tests of the host logic: config parser, C-ABI surface, record rendering, ingest, contig partition, stage checkpoints."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from mir_prefer_amd import config, dist, ingest, pipeline, records, synth
from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mir-prefer_amd", "libmirprefer.so")


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mirprefer.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mirp_[a-z_]+)\s*\(", text)))


def test_cabi_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    lib = C.CDLL(LIB)
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), s
    lib.mirp_abi_version.restype = C.c_int
    assert lib.mirp_abi_version() >= 1


def test_no_gpu_fails_loudly_without_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mir_prefer_amd import capi
    with pytest.raises(capi.MirpError):
        capi.Context(0)


def test_cabi_struct_sizes_match_numpy_dtypes():
    assert synth.ALN_DTYPE.itemsize == 16 and records.PEAK_DTYPE.itemsize == 16 and records.MATURE_DTYPE.itemsize == 16
    assert records.WINDOW_DTYPE.itemsize == 72 and records.LOCUS_DTYPE.itemsize == 48 and records.MIRNA_DTYPE.itemsize == 64


def _write_cfg(tmp_path, extra=""):
    fa = tmp_path / "g.fa"; fa.write_text(">c\nACGT\n")
    sam = tmp_path / "s.sam"; sam.write_text("@SQ\tSN:c\tLN:4\n")
    cfg = tmp_path / "config"
    cfg.write_text("# comment\nFASTA_FILE = %s\nALIGNMENT_FILE = %s\nNAME_PREFIX = t\nOUTFOLDER = %s\n%s" % (fa, sam, tmp_path / "out", extra))
    return str(cfg)


def test_config_defaults_and_values(tmp_path):
    opt = config.parse_configfile(_write_cfg(tmp_path, "PRECURSOR_LEN = 280\nALLOW_3NT_OVERHANG = y\nALLOW_NO_STAR_EXPRESSION=N\n"))
    assert opt["PRECURSOR_LEN"] == 280 and opt["READS_DEPTH_CUTOFF"] == 10 and opt["MAX_GAP"] == 100
    assert opt["MIN_MATURE_LEN"] == 18 and opt["MAX_MATURE_LEN"] == 23 and opt["CHECKPOINT_SIZE"] == 3000
    assert opt["ALLOW_3NT_OVERHANG"] is True and opt["ALLOW_NO_STAR_EXPRESSION"] is False
    assert len(opt["ALIGNMENT_FILE"]) == 1


@pytest.mark.parametrize("extra", ["PRECURSOR_LEN = 50\n", "PRECURSOR_LEN = 3001\n", "READS_DEPTH_CUTOFF = 1\n", "CHECKPOINT_SIZE = 5\n",
                                   "MIN_MATURE_LEN = 30\n", "ALLOW_3NT_OVERHANG = maybe\n", "FASTA_FILE = /nonexistent.fa\n"])
def test_config_validation_exits_like_the_reference(tmp_path, extra):
    with pytest.raises(SystemExit) as e:
        config.parse_configfile(_write_cfg(tmp_path, extra))
    assert e.value.code == -1


def test_ingest_matches_generator_and_sort_is_stable(tmp_path):
    ds = synth.make_dataset([30000, 20000], 20, n_samples=2, seed=4, contig_names=["b", "a"], edge_cases=True)
    sams = ds.write_sams(str(tmp_path))
    ds.write_fasta(str(tmp_path / "g.fa"))
    names, lens, samples, alns = ingest.read_sams(sams)
    assert names == ["b", "a"] and list(lens) == [30000, 20000] and samples == ["S1", "S2"]
    assert np.array_equal(alns, ds.sorted_alns())
    fa = dict(ingest.read_fasta(str(tmp_path / "g.fa")))
    assert all(np.array_equal(fa[n], s) for n, s in ds.contigs)
    key = alns["tid"].astype(np.int64) << 32 | alns["pos"]
    assert (np.diff(key) >= 0).all()


def test_native_sam_ingest_matches_python_parser(tmp_path):
    ds = synth.make_dataset([200000, 120000, 90000], 400, n_samples=3, seed=8, contig_names=["z", "a", "m"], edge_cases=True)
    sams = ds.write_sams(str(tmp_path), sq_order=[2, 0, 1])
    a = ingest.read_sams(sams, native=True)
    b = ingest.read_sams(sams, native=False)
    assert a[0] == b[0] == ["m", "z", "a"] and list(a[1]) == list(b[1]) and a[2] == b[2] == ["S1", "S2", "S3"]
    assert len(a[3]) == len(b[3]) > 3000 and np.array_equal(a[3], b[3])
    from mir_prefer_amd import capi
    for nt in (1, 3):
        assert np.array_equal(capi.ingest_sams(sams, n_threads=nt)[3], b[3])
    # gapped alignments: the record keeps POS and len(SEQ), the M / = / X blocks become coverage segments (SURVEY A-1); both parsers agree
    gap = tmp_path / "gap.sam"
    lines = ["@SQ\tSN:c\tLN:1000", "@SQ\tSN:d\tLN:500",
             "S_r0_x5\t0\tc\t5\t255\t10M2D9M\t*\t0\t0\t" + "A" * 19 + "\t" + "I" * 19,
             "S_r1_x7\t16\td\t40\t255\t3S8M1I4=2X10N5M2H\t*\t0\t0\t" + "C" * 23 + "\t" + "I" * 23,
             "S_r2_x2\t0\tc\t3\t255\t21M\t*\t0\t0\t" + "G" * 21 + "\t" + "I" * 21,
             "S_r3_x9\t4\t*\t0\t0\t*\t*\t0\t0\t" + "G" * 21 + "\t" + "I" * 21]
    gap.write_text("\n".join(lines) + "\n")
    n1 = capi.ingest_sams([str(gap)], with_segments=True)
    n2 = ingest.read_sams([str(gap)], native=False, with_segments=True)
    assert np.array_equal(n1[3], n2[3]) and len(n1[3]) == 3 and [int(x) for x in n1[3]["len"]] == [21, 19, 23]
    key = lambda a: sorted(map(tuple, a.tolist()))
    assert key(n1[4]) == key(n2[4])
    want = [(0, 5, 5, 19, 2, 0), (0, 5, 5, 10, 0, 0), (0, 17, 5, 9, 0, 0),                       # 10M2D9M at 5
            # 3S8M1I4=2X10N5M2H at 40: samtools 0.1.18 piles a read up below bam_calend() = 40 + 8 + 10 + 5 = 63 only (= and X are not added
            # to the end), which cuts the last block [64, 69) off
            (1, 40, 7, 23, 3, 0), (1, 40, 7, 8, 1, 0), (1, 48, 7, 4, 1, 0), (1, 52, 7, 2, 1, 0)]
    assert key(n1[4]) == sorted(want)
    for cig in ("10M2Q9M", "M", "10"):
        bad = tmp_path / "bad.sam"
        bad.write_text("@SQ\tSN:c\tLN:100\nS_r0_x5\t0\tc\t5\t255\t" + cig + "\t*\t0\t0\t" + "A" * 19 + "\t" + "I" * 19 + "\n")
        with pytest.raises(ValueError):
            ingest.read_sams([str(bad)])
        with pytest.raises(ValueError):
            ingest.read_sams([str(bad)], native=False)


def test_partition_contigs_lpt():
    parts = dist.partition_contigs([43, 36, 36, 35, 30, 31, 30, 28, 23, 23, 29, 27], 8)
    assert sorted(t for p in parts for t in p) == list(range(12))
    loads = [sum([43, 36, 36, 35, 30, 31, 30, 28, 23, 23, 29, 27][t] for t in p) for p in parts]
    assert max(loads) <= 59 and min(loads) >= 35
    assert dist.partition_contigs([5], 2) == [[0], []]


def test_gff_writer_matches_reference_fixture():
    for name in ("mini", "mini3"):
        exp = gu.load_json(os.path.join(name, "expected.json.gz"))
        res = []
        for e in gu.unjson(exp["result_raw"]):
            res.append(list(e[:10]) + [dict(e[10])])
        # the fixture's exprinfo subset lacks the imperfect-star keys; they only matter when ALLOW_3NT_OVERHANG is on
        if exp["config"]["ALLOW_3NT_OVERHANG"] == "Y":
            continue
        pipeline.adjust_mature_star(res)
        import tempfile
        with tempfile.NamedTemporaryFile("r", suffix=".gff3") as f:
            pipeline.write_gff(res, f.name)
            assert open(f.name).read() == exp["gff3"]


def test_stage_checkpoint_rules(tmp_path):
    rec = str(tmp_path / "x_recover")
    assert pipeline.detect_stage_last_finished(rec) is None and not pipeline.previous_stage_saved(rec, "prepare")
    f1 = tmp_path / "a"; f1.write_text("x")
    pipeline._save_recover(rec, {"last_stage": "prepare", "finished_stages": {"prepare": {"preparedname": str(f1)}}, "files": {"prepare": [str(f1)]}})
    assert pipeline.previous_stage_saved(rec, "prepare") and pipeline.detect_stage_last_finished(rec) == "prepare"
    f1.unlink()
    assert not pipeline.previous_stage_saved(rec, "prepare") and pipeline.detect_stage_last_finished(rec) is None


def test_fold_stage_folds_once():
    """pipeline.Pipeline._fold_device: windows over the default line capacity are folded again inside mirp_fold (side buffers), so the host
    stage folds exactly once and reports the per-window status it gets back."""
    import numpy as np
    from mir_prefer_amd import pipeline

    class StubCtx:
        def __init__(self):
            self.calls = []

        def fold(self, span, max_lines=96):
            self.calls.append((span, max_lines))

        def fold_status(self):
            return np.zeros(3, dtype=np.int32)

    p = pipeline.Pipeline.__new__(pipeline.Pipeline)
    p.rank, p.world = 0, 1
    p.ctx, p.opt = StubCtx(), {"PRECURSOR_LEN": 300}
    st = p._fold_device()
    assert p.ctx.calls == [(300, 96)] and (st == 0).all()


import pytest


@pytest.mark.parametrize("name", ["mini", "mini3", "mini185", "mini24", "mini400"])
def test_report_writers_match_reference_files(name, tmp_path):
    """mature.fa / precursor.fa / precursor.ss / detail.csv / detail.html / miRNA.stat.txt of the predict stage, byte for byte, from the reference's own
    result list (gen_mirna_fasta_ss_from_result MP:2963-3019, gen_mirna_info MP:2644-2728, gen_csv_table MP:2744-2779, MP:3585-3593)."""
    from mir_prefer_amd import pipeline
    from tests import golden_util as gu
    c = gu.load_pipeline_case(name)
    exp = c["exp"]
    result = [list(m[:10]) + [dict(m[10])] for m in gu.unjson(exp["result_raw"])]
    pipeline.adjust_mature_star(result)
    pipeline.write_gff(result, str(tmp_path / "x.gff3"))           # sorts the list like gen_gff_from_result
    assert open(tmp_path / "x.gff3").read() == exp["gff3"]
    contigs = dict(c["contigs"])
    pipeline.write_fasta_ss(result, contigs, str(tmp_path / "m.fa"), str(tmp_path / "p.fa"), str(tmp_path / "p.ss"))
    counts = pipeline.mirna_read_counts(result, c["contig_names"], c["alns"], len(c["sample_names"]))
    pipeline.write_csv_and_stat(result, contigs, c["sample_names"], counts, str(tmp_path / "d.csv"), str(tmp_path / "s.txt"))
    rep = exp["reports"]
    assert open(tmp_path / "m.fa").read() == rep["mature_fa"]
    assert open(tmp_path / "p.fa").read() == rep["precursor_fa"]
    assert open(tmp_path / "p.ss").read() == rep["precursor_ss"]
    assert open(tmp_path / "d.csv").read() == rep["detail_csv"]
    assert open(tmp_path / "s.txt").read() == rep["stat_txt"]
    pipeline.write_html(result, contigs, c["sample_names"], counts, str(tmp_path / "d.html"))      # gen_html_table_file MP:2793-2904
    assert open(tmp_path / "d.html").read() == rep["detail_html"]
    # the product's writer (mirp_write_reports, one native call for the seven files) against the same reference files
    nat = {k: str(tmp_path / ("n_" + k)) for k in ("gff", "mature", "precursor", "ss", "csv", "html", "stat")}
    pipeline.write_report_files(result, [pipeline._faidx(contigs, m[0], m[1], m[2] - 1) for m in result], c["sample_names"], counts, nat)
    for k, want in (("gff", exp["gff3"]), ("mature", rep["mature_fa"]), ("precursor", rep["precursor_fa"]), ("ss", rep["precursor_ss"]), ("csv", rep["detail_csv"]),
                    ("html", rep["detail_html"]), ("stat", rep["stat_txt"])):
        assert open(nat[k]).read() == want, k
    # per-locus read layouts (gen_map_result MP:2907-2959); the synthetic reads are perfect matches, as bowtie -v 0 produces
    pipeline.write_readmapping(result, contigs, c["contig_names"], c["alns"], c["sample_names"], counts, str(tmp_path / "rm"))
    assert sorted(os.listdir(tmp_path / "rm")) == sorted(exp["readmapping"])
    for fn, text in exp["readmapping"].items():
        assert open(tmp_path / "rm" / fn).read() == text, fn


@pytest.mark.parametrize("name", ["mini", "mini3", "mini185", "mini24", "mini400"])
def test_result_reports_one_call_matches_reference_files(name, tmp_path):
    """mirp_write_result_reports (the lean `pipeline` run's report path): the reference's RAW result list -- queue order, mature / star as the filter
    found them -- as flat records, shuffled; one native call must produce the reference's gff3, fasta / ss / csv / html / stat files and every
    read-mapping file byte for byte (swap MP:2611-2617, resultlist.sort() MP:2622, gen_mirna_info MP:2644-2728, gen_map_result MP:2907-2959)."""
    import random
    from mir_prefer_amd import capi, pipeline, records
    from tests import golden_util as gu
    c = gu.load_pipeline_case(name)
    exp = c["exp"]
    raw = [list(m[:10]) + [dict(m[10])] for m in gu.unjson(exp["result_raw"])]
    random.Random(3).shuffle(raw)
    tid_of = {n: t for t, n in enumerate(c["contig_names"])}
    stride = max(len(m[7]) for m in raw) + 3
    rec = np.zeros(len(raw), dtype=records.MIRNA_DTYPE)
    text = np.zeros((len(raw), stride), dtype=np.uint8)
    for k, m in enumerate(raw):
        e = m[10]
        rec[k] = (k, tid_of[m[0]], m[1], m[2], m[3], m[4], m[5], m[6], 1 if m[8] == "-" else 0, 1 if m[9] else 0, 0, 0, len(m[7]), 0,
                  e["total_depth_mature"], e["total_depth_star"])
        text[k, :len(m[7])] = np.frombuffer(m[7].encode(), dtype=np.uint8)
        text[k, len(m[7]):] = ord("#")          # a row is not NUL-terminated: the record's ss_len bounds it
    mark = "\x00SEQ\x00"
    form = [p for taxon in ("Viridiplantae", "ALL") for p in pipeline._mirbase_form_text(mark, taxon).split(mark)]
    out = tmp_path / "out"
    res, order, counts = capi.write_result_reports(rec, text, c["contig_names"], [sq for _, sq in c["contigs"]], c["alns"], c["sample_names"], form, str(out), name)
    rep = exp["reports"]
    for fn, want in ((name + "_miRNA.gff3", exp["gff3"]), (name + "_miRNA.mature.fa", rep["mature_fa"]), (name + "_miRNA.precursor.fa", rep["precursor_fa"]),
                     (name + "_miRNA.precursor.ss", rep["precursor_ss"]), (name + "_miRNA.detail.csv", rep["detail_csv"]), (name + "_miRNA.detail.html", rep["detail_html"]),
                     ("miRNA.stat.txt", rep["stat_txt"])):
        assert open(out / fn).read() == want, fn
    assert sorted(os.listdir(out / "readmapping")) == sorted(exp["readmapping"])
    for fn, want in exp["readmapping"].items():
        assert open(out / "readmapping" / fn).read() == want, fn
    # the optional outputs: list order, records after the swap, counts as the Python statement computes them
    ref = [list(m[:10]) + [dict(m[10])] for m in raw]
    pipeline.adjust_mature_star(ref)
    want_order = sorted(range(len(ref)), key=lambda k: ref[k][:10])
    assert order.tolist() == want_order
    assert [[c["contig_names"][r["tid"]], int(r["fold_s"]), int(r["fold_e"]), int(r["mat_s"]), int(r["mat_e"]), int(r["star_s"]), int(r["star_e"])] for r in res] == \
           [ref[k][:7] for k in want_order]
    assert np.array_equal(counts, pipeline.mirna_read_counts([ref[k] for k in want_order], c["contig_names"], c["alns"], len(c["sample_names"])))
    # no loci: nothing is written, nothing fails
    res0, _, _ = capi.write_result_reports(rec[:0], text[:0], c["contig_names"], [sq for _, sq in c["contigs"]], c["alns"], c["sample_names"], form, str(tmp_path / "none"), name)
    assert len(res0) == 0 and not os.path.exists(tmp_path / "none")


def test_native_report_writer_equals_python_writers_on_random_loci(tmp_path):
    """mirp_write_reports against the Python statement of the same formats on 3,000 random loci: both strands, star before / after the mature,
    the three overhang forms, 1..3 samples, contig names in any order."""
    import random
    from mir_prefer_amd import pipeline
    rnd = random.Random(5)
    for ns in (1, 3):
        samples = ["s%d.sam" % k for k in range(ns)]
        result, pre = [], []
        for _ in range(3000 if ns == 3 else 40):
            fs = rnd.randrange(1, 10 ** 7)
            ln = rnd.randrange(60, 300)
            a, b = sorted(rnd.sample(range(0, ln - 25), 2))
            if b - a < 26:
                b = min(a + 26, ln - 25)
            ml, sl = rnd.randrange(18, 25), rnd.randrange(18, 25)
            arms = [(fs + a, fs + a + ml), (fs + b, fs + b + sl)]
            if rnd.random() < 0.5:
                arms.reverse()
            e = {"total_depth_mature": rnd.randrange(1, 999), "total_depth_star": rnd.choice([0, 0, 5, 77])}
            if rnd.random() < 0.5:
                e["max_imperfect_star"] = rnd.choice([0, 1])
                e["imperfect_star_which"] = rnd.choice([0, 1, 2])
            result.append(["chr%d" % rnd.randrange(12), fs, fs + ln, arms[0][0], arms[0][1], arms[1][0], arms[1][1],
                           "".join(rnd.choice("(.)") for _ in range(ln)), rnd.choice("+-"), bool(e["total_depth_star"]), e])
        result.sort(key=lambda m: m[:10])
        for m in result:
            pre.append("".join(rnd.choice("ACGU") for _ in range(m[2] - m[1])))
        counts = np.array([[[rnd.randrange(0, 10 ** rnd.randrange(1, 7)) for _ in range(4)] for _ in range(ns)] for _ in result], dtype=np.int64)
        pay = [{"pre": x} for x in pre]
        py = {k: str(tmp_path / ("p_%d_%s" % (ns, k))) for k in ("gff", "mature", "precursor", "ss", "csv", "html", "stat")}
        pipeline.write_gff(list(result), py["gff"])
        pipeline.write_fasta_ss(result, pay, py["mature"], py["precursor"], py["ss"])
        pipeline.write_csv_and_stat(result, pay, samples, counts, py["csv"], py["stat"])
        pipeline.write_html(result, pay, samples, counts, py["html"])
        nat = {k: v.replace("p_", "n_") for k, v in py.items()}
        pipeline.write_report_files(result, pre, samples, counts, nat)
        for k in py:
            assert open(nat[k], "rb").read() == open(py[k], "rb").read(), (ns, k)
    # a path that cannot be opened is an error with the file's name, not a silent skip
    with pytest.raises(Exception, match="no_such_dir"):
        pipeline.write_report_files(result[:2], pre[:2], samples, counts[:2], {"gff": str(tmp_path / "no_such_dir" / "x.gff3")})


def test_gff_keep_regions_match_the_reference_functions(tmp_path):
    """Keep-region BED text of gen_keep_regions_from_exclude_gff / _include_gff (MP:543-652) on seeded GFF files (incl. comments, blank
    lines, a ##FASTA section, features on unknown contigs, overlapping / nested / adjacent features)."""
    from mir_prefer_amd import gffmask
    from tests import golden_util as gu
    gold = gu.load_json("gffmask.json.gz")
    for k, c in enumerate(gold["bed_cases"]):
        p = tmp_path / ("c%d.gff" % k)
        p.write_text(c["gff"])
        dict_len = {n: l for n, l in c["dict_len"]}
        ex = "".join("%s\t%d\t%d\n" % r for r in gffmask.keep_regions_exclude(str(p), dict_len, 55))
        inc = "".join("%s\t%d\t%d\n" % r for r in gffmask.keep_regions_include(str(p), 55))
        assert ex == c["bed_exclude"], k
        assert inc == c["bed_include"], k


def test_gff_mask_keeps_what_samtools_view_L_keeps():
    """apply_keep against `samtools view -L` of the bundled samtools 0.1.18 (names of the kept alignments, in order)."""
    import numpy as np
    from mir_prefer_amd import gffmask, synth
    from tests import golden_util as gu
    gold = gu.load_json("gffmask.json.gz")
    for v in gold["view_cases"]:
        names = [n for n, _ in v["contigs"]]
        a = np.zeros(len(v["records"]), dtype=synth.ALN_DTYPE)
        for k, (c, p, l) in enumerate(v["records"]):
            a[k] = (names.index(c), p, 1, l, 0, 0)
        a["depth"] = np.arange(len(a))                    # record identity
        kept = gffmask.apply_keep(a, names, [tuple(x) for x in v["bed"]])
        want = [int(nm.split("_r")[1].split("_x")[0]) for nm in v["kept_names"]]
        assert [int(x) for x in kept["depth"]] == want
        assert 0 < len(want) < len(a)


def test_prepare_stage_applies_gff_exclude_mask(tmp_path):
    """run_prepare with GFF_FILE_EXCLUDE: the prepared record array is the ingest output filtered by the reference's keep regions."""
    import numpy as np
    from mir_prefer_amd import gffmask, ingest, pipeline, synth
    ds = synth.make_dataset([30000, 20000], 25, n_samples=2, seed=12, contig_names=["cB", "cA"])
    ds.write_fasta(str(tmp_path / "g.fa"))
    import gzip, shutil
    sams = []
    for sp in ds.write_sams(str(tmp_path)):          # compressed inputs take the host parser / host filter (no GPU in this test)
        with open(sp, "rb") as fi, gzip.open(sp + ".gz", "wb") as fo:
            shutil.copyfileobj(fi, fo)
        sams.append(sp + ".gz")
    gff = tmp_path / "ex.gff"
    gff.write_text("##gff-version 3\ncB\ts\tgene\t2000\t9000\t.\t+\t.\tID=a\ncB\ts\tgene\t8000\t12000\t.\t-\t.\tID=b\ncA\ts\tgene\t500\t700\t.\t+\t.\tID=c\n")
    p = pipeline.Pipeline.__new__(pipeline.Pipeline)
    p.rank, p.world = 0, 1
    p.opt = {"ALIGNMENT_FILE": sams, "GFF_FILE_EXCLUDE": str(gff), "GFF_FILE_INCLUDE": "", "NAME_PREFIX": "t", "OUTFOLDER": str(tmp_path)}
    p.tmp = str(tmp_path)
    p.recovername = str(tmp_path / "t_recover")
    p.run_prepare()
    z = np.load(tmp_path / "prepared.npz", allow_pickle=True)
    names, lens, samples, alns = ingest.read_sams(sams)
    want = gffmask.apply_keep(alns, names, gffmask.keep_regions_exclude(str(gff), dict(zip(names, [int(x) for x in lens])), 55))
    assert np.array_equal(z["alns"], want) and 0 < len(want) < len(alns)
    inside = (z["alns"]["tid"] == names.index("cB")) & (z["alns"]["pos"] > 2100) & (z["alns"]["pos"] + z["alns"]["len"] < 11900)
    assert not inside.any()                               # nothing survives strictly inside the merged excluded block


def test_native_fasta_reader_matches_the_python_reader(tmp_path):
    """mirp_read_fasta (host-only entry point of the library; replaces the genome access of `samtools faidx`, MP:1100-1105): names = first word
    of the header, lines joined with surrounding white space stripped, case kept; with a want list the other sequences are skipped."""
    from mir_prefer_amd import capi, ingest
    fa = tmp_path / "g.fa"
    fa.write_bytes(b">chrB some description\nACGTacgtNN\r\n  ACGT  \n\nTT\n>chrA\n>chrC\tx\nGGGG>notaheader\nCC\n>last\nA")
    py = ingest.read_fasta(str(fa))
    nat = capi.read_fasta(str(fa))
    assert [n for n, _ in nat] == [n for n, _ in py] == ["chrB", "chrA", "chrC", "last"]
    for (_, a), (_, b) in zip(nat, py):
        assert a.tobytes() == b.tobytes()
    assert nat[0][1].tobytes() == b"ACGTacgtNNACGTTT" and nat[1][1].tobytes() == b"" and nat[2][1].tobytes() == b"GGGG>notaheaderCC"
    part = capi.read_fasta(str(fa), want=["chrC", "chrB"])
    assert [n for n, _ in part] == ["chrB", "chrA", "chrC", "last"]
    assert part[0][1].tobytes() == b"ACGTacgtNNACGTTT" and part[1][1] is None and part[2][1].tobytes() == b"GGGG>notaheaderCC" and part[3][1] is None
    with pytest.raises(ValueError):
        capi.read_fasta(str(tmp_path / "missing.fa"))


def test_host_keep_mask_uses_the_reference_span_of_gapped_reads(tmp_path):
    """gffmask.keep_mask / ingest.read_sams(regions=...) against the reads the bundled samtools 0.1.18 `view -L` keeps on the gapped fixture:
    the overlap test runs on [POS - 1, bam_calend), the M / D / N span (gen_gapped_golden.py, `view_L`)."""
    from mir_prefer_amd import gffmask, ingest
    from tests import golden_util as gu
    g = gu.load_json("gapped.json.gz")
    sam = tmp_path / "S1.sam"
    sam.write_text(g["sam"])
    names = [c[0] for c in g["contigs"]]
    regions = gffmask.regions_by_tid([tuple(x) for x in g["view_L"]["bed"]], names)
    _, _, _, alns = ingest.read_sams([str(sam)], native=False, regions=regions)
    lines = [l.split("\t") for l in g["sam"].splitlines() if not l.startswith("@")]
    kept = set(g["view_L"]["kept_ids"])
    want = sorted((names.index(f[2]), int(f[3]), int(f[0].rsplit("_x", 1)[1]), len(f[9])) for f in lines if f[0] in kept)
    got = sorted((int(r["tid"]), int(r["pos"]), int(r["depth"]), int(r["len"])) for r in alns)
    assert got == want and 5 < len(want) < len(lines)
    # the len(SEQ) rule (round 2) disagrees with the binary on this fixture: the test has teeth
    _, _, _, a0 = ingest.read_sams([str(sam)], native=False)
    old = sorted((int(r["tid"]), int(r["pos"]), int(r["depth"]), int(r["len"])) for r in a0[gffmask.keep_mask(a0, regions)])
    assert old != want
