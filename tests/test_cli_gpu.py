"""GPU test of the host CLI / stage drivers on the golden datasets: same config file, stage artefacts and gff3 as the reference."""
import gzip
import os
import shutil

import numpy as np
import pytest

from mir_prefer_amd import cli, pipeline
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def _setup(name, tmp_path):
    exp = gu.load_json(os.path.join(name, "expected.json.gz"))
    src = os.path.join(gu.GOLD, name)
    files = {}
    for fn in ["genome.fa"] + [s + ".sam" for s in exp["sample_names"]]:
        dst = tmp_path / fn
        with gzip.open(os.path.join(src, fn + ".gz"), "rb") as fi, open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        files[fn] = str(dst)
    cfg = tmp_path / "config"
    c = exp["config"]
    lines = ["FASTA_FILE = " + files["genome.fa"], "ALIGNMENT_FILE = " + ", ".join(files[s + ".sam"] for s in exp["sample_names"]),
             "OUTFOLDER = " + str(tmp_path / "out")]
    for k in ("PRECURSOR_LEN", "READS_DEPTH_CUTOFF", "MAX_GAP", "MIN_MATURE_LEN", "MAX_MATURE_LEN", "ALLOW_NO_STAR_EXPRESSION", "ALLOW_3NT_OVERHANG",
              "CHECKPOINT_SIZE", "NAME_PREFIX"):
        lines.append("%s = %s" % (k, c[k]))
    cfg.write_text("\n".join(lines) + "\n")
    return exp, str(cfg), tmp_path / "out"


@pytest.mark.parametrize("name", ["mini", "mini3", "mini185", "mini24", "mini400"])
def test_pipeline_verb_reproduces_reference_outputs(name, tmp_path):
    exp, cfg, out = _setup(name, tmp_path)
    assert cli.main(["-k", "-d", "--fold-model", exp.get("fold_model", "vienna-2.1.2"), "pipeline", cfg]) == 0
    prefix = exp["config"]["NAME_PREFIX"]
    tmp = out / (prefix + "_tmp")
    assert open(out / (prefix + "_miRNA.gff3")).read() == exp["gff3"]
    rep = exp["reports"]    # the report files of the predict stage (MP:2963-3019, 2644-2779, 3585-3593), byte for byte
    assert open(out / (prefix + "_miRNA.mature.fa")).read() == rep["mature_fa"]
    assert open(out / (prefix + "_miRNA.precursor.fa")).read() == rep["precursor_fa"]
    assert open(out / (prefix + "_miRNA.precursor.ss")).read() == rep["precursor_ss"]
    assert open(out / (prefix + "_miRNA.detail.csv")).read() == rep["detail_csv"]
    assert open(out / "miRNA.stat.txt").read() == rep["stat_txt"]
    assert open(out / (prefix + "_miRNA.detail.html")).read() == rep["detail_html"]
    for fn, text in exp["readmapping"].items():
        assert open(out / "readmapping" / fn).read() == text, fn
    # -d artefact: why the other regions are not miRNAs -- every block with every failure line and the expression numbers.  Block order is
    # not comparable: the reference appends per-piece results in the order its worker processes finish (MP:2481-2497).
    def blocks(text):
        bl = [b for b in text.split("===========================================================\n") if b.strip()]
        d = {b.split("\n", 1)[0]: b for b in bl}
        assert len(d) == len(bl)
        return d
    got_b, want_b = blocks(open(out / (prefix + "_reason_why_not_miRNA.txt")).read()), blocks(exp["reasons_txt"])
    assert sorted(got_b) == sorted(want_b)
    for k in want_b:
        assert got_b[k] == want_b[k], k
    assert len(want_b) > 50
    # failed_readmapping: the read layouts of the (mature, structure) pairs that failed the expression test.  Files are numbered in block
    # order, so the comparison is over the contents with the running number taken out.
    def unnumbered(texts):
        return sorted(t.split(" ", 1)[1] for t in texts)
    got_f = [open(out / "failed_readmapping" / fn).read() for fn in sorted(os.listdir(out / "failed_readmapping"))]
    assert unnumbered(got_f) == unnumbered(exp["failed_readmapping"].values())
    assert all(t.startswith(">miRNA-precursor_") for t in got_f) and len(got_f) >= 2
    assert open(tmp / ("bam.depth.cut%d" % exp["config"]["READS_DEPTH_CUTOFF"])).read() == exp["depth_cut"]
    assert open(tmp / (prefix + "_ExRegionA.gff3")).read() == exp["exregion_gff"]          # the candidate stage's debug artefact (MP:1357-1369)
    fasta = open(tmp / (prefix + ".rnalfold.in_0.fa")).read().splitlines()
    want = [x for p in exp["pieces"] for e in p["fasta"] for x in e]
    assert fasta == want
    # RNALfold-format text of the fold stage == the reference's RNALfold 2.1.2 output (sequence echo case aside)
    got = open(tmp / (prefix + "_rnalfoldoutput_0")).read().upper().splitlines()
    ref = "".join(p["rnalfold_out"] for p in exp["pieces"]).upper().splitlines()
    assert [l for l in got if not l.startswith(">")] == [l for l in ref if not l.startswith(">")]
    rec = pipeline.load_recover_file(str(tmp / (prefix + "_recover")))
    assert rec["last_stage"] == "predict" and set(rec["finished_stages"]) == {"prepare", "candidate", "fold", "predict"}


@pytest.mark.parametrize("chunks", [0, 7])
@pytest.mark.parametrize("name", ["mini", "mini3", "mini185", "mini24", "mini400"])
def test_lean_pipeline_process_reproduces_reference_outputs(name, chunks, tmp_path):
    """`python -m mir_prefer_amd.cli pipeline <config>` as a user runs it -- a fresh process, no -k, no -d: the device context and the genome read start
    before the heavy imports (early.py), no stage artefact is written (they would be deleted at the end, MP:3630-3639), the report files come from one
    native call.  The outputs are the reference's byte for byte, the temporary folder is gone, the exit status is 0."""
    import subprocess
    import sys
    exp, cfg, out = _setup(name, tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "mir_prefer_amd.cli", "--fold-model", exp.get("fold_model", "vienna-2.1.2"), "pipeline", cfg], cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONPATH=root, MIRP_STREAM_CHUNKS=str(chunks)), capture_output=True, text=True, timeout=600)
    # chunks = 7: the fold / filter / report pipeline (mirp_fold_predict_report_stream) cut into seven chunks of ~40 windows, read-mapping files of one chunk
    # written while the next is folded; 0 = the default (one chunk at this size)
    assert r.returncode == 0, r.stderr[-2000:]
    prefix = exp["config"]["NAME_PREFIX"]
    rep = exp["reports"]
    for fn, want in ((prefix + "_miRNA.gff3", exp["gff3"]), (prefix + "_miRNA.mature.fa", rep["mature_fa"]), (prefix + "_miRNA.precursor.fa", rep["precursor_fa"]),
                     (prefix + "_miRNA.precursor.ss", rep["precursor_ss"]), (prefix + "_miRNA.detail.csv", rep["detail_csv"]), (prefix + "_miRNA.detail.html", rep["detail_html"]),
                     ("miRNA.stat.txt", rep["stat_txt"])):
        assert open(out / fn).read() == want, fn
    assert sorted(os.listdir(out / "readmapping")) == sorted(exp["readmapping"])
    for fn, text in exp["readmapping"].items():
        assert open(out / "readmapping" / fn).read() == text, fn
    assert not os.path.exists(out / (prefix + "_tmp"))
    n = len(exp["readmapping"])
    assert ("%d miRNAs identified." % n) in r.stdout and "Temporary folder removed." in r.stdout
    for stage in ("prepare", "candidate", "fold", "predict"):
        assert ("Done (%s stage)" % stage) in r.stdout
    # the same verb in-process: the stage drivers behave the same without the early start
    shutil.rmtree(out)
    assert cli.main(["--fold-model", exp.get("fold_model", "vienna-2.1.2"), "pipeline", cfg]) == 0
    assert open(out / (prefix + "_miRNA.gff3")).read() == exp["gff3"] and not os.path.exists(out / (prefix + "_tmp"))


def test_stage_verbs_and_recover(tmp_path):
    exp, cfg, out = _setup("mini", tmp_path)
    with pytest.raises(SystemExit):          # fold before candidate: refused like the reference (MP:3446-3448)
        cli.main(["fold", cfg])
    for verb in ("prepare", "candidate"):
        assert cli.main([verb, cfg]) == 0
    assert cli.main(["recover", cfg]) == 0    # continues with fold + predict
    assert open(out / "mini_miRNA.gff3").read() == exp["gff3"]


def _run_ranks(world, args, port, backend="gloo"):
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MIRP_DIST_BACKEND=backend)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-m", "mir_prefer_amd.cli"] + args, cwd=root,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    return [p.returncode for p in procs], logs


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (3, "gloo"), (2, "local"), (3, "local")])
def test_pipeline_verb_sharded_over_ranks(world, backend, tmp_path):
    """Contig sharding: one process per rank (here all on GPU 0) must produce the files of the single-process run, -d output included -- and with
    it the strand-vote quirk at the first run of a shard (mirp_set_contig_shard).  backend gloo: rank 0 ingests, every exchange is a host object;
    backend local: the library's own exchanges -- the sharded ingest (every rank tokenizes a byte range of every SAM file, records are routed
    to the contig's owner: mirp_ingest_sams_shard) and the gather of the loci list (mirp_gather_loci) -- over the transport for ranks that share
    a GPU (on a multi-GPU node the same calls run over RCCL)."""
    exp, cfg, out = _setup("mini", tmp_path)
    codes, logs = _run_ranks(world, ["-k", "-d", "--device", "0", "pipeline", cfg], 29517 + world + (10 if backend == "local" else 0), backend)
    assert all(c == 0 for c in codes), logs
    prefix = exp["config"]["NAME_PREFIX"]
    tmp = out / (prefix + "_tmp")
    assert open(out / (prefix + "_miRNA.gff3")).read() == exp["gff3"]
    rep = exp["reports"]
    assert open(out / (prefix + "_miRNA.detail.csv")).read() == rep["detail_csv"]
    assert open(out / (prefix + "_miRNA.precursor.ss")).read() == rep["precursor_ss"]
    assert open(tmp / ("bam.depth.cut%d" % exp["config"]["READS_DEPTH_CUTOFF"])).read() == exp["depth_cut"]
    assert open(tmp / (prefix + "_ExRegionA.gff3")).read() == exp["exregion_gff"]          # the candidate stage's debug artefact (MP:1357-1369)
    for fn, text in exp["readmapping"].items():
        assert open(out / "readmapping" / fn).read() == text, fn
    got_f = [open(out / "failed_readmapping" / fn).read() for fn in sorted(os.listdir(out / "failed_readmapping"))]
    assert sorted(t.split(" ", 1)[1] for t in got_f) == sorted(t.split(" ", 1)[1] for t in exp["failed_readmapping"].values())
    # every block of the -d reasons file of the single-process reference run, line for line
    def blocks(text):
        return sorted(b for b in text.split("===========================================================\n") if b.strip())
    assert blocks(open(out / (prefix + "_reason_why_not_miRNA.txt")).read()) == blocks(exp["reasons_txt"])
    # all pieces exist and together hold every FASTA entry of the reference run
    got = []
    for r in range(world):
        got += open(tmp / (prefix + ".rnalfold.in_%d.fa" % r)).read().splitlines()
    want = [x for p in exp["pieces"] for e in p["fasta"] for x in e]
    assert sorted(got) == sorted(want) and len(got) == len(want)
    rec = pipeline.load_recover_file(str(tmp / (prefix + "_recover")))
    assert rec["world"] == world
    if backend == "local":          # one prepared file per rank, each holding only the records of the rank's own contigs
        import numpy as np
        from mir_prefer_amd import dist
        files = rec["finished_stages"]["prepare"]["preparedname"]
        assert len(files) == world
        z0 = np.load(files[0], allow_pickle=True)
        parts = dist.partition_contigs(z0["contig_lens"], world)
        total = 0
        for r, fn in enumerate(files):
            a = np.load(fn, allow_pickle=True)["alns"]
            assert set(np.unique(a["tid"]).tolist()) <= set(parts[r])
            total += len(a)
        assert total == sum(len(gu.load_pipeline_case("mini")["alns"]) for _ in range(1))


def _read_outputs(out, prefix):
    files = {}
    for fn in [prefix + "_miRNA.gff3", prefix + "_miRNA.mature.fa", prefix + "_miRNA.precursor.fa", prefix + "_miRNA.precursor.ss", prefix + "_miRNA.detail.csv",
               prefix + "_miRNA.detail.html", "miRNA.stat.txt"]:
        files[fn] = open(out / fn).read()
    for fn in sorted(os.listdir(out / "readmapping")):
        files["readmapping/" + fn] = open(out / "readmapping" / fn).read()
    return files


@pytest.mark.parametrize("world,backend", [(3, "local"), (2, "gloo")])
def test_window_rebalancing_moves_windows_and_keeps_the_files(world, backend, tmp_path):
    """Window-level re-balancing in front of the fold (balance.py): the fullest rank ships the tail of its window list, the helpers fold + filter
    the imported windows through the batch entry points, and every result file equals the single-process run's.  MIRP_BALANCE_MIN_MOVE=1 makes the
    miniature move windows at all; the payloads travel over the library's transport (local) or as host objects (gloo)."""
    exp, cfg, out = _setup("mini", tmp_path)
    os.environ["MIRP_BALANCE_MIN_MOVE"] = "1"
    try:
        codes, logs = _run_ranks(world, ["-k", "--device", "0", "pipeline", cfg], 29561 + world, backend)
    finally:
        del os.environ["MIRP_BALANCE_MIN_MOVE"]
    assert all(c == 0 for c in codes), logs
    assert "Re-balancing the fold" in logs[0], logs[0]
    prefix = exp["config"]["NAME_PREFIX"]
    assert open(out / (prefix + "_miRNA.gff3")).read() == exp["gff3"]
    rep = exp["reports"]
    got = _read_outputs(out, prefix)
    assert got[prefix + "_miRNA.detail.csv"] == rep["detail_csv"] and got[prefix + "_miRNA.precursor.ss"] == rep["precursor_ss"]
    assert got[prefix + "_miRNA.mature.fa"] == rep["mature_fa"] and got[prefix + "_miRNA.precursor.fa"] == rep["precursor_fa"]
    assert got[prefix + "_miRNA.detail.html"] == rep["detail_html"] and got["miRNA.stat.txt"] == rep["stat_txt"]
    for fn, text in exp["readmapping"].items():
        assert got["readmapping/" + fn] == text, fn
    # the fold artefact: every window's RNALfold text exactly once over the ranks' files and the helpers' ".from<src>" parts
    tmp = out / (prefix + "_tmp")
    rec = pipeline.load_recover_file(str(tmp / (prefix + "_recover")))
    names = rec["finished_stages"]["fold"]["foldnames"]
    assert any(".from" in n for n in names)
    lines = []
    for n in names:
        lines += open(n).read().upper().splitlines()
    ref = "".join(p["rnalfold_out"] for p in exp["pieces"]).upper().splitlines()
    assert sorted(l for l in lines if not l.startswith(">")) == sorted(l for l in ref if not l.startswith(">"))
    assert sorted(l for l in lines if l.startswith(">")) == sorted(l for l in ref if l.startswith(">"))


def test_window_rebalancing_at_config3_shape_equals_single_rank(tmp_path):
    """BASELINE config[3]-shaped contig lengths (12 MSU7-like contigs, scaled to 1/150) on 3 ranks over the local transport: whole-contig LPT leaves
    the ranks uneven, windows move, and every output file equals the 1-rank run's byte for byte."""
    from mir_prefer_amd import synth
    msu7 = [43300000, 35900000, 36400000, 35500000, 30000000, 31200000, 29700000, 28400000, 23000000, 23200000, 29000000, 27500000]
    lens = [l // 150 for l in msu7]
    ds = synth.make_dataset(lens, 900, n_samples=2, seed=77, contig_names=["Chr%d" % (k + 1) for k in range(12)])
    # a skewed read set: the generator spreads its loci evenly, so only the four largest contigs keep their reads -- LPT puts two of them on one rank
    ds.alns = ds.alns[ds.alns["tid"] < 4]
    src = tmp_path / "in"
    src.mkdir()
    sams = ds.write_sams(str(src))
    fa = str(src / "genome.fa")
    ds.write_fasta(fa)
    outs = {}
    for tag, world in (("w1", 1), ("w3", 3)):
        cfg = tmp_path / ("config_" + tag)
        out = tmp_path / ("out_" + tag)
        cfg.write_text("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %s\nNAME_PREFIX = c3\nPRECURSOR_LEN = 300\nREADS_DEPTH_CUTOFF = 10\nMAX_GAP = 100\n"
                       % (fa, ", ".join(sams), out))
        if world == 1:
            assert cli.main(["-k", "pipeline", str(cfg)]) == 0
        else:
            codes, logs = _run_ranks(world, ["-k", "--device", "0", "pipeline", str(cfg)], 29577, "local")
            assert all(c == 0 for c in codes), logs
            assert "Re-balancing the fold" in logs[0], logs[0]
        outs[tag] = _read_outputs(out, "c3")
    assert len(outs["w1"]) > 20
    assert outs["w1"].keys() == outs["w3"].keys()
    for k in outs["w1"]:
        assert outs["w1"][k] == outs["w3"][k], k


def test_sharded_run_fails_on_every_rank_together(tmp_path):
    """A failure that only rank 0 can see (the GFF file leaves no region) must end every rank with a non-zero status instead of leaving the
    others in a collective; and stage files of one world size are refused by a run of another."""
    exp, cfg, out = _setup("mini", tmp_path)
    gff = tmp_path / "all.gff"
    names = [l.split("\t")[1][3:] for l in gzip.open(os.path.join(gu.GOLD, "mini", exp["sample_names"][0] + ".sam.gz"), "rt") if l.startswith("@SQ")]
    gff.write_text("".join("%s\tx\tgene\t1\t100000000\t.\t+\t.\tID=g\n" % n for n in names))
    cfg2 = tmp_path / "config_gff"
    cfg2.write_text(open(cfg).read() + "GFF_FILE_EXCLUDE = %s\n" % gff)
    codes, logs = _run_ranks(2, ["-k", "pipeline", str(cfg2)], 29531)
    assert all(c != 0 for c in codes), logs
    # world-size mismatch between stages
    codes, logs = _run_ranks(2, ["-k", "prepare", cfg], 29533)
    assert all(c == 0 for c in codes), logs
    with pytest.raises(SystemExit):
        cli.main(["candidate", cfg])


def _run_cli(args, cwd, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-m", "mir_prefer_amd.cli"] + args, cwd=str(cwd), env=dict(os.environ, PYTHONPATH=root, **(env or {})), capture_output=True,
                          text=True, timeout=600)


def test_lean_pipeline_edge_cases(tmp_path):
    """The lean `pipeline` process on inputs that end early: (1) a SAM file with a malformed read id -- the early host ingest's error surfaces as the
    reference's message and exit status -1 (MP:242-253) while the device-open thread is still running; (2) reads that never reach the depth cutoff -- no
    candidate, "0 miRNA identified", no result files, exit 0; (3) a window over the default line capacity (a planted tandem repeat) -- the streamed fold /
    filter / report call re-folds it like the stage-by-stage run: same files as `-k pipeline`."""
    from mir_prefer_amd import synth
    exp, cfg, out = _setup("mini", tmp_path)
    # (1)
    sam = tmp_path / "S1.sam"
    lines = open(sam).read().splitlines()
    k = next(i for i, l in enumerate(lines) if not l.startswith("@"))
    lines[k + 5] = "badid" + lines[k + 5][lines[k + 5].index("\t"):]
    sam.write_text("\n".join(lines) + "\n")
    r = _run_cli(["pipeline", cfg], tmp_path)
    assert r.returncode == 255 and "Read id must be in" in r.stderr, (r.returncode, r.stderr[-500:])
    assert not os.path.exists(out / "mini_miRNA.gff3")
    # (2)
    d2 = tmp_path / "low"
    d2.mkdir()
    ds = synth.make_dataset([50000], 0, n_samples=1, seed=4, contig_names=["c1"])
    alns = ds.alns.copy()
    ds2 = synth.Dataset(ds.contigs, ds.sample_names, alns[:0], [])
    sams = ds2.write_sams(str(d2))
    ds2.write_fasta(str(d2 / "g.fa"))
    (d2 / "config").write_text("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %s\nNAME_PREFIX = low\n" % (d2 / "g.fa", sams[0], d2 / "out"))
    with open(sams[0], "a") as f:          # one read, depth 3: below READS_DEPTH_CUTOFF
        f.write("S1_r0_x3\t0\tc1\t1000\t255\t21M\t*\t0\t0\t%s\t%s\n" % ("A" * 21, "I" * 21))
    r = _run_cli(["pipeline", str(d2 / "config")], d2)
    assert r.returncode == 0, r.stderr[-800:]
    assert "0 miRNA identified. No result files generated." in r.stdout and not os.path.exists(d2 / "out" / "low_miRNA.gff3")
    # (3)
    d3 = tmp_path / "rep"
    d3.mkdir()
    ds = synth.make_dataset([120000], 40, n_samples=1, seed=8, contig_names=["c1"], edge_cases=True)
    g = ds.contigs[0][1].copy()
    a0 = ds.alns[ds.alns["depth"] > 20][0]          # a covered locus: overwrite its surroundings with a dinucleotide repeat (one structure line per start position)
    lo = max(0, int(a0["pos"]) - 160)
    g[lo:lo + 340] = np.frombuffer((b"AT" * 170), dtype=np.uint8)
    ds3 = synth.Dataset([("c1", g)], ds.sample_names, ds.alns, [])
    sams = ds3.write_sams(str(d3))
    ds3.write_fasta(str(d3 / "g.fa"))
    for name, extra in (("lean", []), ("keep", ["-k"])):
        (d3 / ("config_" + name)).write_text("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %s\nNAME_PREFIX = rep\n" % (d3 / "g.fa", sams[0], d3 / name))
        r = _run_cli(extra + ["pipeline", str(d3 / ("config_" + name))], d3, env={"MIRP_STREAM_CHUNKS": "3"})
        assert r.returncode == 0, r.stderr[-800:]
    import filecmp
    files = [f for f in os.listdir(d3 / "keep") if os.path.isfile(d3 / "keep" / f)]
    assert "rep_miRNA.gff3" in files
    match, mismatch, errors = filecmp.cmpfiles(str(d3 / "keep"), str(d3 / "lean"), files, shallow=False)
    assert not mismatch and not errors, (mismatch, errors)
    rm = sorted(os.listdir(d3 / "keep" / "readmapping"))
    assert rm == sorted(os.listdir(d3 / "lean" / "readmapping"))
    match, mismatch, errors = filecmp.cmpfiles(str(d3 / "keep" / "readmapping"), str(d3 / "lean" / "readmapping"), rm, shallow=False)
    assert not mismatch and not errors


def test_lean_pipeline_relative_paths_and_log(tmp_path):
    """Relative OUTFOLDER / input paths and -L in a lean run: the native report writer creates the folders it needs, the log file is written
    (MP:60-66), the result is the reference's."""
    exp, cfg, out = _setup("mini", tmp_path)
    text = open(cfg).read().replace(str(tmp_path) + "/", "")
    (tmp_path / "rel.cfg").write_text(text.replace("OUTFOLDER = out", "OUTFOLDER = res/deep"))
    r = _run_cli(["-L", "pipeline", "rel.cfg"], tmp_path)
    assert r.returncode == 0, r.stderr[-1500:]
    assert open(tmp_path / "res" / "deep" / "mini_miRNA.gff3").read() == exp["gff3"]
    assert os.path.exists(tmp_path / "res" / "deep" / "mini.log")
    assert sorted(os.listdir(tmp_path / "res" / "deep" / "readmapping")) == sorted(exp["readmapping"])


def test_pipeline_process_at_precursor_length_2000(tmp_path):
    """PRECURSOR_LEN = 2000 (the reference accepts 60 .. 3000, MP:167-184) through the command line: the lean `pipeline` process (streamed fold / filter /
    reports) and `-k pipeline` (stage by stage, artefacts kept) write the same result files; the kept RNALfold text holds windows of about 2,000 nt whose
    structure lines are as long."""
    from mir_prefer_amd import synth
    import filecmp
    ds = synth.make_dataset([60000, 40000], 10, n_samples=2, seed=2000, contig_names=["k2", "k1"], edge_cases=True)
    sams = ds.write_sams(str(tmp_path))
    ds.write_fasta(str(tmp_path / "g.fa"))
    for name, extra in (("lean", []), ("keep", ["-k"])):
        (tmp_path / ("config_" + name)).write_text("FASTA_FILE = %s\nALIGNMENT_FILE = %s\nOUTFOLDER = %s\nNAME_PREFIX = xl\nPRECURSOR_LEN = 2000\nREADS_DEPTH_CUTOFF = 8\n"
                                                   "MAX_GAP = 80\n" % (tmp_path / "g.fa", ", ".join(sams), tmp_path / name))
        r = _run_cli(extra + ["pipeline", str(tmp_path / ("config_" + name))], tmp_path, env={"MIRP_STREAM_CHUNKS": "2"})
        assert r.returncode == 0, r.stderr[-1200:]
    files = [f for f in os.listdir(tmp_path / "keep") if os.path.isfile(tmp_path / "keep" / f)]
    assert "xl_miRNA.gff3" in files and os.path.getsize(tmp_path / "keep" / "xl_miRNA.gff3") > 0
    match, mismatch, errors = filecmp.cmpfiles(str(tmp_path / "keep"), str(tmp_path / "lean"), files, shallow=False)
    assert not mismatch and not errors, (mismatch, errors)
    rm = sorted(os.listdir(tmp_path / "keep" / "readmapping"))
    assert rm and rm == sorted(os.listdir(tmp_path / "lean" / "readmapping"))
    match, mismatch, errors = filecmp.cmpfiles(str(tmp_path / "keep" / "readmapping"), str(tmp_path / "lean" / "readmapping"), rm, shallow=False)
    assert not mismatch and not errors
    fold = open(tmp_path / "keep" / "xl_tmp" / "xl_rnalfoldoutput_0").read().splitlines()
    seqs = [ln for ln in fold if ln and ln[0] in "ACGUTN"]
    assert seqs and max(len(s) for s in seqs) > 1900 and max(len(ln.split(" ")[0]) for ln in fold if ln and ln[0] in ".(") > 1500
