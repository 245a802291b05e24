"""GPU tests of the SAM ingest's device side: the stable (tid, pos) radix sort against the host stable sort on a
tie-heavy input, the keep-region filter against `samtools view -L` semantics, gapped alignments against the bundled samtools 0.1.18's depth
output, and the CLI with GFF_FILE_EXCLUDE taking the device path."""
import gzip
import os
import shutil

import numpy as np
import pytest

from mir_prefer_amd import capi, cli, gffmask, ingest, records, synth
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def _tie_heavy_records(n, n_contigs, span, seed):
    """Many records per (tid, pos): read stacks of isomiRs, depths / lengths / strands differing inside a tie."""
    rng = np.random.RandomState(seed)
    a = np.zeros(n, dtype=synth.ALN_DTYPE)
    a["tid"] = rng.randint(0, n_contigs, size=n)
    a["pos"] = rng.randint(1, span, size=n) // 37 * 37 + 1          # ~ span / 37 distinct positions per contig
    a["depth"] = rng.randint(1, 2000, size=n)
    a["len"] = rng.randint(18, 26, size=n)
    a["strand"] = rng.randint(0, 2, size=n)
    return a


def test_device_sort_equals_host_stable_sort_on_ties(gpu_ctx, tmp_path):
    """Three sample files, 2.4 M records, ~100 records per (tid, pos): the device-sorted array must equal the host stable sort record for
    record -- ties keep sample-then-file order, which decides the first-seen maximum of gen_loci_alignment_info (miR_PREFeR.py:1457)."""
    names, lens = ["c%d" % k for k in range(5)], [300000] * 5
    paths = []
    for s in range(3):
        a = _tie_heavy_records(800000, 5, 290000, seed=50 + s)
        p = str(tmp_path / ("S%d.sam" % (s + 1)))
        synth.write_sam_fast(p, "S%d" % (s + 1), a, names, lens)
        paths.append(p)
    host = capi.ingest_sams(paths)
    cn, cl, sn, alns, segs, sec = gpu_ctx.ingest_sams(paths)
    assert cn == host[0] == names and list(cl) == lens and sn == host[2] == ["S1", "S2", "S3"] and len(segs) == 0
    assert len(alns) == 2400000 and np.array_equal(alns, host[3])
    key = alns["tid"].astype(np.int64) << 32 | alns["pos"].astype(np.int64)
    assert (np.diff(key) >= 0).all() and len(np.unique(key)) < len(alns) / 20
    ties = np.flatnonzero(np.diff(key) == 0)
    assert (np.diff(alns["sample"].astype(np.int64))[ties] >= 0).all()          # inside a tie: sample order
    # the records are resident: the candidate stage runs on them without another upload
    genome = [(n, synth._BASES[np.random.RandomState(k).randint(0, 4, size=l, dtype=np.uint8)]) for k, (n, l) in enumerate(zip(names, lens))]
    gpu_ctx.load_genome(genome)
    npk, nloci, nwin = gpu_ctx.candidate(10, 100, 300, np.arange(5, dtype=np.int32))
    assert npk > 1000
    print("device ingest: %.1f M records/s (tokenize %.3f s, upload+filter %.3f s, sort %.3f s, download %.3f s)" % (
        len(alns) / sum(sec.values()) / 1e6, sec["tokenize_s"], sec["upload_filter_s"], sec["sort_s"], sec["download_s"]))


def test_device_keep_regions_match_samtools_view_L(gpu_ctx, tmp_path):
    """The keep-region filter on the device against the host mask, which is pinned to the bundled `samtools view -L` (tests/golden/gffmask.json.gz)."""
    names, lens = ["k1", "k2", "k3"], [200000, 150000, 90000]
    a = _tie_heavy_records(300000, 3, 88000, seed=7)
    p = str(tmp_path / "S1.sam")
    synth.write_sam_fast(p, "S1", a, names, lens)
    rng = np.random.RandomState(3)
    regions = []
    for _ in range(400):
        t = int(rng.randint(0, 3)); s = int(rng.randint(0, lens[t] - 10))
        regions.append((t, s, min(lens[t], s + int(rng.choice([1, 20, 55, 300, 2000])))))
    host = capi.ingest_sams([p])[3]
    want = host[gffmask.keep_mask(host, regions)]
    got = gpu_ctx.ingest_sams([p], regions=regions)[3]
    assert 0 < len(want) < len(host) and np.array_equal(got, want)
    # the golden of the real tool, through the device path
    for v in gu.load_json("gffmask.json.gz")["view_cases"]:
        cn = [c[0] for c in v["contigs"]]
        sam = tmp_path / "v.sam"
        lines = ["@SQ\tSN:%s\tLN:%d" % (c[0], c[1]) for c in v["contigs"]]
        lines += ["S_r%d_x5\t0\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s" % (i, c, pos, ln, "A" * ln, "I" * ln) for i, (c, pos, ln) in enumerate(v["records"])]
        sam.write_text("\n".join(lines) + "\n")
        regs = [(cn.index(c), s, e) for c, s, e in v["bed"]]
        got = gpu_ctx.ingest_sams([str(sam)], regions=regs)[3]
        kept_pos = sorted((v["records"][int(nm.split("_r")[1].split("_")[0])][0], v["records"][int(nm.split("_r")[1].split("_")[0])][1]) for nm in v["kept_names"])
        assert sorted((cn[r["tid"]], int(r["pos"])) for r in got) == kept_pos


def test_gapped_alignments_depth_on_device_matches_samtools(gpu_ctx, tmp_path):
    """CIGARs with I / D / N / S / H / = / X: the device's thresholded depth lines against the bundled samtools 0.1.18 (expanded, strand-split
    BAMs -> `samtools depth | awk`, tests/golden/tools/gen_gapped_golden.py)."""
    g = gu.load_json("gapped.json.gz")
    sam = tmp_path / "S1.sam"
    sam.write_text(g["sam"])
    names, lens, samples, alns, segs, _ = gpu_ctx.ingest_sams([str(sam)])
    assert names == [c[0] for c in g["contigs"]] and len(segs) > 50
    genome = [(n, synth._BASES[np.random.RandomState(k).randint(0, 4, size=int(l), dtype=np.uint8)]) for k, (n, l) in enumerate(g["contigs"])]
    gpu_ctx.load_genome(genome)
    gpu_ctx.candidate(g["cutoff"], 100, 300, np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32))
    assert records.depth_text(gpu_ctx.get_depth(), names) == g["depth_cut"]
    # the same through the explicit upload path (host ingest -> load_alignments + load_coverage_segments)
    h = ingest.read_sams([str(sam)], with_segments=True)
    assert np.array_equal(h[3], alns) and sorted(map(tuple, h[4].tolist())) == sorted(map(tuple, segs.tolist()))
    gpu_ctx.load_alignments(h[3])
    gpu_ctx.load_coverage_segments(h[4])
    gpu_ctx.candidate(g["cutoff"], 100, 300, np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32))
    assert records.depth_text(gpu_ctx.get_depth(), names) == g["depth_cut"]
    # the read bookkeeping keeps POS and len(SEQ) of a gapped read (samtools view fields 3 and 9, miR_PREFeR.py:1439-1457)
    rt = gpu_ctx.get_window_readtable()
    W = gpu_ctx.get_windows()["windows"]
    for k in range(len(W)):
        st = int(W[k]["strand"])
        for x in np.nonzero(rt[k, :, 1])[0]:
            pos = int(W[k]["ws"]) + int(x)
            here = alns[(alns["tid"] == W[k]["tid"]) & (alns["pos"] == pos) & (alns["strand"] == st)]
            assert len(here) and int(rt[k, x, 2]) == int(here["depth"].sum()) and int(rt[k, x, 0]) in set(int(v) for v in here["len"])


def _setup_gff(name, tmp_path, compressed):
    exp = gu.load_json(os.path.join(name, "expected.json.gz"))
    src = os.path.join(gu.GOLD, name)
    tmp_path.mkdir(parents=True, exist_ok=True)
    fa = tmp_path / "genome.fa"
    with gzip.open(os.path.join(src, "genome.fa.gz"), "rb") as fi, open(fa, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    sams = []
    for s in exp["sample_names"]:
        if compressed:
            dst = tmp_path / (s + ".sam.gz")
            shutil.copyfile(os.path.join(src, s + ".sam.gz"), dst)
        else:
            dst = tmp_path / (s + ".sam")
            with gzip.open(os.path.join(src, s + ".sam.gz"), "rb") as fi, open(dst, "wb") as fo:
                shutil.copyfileobj(fi, fo)
        sams.append(str(dst))
    names, lens = ingest.read_sam_header(sams[0])
    gff = tmp_path / "ex.gff"
    feats = []
    for m in gu.unjson(exp["result_raw"])[:6]:          # features over the first reported loci: their reads go, and the loci with them
        feats.append("%s\tsrc\tgene\t%d\t%d\t.\t+\t.\tID=f%d" % (m[0], max(1, m[1] - 150), m[2] + 150, len(feats)))
    gff.write_text("##gff-version 3\n" + "\n".join(feats) + "\n")
    c = exp["config"]
    lines = ["FASTA_FILE = " + str(fa), "ALIGNMENT_FILE = " + ", ".join(sams), "OUTFOLDER = " + str(tmp_path / "out"), "GFF_FILE_EXCLUDE = " + str(gff)]
    for k in ("PRECURSOR_LEN", "READS_DEPTH_CUTOFF", "MAX_GAP", "MIN_MATURE_LEN", "MAX_MATURE_LEN", "ALLOW_NO_STAR_EXPRESSION", "ALLOW_3NT_OVERHANG",
              "CHECKPOINT_SIZE", "NAME_PREFIX"):
        lines.append("%s = %s" % (k, c[k]))
    cfg = tmp_path / "config"
    cfg.write_text("\n".join(lines) + "\n")
    return exp, str(cfg), tmp_path / "out", str(gff), sams


def test_cli_with_gff_exclude_on_the_device_path(tmp_path):
    """`pipeline` with GFF_FILE_EXCLUDE (miR_PREFeR.py:543-652, 817-859): plain SAM files take the device path (tokenizer -> keep-region filter
    and sort on the GPU); the result must be the one of the host path (compressed inputs: Python parser + numpy mask, pinned to the reference's
    BED text and to `samtools view -L` by tests/test_host_cpu.py) and differ from the run without a GFF file."""
    exp, cfg_d, out_d, gff, sams_d = _setup_gff("mini", tmp_path / "dev", compressed=False)
    _, cfg_h, out_h, _, sams_h = _setup_gff("mini", tmp_path / "host", compressed=True)
    assert cli.main(["-k", "pipeline", cfg_d]) == 0
    assert cli.main(["-k", "pipeline", cfg_h]) == 0
    prefix = exp["config"]["NAME_PREFIX"]
    zd = np.load(out_d / (prefix + "_tmp") / "prepared.npz", allow_pickle=True)
    zh = np.load(out_h / (prefix + "_tmp") / "prepared.npz", allow_pickle=True)
    full = ingest.read_sams(sams_d)[3]
    assert np.array_equal(zd["alns"], zh["alns"]) and 0 < len(zd["alns"]) < len(full)
    names, lens = ingest.read_sam_header(sams_d[0])
    want = gffmask.apply_keep(full, names, gffmask.keep_regions_exclude(gff, dict(zip(names, lens)), 55))
    assert np.array_equal(zd["alns"], want)
    gd, gh = open(out_d / (prefix + "_miRNA.gff3")).read(), open(out_h / (prefix + "_miRNA.gff3")).read()
    assert gd == gh and gd != exp["gff3"] and len(gd) > 0
    for fn in (prefix + "_miRNA.detail.csv", prefix + "_miRNA.precursor.ss"):
        assert open(out_d / fn).read() == open(out_h / fn).read()
    dd = open(out_d / (prefix + "_tmp") / ("bam.depth.cut%d" % exp["config"]["READS_DEPTH_CUTOFF"])).read()
    assert dd == open(out_h / (prefix + "_tmp") / ("bam.depth.cut%d" % exp["config"]["READS_DEPTH_CUTOFF"])).read() and dd != exp["depth_cut"]


def test_keep_regions_on_gapped_alignments_match_samtools_view_L(gpu_ctx, tmp_path):
    """`samtools view -L` of the bundled 0.1.18 tests [POS - 1, bam_calend) -- the M / D / N span of a gapped alignment, not len(SEQ): a region inside
    an intron keeps the read, a region next to a soft-clipped read does not (golden: gen_gapped_golden.py, `view_L`).  Device filter
    (mask_keep_kernel) and host filter (gffmask.keep_mask) both keep exactly the reads the binary keeps, and the segments of the dropped ones go."""
    g = gu.load_json("gapped.json.gz")
    sam = tmp_path / "S1.sam"
    sam.write_text(g["sam"])
    names = [c[0] for c in g["contigs"]]
    regions = gffmask.regions_by_tid([tuple(x) for x in g["view_L"]["bed"]], names)
    ids = [l.split("\t")[0] for l in g["sam"].splitlines() if not l.startswith("@")]
    want = sorted(g["view_L"]["kept_ids"])
    assert 5 < len(want) < len(ids)

    def kept_ids(alns):
        # a record is identified by (contig, pos, depth, len, strand): map back to the read ids of the input
        key = {}
        for l in g["sam"].splitlines():
            if l.startswith("@"):
                continue
            f = l.split("\t")
            key.setdefault((names.index(f[2]), int(f[3]), int(f[0].rsplit("_x", 1)[1]), len(f[9]), 1 if int(f[1]) & 16 else 0), []).append(f[0])
        out = []
        for r in alns:
            out.append(key[(int(r["tid"]), int(r["pos"]), int(r["depth"]), int(r["len"]), int(r["strand"]))].pop(0))
        return sorted(out)
    _, _, _, a_dev, s_dev, _ = gpu_ctx.ingest_sams([str(sam)], regions=regions)
    _, _, _, a_host, s_host = ingest.read_sams([str(sam)], native=False, regions=regions, with_segments=True)
    assert kept_ids(a_dev) == want and kept_ids(a_host) == want
    assert np.array_equal(a_dev, a_host) and sorted(map(tuple, s_dev.tolist())) == sorted(map(tuple, s_host.tolist()))
    _, _, _, a_all, s_all, _ = gpu_ctx.ingest_sams([str(sam)])
    assert len(s_dev) < len(s_all)
