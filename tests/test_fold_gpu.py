"""GPU parity: HIP local fold (through the C-ABI) vs the CPU oracle, structure lines bit-exact."""
import pytest

from tests import seqgen

pytestmark = pytest.mark.gpu


def _compare(gpu_ctx, oracle, seqs, span):
    got = gpu_ctx.fold_batch(seqs, span)
    assert len(got) == len(seqs)
    for s, g in zip(seqs, got):
        want = oracle.lfold(s, span)
        assert g["status"] == 0, (s, g["status"])
        assert g["mfe"] == want["mfe"], s
        assert g["lines"] == want["lines"], s


def test_fold_small_windows(gpu_ctx, oracle):
    _compare(gpu_ctx, oracle, seqgen.windows(11, 200, 5, 120), 300)


def test_fold_every_window_length(gpu_ctx, oracle):
    """One window of every length 5..350: every row-block / tile-edge case of the tiled archive the fill kernel hands to the epilogue (8 x 8 tiles over
    (row, diagonal): partial last row block, partial last tile of a row block, blocks of 32 rows in the exterior sweep that end past the window)."""
    import random
    r = random.Random(2024)
    seqs = []
    for n in range(5, 351):
        w = seqgen.window(r, n, n)
        seqs.append(w[:n] if len(w) >= n else w + "A" * (n - len(w)))
    assert sorted(set(len(s) for s in seqs)) == list(range(5, 351))
    _compare(gpu_ctx, oracle, seqs, 300)
    _compare(gpu_ctx, oracle, seqs[60:], 64)


def test_fold_span_shorter_than_window(gpu_ctx, oracle):
    _compare(gpu_ctx, oracle, seqgen.windows(12, 100, 60, 200), 40)


def test_fold_production_windows(gpu_ctx, oracle):
    _compare(gpu_ctx, oracle, seqgen.windows(13, 64, 300, 350), 300)


def test_fold_maximal_asymmetric_interior_loop_tie(gpu_ctx, oracle):
    """Regression (found by profiles/tools/stress_fold.py): a 28 x 2 interior loop -- the last admissible n1 of the largest loop size --
    ties with a multiloop; RNALfold's backtrack takes the interior loop, so the fill kernel must rank that shape too."""
    s = ("GUNUCGUAGCACGGGNCUACCCUACACCACAUUCCNNUNAUCANCGUCUUAGUANNCCAGCANNGAAANGGCGNGUAUUCAAGGCCUGCCUGUGAAUUUGCAGUCGNGUUAGUNGGCUUGCUCUUNAAGAAAUACCGCAAUCGAGCU"
         "NCNGAUCAGGUANNUANGCUAAUCCCGCUGUACCNNUAUCNNAAANAGGUUCUGGACACAAACUCGCUAAAGUGUGAACUAUCNUGGUAACNGGAAAUUAACUUUUCGANAAAGUAUCUGCGCACCAGCGUGUAGCCGGGACAAUUAUGCC"
         "UGGUUUCUUGCACUAGACCU")
    _compare(gpu_ctx, oracle, [s], 150)
    _compare(gpu_ctx, oracle, [s], 300)


_FILL2_WORKER = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
from mir_prefer_amd import capi
seqs = json.load(open(sys.argv[2]))
ctx = capi.Context(0)
out = ctx.fold_batch(seqs, 300)
json.dump([[g["status"], g["mfe"], g["lines"]] for g in out], open(sys.argv[3], "w"))
"""


def test_two_diagonal_fill_schedule_equals_the_product_kernel(gpu_ctx, oracle, tmp_path):
    """fold_lds2_kernel.hip (two anti-diagonals per barrier interval: stacked pairs / 1-bulges finished in phase B, multiloop closings pushed two
    diagonals ahead, fML handed over by DPP) is a second, independently scheduled implementation of the fill.  Its build
    (libmirprefer_vfill2.so, make VARIANT=fill2 VFLAGS="-DMIRP_FILL2 -DMIRP_E1", built by __graft_entry__.build) runs in a child process through MIRP_LIB and must
    produce the product kernel's lines and the oracle's: GU-rich windows put hundreds of paired cells on a diagonal, short windows end on an odd
    number of diagonals, n = 350 fills the LDS layout."""
    import json
    import os
    import random
    import subprocess
    import sys
    from mir_prefer_amd import capi
    lib = os.path.join(os.path.dirname(capi.LIB_PATH), "libmirprefer_vfill2.so")
    if not os.path.exists(lib):
        pytest.fail("libmirprefer_vfill2.so is not built: run __graft_entry__.build()")
    r = random.Random(5)
    seqs = seqgen.windows(14, 24, 280, 350) + ["".join(r.choice("GU") for _ in range(330)), "".join(r.choice("GGGUUC") for _ in range(300))]
    seqs += seqgen.windows(15, 30, 5, 120) + ["".join(r.choice("ACGU") for _ in range(n)) for n in (348, 349, 350, 301, 300, 299)]
    product = gpu_ctx.fold_batch(seqs, 300)
    (tmp_path / "w.py").write_text(_FILL2_WORKER)
    (tmp_path / "in.json").write_text(json.dumps(seqs))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, str(tmp_path / "w.py"), root, str(tmp_path / "in.json"), str(tmp_path / "out.json")], env=dict(os.environ, MIRP_LIB=lib),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    other = json.load(open(tmp_path / "out.json"))
    for s, g, h in zip(seqs, product, other):
        want = oracle.lfold(s, 300)
        assert g["status"] == 0 and h[0] == 0
        assert g["lines"] == want["lines"] and [tuple(x) for x in h[2]] == want["lines"] and g["mfe"] == h[1] == want["mfe"], s


def test_fold_edge_cases(gpu_ctx, oracle):
    seqs = ["A", "ACGU", "GGGGAAAACCCC", "A" * 24, "N" * 30, "GGGAAAUCCCGGGAAAUCCCAAAAGGGGGGAUUUCCCCCCUUUUGGGAUUUCCCGGAUUUCCC",
            "GC" * 150, "G" * 150 + "C" * 150, ""]
    _compare(gpu_ctx, oracle, seqs, 300)


def _compare185(gpu_ctx, oracle, seqs, span, max_lines=96):
    gpu_ctx.set_fold_model("vienna-1.8.5")
    try:
        got = gpu_ctx.fold_batch(seqs, span, max_lines=max_lines)
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")
    for s, g in zip(seqs, got):
        want = oracle.lfold(s, span, model="vienna-1.8.5")
        assert g["status"] == 0, (s, g["status"])
        assert g["mfe"] == want["mfe"], s
        assert g["lines"] == want["lines"], s


def test_fold_vienna185_mode_matches_oracle(gpu_ctx, oracle):
    """Compatibility mode: the RNALfold 1.8.5 the reference bundles for Linux (Turner-1999, dangles 1, multi-component strings)."""
    _compare185(gpu_ctx, oracle, seqgen.windows(31, 150, 5, 120), 300)
    _compare185(gpu_ctx, oracle, seqgen.windows(32, 60, 60, 200), 40)
    _compare185(gpu_ctx, oracle, seqgen.windows(33, 40, 300, 350), 300)
    _compare185(gpu_ctx, oracle, ["A", "ACGU", "AAAAAAAAAA", "GGGAAAACCC", "N" * 30, "GC" * 100, "G" * 150 + "C" * 150, ""], 300)
    _compare185(gpu_ctx, oracle, ["GUGG" * 59], 300, max_lines=352)


def test_fold_vienna185_mode_matches_the_bundled_binary(gpu_ctx):
    from tests import golden_util as gu
    gold = gu.load_json("fold_rnalfold185.json.gz")
    gpu_ctx.set_fold_model("vienna-1.8.5")
    try:
        n = 0
        for case in gold["cases"]:
            got = gpu_ctx.fold_batch(case["seqs"], case["span"], max_lines=352)
            for g, exp, seq in zip(got, case["expected"], case["seqs"]):
                assert g["status"] == 0 and g["mfe"] == exp["mfe"], seq
                assert [list(l) for l in g["lines"]] == exp["lines"], seq
                n += 1
        assert n >= 330
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")


def test_fold_vienna212_matches_the_bundled_binary(gpu_ctx):
    """Default model against the golden of the REAL RNALfold 2.1.2 (dependency/Mac/osx-10.9/RNALfold-2.1.2 run through the dev Mach-O loader,
    tests/golden/tools/gen_golden.py): every structure line (text, energy, start column) and the MFE, all span groups, through the C-ABI."""
    from tests import golden_util as gu
    gold = gu.load_json("fold_rnalfold212.json.gz")
    gpu_ctx.set_fold_model("vienna-2.1.2")
    n, spans = 0, set()
    for case in gold["cases"]:
        got = gpu_ctx.fold_batch(case["seqs"], case["span"], max_lines=352)
        spans.add(case["span"])
        for g, exp, seq in zip(got, case["expected"], case["seqs"]):
            assert g["status"] == 0 and g["mfe"] == exp["mfe"], seq
            assert [list(l) for l in g["lines"]] == exp["lines"], seq
            n += 1
    assert n >= 300 and {300, 100, 40, 20} <= spans


def test_generic_kernels_match_the_real_binaries_beyond_300(gpu_ctx):
    """The generic kernels (span > 300, windows > 350 nt; fill with split candidates, lane = paired cell, trace-back codes; fill and epilogue as two kernels)
    against digests of the real RNALfold 2.1.2 / 1.8.5 output: 200 windows of 360 .. 480 nt at spans 400 and 330 (tests/golden/long_folds.json.gz)."""
    from tests.test_oracle_golden import check_long_folds, long_fold_fixture
    fix, seqs = long_fold_fixture()

    def fold_many(s, span, model):
        gpu_ctx.set_fold_model(model)
        out = gpu_ctx.fold_batch(s, span, max_lines=500)
        assert all(g["status"] == 0 for g in out)
        return [(g["lines"], g["mfe"]) for g in out]
    try:
        assert check_long_folds(fix, seqs, fold_many) == 800
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")


@pytest.mark.parametrize("model", ["vienna-2.1.2", "vienna-1.8.5"])
def test_generic_kernels_on_long_windows_with_sparse_pairs(gpu_ctx, oracle, model):
    """Windows of 600 .. 1,000 nt at spans of 500 .. 700 -- beyond the staging buffer of the generic kernels' interior-loop interval when the paired cells of a
    diagonal lie far apart (A-rich sequence with a few U: 64 paired cells then span more than 480 positions, and the block takes the load-per-candidate form) --
    and ordinary mixed sequence of the same lengths (two staged pieces per segment): structure lines and MFE equal the oracle's."""
    import random
    r = random.Random(20265)
    seqs = []
    for k in range(10):
        n = r.randint(600, 1000)
        if k % 2 == 0:      # sparse pairs: mostly A and C, a U or G here and there, one planted hairpin so that there is something to fold
            s = [r.choice("AAAC") if r.random() < 0.93 else r.choice("UG") for _ in range(n)]
            arm = "".join(r.choice("ACGU") for _ in range(25))
            rc = "".join({"A": "U", "C": "G", "G": "C", "U": "A"}[c] for c in reversed(arm))
            at = r.randint(50, n - 120)
            s[at:at + 25] = arm; s[at + 40:at + 65] = rc
            seqs.append("".join(s))
        else:
            seqs.append("".join(r.choice("ACGU") for _ in range(n)))
    gpu_ctx.set_fold_model(model)
    try:
        for span in (500, 700):
            got = gpu_ctx.fold_batch(seqs, span, max_lines=600)
            for s, g in zip(seqs, got):
                want = oracle.lfold(s.encode(), span, model=model)
                assert g["status"] == 0 and g["mfe"] == want["mfe"] and g["lines"] == want["lines"], (model, span, len(s))
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")


def test_fold_at_the_largest_precursor_length(gpu_ctx):
    """PRECURSOR_LEN = 3000, the reference's upper limit (MP:167-184): a window of 3,020 nt and one of 1,500 nt at span 3000 through the generic kernels (more
    than 64 KB of LDS per workgroup, 200 MB of workspace per window) against digests of the real RNALfold 2.1.2 / 1.8.5 output (tests/golden/xl_folds.json.gz)."""
    from tests.test_oracle_golden import check_long_folds, xl_fold_fixture
    fix, seqs = xl_fold_fixture()

    def fold_many(s, span, model):
        gpu_ctx.set_fold_model(model)
        out = gpu_ctx.fold_batch(s, span, max_lines=1200)
        assert all(g["status"] == 0 for g in out)
        return [(g["lines"], g["mfe"]) for g in out]
    try:
        assert check_long_folds(fix, seqs, fold_many) == 4
    finally:
        gpu_ctx.set_fold_model("vienna-2.1.2")
