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


def test_fold_span_shorter_than_window(gpu_ctx, oracle):
    _compare(gpu_ctx, oracle, seqgen.windows(12, 100, 60, 200), 40)


def test_fold_production_windows(gpu_ctx, oracle):
    _compare(gpu_ctx, oracle, seqgen.windows(13, 64, 300, 350), 300)


def test_fold_edge_cases(gpu_ctx, oracle):
    seqs = ["A", "ACGU", "GGGGAAAACCCC", "A" * 24, "N" * 30, "GGGAAAUCCCGGGAAAUCCCAAAAGGGGGGAUUUCCCCCCUUUUGGGAUUUCCCGGAUUUCCC",
            "GC" * 150, "G" * 150 + "C" * 150, ""]
    _compare(gpu_ctx, oracle, seqs, 300)
