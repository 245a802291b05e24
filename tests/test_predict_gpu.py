"""GPU parity: fold + predict-stage filter kernels (through the C-ABI) on the golden pipeline cases,
against the reference's own decisions and result list (tests/golden/*/expected.json.gz)."""
import numpy as np
import pytest

from mir_prefer_amd import records, synth
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["mini", "mini3"])
def test_fold_then_predict_matches_reference(name, gpu_ctx, oracle):
    c = gu.load_pipeline_case(name)
    cfg = c["exp"]["config"]
    # windows/matures come from the oracle here (the candidate kernels have their own parity tests)
    _, peaks = oracle.coverage_peaks(c["alns"], c["contig_lens"], cfg["READS_DEPTH_CUTOFF"])
    order = np.argsort(np.array(c["contig_names"], dtype=object), kind="stable").astype(np.int32)
    win = oracle.make_windows(peaks, c["alns"], c["contigs"], order, cfg["MAX_GAP"], cfg["PRECURSOR_LEN"], cfg["READS_DEPTH_CUTOFF"] * 0.5)
    W = win["windows"]
    seqs = [win["seq"][w["seq_off"]:w["seq_off"] + w["seq_len"]].tobytes() for w in W]
    raw = gpu_ctx.fold_batch_raw(seqs, cfg["PRECURSOR_LEN"])
    assert (raw["status"] == 0).all()
    params = (len(c["sample_names"]), cfg["MIN_MATURE_LEN"], cfg["MAX_MATURE_LEN"], 1 if cfg["ALLOW_3NT_OVERHANG"] == "Y" else 0,
              1 if cfg["ALLOW_NO_STAR_EXPRESSION"] == "Y" else 0, 55)
    mir, nm, st = gpu_ctx.predict_batch(W, win["matures"], c["alns"], raw, params)
    assert (st == 0).all()

    def rec(w, m):
        ss = raw["ss"][w, m["line"], m["ss_off"]:m["ss_off"] + m["ss_len"]].tobytes().decode()
        return [c["contig_names"][m["tid"]], int(m["fold_s"]), int(m["fold_e"]), int(m["mat_s"]), int(m["mat_e"]), int(m["star_s"]),
                int(m["star_e"]), ss, records.STRAND[m["strand"]], bool(m["has_star"])]

    # replay filter_next_loci's 0 / (L,R) pairing (miR_PREFeR.py:2373-2432) over the per-window results
    exp_dec = [d for p in c["exp"]["pieces"] for d in p["decisions"]]
    got_dec, result = [], []
    k = 0
    while k < len(W):
        if W[k]["tag"] == 0:
            got_dec.append(k); k += 1
        else:
            got_dec.append(k)
            if nm[k] == 0:
                got_dec.append(k + 1)
            k += 2
    assert len(got_dec) == len(exp_dec)
    for w, e in zip(got_dec, exp_dec):
        assert (nm[w] > 0) == e["pass"], w
        if e["pass"]:
            em = gu.unjson(e["mirnas"])
            assert nm[w] == len(em)
            for j in range(nm[w]):
                assert rec(w, mir[w, j]) == em[j][:10]
                assert mir[w, j]["total_depth_mature"] == em[j][10]["total_depth_mature"]
                assert mir[w, j]["total_depth_star"] == em[j][10]["total_depth_star"]
            result.append(rec(w, mir[w, 0]))
    exp_res = [e[:10] for e in gu.unjson(c["exp"]["result_raw"])]
    assert result == exp_res and len(result) > 5


def _batch(cases, max_len=352):
    """One window per case: a single RNALfold line (the case's dot-bracket) and a single candidate mature."""
    n = len(cases)
    W = np.zeros(n, dtype=records.WINDOW_DTYPE)
    M = np.zeros(n, dtype=records.MATURE_DTYPE)
    lines = np.zeros((n, 1), dtype=[("start", "<i4"), ("len", "<i4"), ("energy", "<i4"), ("printed", "<i4")])
    ss = np.zeros((n, 1, max_len), dtype=np.uint8)
    for k, c in enumerate(cases):
        st = 1 if c["strand"] == "-" else 0
        W[k] = (0, c["regionstart"], c["regionend"], st, c["regionstart"], c["regionend"], 0, 0, 0, 1, 0, k, 0, c["regionend"] - c["regionstart"], 0)
        M[k] = (c["mature"][0], c["mature"][1], st, 100)
        b = c["ss"].encode()
        lines[k, 0] = (c["foldstart"], len(b), c.get("energy_dcal", -1000), 1)
        ss[k, 0, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    raw = {"lines": lines, "ss": ss, "n_lines": np.ones(n, dtype=np.int32), "stride": max_len, "max_lines": 1}
    return W, M, raw


def _is_stem_loop(ss):          # MP:1602-1608 with minloopsize 3
    return ss.find(")") - ss.rfind("(") - 1 >= 3


def test_duplex_rules_on_device_match_direct_reference_calls(gpu_ctx):
    """predict_kernel's a8 / a9 against DIRECT calls of the reference's functions (tests/golden/struct_rules.json.gz), through
    the -d records of mirp_predict_batch_reasons: the get_maturestar_info code of every (mature, structure) pair -- each failure code of
    MP:1848-1999 at least 20 times, FAIL_STRUCTURE_TOO_MANY_BULGE_OR_LOOP and FAIL_STRUCTURE_MATCHED_BASES included -- the star / fold coordinates of
    the passing ones, and the structure list (pieces of filter_ss, MP:1685-1724) of 4,000 seeded lines."""
    from tests.test_oracle_golden import MS_CODE
    g = gu.load_json("struct_rules.json.gz")
    alns = np.zeros(1, dtype=synth.ALN_DTYPE)
    alns["tid"] = 0; alns["pos"] = 1; alns["len"] = 20; alns["depth"] = 1
    # ---- a9: cases whose string the predict stage hands to get_maturestar_info as it is: a line that is ONE structure by itself (`whole`: a
    # stem-loop, type 0, or a good bifurcation that filter_ss leaves in one piece, type 1)
    cases = [c for c in g["maturestar"] if c["whole"] is not None and "raises" not in c["result"]]
    assert sum(c["whole"] == 1 for c in cases) > 100 and all(_is_stem_loop(c["ss"]) == (c["whole"] == 0) for c in cases)
    W, M, raw = _batch(cases)
    params = (1, 1, 100, 0, 1, 55)          # every mature length counts
    mir, nm, st, rec = gpu_ctx.predict_batch_reasons(W, M, alns, raw, params)
    assert (st == 0).all()
    pair = {int(r[0]): r for r in rec if r[1] >= 0}
    perw = {int(r[0]): r for r in rec if r[1] < 0}
    assert len(perw) == len(cases)
    seen = {}
    for k, c in enumerate(cases):
        want = c["result"]["ok"]
        assert perw[k][2] == 1 and k in pair, (k, c["ss"])
        r = pair[k]
        assert (int(r[4]), int(r[5])) == (0, len(c["ss"]))
        if isinstance(want, str):
            assert int(r[6]) == MS_CODE[want], (c, int(r[6]))
            seen[want] = seen.get(want, 0) + 1
        else:
            assert int(r[6]) == 0, (c, int(r[6]))
            assert [int(r[10]), int(r[11]), int(r[8]), int(r[9])] == want[:4], c
            seen["OK"] = seen.get("OK", 0) + 1
    assert all(seen.get(code, 0) >= 20 for code in MS_CODE), seen
    assert seen["OK"] > 500
    # ---- a8: structure list of a line (stem-loop -> the line itself; else the pieces filter_ss keeps, stem-loops and good bifurcations)
    cases = []
    for c in g["structures"]:
        n = len(c["ss"])
        cases.append({"ss": c["ss"], "strand": "+", "foldstart": c["start"], "regionstart": 1000, "regionend": 1000 + c["start"] + n + 5,
                      "mature": [1000 + c["start"] - 1 + 3, 1000 + c["start"] - 1 + 24], "energy_dcal": int(round(c["energy"] * 100)), "extend": c["extend"]})
    W, M, raw = _batch(cases)
    mir, nm, st, rec = gpu_ctx.predict_batch_reasons(W, M, alns, raw, (1, 1, 100, 0, 1, 55))
    assert (st == 0).all()
    pairs, perw = {}, {}
    for r in rec:
        if r[1] < 0:
            perw[int(r[0])] = r
        else:
            pairs.setdefault(int(r[0]), []).append(r)
    n_split = 0
    for k, c in enumerate(cases):
        want = [(s, x) for _, s, x, _ in c["extend"]]
        assert int(perw[k][2]) == len(want), (c["ss"], int(perw[k][2]), want)
        got = sorted((int(r[2]), c["foldstart"] + int(r[4]), c["ss"][int(r[4]):int(r[4]) + int(r[5])]) for r in pairs.get(k, []))
        assert [(s, x) for _, s, x in got] == want, (c["ss"], got, want)
        n_split += len(want) > 1
    assert n_split > 100
