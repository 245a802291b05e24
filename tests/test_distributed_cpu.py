"""world_size-2 gloo test of the N>1 path: contig partition + gather of the loci records to rank 0."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mir_prefer_amd import dist as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parts = mdist.partition_contigs([30, 20, 25, 10, 18], world)
    mine = parts[rank]
    # each rank's "loci records": 16 int32 per record, 3 records per owned contig, tagged with contig and rank
    rec = np.array([[t, rank, k] + [0] * 13 for t in mine for k in range(3)], dtype=np.int32).reshape(-1, 16)
    if rank == 1:
        rec = rec[:-1]   # ragged counts
    out = mdist.gather_records(rec)
    q.put((rank, parts, None if out is None else out.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_and_gather_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, parts, out = q.get(timeout=120)
        res[rank] = (parts, out)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = res[0][0]
    assert sorted(t for p in parts for t in p) == [0, 1, 2, 3, 4] and res[1][0] == parts
    assert res[1][1] is None
    got = res[0][1]
    want = [[t, 0, k] for t in parts[0] for k in range(3)] + [[t, 1, k] for t in parts[1] for k in range(3)][:-1]
    assert [g[:3] for g in got] == want


# ---- window-level re-balancing (mir-prefer_amd/balance.py): plan, cuts and payloads are host logic, checked without a GPU
def test_balance_plan_moves_surplus_to_deficit_and_leaves_even_loads_alone():
    sys.path.insert(0, ROOT)
    from mir_prefer_amd import balance
    assert balance.plan([100, 100, 100]) == [] and balance.plan([5000]) == [] and balance.plan([0, 0]) == []
    assert balance.plan([1000, 1050, 1080, 990]) == []                       # within 1.1 x the mean
    for counts in ([5700, 4600, 4600, 4600, 4600, 4600, 4600, 4000], [20000, 15000, 13000, 12000, 11000, 0, 0, 0], [300, 100, 50], [17910, 26964, 25370]):
        moves = balance.plan(counts)
        assert moves
        after = list(counts)
        for s, d, m in moves:
            assert s != d and m > 0
            after[s] -= m; after[d] += m
        assert sum(after) == sum(counts) and min(after) >= 0
        target = -(-sum(counts) // len(counts))
        assert max(after) <= target + 16 * len(counts), (counts, after)
        assert max(after) < max(counts)
        assert all(after[s] >= target for s in set(s for s, _, _ in moves))      # a donor never drops below the target


def test_balance_cuts_respect_pairs_and_payload_round_trip():
    sys.path.insert(0, ROOT)
    import numpy as np
    from mir_prefer_amd import balance, synth
    from tests import oracle_binding
    o = oracle_binding.load()
    ds = synth.make_dataset([120000, 80000], 160, n_samples=2, seed=11, contig_names=["cB", "cA"], edge_cases=True)
    alns = ds.sorted_alns()
    _, peaks = o.coverage_peaks(alns, ds.contig_lens, 10)
    win = o.make_windows(peaks, alns, ds.contigs, np.array([1, 0], dtype=np.int32), 100, 300, 5.0)
    W = win["windows"]
    n = len(W)
    assert n > 100 and (W["tag"] != 0).any()
    units = balance.parse_units(W["tag"])
    assert units[0] == 0 and units[-1] == n and (np.diff(units) >= 1).all() and (np.diff(units) <= 2).all()
    for u in range(len(units) - 1):          # a unit is a tag-0 window or two tagged ones
        assert (units[u + 1] - units[u] == 1) == (W["tag"][units[u]] == 0) or units[u + 1] == n
    keep, ranges = balance.export_ranges([(0, 1, n // 3), (0, 2, n // 4)], 0, n, W["tag"])
    assert keep in units and [r[0] for r in ranges] == [1, 2]
    assert ranges[0][1] == keep and ranges[0][2] == ranges[1][1] and ranges[1][2] == n and all(r[1] in units and r[2] in units for r in ranges)
    for dst, a, b in ranges:
        p = balance.unpack(balance.pack(win, alns, a, b, 0))
        assert p["meta"].tolist() == [0, a, b] and len(p["windows"]) == b - a
        for k in range(b - a):
            w, g = W[a + k], p["windows"][k]
            for f in ("tid", "ws", "we", "strand", "loc_s", "loc_e", "tag", "n_peaks", "n_matures", "seq_len"):
                assert w[f] == g[f]
            assert np.array_equal(p["matures"][g["mature_off"]:g["mature_off"] + g["n_matures"]], win["matures"][w["mature_off"]:w["mature_off"] + w["n_matures"]])
            assert np.array_equal(p["wpeaks"][g["peak_off"]:g["peak_off"] + g["n_peaks"]], win["wpeaks"][w["peak_off"]:w["peak_off"] + w["n_peaks"]])
            assert p["seq"][g["seq_off"]:g["seq_off"] + g["seq_len"]].tobytes() == win["seq"][w["seq_off"]:w["seq_off"] + w["seq_len"]].tobytes()
            # every record the window can see is in the slice, in the order of the sorted array
            sel = alns[(alns["tid"] == w["tid"]) & (alns["pos"] >= w["ws"]) & (alns["pos"] <= w["we"])]
            got = p["alns"][(p["alns"]["tid"] == w["tid"]) & (p["alns"]["pos"] >= w["ws"]) & (p["alns"]["pos"] <= w["we"])]
            assert sel.tobytes() == got.tobytes()
        key = p["alns"]["tid"].astype(np.int64) << 32 | p["alns"]["pos"].astype(np.int64)
        assert (np.diff(key) >= 0).all()


def _balance_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from mir_prefer_amd import balance, records, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 0 holds 400 single windows, rank 1 holds 40: the exchange moves windows from 0 to 1 as host objects
    n = 400 if rank == 0 else 40
    W = np.zeros(n, dtype=records.WINDOW_DTYPE)
    W["tid"] = rank; W["ws"] = 1000 * np.arange(n) + 1; W["we"] = W["ws"] + 300; W["seq_len"] = 300; W["seq_off"] = 304 * np.arange(n)
    W["n_matures"] = 1; W["mature_off"] = np.arange(n); W["n_peaks"] = 1; W["peak_off"] = np.arange(n)
    win = {"windows": W, "wpeaks": np.zeros(n, dtype=records.PEAK_DTYPE), "matures": np.zeros(n, dtype=records.MATURE_DTYPE),
           "seq": np.full(304 * n, 65 + rank, dtype=np.uint8)}
    alns = np.zeros(n, dtype=synth.ALN_DTYPE)
    alns["tid"] = rank; alns["pos"] = W["ws"] + 5; alns["len"] = 21; alns["depth"] = 7

    def xchg(blocks):
        box = [None] * world
        dist.all_gather_object(box, blocks)
        return [b[rank] for b in box]
    counts = [None] * world
    dist.all_gather_object(counts, n)
    keep, imported, moves = balance.exchange(xchg, rank, world, lambda: win, alns, counts)
    q.put((rank, keep, [(int(p["meta"][0]), len(p["windows"]), len(p["alns"]), bytes(p["seq"][:2])) for p in imported], moves))
    dist.barrier()
    dist.destroy_process_group()


def test_balance_exchange_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_balance_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, keep, imp, moves = q.get(timeout=120)
        res[rank] = (keep, imp, moves)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] == res[1][2] == [(0, 1, 180)]
    assert res[0][0] == 220 and res[0][1] == []
    assert res[1][0] == 40 and res[1][1] == [(0, 180, 180, b"AA")]
