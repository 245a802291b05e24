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
