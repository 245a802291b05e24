"""Contig sharding across ranks and the one exchange step of the path: the gather of the final loci list to rank 0
(the reference's analogue is `multiprocessing.Queue.put(list)` per piece, /root/reference/miR_PREFeR.py:2461-2499).
On MI355X the gather runs over RCCL inside libmirprefer.so (mirp_gather_loci / mirp_gather_records on the context's own communicator,
init_context below); `gather_records` over a `gloo` group is the CPU-side stand-in the tests use."""
import numpy as np


def init_host_group():
    """The CPU-side `gloo` process group of a multi-rank run (host objects, barriers).  gloo announces its connections on the process' stdout
    file descriptor; for the duration of the rendezvous that descriptor points at stderr, so stdout stays what the caller prints (the bench's
    one JSON line, the CLI's messages)."""
    import os
    import sys
    import torch.distributed as tdist
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    try:
        tdist.init_process_group("gloo")
        tdist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(keep, 1)
        os.close(keep)
    return tdist


def init_context(ctx, rank, world):
    """Gives `ctx` its RCCL communicator: rank 0 draws the ncclUniqueId (mirp_dist_unique_id) and the host's own process group -- any backend,
    `gloo` in the CLI and the bench -- carries the 128 bytes to the other ranks; every rank then joins with mirp_dist_init.  No torch tensor
    ever touches the GPU here: the process keeps the one HIP runtime and the one RCCL instance of the library."""
    import torch.distributed as tdist
    box = [ctx.dist_unique_id() if rank == 0 else None]
    if world > 1:
        tdist.broadcast_object_list(box, src=0)
    ctx.dist_init(box[0], rank, world)
    return ctx


def partition_contigs(contig_lens, world):
    """Longest-processing-time assignment of contigs to ranks; returns a list (per rank) of contig indices (ascending)."""
    loads = [0] * world
    parts = [[] for _ in range(world)]
    for t in sorted(range(len(contig_lens)), key=lambda t: (-int(contig_lens[t]), t)):
        r = min(range(world), key=lambda r: (loads[r], r))
        parts[r].append(t)
        loads[r] += int(contig_lens[t])
    return [sorted(p) for p in parts]


def gather_records(rec, device=None, dst=0):
    """rec: int32 array [k, w] of this rank (k may differ per rank). Returns the concatenation over ranks on `dst`
    (rank order), None elsewhere.  all_gather of the counts, then gather of max-padded blocks."""
    import torch
    import torch.distributed as dist
    rec = np.ascontiguousarray(rec, dtype=np.int32)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    w = rec.shape[1]
    cnt = torch.tensor([rec.shape[0]], device=dev, dtype=torch.int64)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    counts = [int(c.item()) for c in cnts]
    mx = max(max(counts), 1)
    pad = torch.zeros((mx, w), device=dev, dtype=torch.int32)
    if rec.shape[0]:
        pad[:rec.shape[0]] = torch.from_numpy(rec).to(dev)
    bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return np.concatenate([b[:c].cpu().numpy() for b, c in zip(bufs, counts)], axis=0)
