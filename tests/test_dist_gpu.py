"""GPU tests of the multi-GPU entry points of the C-ABI (include/mirprefer.h, "Multi-GPU"): the library's own RCCL communicator (first execution of
librccl through mirp_dist_init on the one GPU of the box), the gather of the loci list (mirp_gather_loci, the reference's result queue
MP:2461-2499) and the sharded SAM ingest (mirp_ingest_sams_shard, replaces the serial prepare_data MP:772-874) with several ranks on one GPU
over the local transport."""
import os
import subprocess
import sys

import numpy as np
import pytest

from mir_prefer_amd import capi, dist, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dataset(tmp_path, seed=5):
    ds = synth.make_dataset([90000, 40000, 70000, 20000, 55000], 90, n_samples=3, seed=seed, contig_names=["c3", "c1", "c5", "c2", "c4"], edge_cases=True)
    return ds, ds.write_sams(str(tmp_path))


def test_rccl_communicator_world1_gather_and_ingest(tmp_path):
    """ncclGetUniqueId / ncclCommInitRank / all-reduce / all-gather / grouped send-recv paths of the library at world = 1 on the GPU, and the
    loci gather straight from the device-resident result: equal to what mirp_predict returned."""
    ds, sams = _dataset(tmp_path)
    ctx = capi.Context(0)
    ctx.dist_init(ctx.dist_unique_id(), 0, 1)
    assert ctx.dist_world() == 1
    ctx.dist_barrier()
    assert ctx.dist_allreduce_sum([3, -4, 1 << 40]).tolist() == [3, -4, 1 << 40]
    rec = np.arange(35, dtype=np.int32).reshape(7, 5)
    assert (ctx.gather_records(rec) == rec).all()
    assert len(ctx.gather_records(rec[:0])) == 0
    # sharded ingest with one rank == plain device ingest
    names, lens, samples, alns, segs, _ = ctx.ingest_sams_shard(sams, np.zeros(len(ds.contig_names), dtype=np.int32))
    ref = capi.Context(0)
    n2, l2, s2, a2, g2, _ = ref.ingest_sams(sams)
    assert names == n2 and samples == s2 and (lens == l2).all() and alns.tobytes() == a2.tobytes() and segs.tobytes() == g2.tobytes()
    ref.close()
    ctx.load_genome(ds.contigs)
    order = np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)
    ctx.candidate(10, 100, 300, order)
    ctx.fold(300)
    out = ctx.predict(3, 18, 23, False, True)
    g = ctx.gather_loci(0)
    assert len(out["result"]) > 3
    assert g["result"].tobytes() == out["result"].tobytes() and g["ss"] == out["ss"]
    ctx.dist_finalize()
    ctx.close()


_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
from mir_prefer_amd import capi, dist, synth
rank, world, xdir, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
sams = sys.argv[6:]
ctx = capi.Context(0)
ctx.dist_init_local(xdir, rank, world)
names, lens = None, None
from mir_prefer_amd import ingest
names, lens = ingest.read_sam_header(sams[0])
owner = np.zeros(len(names), dtype=np.int32)
for r, part in enumerate(dist.partition_contigs(lens, world)):
    owner[part] = r
regions = json.load(open(os.path.join(out, "regions.json")))
try:
    cn, cl, sn, alns, segs, sec = ctx.ingest_sams_shard(sams, owner, regions=regions or None)
except ValueError as e:
    open(os.path.join(out, "failed_%d.txt" % rank), "w").write(str(e))
    ctx.close()
    sys.exit(3)
np.savez(os.path.join(out, "shard_%d.npz" % rank), alns=alns, segs=segs, owner=owner)
tot = ctx.dist_allreduce_sum([len(alns), rank])
g = ctx.gather_records(np.full((rank + 2, 3), rank, dtype=np.int32), dst=world - 1)
if rank == world - 1:
    np.save(os.path.join(out, "gathered.npy"), g)
np.save(os.path.join(out, "tot_%d.npy" % rank), tot)
ctx.dist_barrier()
ctx.close()
"""


@pytest.mark.parametrize("world,with_regions", [(2, False), (3, True), (4, False)])
def test_sharded_ingest_equals_single_rank_ingest(world, with_regions, tmp_path):
    """Every rank tokenizes its own byte range of every SAM file; after the all-to-all each rank holds exactly the single-process ingest's
    records of its contigs, in the same order (ties included: the stable sort sees file-then-offset order), with gapped reads' coverage
    segments following their records -- with and without GFF keep regions."""
    rng = np.random.default_rng(3)
    ds, sams = _dataset(tmp_path, seed=11)
    # a few gapped alignments, so that segments and their owner indices travel too
    extra = tmp_path / "s_gapped.sam"
    lines = open(sams[0]).read().splitlines()
    head = [l for l in lines if l.startswith("@")]
    body = [l for l in lines if not l.startswith("@")]
    sname = body[0].split("\t")[0].rsplit("_", 2)[0]
    gl = []
    for k in range(200):
        t = int(rng.integers(0, len(ds.contig_names)))
        pos = int(rng.integers(1, ds.contig_lens[t] - 200))
        gl.append("%s_r%d_x%d\t%d\t%s\t%d\t255\t10M%dN11M\t*\t0\t0\t%s\t*" % (sname, 900000 + k, int(rng.integers(1, 30)), 16 if k % 3 == 0 else 0,
                                                                             ds.contig_names[t], pos, int(rng.integers(1, 90)), "A" * 21))
    mixed = body + gl
    perm = rng.permutation(len(mixed))
    extra.write_text("\n".join(head + [mixed[k] for k in perm]) + "\n")
    sams = [str(extra)] + sams[1:]
    regions = []
    if with_regions:
        regions = [[0, 1000, 30000], [0, 50000, 51000], [2, 0, 70000], [4, 100, 25000], [3, 5000, 5100]]
    import json
    (tmp_path / "regions.json").write_text(json.dumps(regions))
    xdir = tmp_path / "x"
    xdir.mkdir()
    wk = tmp_path / "worker.py"
    wk.write_text(_WORKER)
    procs = [subprocess.Popen([sys.executable, str(wk), ROOT, str(r), str(world), str(xdir), str(tmp_path)] + sams, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    ctx = capi.Context(0)
    _, lens, _, alns, segs, _ = ctx.ingest_sams(sams, regions=regions or None)
    ctx.close()
    assert len(segs) > 100
    parts = dist.partition_contigs(lens, world)
    seen = 0
    for r in range(world):
        z = np.load(tmp_path / ("shard_%d.npz" % r))
        mine = np.isin(alns["tid"], parts[r])
        assert z["alns"].tobytes() == alns[mine].tobytes(), r
        want_segs = segs[np.isin(segs["tid"], parts[r])]
        key = lambda a: sorted(a.tolist())        # segments are an unordered side array
        assert key(z["segs"]) == key(want_segs), r
        seen += len(z["alns"])
        assert np.load(tmp_path / ("tot_%d.npy" % r)).tolist() == [len(alns), world * (world - 1) // 2]
    assert seen == len(alns)
    g = np.load(tmp_path / "gathered.npy")
    assert g.tolist() == [[r] * 3 for r in range(world) for _ in range(r + 2)]


def test_sharded_ingest_fails_on_every_rank_together(tmp_path):
    """A malformed line in ONE rank's byte range: every rank returns the error (agreed before the exchange) instead of the others waiting in it."""
    ds, sams = _dataset(tmp_path, seed=21)
    lines = open(sams[0]).read().splitlines()
    k = len(lines) * 3 // 4          # well inside the second rank's half
    f = lines[k].split("\t")
    f[3] = "notanumber"
    lines[k] = "\t".join(f)
    open(sams[0], "w").write("\n".join(lines) + "\n")
    import json
    (tmp_path / "regions.json").write_text(json.dumps([]))
    xdir = tmp_path / "x"
    xdir.mkdir()
    wk = tmp_path / "worker.py"
    wk.write_text(_WORKER)
    procs = [subprocess.Popen([sys.executable, str(wk), ROOT, str(r), "2", str(xdir), str(tmp_path)] + sams, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert [p.returncode for p in procs] == [3, 3], logs
    msgs = [open(tmp_path / ("failed_%d.txt" % r)).read() for r in range(2)]
    assert "SAM position is not a number" in msgs[1] and "rank 1 failed" in msgs[0]


def test_exported_window_payload_through_the_batch_path_equals_the_resident_path(gpu_ctx):
    """balance.py's unit of work on ONE rank: a range of windows packed into a payload, folded and filtered through mirp_fold_batch /
    mirp_predict_batch, must give the records the resident path (mirp_fold / mirp_predict) reports for those windows; and the rest of the list,
    after mirp_limit_windows, the records of the rest."""
    from mir_prefer_amd import balance
    from tests import golden_util as gu
    case = gu.load_pipeline_case("mini")
    names = case["contig_names"]
    order = np.argsort(np.array(names, dtype=object), kind="stable").astype(np.int32)
    gpu_ctx.load_genome(case["contigs"])
    gpu_ctx.load_alignments(case["alns"])
    cfg = case["exp"]["config"]
    _, _, nwin = gpu_ctx.candidate(cfg["READS_DEPTH_CUTOFF"], cfg["MAX_GAP"], cfg["PRECURSOR_LEN"], order)
    gpu_ctx.fold(cfg["PRECURSOR_LEN"])
    ns = len(case["sample_names"])
    a3, ans = cfg["ALLOW_3NT_OVERHANG"] == "Y", cfg["ALLOW_NO_STAR_EXPRESSION"] == "Y"
    full = gpu_ctx.predict(ns, cfg["MIN_MATURE_LEN"], cfg["MAX_MATURE_LEN"], a3, ans)
    win = gpu_ctx.get_windows()
    keep, ranges = balance.export_ranges([(0, 1, nwin // 2)], 0, nwin, win["windows"]["tag"])
    (dst, a, b), = ranges
    assert a == keep and b == nwin and 0 < keep < nwin
    p = balance.unpack(balance.pack(win, case["alns"], a, b, 0))
    balance.fold_imported(gpu_ctx, p, cfg["PRECURSOR_LEN"])
    params = (ns, cfg["MIN_MATURE_LEN"], cfg["MAX_MATURE_LEN"], 1 if a3 else 0, 1 if ans else 0, 55)
    got = balance.predict_imported(gpu_ctx, p, params)
    tail = full["result"]["window"] >= keep
    cols = [f for f in full["result"].dtype.names if f not in ("window",)]
    assert len(got["result"]) == tail.sum() > 3
    for f in cols:
        assert np.array_equal(got["result"][f], full["result"][tail][f]), f
    assert got["ss"] == [s for s, t in zip(full["ss"], tail) if t]
    # the head of the list after the cut; a batch fold between the resident fold and the resident filter (what a helper rank does) must leave
    # the resident fold output alone
    gpu_ctx.limit_windows(keep)
    gpu_ctx.fold(cfg["PRECURSOR_LEN"])
    balance.fold_imported(gpu_ctx, p, cfg["PRECURSOR_LEN"])
    head = gpu_ctx.predict(ns, cfg["MIN_MATURE_LEN"], cfg["MAX_MATURE_LEN"], a3, ans)
    assert head["result"].tobytes() == full["result"][~tail].tobytes() and head["ss"] == [s for s, t in zip(full["ss"], tail) if not t]


def _bench_line(args, env=None, timeout=900):
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_gpus_n_on_one_gpu(tmp_path):
    """`python bench.py --gpus N` end to end, the way the driver's scaling runs start it (the relaunch under torch.distributed.run, the gloo host
    group, the library's exchanges -- here over the local transport because RCCL refuses two ranks on one device): a scaled-down config[2] (5 contigs,
    3 samples, sharded by contig, windows re-balanced) at N = 1, 2, 3 must report the same windows and the same number of miRNA loci, and a JSON
    line whose `ranks` block accounts for every window."""
    common = ["--workload", "config2", "--genome-scale", "0.04", "--steps", "2", "--warmup", "1", "--no-configs", "--no-e2e", "--no-cpu-baseline", "--no-ingest"]
    one = _bench_line(["--gpus", "1"] + common)
    assert one["n_gpus"] == 1 and one["config"]["loci_found"] > 50 and "SCALED" in one["config"]["workload"]
    for n in (2, 3):
        share = tmp_path / ("share%d" % n)
        share.mkdir()
        line = _bench_line(["--gpus", str(n)] + common, env={"MIRP_BENCH_SHARE_GPU": str(share)})
        assert line["n_gpus"] == n and line["scaling"] == "strong" and line["steps"] == 2
        assert line["config"]["windows_total"] == one["config"]["windows_total"]
        assert line["config"]["loci_found"] == one["config"]["loci_found"]
        rk = line["ranks"]
        assert len(rk["windows"]) == n and sum(rk["windows"]) == one["config"]["windows_total"]
        assert sum(rk["windows_shipped"]) == sum(rk["windows_received"])
        assert sum(rk["contigs"]) == 5 and len(set(rk["pid"])) == n
        assert rk["rccl_comm_count"] == [-1] * n and rk["context_device"] == [0] * n          # local transport: no RCCL communicator on a shared GPU
        assert line["value"] > 0 and abs(line["ms_per_step"] * line["value"] / 1e3 - line["config"]["windows_total"]) < 1e-6 * line["config"]["windows_total"] + 1


def test_lost_rank_ends_the_exchange_with_an_error(tmp_path):
    """A rank that never joins: the waiting rank's exchange returns an error after MIRP_DIST_TIMEOUT_S instead of hanging (the reference's parent waits
    for ever on a crashed child), and every later exchange on the context fails at once."""
    code = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from mir_prefer_amd import capi
ctx = capi.Context(0)
ctx.dist_init_local(sys.argv[2], 0, 2)          # rank 1 never starts
t = time.time()
try:
    ctx.gather_records(np.arange(8, dtype=np.int32).reshape(2, 4))
    print("NOERROR")
except capi.MirpError as e:
    print("ERROR1 %.1f %s" % (time.time() - t, e))
t = time.time()
try:
    ctx.dist_barrier()
    print("NOERROR")
except capi.MirpError as e:
    print("ERROR2 %.1f %s" % (time.time() - t, e))
"""
    r = subprocess.run([sys.executable, "-c", code, ROOT, str(tmp_path)], env=dict(os.environ, MIRP_DIST_TIMEOUT_S="2"), capture_output=True, text=True, timeout=120)
    out = r.stdout.splitlines()
    assert r.returncode == 0 and len(out) == 2, (r.stdout, r.stderr[-1500:])
    assert out[0].startswith("ERROR1 ") and 1.5 < float(out[0].split()[1]) < 10 and "timed out waiting for rank 1" in out[0]
    assert out[1].startswith("ERROR2 0.0") and "aborted by an earlier failure" in out[1]


def test_bench_eight_ranks_config3_on_one_gpu(tmp_path):
    """BASELINE config[3]'s shape as the driver's 8-GPU run deals it -- 12 MSU7-sized contigs over 8 ranks, longest first, windows re-balanced before the fold, loci
    gathered on rank 0 -- with all 8 ranks on this one GPU over the local transport, at 5 % of the size: the same windows and the same loci as the 1-rank run.
    (The full-size run was done by hand in round 5: 218,293 windows, 50,074 loci on 8 ranks and on 1, profiles/r5_config3_8_ranks_one_gpu.json.)"""
    common = ["--workload", "config3", "--genome-scale", "0.05", "--steps", "1", "--warmup", "1", "--no-configs", "--no-e2e", "--no-cpu-baseline", "--no-ingest"]
    one = _bench_line(["--gpus", "1"] + common)
    share = tmp_path / "share8"
    share.mkdir()
    line = _bench_line(["--gpus", "8"] + common, env={"MIRP_BENCH_SHARE_GPU": str(share)}, timeout=1500)
    assert line["n_gpus"] == 8 and line["config"]["windows_total"] == one["config"]["windows_total"] > 5000
    assert line["config"]["loci_found"] == one["config"]["loci_found"] > 1000
    rk = line["ranks"]
    assert sum(rk["contigs"]) == 12 and sum(rk["windows"]) == one["config"]["windows_total"] and sum(rk["windows_shipped"]) == sum(rk["windows_received"]) > 0
    assert max(rk["windows"]) <= 1.12 * (sum(rk["windows"]) / 8.0)          # re-balanced to within the plan's 10 % band


def test_rccl_two_rank_rendezvous_on_one_gpu_is_refused_on_both_ranks(tmp_path):
    """The one part of an N > 1 RCCL start this one-GPU box can execute for real: two processes draw / read the same ncclUniqueId and call
    mirp_dist_init -- RCCL's socket bootstrap between the two ranks runs (interface discovery, the all-gather of the peers' device info), then RCCL
    itself refuses the communicator because both ranks sit on device 0.  Both ranks must come back with that error (no hang, no half-made
    communicator), and the context must still work as a 1-rank context afterwards."""
    code = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from mir_prefer_amd import capi
share, rank = sys.argv[2], int(sys.argv[3])
ctx = capi.Context(0)
idf = os.path.join(share, "id")
if rank == 0:
    uid = ctx.dist_unique_id()
    with open(idf + ".tmp", "wb") as f:
        f.write(bytes(uid))
    os.rename(idf + ".tmp", idf)
else:
    t_end = time.time() + 60
    while not os.path.exists(idf):
        assert time.time() < t_end
        time.sleep(0.01)
    uid = open(idf, "rb").read()
t = time.time()
try:
    ctx.dist_init(uid, rank, 2)
    print("JOINED world=%d" % ctx.dist_world())
except capi.MirpError as e:
    print("REFUSED %.1f %s" % (time.time() - t, str(e).replace("\n", " ")))
print("WORLD %d INFO %s" % (ctx.dist_world(), ctx.dist_comm_info()))
out = ctx.gather_records(np.arange(8, dtype=np.int32).reshape(2, 4))          # a 1-rank context again
print("GATHER %s" % out.ravel().tolist())
"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", code, ROOT, str(tmp_path), str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the two-rank RCCL rendezvous did not come back")
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        lines = [ln for ln in o.splitlines() if ln.split(" ")[0] in ("REFUSED", "JOINED", "WORLD", "GATHER")]          # RCCL prints its own warning on stdout
        assert rc == 0 and len(lines) == 3, (rc, o[-3000:], e[-2000:])
        assert lines[0].startswith("REFUSED ") and "CommInitRank" in lines[0], (lines[0], e[-1500:])
        assert lines[1].startswith("WORLD 1 INFO") and "-1" in lines[1], lines[1]
        assert lines[2] == "GATHER [0, 1, 2, 3, 4, 5, 6, 7]"


def test_bench_agrees_on_a_refused_rccl_communicator():
    """`bench.py --gpus 2` with both ranks on device 0 and RCCL asked for (MIRP_BENCH_ONE_DEVICE): RCCL refuses the communicator on both ranks; the ranks
    agree on that over the host group and the run ends with exit code 3 and a message -- no line -- unless --allow-gloo asks for the loci lists to go
    over the gloo host group, in which case the line says so and counts the same loci as the 1-rank run."""
    import json
    common = ["--workload", "config2", "--genome-scale", "0.04", "--steps", "1", "--warmup", "1", "--no-configs", "--no-e2e", "--no-cpu-baseline", "--no-ingest"]
    env = dict(os.environ, MIRP_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "RCCL communicator not available" in r.stderr and "--allow-gloo" in r.stderr, (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    one = _bench_line(["--gpus", "1"] + common)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--allow-gloo"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["loci_found"] == one["config"]["loci_found"] and line["config"]["windows_total"] == one["config"]["windows_total"]
    assert "gloo" in json.dumps(line["config"]) + json.dumps(line.get("ranks", {}))
