"""N > 1 path on CPU: world_size-2 gloo process group.  Each rank traces its contiguous shard (with the oracle here — the
GPU engine is exercised by tests marked gpu), the detector hit buffers are all-gathered with the same
`distributed.all_gather_hits` that bench.py uses over RCCL, and rank 0 checks that the concatenation equals the
un-sharded solve bit for bit (reference push! order is preserved by contiguous shards)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ragged, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bmo_amd as bmo
    import pyoracle
    from bmo_amd import distributed as bd
    from scenes import c2_bundle, c2_scene

    system, _ = c2_scene()
    full = c2_bundle(n)
    if ragged:  # make some rays miss everything so shards have different hit counts (and one shard may have none)
        full.planes[3:6, : n // 3] = np.array([[1.0], [0.0], [0.0]])
    lo, hi = bd.shard_bounds(n, rank, world)
    shard = bmo.RayBundle(full.kind, full.planes[:, lo:hi])
    scene = bmo.CompiledScene(system, full.lambdas)
    res = pyoracle.trace(scene, shard, 100, threads=2)
    gathered = []
    for slot in range(len(scene.detectors)):
        hits, counts = bd.all_gather_hits(torch.from_numpy(res.detector_hits(slot).copy()))
        gathered.append(hits.numpy())
        assert int(counts.sum()) == hits.shape[0]
    if rank == 0:
        ref = pyoracle.trace(scene, full, 100, threads=2)
        for slot in range(len(scene.detectors)):
            assert np.array_equal(gathered[slot], ref.detector_hits(slot)), slot
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_sharded_hits_equal_unsharded(tmp_path, ragged):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 96, ragged, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_bounds_cover():
    sys.path.insert(0, ROOT)
    from bmo_amd.distributed import shard_bounds

    for n in (0, 1, 7, 1 << 20):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _pd_worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bmo_amd as bmo
    import pyoracle
    from bmo_amd import distributed as bd
    from test_photodetector import pd_scene

    system, pd, full = pd_scene(n)
    lo, hi = bd.shard_bounds(n, rank, world)
    shard = bmo.RayBundle(full.kind, full.planes[:, lo:hi])
    scene = bmo.CompiledScene(system, full.lambdas)
    res, sol = pyoracle.trace(scene, shard, 20, threads=2, keep=True)
    f = np.zeros((len(pd.x), len(pd.y)), dtype=np.complex128)
    sol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, f)
    total = bd.all_reduce_field(torch.from_numpy(f.copy())).numpy()  # the exchange step of the Photodetector path (§8e)
    if rank == 0:
        ref, rsol = pyoracle.trace(scene, full, 20, threads=2, keep=True)
        fr = np.zeros_like(f)
        rsol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, fr)
        assert np.abs(total - fr).max() <= 1e-12 * np.abs(fr).max()
        assert np.abs(fr).max() > 0
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_photodetector_field_all_reduce(tmp_path):
    world = 2
    mp.spawn(_pd_worker, args=(world, _free_port(), 48, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def _lists_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmo_amd import distributed as bd

    # two detectors with different widths and ragged (one empty) per-rank counts
    a = torch.arange((3 if rank == 0 else 0) * 2, dtype=torch.float64).reshape(-1, 2) + 100 * rank
    b = torch.arange((2 + rank) * 9, dtype=torch.float64).reshape(-1, 9) + 1000 * rank
    pend = bd.all_gather_hit_lists([a, b])
    (ha, ca), (hb, cb) = [p.wait() for p in pend]
    assert ca.tolist() == [3, 0] and cb.tolist() == [2, 3]
    assert ha.shape == (3, 2) and hb.shape == (5, 9)
    assert torch.equal(hb[:2], torch.arange(18, dtype=torch.float64).reshape(2, 9)) and torch.equal(hb[2:], torch.arange(27, dtype=torch.float64).reshape(3, 9) + 1000)
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_hit_lists(tmp_path):
    mp.spawn(_lists_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _shared_width_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmo_amd import distributed as bd

    # three detectors, two of them with the same width (they share one payload collective), ragged counts incl. empty
    a = torch.arange((2 + rank) * 2, dtype=torch.float64).reshape(-1, 2) + 100 * rank
    b = torch.arange((0 if rank == 0 else 4) * 2, dtype=torch.float64).reshape(-1, 2) + 1000 * rank + 7
    c = torch.arange((1 + 2 * rank) * 9, dtype=torch.float64).reshape(-1, 9) + 5000 * rank
    pend = bd.all_gather_hit_lists([a, c, b])
    (ha, ca), (hc, cc), (hb, cb) = [p.wait() for p in pend]
    ref_a = torch.cat([torch.arange(4, dtype=torch.float64).reshape(-1, 2), torch.arange(6, dtype=torch.float64).reshape(-1, 2) + 100])
    ref_b = torch.arange(8, dtype=torch.float64).reshape(-1, 2) + 1007
    ref_c = torch.cat([torch.arange(9, dtype=torch.float64).reshape(-1, 9), torch.arange(27, dtype=torch.float64).reshape(-1, 9) + 5000])
    assert torch.equal(ha, ref_a) and ca.tolist() == [2, 3]
    assert torch.equal(hb, ref_b) and cb.tolist() == [0, 4]
    assert torch.equal(hc, ref_c) and cc.tolist() == [1, 3]
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_hit_lists_shared_width(tmp_path):
    mp.spawn(_shared_width_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _engine_worker(rank, world, port, n, out_dir):
    """The N > 1 path with the HIP engine itself: every rank shards with bmo_trace_device on the one GPU of the test box, packs the
    columns its detectors keep (bmo_result_copy_hit_columns into torch memory) and all-gathers them (gloo here, RCCL in bench.py)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bmo_amd as bmo
    from bmo_amd import distributed as bd
    from scenes import c5_bundle, c5_scene

    system, _ = c5_scene()
    full = c5_bundle(n)
    full.planes[3:6, : n // 4] = np.array([[1.0], [0.0], [0.0]])  # rank 0's first rays leave sideways: ragged hit counts
    lo, hi = bd.shard_bounds(n, rank, world)
    shard = bmo.RayBundle(full.kind, full.planes[:, lo:hi])
    scene = bmo.CompiledScene(system, full.lambdas)
    eng = bmo.Engine(scene, 0)
    res = eng.trace_device(eng.upload(shard), 100)
    payloads = []
    for slot in range(len(scene.detectors)):
        cnt = eng.result_device_hits(res, slot)[1]
        local = torch.empty((cnt, 2), dtype=torch.float64, device="cuda")
        eng.result_copy_hit_columns(res, slot, 2, local.data_ptr(), cnt)
        payloads.append(local.cpu())
    gathered = [p.wait()[0].numpy() for p in bd.all_gather_hit_lists(payloads)]
    eng.free_result(res)
    if rank == 0:
        import pyoracle

        whole = eng.trace(full, 100)
        ref = pyoracle.trace(scene, full, 100, threads=8)
        for slot in range(len(scene.detectors)):
            assert np.array_equal(gathered[slot], whole.detector_hits(slot)[:, :2]), slot
            assert np.array_equal(gathered[slot], ref.detector_hits(slot)[:, :2]), slot
        assert sum(len(g) for g in gathered) > 0
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_engine_sharded_hits_equal_unsharded(tmp_path):
    world = 2
    mp.spawn(_engine_worker, args=(world, _free_port(), 3072, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


@pytest.mark.gpu
def test_bench_distributed_path_over_rccl_with_one_rank(tmp_path):
    """bench.py's N > 1 code path — process group on the `nccl` backend (= RCCL), the exchange step with `all_gather_into_tensor` on
    device tensors left in flight across solves, the barrier / max-over-ranks timing, the aggregate all-reduce — run for real with a
    world of ONE rank under torchrun (two ranks cannot share the one GPU of the test box under RCCL; the two-rank logic is covered with
    gloo above).  Catches API misuse of the collectives on device tensors before the driver's multi-GPU run does."""
    import json
    import subprocess

    env = dict(os.environ, BMO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--rays", "65536", "--cpu-sample", "0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "RCCL all-gather" in line["config"]["parallelism"]
    assert line["config"]["name"] == "c5" and line["one_gpu_same_workload"]["value"] > 0


def _exchange_worker(rank, world, port, out_dir):
    """HitExchange over a loop of steps: counts ride in the payload, block size from the previous step, one step that outgrows it."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmo_amd import distributed as bd

    def payloads(step):  # two detectors (2 and 9 columns), ragged counts that depend on rank and step; step 3 outgrows the block
        na = 3 + rank + (step % 2) + (200 if step == 3 and rank == 1 else 0)
        nb = (0 if (rank + step) % 3 == 0 else 2 + step)
        a = torch.arange(na * 2, dtype=torch.float64).reshape(-1, 2) + 1000 * rank + 10 * step
        b = torch.arange(nb * 9, dtype=torch.float64).reshape(-1, 9) + 7000 * rank + 100 * step
        return [a, b]

    ex = bd.HitExchange([2, 9])
    pending = None
    for step in range(6):
        mine = payloads(step)
        nxt = ex.start(mine)
        if pending is not None:
            got, st = pending
            # what every rank must see: the concatenation in rank order of what each rank sent at that step
            sent = [_exchange_payloads(r, st) for r in range(world)]
            for d, (hits, counts) in enumerate(got.wait()):
                want = torch.cat([sent[r][d] for r in range(world)])
                assert torch.equal(hits, want), (st, d)
                assert counts.tolist() == [sent[r][d].shape[0] for r in range(world)]
        pending = (nxt, step)
    pending[0].wait()
    assert ex.rows is not None and ex.rows >= 200  # step 3 re-established the block size
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def _exchange_payloads(rank, step):
    na = 3 + rank + (step % 2) + (200 if step == 3 and rank == 1 else 0)
    nb = (0 if (rank + step) % 3 == 0 else 2 + step)
    a = torch.arange(na * 2, dtype=torch.float64).reshape(-1, 2) + 1000 * rank + 10 * step
    b = torch.arange(nb * 9, dtype=torch.float64).reshape(-1, 9) + 7000 * rank + 100 * step
    return [a, b]


def test_hit_exchange_without_per_step_sync(tmp_path):
    mp.spawn(_exchange_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
