#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on the HIP engine: ray-surface intersections/s (and rays/s).

A "step" = one full pass of the hot path over one batch: solve_system!(system, bundle) for 2^20 fresh
geometric Rays through the 10-element mesh+SDF miniscope scene with one beamsplitter (BASELINE config
"1M Rays, single MI355X, 10-element mesh+SDF system with one beamsplitter").  The bundle is uploaded
once; the timed region runs bmo_trace_device K times (all bounce-step kernels, child spawning, node
ordering, detector-hit compaction; results stay in HBM) and, for N > 1, the RCCL all-gather of the
per-GPU detector hit buffers.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R]

N > 1 is launched by torch.distributed.run with one rank per GPU (weak scaling: every rank traces its own
contiguous shard of R rays of one global Fibonacci bundle).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_BOUNCE = 184  # 64+8 read, 40+64+8 written per bounce of a geometric Ray
BYTES_PER_HIT = 16  # Spotdetector record (Point2{Float64})


def cpu_baseline(scene, bundle_fn, sample, r_max):
    """Reference-algorithm CPU restatement (oracle, kind 'port') on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle

    threads = min(os.cpu_count() or 1, 32)
    b = bundle_fn(sample)  # strided subsample of the benchmark bundle
    t = time.perf_counter()
    ref = pyoracle.trace(scene, b, r_max, threads=threads)
    dt = time.perf_counter() - t
    return {"value": ref.n_intersect_calls / dt, "unit": "intersections/s", "cores": threads, "kind": "port",
            "sample": f"every (N/{sample})-th ray of the same bundle, same scene ({ref.n_intersect_calls} reference intersect3d calls, {dt:.1f} s wall, "
                      f"{threads} threads, plain parallel-for over rays)",
            "rays_per_s": sample / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays", type=int, default=1 << 20, help="root rays per GPU")
    ap.add_argument("--r-max", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=16384, help="rays for the CPU baseline (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist

        backend = os.environ.get("BMO_BENCH_BACKEND", "nccl")  # "gloo": rehearse the N > 1 path on a box with fewer GPUs
        ndev = max(torch.cuda.device_count(), 1)
        device_ord = local_rank % ndev
        torch.cuda.set_device(device_ord)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_ord))
        else:
            dist.init_process_group(backend)

    import bmo_amd as bmo
    from scenes import c2_bundle, c2_scene

    if world > 1:
        from bmo_amd import distributed as bd

    system, _ = c2_scene()
    n_local = args.rays
    n_global = n_local * world

    # Weak scaling: the global bundle is the concatenation, in rank order, of one complete C2 bundle per GPU (same disc and cone
    # distribution, the cone directions drawn from seed + rank), so every rank traces the N = 1 workload and its contiguous shard
    # keeps the reference's detector order.  Slicing ONE disc into rank-sized rings would give the ranks unequal work.
    from scenes import SEED

    bundle = c2_bundle(n_local, seed=SEED + rank)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, device_ord if world > 1 else 0)
    dev_batch = eng.upload(bundle)
    n_det = len(scene.detectors)

    det_width = [2 if d.kind == bmo.components.O_SPOT else 9 for d in scene.detectors]  # a Spotdetector stores (x, y) only
    in_flight = []  # exchange step of the previous trace, still travelling

    def finish_exchange():
        out = [g.wait() for g in in_flight]  # (hits in reference order, counts) per detector
        in_flight.clear()
        return out

    def one_step():
        res = eng.trace_device(dev_batch, args.r_max)
        kms, tms, nl = eng.result_timing(res)
        counts = [eng.result_device_hits(res, s)[1] for s in range(n_det)]
        if world > 1:
            # Exchange step: all-gather of the detector hit lists over xGMI — counts first, then the payload
            # (the columns the detector keeps), left in flight so that it overlaps the NEXT trace (separate RCCL stream);
            # the previous step's exchange is completed first, and the last one inside the timed region (sync()).
            finish_exchange()
            payloads = []
            for s in range(n_det):
                local = torch.empty((counts[s], 9), dtype=torch.float64, device="cuda")
                eng.result_copy_hits(res, s, local.data_ptr(), counts[s])
                payload = local[:, : det_width[s]].contiguous()
                payloads.append(payload if backend == "nccl" else payload.cpu())
            if payloads:
                in_flight.extend(bd.all_gather_hit_lists(payloads))
        stats = dict(kernel_ms=kms, total_ms=tms, launches=nl, hits=counts)
        return res, stats

    def sync():
        if world > 1:
            finish_exchange()
            dist.barrier()
            torch.cuda.synchronize()

    totals = None
    for _ in range(args.warmup):
        res, st = one_step()
        eng.free_result(res)
    sync()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    launches = 0
    last = None
    for _ in range(args.steps):
        if last is not None:
            eng.free_result(last)
        last, st = one_step()
        kernel_ms += st["kernel_ms"]
        launches += st["launches"]
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # workload counters of ONE step (identical every step: the trace is deterministic)
    calls, nrec, nnodes, hits = eng.result_size(last)  # bmo_result_counts: nothing is downloaded
    traced = int(nrec)  # every record is one tracing_step
    eng.free_result(last)
    if world > 1:
        agg = torch.tensor([calls, traced, hits, n_local], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(agg)
        calls_all, traced_all, hits_all, rays_all = (float(x) for x in agg.cpu())
    else:
        calls_all, traced_all, hits_all, rays_all = float(calls), float(traced), float(hits), float(n_local)

    if rank == 0:
        value = calls_all * args.steps / dt
        alg_bytes_step = traced * BYTES_PER_BOUNCE + hits * BYTES_PER_HIT  # this rank, one step
        avg_launch_ms = kernel_ms / max(launches, 1)
        achieved = alg_bytes_step * args.steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # HBM traffic of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this timed
        # process): bytes per launch, measured with the same command at the default workload; null for any other workload.
        traffic, traffic_src = None, None
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic_c2_1M.json")
        if os.path.exists(tp) and n_local == (1 << 20) and args.r_max == 100:
            tj = json.load(open(tp))
            traffic = (tj["fetch_bytes"] + tj["write_bytes"]) / tj["launches"]
            traffic_src = tj["source"]
        out = {
            "metric": "ray-surface intersections/s",
            "value": value,
            "unit": "intersections/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "rays_per_s": rays_all * args.steps / dt,
            "config": {
                "workload": "1M geometric Rays (2^20 per GPU, Fibonacci disc + 0.25 rad cone, seed 20251003 + rank) through the 10-element "
                            "mesh+SDF miniscope scene with one ThinBeamsplitter and two Spotdetectors; r_max=100; full segment log kept",
                "rays_per_gpu": n_local, "elements": scene.n_objects, "shapes": len(scene.shape_list),
                "segments_per_step": int(traced_all), "beam_nodes": int(nnodes), "detector_hits": int(hits_all),
                "intersect3d_calls_per_step": int(calls_all), "parallelism": f"ray-shard x{world}" + (f" + {'RCCL' if backend == 'nccl' else backend} all-gather of detector hits" if world > 1 else ""),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                "kernel": "step_kernel<RAY>", "launches_per_step": launches // max(args.steps, 1), "avg_launch_ms": avg_launch_ms,
                "algorithmic_bytes_per_launch": alg_bytes_step / max(launches // max(args.steps, 1), 1),
                "note": "path is FP64-VALU/latency bound (SURVEY.md §8d): algorithmic HBM bytes are 184 B/bounce + 16 B/hit; "
                        "see DESIGN.md for the VALU-side accounting",
            },
        }
        if args.cpu_sample > 0 and world == 1:  # the CPU baseline is reported at N = 1 only
            def strided(sample):
                g = c2_bundle(n_local)
                idx = (np.arange(sample) * (n_local // sample)).astype(np.int64)
                return bmo.RayBundle(g.kind, g.planes[:, idx])

            out["cpu_baseline"] = cpu_baseline(scene, strided, min(args.cpu_sample, n_local), args.r_max)
        print(json.dumps(out))
    eng.free_batch(dev_batch)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
