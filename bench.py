#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on the HIP engine: ray-surface intersections/s (and rays/s, beamlets/s).

A "step" = one full pass of the hot path over one batch: solve_system!(system, bundle) on fresh beams.  The bundle is uploaded
once; the timed region runs bmo_trace_device K times (all bounce-step kernels, child spawning, node ordering, detector-hit
compaction; results stay in HBM) and, for N > 1, the RCCL all-gather of the per-GPU detector hit buffers.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R] [--workload c2|c2v|c3|c4|c5]

N = 1 (default): BASELINE config 2 on SURVEY §8(d)'s literal bundle (`c2s`) — 2^20 geometric Rays, Fibonacci disc of 0.8 x the first
                 clear aperture along the axis + 2 mrad jitter, through the 10-element mesh+SDF miniscope scene with one beamsplitter.
                 The same line also carries the other BASELINE configs at their per-GPU sizes (`configs`; `c2_cone` is the point-source
                 cone bundle rounds 1-3 headlined), a vignetted C2 bundle, the PCIe-inclusive rates of the host-buffer boundary
                 (`pcie`) and the CPU baseline (`cpu_baseline`, an object: the reference algorithm on one thread, the other two nested).
N > 1:           BASELINE config 5 — the 32-element scene, 2^21 Rays per GPU (weak scaling: 2^24 at 8 GPUs), contiguous shards,
                 RCCL all-gather of the detector hit lists.  Started either by the driver under torch.distributed.run, or as
                 plain `python bench.py --gpus N`: the N ranks are then started as child processes (before this process touches
                 a GPU) and rank 0's JSON line is relayed.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# SURVEY.md §8d algorithmic bytes per bounce (segment 64 + hint 8 read; intersection 40 + segment 64 + hint 8 written) by beam kind
BYTES_PER_BOUNCE = {0: 184, 1: 280, 2: 584}
# the bound the path actually runs into: one vector instruction of a 64-wide wave occupies a SIMD's 16 FP64 lanes for 4 cycles,
# 256 CUs x 4 SIMDs at the 2.4 GHz maximum clock (MI355X_MICROARCH.md) = 6.1e11 wave-instructions/s = the 78.6 TFLOP/s vector-FP64 peak in FMAs
VALU_PEAK_WAVE_INST_PER_S = 256 * 4 * 2.4e9 / 4
KERNEL_NAME = {0: "step_kernel<RAY>", 1: "step_kernel<POLARIZED>", 2: "step_kernel_gauss"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays", type=int, default=0, help="root beams per GPU (0 = the BASELINE size of the workload)")
    ap.add_argument("--workload", default="", help="c2s | c2 | c2v | c3 | c4 | c5 (default: c2s at N = 1, c5 at N > 1)")
    ap.add_argument("--r-max", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=16384, help="rays for the CPU baseline (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the other configs and the PCIe-inclusive rates")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (nothing here has touched a GPU or
    imported torch) under torch.distributed.run and relay rank 0's JSON line.  Never exec: the child's exit code is ours."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    else:
        sys.stderr.write(p.stdout)
    return p.returncode if p.returncode != 0 or line is not None else 1


# ------------------------------------------------------------------------------------------------------------ workloads
def workload(name, n, rank=0):
    """(system, bundle, description) of one BASELINE config; the bundle of rank r is the r-th contiguous shard of the global one."""
    import scenes
    from scenes import SEED

    lg = "2^%d" % (n.bit_length() - 1) if n & (n - 1) == 0 else str(n)
    if name == "c2":
        return scenes.c2_scene()[0], scenes.c2_bundle(n, seed=SEED + rank), (
            f"config 2 scene, point-source cone bundle (the headline of rounds 1-3): {lg} geometric Rays per GPU (0.3 mm object disc, 0.25 rad cone, seed {SEED} + rank) "
            "through the 10-element mesh+SDF miniscope scene with one ThinBeamsplitter and two Spotdetectors")
    if name == "c2s":
        return scenes.c2_scene()[0], scenes.c2_survey_bundle(n, seed=SEED + rank), (
            f"BASELINE config 2 on SURVEY 8(d)'s literal bundle: {lg} geometric Rays per GPU, Fibonacci disc (BeamGroups.jl:232-243) of 0.8 x the first clear aperture "
            f"(1.83 mm) at the object plane, directions along the optical axis + a per-ray jitter of at most 2 mrad (PCG64 seed {SEED} + rank), through the 10-element "
            "mesh+SDF miniscope scene with one ThinBeamsplitter and two Spotdetectors")
    if name == "c2v":
        return scenes.c2_scene()[0], scenes.c2_vignetted_bundle(n, seed=SEED + rank), (
            f"config 2 scene, vignetted bundle: {lg} geometric Rays per GPU, 2.0 mm object disc (0.87 x the first aperture) and 0.6 rad cone: 10 % of the rays miss the first lens, 45 % reach the splitter, the rest "
            "are clipped at lens rims and rings or leave the train after 3-26 segments")
    if name == "c3":
        return scenes.c2_scene()[0], scenes.c3_bundle(n), (
            f"BASELINE config 3: {lg} GaussianBeamlets (TEM00, w0 = 50 um; 3 rays each) per GPU, same 10-element scene")
    if name == "c4":
        return scenes.c4_scene()[0], scenes.c4_bundle(n), (
            f"BASELINE config 4: {lg} PolarizedRays per GPU through three singlets (6 refracting surfaces) + end stop")
    if name == "c5":
        return scenes.c5_scene()[0], scenes.c5_bundle(n, seed=SEED + rank), (
            f"BASELINE config 5: {lg} geometric Rays per GPU (contiguous shard of the global bundle, seed {SEED} + rank) through the "
            "32-element scene (3 miniscope trains, fold mirrors, prisms, one ThinBeamsplitter, two Spotdetectors, two baffles)")
    raise SystemExit(f"unknown workload {name!r}")


CONFIG_KEY = {"c2": "c2_cone"}  # key of a workload in the line's `configs`
DEFAULT_RAYS = {"c2": 1 << 20, "c2s": 1 << 20, "c2v": 1 << 20, "c3": 1 << 20, "c4": 1 << 18, "c5": 1 << 21}


class Case:
    """One compiled workload resident on one GPU."""

    def __init__(self, bmo, name, n, device, rank=0):
        self.bmo, self.name = bmo, name
        self.system, self.bundle, self.text = workload(name, n, rank)
        self.scene = bmo.CompiledScene(self.system, self.bundle.lambdas)
        self.eng = bmo.Engine(self.scene, device)
        self.dev_batch = self.eng.upload(self.bundle)
        self.kind = self.bundle.kind
        self.n_det = len(self.scene.detectors)
        # bytes of one detector record as the reference stores it: Spotdetector Point2 (16 B), PSF hit (72 B); x3 rays for a beamlet
        sub = 3 if self.kind == 2 else 1
        self.det_cols = [2 if d.kind == bmo.components.O_SPOT else 9 for d in self.scene.detectors]
        self.det_bytes = [8 * c * sub for c in self.det_cols]

    def close(self):
        self.eng.free_batch(self.dev_batch)
        self.eng.close()

    def solve(self, r_max):
        res = self.eng.trace_device(self.dev_batch, r_max)
        kms, tms, nl = self.eng.result_timing(res)
        return res, kms, nl

    def measure(self, r_max, steps, warmup):
        """K resident solves: (seconds, kernel ms, launches, counters of one solve)."""
        for _ in range(warmup):
            res, _, _ = self.solve(r_max)
            self.eng.free_result(res)
        t0 = time.perf_counter()
        kms_sum, nl_sum, last = 0.0, 0, None
        for _ in range(steps):
            if last is not None:
                self.eng.free_result(last)
            last, kms, nl = self.solve(r_max)
            kms_sum += kms
            nl_sum += nl
        dt = time.perf_counter() - t0
        calls, nrec, nnodes, hits = self.eng.result_size(last)
        counts = self.eng.result_counts(last)
        self.eng.free_result(last)
        return dt, kms_sum, nl_sum, dict(calls=calls, segments=nrec, nodes=nnodes, hits=hits, det_counts=counts)

    def algorithmic_bytes(self, c):
        sub = 3 if self.kind == 2 else 1
        return c["segments"] * BYTES_PER_BOUNCE[self.kind] + sum(n // sub * b for n, b in zip(c["det_counts"], self.det_bytes))


def measured_traffic(name, n, r_max):
    """HBM bytes per SOLVE of the workload's step kernels from the PMC passes committed under profiles/ (rocprofv3 cannot run inside
    this timed process): FETCH_SIZE x 2 + WRITE_SIZE, separate passes (MI355X_MICROARCH.md "HBM"), same command, default size only.
    roofline_of divides by the launches per solve of THIS run, like the algorithmic bytes."""
    tp = os.path.join(ROOT, "profiles", "r04_traffic.json")
    if not os.path.exists(tp):
        tp = os.path.join(ROOT, "profiles", "r03_traffic.json")
    if not os.path.exists(tp) or n != DEFAULT_RAYS.get(name) or r_max != 100:
        return None, None
    tj = json.load(open(tp)).get(name)
    if not tj:
        return None, None
    return 2 * tj["fetch_size_bytes"] + tj["write_size_bytes"], tj["source"]


def valu_issue(name, n, r_max, kernel_ms_per_solve):
    """The VALU side of the accounting: vector instructions the step kernels issue per solve (a property of the workload and the build: the
    SQ_INSTS_VALU pass committed under profiles/) over the kernel time measured live in THIS run, against the SIMDs' issue peak."""
    tp = os.path.join(ROOT, "profiles", "r04_traffic.json")
    if not os.path.exists(tp) or n != DEFAULT_RAYS.get(name) or r_max != 100 or kernel_ms_per_solve <= 0:
        return None
    tj = json.load(open(tp)).get(name) or {}
    if not tj.get("valu_wave_instructions"):
        return None
    rate = tj["valu_wave_instructions"] / (kernel_ms_per_solve * 1e-3)
    return {"wave_instructions_per_solve": tj["valu_wave_instructions"], "achieved": rate, "peak": VALU_PEAK_WAVE_INST_PER_S, "unit": "wave-instructions/s",
            "frac": rate / VALU_PEAK_WAVE_INST_PER_S, "fp64_arithmetic_share": tj["valu_fp64_arithmetic"] / tj["valu_wave_instructions"],
            "source": tj["valu_source"], "note": "peak at the 2.4 GHz maximum clock; every vector instruction (FP64 arithmetic, selects, compares, moves) takes one issue slot"}


def roofline_of(case, c, kernel_ms, launches, steps, traffic=None, traffic_src=None):
    alg = case.algorithmic_bytes(c)  # one solve on this rank
    achieved = alg * steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    per_step = max(launches // max(steps, 1), 1)
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic / per_step if traffic is not None else None,
        "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
        "kernel": KERNEL_NAME[case.kind], "launches_per_step": per_step, "avg_launch_ms": kernel_ms / max(launches, 1),
        # (the launches of a solve are very unequal — the first one carries the roots through up to 32 levels, the later ones the
        #  reflected children —, so the per-SOLVE kernel time is the basis of `achieved`; avg_launch_ms is informational)
        "kernel_ms_per_solve": kernel_ms / max(steps, 1), "algorithmic_bytes_per_solve": alg,
        "algorithmic_bytes_per_launch": alg / per_step,
        "note": "path is FP64-VALU/latency bound (SURVEY.md §8d): algorithmic HBM bytes are %d B/bounce + the detector records; "
                "see DESIGN.md §4 for the VALU-side accounting" % BYTES_PER_BOUNCE[case.kind],
    }


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(case, sample, r_max):
    """ONE object holding three CPU figures on bounded samples of the same workload, timed on this box's host cores (rank 0, N = 1 only):
      headline: the oracle (op-for-op restatement of the reference algorithm, incl. its 1000-iteration misses) on ONE thread — the
          reference's trace loop is serial (System.jl:463-468);                                      kind "port"
      "all_cores": the oracle, plain parallel-for over rays on every host core;                           kind "port"
      "lane_code_all_cores": the engine's own lane code compiled for the host (tests/emu: same culls, same shortcuts, same arithmetic as the HIP
          kernels) on every host core — the "same algorithm on a CPU" figure that separates what the GPU buys from what the
          result-preserving shortcuts buy.                                                             kind "port"
    """
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    import parity
    import pyoracle

    nproc = os.cpu_count() or 1
    workers = min(nproc, 32)  # a GPU box shares its host: 32 threads were the fastest the oracle ran with (round 2), more only contend
    model = cpu_model()
    g = case.bundle

    def strided(k):
        idx = (np.arange(k) * (g.n // k)).astype(np.int64)  # strided subsample of the benchmark bundle
        return case.bmo.RayBundle(g.kind, g.planes[:, idx])

    out = []
    for threads, k in ((1, max(sample // 8, 64)), (workers, sample)):
        b = strided(min(k, g.n))
        t = time.perf_counter()
        ref = pyoracle.trace(case.scene, b, r_max, threads=threads)
        dt = time.perf_counter() - t
        out.append({"value": ref.n_intersect_calls / dt, "unit": "intersections/s", "cores": threads, "kind": "port",
                    "what": "oracle: reference algorithm restated op for op (misses burn the reference's 1000 sdf evaluations)" + (
                        ", one thread: the reference's trace loop is serial (System.jl:463-468)" if threads == 1 else f", plain parallel-for over rays on {workers} threads"),
                    "sample": f"every (N/{b.n})-th ray of the same bundle, same scene ({ref.n_intersect_calls} reference intersect3d calls, {dt:.1f} s wall)",
                    "rays_per_s": b.n / dt, "nproc": nproc, "cpu": model})
    # the lane code on the host: a sample 16 x larger (it skips what the engine skips), split over the cores
    b = strided(min(sample * 16, g.n))
    parts = max(workers, 1)
    cuts = [b.n * i // parts for i in range(parts + 1)]
    subs = [case.bmo.RayBundle(b.kind, b.planes[:, cuts[i]:cuts[i + 1]]) for i in range(parts) if cuts[i + 1] > cuts[i]]
    parity.emu_trace(case.scene, case.bmo.RayBundle(b.kind, b.planes[:, :8]), r_max)  # build / load the host library outside the timed region
    t = time.perf_counter()
    with ThreadPoolExecutor(max_workers=parts) as ex:
        calls = sum(r.n_intersect_calls for r in ex.map(lambda sb: parity.emu_trace(case.scene, sb, r_max), subs))
    dt = time.perf_counter() - t
    out.append({"value": calls / dt, "unit": "intersections/s", "cores": workers, "kind": "port",
                "what": "the engine's lane code (csrc/bmo_lane.hpp: culls, prunes and child skips on) compiled for the host, one sub-bundle per core",
                "sample": f"every (N/{b.n})-th ray of the same bundle, same scene ({calls} reference intersect3d calls counted, {dt:.1f} s wall)",
                "rays_per_s": b.n / dt, "nproc": nproc, "cpu": model})
    # ONE object (the driver's parser keeps `cpu_baseline` only as an object): the headline is the reference algorithm on one thread —
    # the reference's trace loop is serial (System.jl:463-468) —, the other two figures are nested inside it
    head = dict(out[0])
    head["all_cores"] = out[1]
    head["lane_code_all_cores"] = out[2]
    return head


def pcie_rates(case, r_max, calls):
    """The host-buffer form of the boundary on the headline workload (never `value`): best of 3 per variant."""
    import ctypes as C

    import numpy as np

    bmo, eng, scene = case.bmo, case.eng, case.scene
    lib = eng.lib
    abi = bmo.abi
    batch, keep = bmo.make_batch(scene, case.bundle)
    o = eng.opts(r_max)
    hit_bufs = {}

    def host_solve():
        res = C.c_void_p()
        abi.check(lib, lib.bmo_trace(eng.handle, C.byref(batch), C.byref(o), C.byref(res)), "bmo_trace")
        return res

    def with_h2d():
        lib.bmo_result_free(host_solve())

    def with_hits():
        res = host_solve()
        copy_hits(res)
        lib.bmo_result_free(res)

    def copy_hits(res):
        for s in range(case.n_det):
            cnt = eng.result_device_hits(res, s)[1]
            w = case.det_cols[s]
            if s not in hit_bufs or hit_bufs[s].shape[0] < cnt:
                hit_bufs[s] = np.zeros((max(cnt, 1), w))  # the caller's buffer, reused across solves
            eng.result_copy_hit_columns(res, s, w, hit_bufs[s].ctypes.data, cnt)

    def with_view(what, packed_hits=False):
        def f():
            res = host_solve()
            v = abi.ResultView()
            abi.check(lib, lib.bmo_result_view_select(res, what, C.byref(v)), "bmo_result_view_select")
            if packed_hits:
                copy_hits(res)
            lib.bmo_result_free(res)
        return f

    out = {}
    for name, fn, reps in (("with_h2d", with_h2d, 3), ("with_hits_d2h", with_hits, 3),
                           ("with_last_segment_view", with_view(abi.VIEW_LAST_SEGMENT, packed_hits=True), 3),
                           ("with_full_view", with_view(abi.VIEW_HITS | abi.VIEW_SEGMENTS), 2)):
        fn()  # warm-up: pinned pools, page faults
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        out[name] = {"ms": best * 1e3, "intersections_per_s": calls / best}
    out["what"] = {"with_h2d": "bmo_trace: host ray batch in, solution left in HBM",
                   "with_hits_d2h": "+ every detector's hit table (the columns the detector keeps) copied to host memory",
                   "with_last_segment_view": "+ bmo_result_view_select(LAST_SEGMENT) + the hit columns: beam tree, last ray of every beam and the detectors' data on the host",
                   "with_full_view": "+ bmo_result_view: the whole segment log on the host (PCIe-bound)"}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = torch = None
    backend = None
    device_ord = 0
    # BMO_BENCH_FORCE_DIST=1 (under torchrun with one rank): take the N > 1 code path — process group, exchange step, collectives — with a
    # world of one, e.g. to rehearse the RCCL calls on a one-GPU box (tests/test_multi_gpu.py)
    multi = world > 1 or bool(os.environ.get("BMO_BENCH_FORCE_DIST"))
    if multi:
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

    if multi:
        from bmo_amd import distributed as bd

    name = args.workload or ("c5" if multi else "c2s")
    n_local = args.rays or DEFAULT_RAYS[name]
    # Weak scaling: the global bundle is the concatenation, in rank order, of one complete bundle per GPU (same disc and cone
    # distribution, directions drawn from seed + rank), so every rank traces the same amount of work and its contiguous shard keeps
    # the reference's detector order.  Slicing ONE disc into rank-sized rings would give the ranks unequal work.
    case = Case(bmo, name, n_local, device_ord, rank)
    eng, n_det = case.eng, case.n_det
    in_flight = []  # exchange step of the previous trace, still travelling
    exchange_state = bd.HitExchange(case.det_cols) if multi and n_det else None  # counts ride in the payload: no host round trip per step

    def finish_exchange():
        out = [g.wait() for g in in_flight]  # [(hits in reference order, counts) per detector] per pending exchange
        in_flight.clear()
        return out

    def one_step(exchange):
        res, kms, nl = case.solve(args.r_max)
        if exchange:
            # Exchange step: all-gather of the detector hit lists over xGMI — counts first, then the payload
            # (only the columns the detector keeps: 16 B per Spotdetector hit), left in flight so that it overlaps the NEXT
            # trace (separate RCCL stream); the previous step's exchange is completed first, the last one inside the timed region.
            finish_exchange()
            payloads = []
            for s in range(n_det):
                cnt = eng.result_device_hits(res, s)[1]
                local = torch.empty((cnt, case.det_cols[s]), dtype=torch.float64, device="cuda")
                eng.result_copy_hit_columns(res, s, case.det_cols[s], local.data_ptr(), cnt)
                payloads.append(local if backend == "nccl" else local.cpu())
            if payloads:
                in_flight.append(exchange_state.start(payloads))
        return res, kms, nl

    def sync():
        if multi:
            finish_exchange()
            dist.barrier()
            torch.cuda.synchronize()

    # N > 1: this rank's rate WITHOUT the exchange step, same workload, measured before the timed region — the one-GPU reference
    # point of the same config, so that scaling can be read against the same work (N = 1 of the driver's curve runs config 2).
    local_only = None
    if multi:
        dt0, _, _, c0 = case.measure(args.r_max, max(2, min(args.steps, 3)), 1)
        local_only = c0["calls"] * max(2, min(args.steps, 3)) / dt0

    for _ in range(args.warmup):
        res, _, _ = one_step(multi)
        eng.free_result(res)
    sync()
    t0 = time.perf_counter()
    kernel_ms, launches, last = 0.0, 0, None
    for _ in range(args.steps):
        if last is not None:
            eng.free_result(last)
        last, kms, nl = one_step(multi)
        kernel_ms += kms
        launches += nl
    sync()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # workload counters of ONE step (identical every step: the trace is deterministic)
    calls, nrec, nnodes, hits = eng.result_size(last)  # bmo_result_counts: nothing is downloaded
    det_counts = eng.result_counts(last)
    eng.free_result(last)
    mine = dict(calls=calls, segments=nrec, nodes=nnodes, hits=hits, det_counts=det_counts)
    if multi:
        agg = torch.tensor([calls, nrec, hits, n_local, nnodes], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(agg)
        calls_all, traced_all, hits_all, rays_all, nodes_all = (float(x) for x in agg.cpu())
    else:
        calls_all, traced_all, hits_all, rays_all, nodes_all = float(calls), float(nrec), float(hits), float(n_local), float(nnodes)

    if rank == 0:
        # HBM traffic of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this timed
        # process): bytes per launch, measured with the same command at the default workload; null for any other workload.
        traffic, traffic_src = measured_traffic(name, n_local, args.r_max)
        unit_name = "beamlets_per_s" if case.kind == 2 else "rays_per_s"
        out = {
            "metric": "ray-surface intersections/s",
            "value": calls_all * args.steps / dt,
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
            unit_name: rays_all * args.steps / dt,
            "config": {
                "workload": case.text + f"; r_max={args.r_max}; full segment log kept",
                "name": name, "rays_per_gpu": n_local, "elements": case.scene.n_objects, "shapes": len(case.scene.shape_list),
                "segments_per_step": int(traced_all), "beam_nodes": int(nodes_all), "detector_hits": int(hits_all),
                "intersect3d_calls_per_step": int(calls_all),
                "parallelism": f"ray-shard x{world}" + (f" + {'RCCL' if backend == 'nccl' else backend} all-gather of detector hits" if multi else ""),
            },
            "roofline": roofline_of(case, mine, kernel_ms, launches, args.steps, traffic, traffic_src),
        }
        out["roofline"]["valu_issue"] = valu_issue(name, n_local, args.r_max, kernel_ms / max(args.steps, 1))
        if local_only is not None:
            out["one_gpu_same_workload"] = {"value": local_only, "unit": "intersections/s",
                                            "what": "rank 0, same shard, solves only (no exchange step), measured before the timed region"}
        if not multi and not args.no_extras:
            out["pcie"] = pcie_rates(case, args.r_max, calls)
        if args.cpu_sample > 0 and not multi:  # the CPU baseline is reported at N = 1 only
            cb = cpu_baseline(case, min(args.cpu_sample, n_local), args.r_max)
            out["cpu_baseline"] = cb
            # BASELINE.md holds no published number for this metric (the reference publishes none); its §3 names the baseline to time
            # beside the GPU: B1, the reference algorithm on one host thread.  vs_baseline is the ratio to THAT measurement.
            out["vs_baseline"] = out["value"] / cb["value"] if cb["value"] > 0 else None
            out["vs_baseline_of"] = "cpu_baseline.value (BASELINE.md §3 B1: the reference algorithm restated op for op, one host thread; nothing is published upstream)"
    case.close()
    if rank == 0 and not multi and not args.no_extras:
        # the other BASELINE configs at their per-GPU sizes, and the vignetted C2 bundle: 3 resident solves each after 1 warm-up
        cfgs = {}
        for other in ("c2s", "c2", "c2v", "c3", "c4", "c5"):
            if other == name:
                continue
            oc = Case(bmo, other, DEFAULT_RAYS[other], device_ord)
            odt, okms, onl, c = oc.measure(args.r_max, 3, 1)
            rl = roofline_of(oc, c, okms, onl, 3, *measured_traffic(other, DEFAULT_RAYS[other], args.r_max))
            cfgs[CONFIG_KEY.get(other, other)] = {"workload": oc.text, "beams": oc.bundle.n, "ms": odt / 3 * 1e3, "kernel_ms": okms / 3, "launches": onl // 3,
                           "intersections_per_s": c["calls"] * 3 / odt, ("beamlets_per_s" if oc.kind == 2 else "rays_per_s"): oc.bundle.n * 3 / odt,
                           "segments": c["segments"], "beam_nodes": c["nodes"], "detector_hits": c["hits"],
                           "roofline_frac": rl["frac"], "achieved_GBps": rl["achieved"], "traffic": rl["traffic"], "traffic_unit": "bytes per launch",
                           "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"]}
            oc.close()
        out["configs"] = cfgs
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
