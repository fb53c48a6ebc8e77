"""Multi-GPU partitioning of the batch axis: one process per GPU, scene replicated, root rays
sharded contiguously, ONE exchange step — the all-gather of the per-detector hit buffers (RCCL over xGMI when the
process group is `nccl`; `gloo` on CPU in the tests).

Contiguity matters: the reference appends detector data in bundle order x BFS order (Spotdetector.jl:27,59;
System.jl:446-458).  Every rank's buffer is already in that order for its shard, so concatenating the gathered
buffers in rank order reproduces `sd.data` of the un-sharded solve exactly.
"""
import torch
import torch.distributed as dist

HIT_WIDTH = 9  # doubles per hit record (include/bmo.h det_data)


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n root beams for `rank` of `world` (sizes differ by at most one)."""
    return n * rank // world, n * (rank + 1) // world


def all_gather_hits(local_hits, group=None):
    """All-gather variable-length hit buffers.

    local_hits: tensor [count, 9] float64 on this rank's device (CUDA for nccl, CPU for gloo).
    Returns (hits [total, 9] in reference order, counts [world]).  Counts are exchanged first, then buffers padded to
    the maximum count (one collective each), then trimmed and concatenated in rank order.
    """
    world = dist.get_world_size(group)
    dev = local_hits.device
    cnt = torch.tensor([local_hits.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = torch.cat(counts).cpu()
    mx = max(int(counts.max()), 1)
    buf = torch.zeros((mx, HIT_WIDTH), dtype=torch.float64, device=dev)
    buf[: local_hits.shape[0]] = local_hits
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    hits = torch.cat([outs[r][: int(counts[r])] for r in range(world)], dim=0)
    return hits, counts


def all_reduce_field(field, group=None):
    """Sum of the per-rank Photodetector fields: every rank accumulates the field of ITS shard of beamlets with
    bmo_photodetector_field; the detector's field is their sum, one all-reduce of the nx x ny complex grid (viewed as float64
    pairs; RCCL ring all-reduce over xGMI for `nccl`).  Summation order differs from the single-process solve, so the result
    agrees with it to FP64 re-association tolerance.  `field`: complex128 tensor [nx, ny]; reduced in place and returned."""
    buf = torch.view_as_real(field)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return field
