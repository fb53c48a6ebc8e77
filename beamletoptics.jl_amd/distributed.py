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


class PendingGather:
    """An all-gather of hit buffers in flight (async_op): wait() -> (hits [total, width] in reference order, counts [world])."""

    def __init__(self, work, outs, counts):
        self.work, self.outs, self.counts = work, outs, counts

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        world = len(self.outs)
        hits = torch.cat([self.outs[r][: int(self.counts[r])] for r in range(world)], dim=0)
        return hits, self.counts


def all_gather_hits(local_hits, group=None, async_op=False):
    """All-gather variable-length hit buffers.

    local_hits: tensor [count, width] float64 on this rank's device (CUDA for nccl, CPU for gloo); width is 9 for the full hit
    record (include/bmo.h det_data) or fewer leading columns (a Spotdetector only uses x, y).
    Returns (hits [total, width] in reference order, counts [world]).  Counts are exchanged first (one tiny collective), then
    buffers padded to the maximum count, trimmed and concatenated in rank order.  With async_op the payload collective is
    left in flight (PendingGather) so that it overlaps the next trace; call .wait() before using the result.
    """
    pending = all_gather_hit_lists([local_hits], group)[0]
    return pending if async_op else pending.wait()


class _GroupGather:
    """One payload collective shared by the detectors of equal record width: every rank's block holds its hits of those detectors
    back to back.  wait() -> per-rank blocks; PendingGather objects slice them."""

    def __init__(self, work, outs):
        self.work, self.outs = work, outs

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.outs


class PendingGatherSlice(PendingGather):
    """One detector's share of a _GroupGather: rows [start[r], start[r] + counts[r]) of every rank's block."""

    def __init__(self, grp, starts, counts):
        self.grp, self.starts, self.counts = grp, starts, counts

    def wait(self):
        outs = self.grp.wait()
        hits = torch.cat([outs[r][int(self.starts[r]): int(self.starts[r]) + int(self.counts[r])] for r in range(len(outs))], dim=0)
        return hits, self.counts


def all_gather_hit_lists(payloads, group=None):
    """The exchange step for several detectors at once: ONE collective for all their counts, then one payload all-gather per record
    width (detectors of equal width share it) left in flight.  payloads: list of [count_d, width_d] float64 tensors.  Returns a
    list of pending gathers, one per detector, whose wait() gives (hits [total, width] in reference order, counts [world])."""
    world = dist.get_world_size(group)
    dev = payloads[0].device
    cnt = torch.tensor([p.shape[0] for p in payloads], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(allc, cnt, group=group)
    allc = torch.stack(allc).cpu()  # [world, n_det]
    out = [None] * len(payloads)
    widths = sorted({p.shape[1] for p in payloads})
    for width in widths:
        members = [d for d, p in enumerate(payloads) if p.shape[1] == width]
        tot = allc[:, members].sum(dim=1)  # rows of every rank's block
        mx = max(int(tot.max()), 1)
        buf = torch.zeros((mx, width), dtype=torch.float64, device=dev)
        at = 0
        for d in members:
            n = payloads[d].shape[0]
            buf[at: at + n] = payloads[d]
            at += n
        if dist.get_backend(group) == "nccl":  # one contiguous receive buffer: no per-rank copy kernels after the ring all-gather
            flat = torch.empty((world * mx, width), dtype=torch.float64, device=dev)
            work = dist.all_gather_into_tensor(flat, buf, group=group, async_op=True)
            outs = [flat[r * mx:(r + 1) * mx] for r in range(world)]
        else:
            outs = [torch.empty_like(buf) for _ in range(world)]
            work = dist.all_gather(outs, buf, group=group, async_op=True)
        grp = _GroupGather(work, outs)
        starts = torch.zeros(world, dtype=torch.int64)
        for d in members:
            out[d] = PendingGatherSlice(grp, starts.clone(), allc[:, d].contiguous())
            starts = starts + allc[:, d]
    return out


def all_reduce_field(field, group=None):
    """Sum of the per-rank Photodetector fields: every rank accumulates the field of ITS shard of beamlets with
    bmo_photodetector_field; the detector's field is their sum, one all-reduce of the nx x ny complex grid (viewed as float64
    pairs; RCCL ring all-reduce over xGMI for `nccl`).  Summation order differs from the single-process solve, so the result
    agrees with it to FP64 re-association tolerance.  `field`: complex128 tensor [nx, ny]; reduced in place and returned."""
    buf = torch.view_as_real(field)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return field
