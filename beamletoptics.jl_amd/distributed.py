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


class HitExchange:
    """The exchange step of a LOOP of solves without a host synchronisation per step.

    all_gather_hit_lists learns the other ranks' hit counts with a small collective and a `.cpu()` before it can size the payload
    buffers: one host round trip inside every step.  Here the counts travel IN the payload — row 0 of every rank's block holds its
    per-detector counts — and the block size is the one the PREVIOUS step established (all ranks derived it from the same gathered
    counts, so they agree on it without talking).  The gathered blocks are read on the host one step late, when the next solve has
    been queued already.  A step whose hits do not fit the established block size (first step, or a workload that changed) falls
    back to the synchronous exchange once and establishes the new size.

        ex = HitExchange(widths)            # record width (columns) of every detector
        p = ex.start(payloads)              # list of [count_d, width_d] float64 tensors of this rank, returns a pending object
        ... queue the next solve ...
        per_detector = p.wait()             # [(hits [total, width] in reference order, counts [world]) for every detector]
    """

    def __init__(self, widths, group=None):
        self.widths, self.group = list(widths), group
        self.world = dist.get_world_size(group)
        self.rows = None  # established payload rows per rank block (without the header row)

    def _pack(self, payloads, rows):
        wmax = max(max(self.widths), len(self.widths))
        dev = payloads[0].device
        buf = torch.zeros((rows + 1, wmax), dtype=torch.float64, device=dev)
        for d, p in enumerate(payloads):  # header: this rank's counts — host ints written by device-side fills (no pageable H2D copy per step)
            buf[0, d].fill_(float(p.shape[0]))
        at = 1
        for p in payloads:
            buf[at: at + p.shape[0], : p.shape[1]] = p
            at += p.shape[0]
        return buf

    def start(self, payloads):
        own = sum(int(p.shape[0]) for p in payloads)
        if self.rows is None:
            return self._sync(payloads)
        fits = own <= self.rows
        buf = self._pack(payloads if fits else [p[:0] for p in payloads], self.rows)
        if not fits:
            buf[0, : len(payloads)] = -1.0  # tells every rank (itself included) that this step has to be redone synchronously
        if dist.get_backend(self.group) == "nccl":
            flat = torch.empty((self.world * (self.rows + 1), buf.shape[1]), dtype=torch.float64, device=buf.device)
            work = dist.all_gather_into_tensor(flat, buf, group=self.group, async_op=True)
            outs = [flat[r * (self.rows + 1):(r + 1) * (self.rows + 1)] for r in range(self.world)]
        else:
            outs = [torch.empty_like(buf) for _ in range(self.world)]
            work = dist.all_gather(outs, buf, group=self.group, async_op=True)
        return _PendingExchange(self, work, outs, payloads)

    def _sync(self, payloads):
        pend = all_gather_hit_lists(payloads, self.group)
        res = [p.wait() for p in pend]
        tot = torch.stack([c for _, c in res]).sum(dim=0)  # rows per rank
        self.rows = int(tot.max()) + int(tot.max()) // 64 + 16  # a little head-room: later steps of the same workload fit
        return _DoneExchange(res)


class _DoneExchange:
    def __init__(self, res):
        self.res = res

    def wait(self):
        return self.res


class _PendingExchange:
    def __init__(self, ex, work, outs, payloads):
        self.ex, self.work, self.outs, self.payloads = ex, work, outs, payloads

    def wait(self):
        self.work.wait()
        nd = len(self.payloads)
        heads = torch.stack([o[0, :nd] for o in self.outs]).cpu()  # [world, n_det]: read one step late, the collective is long done
        if bool((heads < 0).any()):  # some rank's hits did not fit: every rank sees it in the same gathered header and redoes the step
            return self.ex._sync(self.payloads).wait()
        counts = heads.to(torch.int64)
        res = []
        starts = torch.ones(len(self.outs), dtype=torch.int64)
        for d in range(nd):
            w = self.payloads[d].shape[1]
            hits = torch.cat([self.outs[r][int(starts[r]): int(starts[r]) + int(counts[r, d]), :w] for r in range(len(self.outs))], dim=0)
            res.append((hits, counts[:, d].contiguous()))
            starts = starts + counts[:, d]
        return res


def all_reduce_field(field, group=None):
    """Sum of the per-rank Photodetector fields: every rank accumulates the field of ITS shard of beamlets with
    bmo_photodetector_field; the detector's field is their sum, one all-reduce of the nx x ny complex grid (viewed as float64
    pairs; RCCL ring all-reduce over xGMI for `nccl`).  Summation order differs from the single-process solve, so the result
    agrees with it to FP64 re-association tolerance.  `field`: complex128 tensor [nx, ny]; reduced in place and returned."""
    buf = torch.view_as_real(field)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return field
