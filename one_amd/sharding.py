"""Multi-GPU plumbing: one process per GPU, contiguous shards, result gather over RCCL.

The reference has no distributed layer (its only concurrency is N std::threads over one shared
read-only Executable, tools/thr_red.cpp:86-91).  Here the batch of input lines is cut into
contiguous index ranges, one per rank; every rank scans its own shard with its own copy of the
DFA image (no data-path collective), and the per-line Outcome fields are gathered to rank 0
once, at the end - `torch.distributed` backend "nccl" (= RCCL over xGMI) for device tensors,
"gloo" for the CPU rehearsal in tests.

Wire format: xGMI is point-to-point (7 links x ~50-60 GB/s usable per direction into the
root), so records travel in the narrowest integer type that holds them - result in 1/2/4
bytes by the DFA's max result, start/end in 1/2/4/8 bytes by the longest line - and are widened
on rank 0 back to the reference's Outcome types (int32 result, 64-bit start/end), bit-exact.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_lines: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of lines for `rank`: sizes differ by at most one line."""
    base, extra = divmod(n_lines, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_range_by_bytes(offsets: torch.Tensor, world: int, rank: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of lines for `rank`, balanced by BYTES for ragged lines
    (offsets: int64[n+1], monotone)."""
    n = offsets.numel() - 1
    total = int(offsets[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        cuts.append(int(torch.searchsorted(offsets[: n + 1].contiguous(),
                                           torch.tensor([target], dtype=offsets.dtype,
                                                        device=offsets.device)).item()))
    cuts.append(n)
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def _width_for(max_value: int) -> int:
    for w, lim in ((1, 0xFF), (2, 0xFFFF), (4, 0xFFFFFFFF)):
        if max_value <= lim:
            return w
    return 8


_DT = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
_MASK = {1: 0xFF, 2: 0xFFFF, 4: 0xFFFFFFFF}


def _narrow(t: torch.Tensor, width: int) -> torch.Tensor:
    """int32/int64 values known to fit `width` bytes -> raw little-endian bytes [n, width]."""
    if width == 8:
        return t.to(torch.int64).contiguous().view(torch.uint8).view(-1, 8)
    if width == 1:
        return t.to(torch.uint8).view(-1, 1)
    # two's-complement wrap keeps the low bytes; the widener masks them back to unsigned
    return t.to(_DT[width]).contiguous().view(torch.uint8).view(-1, width)


def _widen(b: torch.Tensor, width: int, out_dtype) -> torch.Tensor:
    if width == 8:
        return b.contiguous().view(torch.int64).view(-1).to(out_dtype)
    if width == 1:
        return b[:, 0].to(out_dtype)  # one strided kernel (no copy first): launches are what counts
    v = b.contiguous().view(_DT[width]).view(-1).to(torch.int64) & _MASK[width]
    return v.to(out_dtype)


class RecordFormat:
    """Compact per-line record: [result | start | end], little-endian, fixed widths."""

    def __init__(self, max_result: int, max_line_len: int, with_start: bool = True):
        self.rw = _width_for(max(int(max_result), 0))
        self.pw = _width_for(max(int(max_line_len), 0))
        self.with_start = with_start
        self.bytes = self.rw + self.pw * (2 if with_start else 1)

    def pack(self, result, start, end) -> torch.Tensor:
        parts = [_narrow(result, self.rw)]
        if self.with_start:
            parts.append(_narrow(start, self.pw))
        parts.append(_narrow(end, self.pw))
        return torch.cat(parts, dim=1).contiguous()

    def unpack(self, rec: torch.Tensor):
        rec = rec.view(-1, self.bytes)
        o = 0
        result = _widen(rec[:, o:o + self.rw], self.rw, torch.int32)
        o += self.rw
        start = None
        if self.with_start:
            start = _widen(rec[:, o:o + self.pw], self.pw, torch.int64)
            o += self.pw
        end = _widen(rec[:, o:o + self.pw], self.pw, torch.int64)
        return result, start, end


def gather_outcomes(result, start, end, *, max_result: int, max_line_len: int, dst: int = 0,
                    group=None, async_op: bool = False, equal_counts: bool = False):
    """Gathers every rank's per-line (result, start, end) to rank `dst`, in rank order.

    All ranks pass tensors on the same kind of device (CUDA -> RCCL, CPU -> gloo); shard sizes
    may differ (equal_counts=True promises they do not and saves the count exchange - one
    collective and a host sync).  Returns on `dst` a callable `finish()` -> (result int32[N], start int64[N] |
    None, end int64[N]) covering all shards concatenated; on other ranks `finish()` -> None.
    With async_op=False the collective has completed when this function returns."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    fmt = RecordFormat(max_result, max_line_len, with_start=start is not None)
    n_here = int(result.numel())
    if equal_counts:
        counts = [n_here] * world
    else:
        n_local = torch.tensor([n_here], dtype=torch.int64, device=result.device)
        counts = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(counts, n_local, group=group)
        counts = [int(c.item()) for c in counts]
    n_max = max(counts) if counts else 0
    if result.is_cuda:
        return _gather_native(result, start, end, fmt, counts, n_max, rank, world, dst, group,
                              async_op)
    rec = fmt.pack(result, start, end)
    # equal-size gather (one collective, every link busy at once); pad the short shards
    padded = rec
    if rec.shape[0] < n_max:
        padded = torch.zeros((n_max, fmt.bytes), dtype=torch.uint8, device=rec.device)
        padded[: rec.shape[0]] = rec
    bufs = whole = None
    if rank == dst:
        # one allocation, one view per rank: equal shards are then widened in ONE pass over the
        # whole buffer (a dozen kernel launches on the root instead of a dozen per rank - the
        # launches, not the bytes, are what a short timed region sees of the gather)
        whole = torch.empty((world, n_max, fmt.bytes), dtype=torch.uint8, device=rec.device)
        bufs = list(whole.unbind(0))
    work = dist.gather(padded, gather_list=bufs, dst=dst, group=group, async_op=async_op)

    def finish():
        if async_op and work is not None:
            work.wait()
        if rank != dst:
            return None
        if all(c == n_max for c in counts):
            return fmt.unpack(whole.view(-1, fmt.bytes))
        parts = [fmt.unpack(bufs[r][: counts[r]]) for r in range(world)]
        res = torch.cat([p[0] for p in parts])
        st = torch.cat([p[1] for p in parts]) if fmt.with_start else None
        en = torch.cat([p[2] for p in parts])
        return res, st, en

    return finish


def _gather_native(result, start, end, fmt, counts, n_max, rank, world, dst, group, async_op):
    """Device tensors: the records are packed and widened by the library's own kernels
    (redgpu_records_pack_dev / _unpack_dev, include/redgpu.h) - one launch per shard instead of
    a dozen elementwise torch kernels per field, which is what the root's GPU would otherwise
    spend a fifth of a step on.  Wire format: the plane layout of redgpu_records_bytes; every
    rank packs with its own line count, the root unpacks shard r with counts[r]."""
    import ctypes as C
    from . import _lib
    l = _lib.lib()
    dev = result.device
    with_start = start is not None
    nbytes = int(l.redgpu_records_bytes(n_max, fmt.rw, fmt.pw, int(with_start)))
    rec = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    res32 = result.to(torch.int32).contiguous()
    st64 = start.to(torch.int64).contiguous() if with_start else None
    en64 = end.to(torch.int64).contiguous()
    stream = torch.cuda.current_stream(dev).cuda_stream

    def check(rc):
        if rc != 0:
            raise RuntimeError(l.redgpu_last_error().decode())

    check(l.redgpu_records_pack_dev(dev.index, res32.data_ptr(),
                                    st64.data_ptr() if with_start else None, en64.data_ptr(),
                                    res32.numel(), fmt.rw, fmt.pw, rec.data_ptr(), C.c_void_p(stream)))
    bufs = whole = None
    if rank == dst:
        whole = torch.empty((world, rec.numel()), dtype=torch.uint8, device=dev)
        bufs = list(whole.unbind(0))
    work = dist.gather(rec, gather_list=bufs, dst=dst, group=group, async_op=async_op)

    def finish():
        if async_op and work is not None:
            work.wait()
        if rank != dst:
            return None
        total = sum(counts)
        out_r = torch.empty(total, dtype=torch.int32, device=dev)
        out_s = torch.empty(total, dtype=torch.int64, device=dev) if with_start else None
        out_e = torch.empty(total, dtype=torch.int64, device=dev)
        s2 = torch.cuda.current_stream(dev).cuda_stream
        at = 0
        for r in range(world):
            if counts[r]:
                check(l.redgpu_records_unpack_dev(
                    dev.index, bufs[r].data_ptr(), counts[r], fmt.rw, fmt.pw,
                    out_r.data_ptr() + 4 * at, (out_s.data_ptr() + 8 * at) if with_start else None,
                    out_e.data_ptr() + 8 * at, C.c_void_p(s2)))
            at += counts[r]
        return out_r, out_s, out_e

    return finish


def assert_equal_shards(n_local: int, device=None, group=None) -> None:
    """equal_counts=True skips the count exchange of gather_outcomes: make sure, once, that the
    promise holds on every rank (a mismatch would hang or corrupt the gather)."""
    world = dist.get_world_size(group)
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device or "cpu")
    all_n = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(all_n, t, group=group)
    counts = [int(x.item()) for x in all_n]
    if any(c != counts[0] for c in counts):
        raise ValueError("shards differ in size (%s): gather with equal_counts=False" % counts)


class StepGather:
    """Gathers EVERY step's per-line Outcomes to rank 0, pipelined against the scans that follow:
    push(step outputs) packs the compact records and starts the collective asynchronously; up to
    `depth` gathers are in flight before the oldest is finished (widened on rank 0).  With RCCL
    the collective runs on the process group's own stream, so the next step's scan overlaps it.
    Every shard must have the same size (checked once, collectively, at construction)."""

    def __init__(self, n_local: int, max_result: int, max_line_len: int, with_start: bool,
                 depth: int = 2, via_host: bool = False, device=None, group=None):
        self.kw = dict(max_result=max_result, max_line_len=max_line_len, equal_counts=True,
                       async_op=True, group=group)
        self.with_start, self.depth, self.via_host = with_start, depth, via_host
        self.inflight = []
        self.last = None      # rank 0: the most recently finished step's (result, start, end)
        self.finished = 0
        assert_equal_shards(n_local, device=device, group=group)

    def push(self, outputs):
        r, s, e = outputs
        if self.via_host:  # gloo rehearsal: records travel through host memory
            r, e = r.cpu(), e.cpu()
            s = s.cpu() if s is not None else None
        self.inflight.append(gather_outcomes(r, s if self.with_start else None, e, **self.kw))
        while len(self.inflight) > self.depth:
            self._finish_one()

    def _finish_one(self):
        got = self.inflight.pop(0)()
        self.finished += 1
        if got is not None:
            self.last = got

    def flush(self):
        while self.inflight:
            self._finish_one()
        return self.last
