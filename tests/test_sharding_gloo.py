"""The N>1 path on CPU: world_size-2 gloo rehearsal of shard -> per-rank outcomes -> compact
gather -> widen on rank 0, compared with the single-rank result.  (Per-rank outcomes come from
the CPU oracle here - there is no GPU in this container; on the GPU the same sharding code
moves the kernels' outputs over RCCL.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from one_amd import sharding
from one_amd import workloads as W
from golden_util import load_dfa


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_lines, stride, out_path):
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blob = load_dfa("uri")
        cpu = O.CpuOracle(blob)
        lo, hi = sharding.shard_range(n_lines, world, rank)
        # every rank generates exactly its own shard (counter-based generator)
        data = W.fixed_lines(hi - lo, stride, 7, plant=W.URI_PLANT, first_line=lo)
        r, s, e = cpu.batch("match", "last", 0, data, stride=stride, n=hi - lo)
        fin = sharding.gather_outcomes(torch.from_numpy(r), torch.from_numpy(s.astype(np.int64)),
                                       torch.from_numpy(e.astype(np.int64)), max_result=1,
                                       max_line_len=stride)
        got = fin()
        fin2 = sharding.gather_outcomes(torch.from_numpy(r), None,
                                        torch.from_numpy(e.astype(np.int64)), max_result=70000,
                                        max_line_len=1 << 20, async_op=True,
                                        equal_counts=(n_lines % world == 0))
        got2 = fin2()
        if rank == 0:
            np.savez(out_path, r=got[0].numpy(), s=got[1].numpy(), e=got[2].numpy(),
                     r2=got2[0].numpy(), e2=got2[2].numpy())
        else:
            assert got is None and got2 is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_lines", [2000, 2001])
def test_two_rank_gather_equals_single_rank(tmp_path, n_lines):
    import oracle as O
    stride, world = 64, 2
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(world, _free_port(), n_lines, stride, out), nprocs=world, join=True)
    g = np.load(out)
    data = W.fixed_lines(n_lines, stride, 7, plant=W.URI_PLANT)
    r, s, e = O.CpuOracle(load_dfa("uri")).batch("match", "last", 0, data, stride=stride,
                                                 n=n_lines)
    assert np.array_equal(g["r"], r) and int((r > 0).sum()) > 100
    assert np.array_equal(g["s"].astype(np.uint64), s)
    assert np.array_equal(g["e"].astype(np.uint64), e)
    assert np.array_equal(g["r2"], r) and np.array_equal(g["e2"].astype(np.uint64), e)


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            rs = [sharding.shard_range(n, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    offs = torch.tensor([0, 10, 10, 500, 520, 1000, 1001, 2000], dtype=torch.int64)
    rs = [sharding.shard_range_by_bytes(offs, 2, r) for r in range(2)]
    assert rs[0][0] == 0 and rs[0][1] == rs[1][0] and rs[1][1] == 7
    assert abs(int(offs[rs[0][1]]) - 1000) <= 500


def test_record_format_roundtrip():
    g = torch.Generator().manual_seed(1)
    for max_res, max_len in ((1, 64), (200, 255), (300, 4096), (100, 70000), (1 << 20, 1 << 33)):
        n = 1000
        res = torch.randint(0, max_res + 1, (n,), generator=g, dtype=torch.int64).to(torch.int32)
        st = torch.randint(0, max_len + 1, (n,), generator=g, dtype=torch.int64)
        en = torch.randint(0, max_len + 1, (n,), generator=g, dtype=torch.int64)
        for with_start in (True, False):
            fmt = sharding.RecordFormat(max_res, max_len, with_start)
            r2, s2, e2 = fmt.unpack(fmt.pack(res, st if with_start else None, en))
            assert torch.equal(r2, res) and torch.equal(e2, en)
            assert (s2 is None) == (not with_start) and (s2 is None or torch.equal(s2, st))


def _worker_steps(rank, world, port, out_path):
    """Every step's outcomes gathered (StepGather, pipelined) + ragged shards of unequal line
    counts (balanced by bytes) through gather_outcomes."""
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blob = load_dfa("uri")
        cpu = O.CpuOracle(blob)
        n_per, stride, steps = 700, 64, 5
        sg = sharding.StepGather(n_per, max_result=1, max_line_len=stride, with_start=True, depth=2)
        got = []
        for k in range(steps):
            data = W.fixed_lines(n_per, stride, 100 + k, plant=W.URI_PLANT,
                                 first_line=rank * n_per)
            r, s, e = cpu.batch("match", "last", 0, data, stride=stride, n=n_per)
            sg.push((torch.from_numpy(r), torch.from_numpy(s.astype(np.int64)),
                     torch.from_numpy(e.astype(np.int64))))
            if rank == 0 and sg.last is not None:
                got.append(tuple(t.clone() for t in sg.last))
        sg.flush()
        assert sg.finished == steps
        last = sg.last
        # unequal shards are refused up front when equal counts were promised
        try:
            sharding.StepGather(n_per + rank, 1, stride, True)
            mismatch_caught = False
        except ValueError:
            mismatch_caught = True
        assert mismatch_caught
        # ragged lines: shards by BYTES, unequal line counts, widths by the longest line
        data, offsets = W.ragged_lines(3001, 1, 900, 77, heads=[W.URI_PLANT], head_every=3)
        lo, hi = sharding.shard_range_by_bytes(torch.from_numpy(offsets.astype(np.int64)), world, rank)
        so = offsets[lo:hi + 1]
        r, s, e = cpu.batch("match", "last", 0, data[int(so[0]):int(so[-1])], offsets=so - so[0])
        fin = sharding.gather_outcomes(torch.from_numpy(r), torch.from_numpy(s.astype(np.int64)),
                                       torch.from_numpy(e.astype(np.int64)), max_result=1,
                                       max_line_len=900)
        rag = fin()
        if rank == 0:
            np.savez(out_path, r=last[0].numpy(), s=last[1].numpy(), e=last[2].numpy(),
                     rr=rag[0].numpy(), rs=rag[1].numpy(), re=rag[2].numpy(), nshard=hi - lo)
    finally:
        dist.destroy_process_group()


def test_every_step_gathered_and_ragged_shards_by_bytes(tmp_path):
    import oracle as O
    world = 2
    out = str(tmp_path / "steps.npz")
    mp.spawn(_worker_steps, args=(world, _free_port(), out), nprocs=world, join=True)
    g = np.load(out)
    cpu = O.CpuOracle(load_dfa("uri"))
    n_per, stride = 700, 64
    # the last step (seed 104), both ranks' shards in rank order
    data = W.fixed_lines(n_per * world, stride, 104, plant=W.URI_PLANT)
    r, s, e = cpu.batch("match", "last", 0, data, stride=stride, n=n_per * world)
    assert np.array_equal(g["r"], r) and np.array_equal(g["s"].astype(np.uint64), s)
    assert np.array_equal(g["e"].astype(np.uint64), e)
    data, offsets = W.ragged_lines(3001, 1, 900, 77, heads=[W.URI_PLANT], head_every=3)
    r, s, e = cpu.batch("match", "last", 0, data, offsets=offsets)
    assert np.array_equal(g["rr"], r) and np.array_equal(g["rs"].astype(np.uint64), s)
    assert np.array_equal(g["re"].astype(np.uint64), e)
    assert int(g["nshard"]) != 3001 - int(g["nshard"])  # the byte-balanced shards differ in lines
