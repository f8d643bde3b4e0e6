"""GPU parity of redgpu_match_batches_dev / redgpu_check_batches_dev: several device-resident
batches per call (the caller's loop over its inputs, tools/bench.cpp:60-71), runs of
streaming-kernel batches folded into one launch (k_stream_multi.h).  Every batch's Outcomes
against the CPU oracle, bit-exact, and against the single-batch entry point."""
import numpy as np
import pytest

import one_amd
import oracle as O
from golden_util import load_dfa
from one_amd import workloads as W

pytestmark = pytest.mark.gpu


def _lines(name, n, stride, seed):
    if name == "syn256":
        return W.fixed_lines(n, stride, seed, alphabet=False)
    plant = {"uri": W.URI_PLANT, "dotstar_err": b"an error", "err": b"error"}[name]
    buf = W.fixed_lines(n, stride, seed, plant=plant, plant_every=3, plant_at=min(7, stride - 8))
    v = buf.reshape(n, stride)
    v[1::5, :len(plant[:stride])] = np.frombuffer(plant[:stride], dtype=np.uint8)
    return buf


def _same(got, exp, what):
    r, s, e = got
    er, es, ee = exp
    assert np.array_equal(r.cpu().numpy(), er), what
    if s is not None:
        assert np.array_equal(s.cpu().numpy().astype(np.uint64), es), what
    if e is not None:
        assert np.array_equal(e.cpu().numpy().astype(np.uint64), ee), what


# line counts: tiles of 1024 lines - whole tiles, a ragged last tile, one line, fewer lines than
# a wave; more tiles in all than the 256 workgroups of the grid and fewer
SHAPES = [
    (64, [300000, 1024 * 7 + 5, 1, 63, 262144, 2048]),
    (128, [150000, 1, 40000, 65537]),
    (4096, [3000, 2500, 1025, 700]),
    (64, [200, 100]),          # fewer tiles than CUs: runs as single launches
    (192, [90000, 90001, 17]),
]


@pytest.mark.parametrize("stride,counts", SHAPES)
@pytest.mark.parametrize("name", ["syn256", "uri", "dotstar_err"])
def test_match_batches_vs_oracle(name, stride, counts):
    import torch
    blob = load_dfa(name)
    exe = one_amd.Executable(blob)
    cpu = O.CpuOracle(blob)
    hosts = [_lines(name, n, stride, seed=7 * k + n % 1000) for k, n in enumerate(counts)]
    devs = [torch.from_numpy(h).cuda() for h in hosts]
    many = sum(-(-n // 1024) for n in counts) >= 256
    for si in (4, 5):
        for lead in (0, 1):
            for want_start in (True, False):
                outs = one_amd.match_batches(exe, devs, si, lead, stride=stride,
                                             want_start=want_start)
                k = one_amd.last_kernel()
                if many:
                    assert k.startswith("k_stream_multi<"), k
                torch.cuda.synchronize()
                for i, (h, n) in enumerate(zip(hosts, counts)):
                    exp = cpu.batch("match", si, lead, h, stride=stride, n=n, threads=4)
                    _same(outs[i], exp, (name, stride, si, lead, want_start, i, k))
                    one = one_amd.match_batch(exe, devs[i], si, lead, stride=stride, n=n,
                                              want_start=want_start)
                    for a, b in zip(outs[i], one):
                        assert (a is None and b is None) or bool((a == b).all())
            res = one_amd.check_batches(exe, devs, si, 0, stride=stride)
            torch.cuda.synchronize()
            for i, (h, n) in enumerate(zip(hosts, counts)):
                cr = cpu.batch("check", si, 0, h, stride=stride, n=n, threads=4)[0]
                assert np.array_equal(res[i].cpu().numpy(), cr), (name, stride, si, i)


def test_match_batches_mixed_list_falls_back_per_batch():
    """A list the streaming kernel cannot take as one launch - two strides, a ragged batch, an
    early-exit style - gives what the single-batch entry point gives, batch by batch."""
    import torch
    blob = load_dfa("uri")
    exe = one_amd.Executable(blob)
    cpu = O.CpuOracle(blob)
    a = _lines("uri", 300000, 64, 1)
    b = _lines("uri", 300000, 64, 2)
    c = _lines("uri", 9000, 4096, 3)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    # same stride run + a different stride: descriptors built by hand through the C-ABI struct
    from one_amd import _lib
    descs, outs = one_amd.batch_descs([da, db], stride=64)
    d2, o2 = one_amd.batch_descs([dc], stride=4096)
    arr = (_lib.BatchDesc * 3)(descs[0], descs[1], d2[0])
    rc = _lib.lib().redgpu_match_batches_dev(exe._h, 4, 0, arr, 3,
                                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.lib().redgpu_last_error()
    torch.cuda.synchronize()
    _same(outs[0], cpu.batch("match", 4, 0, a, stride=64, n=300000, threads=4), "a")
    _same(outs[1], cpu.batch("match", 4, 0, b, stride=64, n=300000, threads=4), "b")
    _same(o2[0], cpu.batch("match", 4, 0, c, stride=4096, n=9000, threads=4), "c")
    # an early-exit style: every batch on its own kernel
    outs = one_amd.match_batches(exe, [da, db], 1, 0, stride=64)
    torch.cuda.synchronize()
    assert not one_amd.last_kernel().startswith("k_stream_multi")
    _same(outs[0], cpu.batch("match", 1, 0, a, stride=64, n=300000, threads=4), "instant a")
    _same(outs[1], cpu.batch("match", 1, 0, b, stride=64, n=300000, threads=4), "instant b")
    # ragged lines in the list
    data, offsets = W.ragged_lines(50000, 1, 200, 5)
    dd, do = torch.from_numpy(data).cuda(), torch.from_numpy(offsets.astype(np.int64)).cuda()
    outs = one_amd.match_batches(exe, [(dd, do), (dd, do)], 4, 0)
    torch.cuda.synchronize()
    exp = cpu.batch("match", 4, 0, data, offsets=offsets, threads=4)
    _same(outs[0], exp, "ragged 0")
    _same(outs[1], exp, "ragged 1")


def test_match_batches_arguments():
    import torch
    from one_amd import _lib
    exe = one_amd.Executable(load_dfa("uri"))
    l = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    assert l.redgpu_match_batches_dev(exe._h, 4, 0, None, 0, st) == 0          # empty list
    assert l.redgpu_match_batches_dev(exe._h, 4, 0, None, 2, st) == _lib.EAPI  # null descriptors
    d = torch.zeros(64 * 10, dtype=torch.uint8, device="cuda")
    descs, outs = one_amd.batch_descs([d], stride=64)
    assert l.redgpu_match_batches_dev(exe._h, 9, 0, descs, 1, st) == _lib.EEXEC  # bad style
    descs[0].result = None
    assert l.redgpu_match_batches_dev(exe._h, 4, 0, descs, 1, st) == _lib.EAPI
    descs[0].n = 0   # an empty batch is skipped, whatever its pointers
    assert l.redgpu_match_batches_dev(exe._h, 4, 0, descs, 1, st) == 0
    torch.cuda.synchronize()


def test_full_size_config1_twenty_batches_one_launch():
    """BASELINE configs[1] as bench.py times it: 20 batches of 2^20 x 64 B through ONE call,
    every batch's Outcomes line for line against the CPU oracle (6 rotating inputs)."""
    import torch
    blob = load_dfa("syn256")
    exe = one_amd.Executable(blob)
    cpu = O.CpuOracle(blob)
    n, L, nbuf, steps = 1 << 20, 64, 6, 20
    hosts = [W.fixed_lines(n, L, 42 + k, alphabet=False) for k in range(nbuf)]
    devs = [torch.from_numpy(h).cuda() for h in hosts]
    exps = [cpu.batch("match", 4, 0, h, stride=L, n=n, threads=8) for h in hosts]
    outs = one_amd.match_batches(exe, [devs[i % nbuf] for i in range(steps)], 4, 0, stride=L)
    assert one_amd.last_kernel() == "k_stream_multi<last,start,end>"
    torch.cuda.synchronize()
    for i in range(steps):
        _same(outs[i], exps[i % nbuf], ("step", i))


LEAN_SHAPES = [(512, 70000), (4096, 9000), (1024, 33000), (576, 60001), (8192, 4097)]


@pytest.mark.parametrize("chains", [2, 4])
@pytest.mark.parametrize("stride,n", LEAN_SHAPES)
@pytest.mark.parametrize("name", ["syn256", "uri", "dotstar_err", "newyork"])
def test_lean_step_long_lines_vs_oracle(name, stride, n, chains):
    """REDGPU_F_FORCE_LEAN: lines of 512 bytes and more take k_stream_multi's deferred-bookkeeping
    step (k_stream_lean.h), two or four lines per lane: per byte it only notes in which 16-byte piece the last accept and the
    last "left the initial state" lie, the two pieces are re-walked exactly at the line's end.
    Styles Last / Full of match and check, with and without start, with the leader, against the
    oracle (Matcher.h:413-495); strides that are / are not multiples of 128."""
    import torch
    blob = load_dfa(name)
    exe = one_amd.Executable(blob, no_chunking=True, force_lean=True, lean_chains=chains)
    cpu = O.CpuOracle(blob)
    if name == "syn256":
        h = W.fixed_lines(n, stride, stride + n, alphabet=False)
    else:
        plant = {"uri": W.URI_PLANT, "dotstar_err": b"an error", "newyork": b"New York"}[name]
        h = W.fixed_lines(n, stride, stride + n, plant=plant, plant_every=3, plant_at=stride - 200)
        v = h.reshape(n, stride)
        v[1::5, :len(plant)] = np.frombuffer(plant, dtype=np.uint8)          # at the line's start
        v[2::7, stride - len(plant):] = np.frombuffer(plant, dtype=np.uint8)  # ... and at its very end
        v[3::11, 15:15 + len(plant)] = np.frombuffer(plant, dtype=np.uint8)   # across a piece border
    d = torch.from_numpy(h).cuda()
    for si in (4, 5):
        for lead in (0, 1):
            exp = cpu.batch("match", si, lead, h, stride=stride, n=n, threads=8)
            got = one_amd.match_batch(exe, d, si, lead, stride=stride, n=n)
            k = one_amd.last_kernel()
            assert "lean" in k, k
            torch.cuda.synchronize()
            _same(got, exp, (name, stride, si, lead, k))
            if si == 4:
                got = one_amd.match_batch(exe, d, si, lead, stride=stride, n=n, want_start=False)
                assert "lean" in one_amd.last_kernel()
                torch.cuda.synchronize()
                _same(got, (exp[0], None, exp[2]), (name, stride, si, lead, "no start"))
        cr = cpu.batch("check", si, 0, h, stride=stride, n=n, threads=8)[0]
        r = one_amd.check_batch(exe, d, si, 0, stride=stride, n=n)
        torch.cuda.synchronize()
        assert np.array_equal(r.cpu().numpy(), cr), (name, stride, si, one_amd.last_kernel())
    # two such batches in one call
    outs = one_amd.match_batches(exe, [d, d], 4, 0, stride=stride)
    torch.cuda.synchronize()
    exp = cpu.batch("match", 4, 0, h, stride=stride, n=n, threads=8)
    _same(outs[0], exp, "two batches, first")
    _same(outs[1], exp, "two batches, second")


@pytest.mark.parametrize("stride,n", [(64, 70000), (128, 9000), (4096, 2100), (192, 5000)])
@pytest.mark.parametrize("name", ["err", "newyork", "aab"])
def test_check_with_leader_on_the_streaming_kernel(name, stride, n):
    """check<styLast / styFull, true> (include/Matcher.h:370-381: compare the leader, start in the
    post-leader state) over a DFA whose leader is its table's own forced prefix chain runs on
    k_stream - the plain walk, deaf to whatever accepts up to the leader's end.  The quirk that
    makes this more than check<..., false>: the post-leader state's own result is never looked at
    unless the line ends there - ERR ('error', leader = the whole pattern) reports 0 for
    'error foo'.  Lines that start with the pattern, hold it later, or hold a broken prefix."""
    import torch
    blob = load_dfa(name)
    exe = one_amd.Executable(blob, force_stream=True, no_chunking=True)
    cpu = O.CpuOracle(blob)
    plant = {"err": b"error", "newyork": b"New York", "aab": b"aab"}[name]
    h = W.fixed_lines(n, stride, 5 * stride + n, plant=plant, plant_every=3, plant_at=20)
    v = h.reshape(n, stride)
    v[1::4, :len(plant)] = np.frombuffer(plant, dtype=np.uint8)              # the pattern at the start
    v[2::9, :len(plant) - 1] = np.frombuffer(plant[:-1], dtype=np.uint8)     # all of it but its last byte
    v[5::13, :len(plant)] = np.frombuffer(plant, dtype=np.uint8)
    v[5::13, len(plant):2 * len(plant)] = np.frombuffer(plant, dtype=np.uint8)  # twice in a row
    d = torch.from_numpy(h).cuda()
    for si in (4, 5):
        cr = cpu.batch("check", si, 1, h, stride=stride, n=n, threads=4)[0]
        r = one_amd.check_batch(exe, d, si, 1, stride=stride, n=n)
        k = one_amd.last_kernel()
        torch.cuda.synchronize()
        assert k.startswith("k_stream<"), k
        assert np.array_equal(r.cpu().numpy(), cr), (name, stride, si, k)
    # the quirk is in the data: with and without the leader differ somewhere for ERR
    if name == "err":
        a = cpu.batch("check", 4, 1, h, stride=stride, n=n, threads=4)[0]
        b = cpu.batch("check", 4, 0, h, stride=stride, n=n, threads=4)[0]
        assert not np.array_equal(a, b)
