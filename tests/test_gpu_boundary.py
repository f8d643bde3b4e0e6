"""GPU tests of the boundary itself (all through the C-ABI): the reference's threading contract
(doc/Performance.md:81-84, tools/thr_red.cpp:84-91 - N host threads over one shared handle), the
chunk-pipelined host-buffer path, per-thread staging, and the several-GPU group (shards, compact
record gather) rehearsed on one device."""
import threading

import numpy as np
import pytest

import one_amd
import oracle as O
from golden_util import load_dfa
from one_amd import _lib
from one_amd import workloads as W

pytestmark = pytest.mark.gpu


def _eq(got, exp):
    return all((g is None and e is None) or np.array_equal(np.asarray(g).astype(np.uint64),
                                                           np.asarray(e).astype(np.uint64))
               for g, e in zip(got, exp))


def test_eight_host_threads_share_one_handle():
    """8 host threads hammer ONE redgpu_dfa: host-buffer calls (fixed and ragged lines, two
    handles' worth of table kinds) and device-pointer calls on distinct streams, every result
    compared with the oracle.  Staging and scratch are per thread / per stream."""
    import torch
    blob = load_dfa("syn256")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    uri = load_dfa("uri")
    exe2, cpu2 = one_amd.Executable(uri), O.CpuOracle(uri)
    n, stride = 6000, 64
    fixed = [W.fixed_lines(n, stride, 100 + t, alphabet=False) for t in range(8)]
    ragged = [W.ragged_lines(3000 + 17 * t, 1, 300, 200 + t, heads=[W.URI_PLANT], head_every=3)
              for t in range(8)]
    exp_fixed = [cpu.batch("match", 4, 0, f, stride=stride, n=n, threads=4) for f in fixed]
    exp_ragged = [cpu2.batch("match", 4, 0, d, offsets=o, threads=4) for d, o in ragged]
    exp_scan = [cpu2.batch("scan", 1, 1, d, offsets=o, threads=4)[0] for d, o in ragged]
    dev_in = [torch.from_numpy(f).cuda() for f in fixed]
    errors = []

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            for rep in range(6):
                got = one_amd.match_batch(exe, fixed[t], 4, 0, stride=stride, n=n)
                assert _eq(got, exp_fixed[t]), ("host fixed", t, rep)
                d, o = ragged[t]
                got = one_amd.match_batch(exe2, d, 4, 0, offsets=o)
                assert _eq(got, exp_ragged[t]), ("host ragged", t, rep)
                assert np.array_equal(one_amd.scan_batch(exe2, d, 1, 1, offsets=o), exp_scan[t])
                with torch.cuda.stream(stream):
                    r, s, e = one_amd.match_batch(exe, dev_in[t], 4, 0, stride=stride, n=n)
                    stream.synchronize()
                assert _eq((r.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy()), exp_fixed[t]), \
                    ("dev fixed", t, rep)
            _lib.lib().redgpu_thread_release()
        except BaseException as ex:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    # scratch is pooled per thread and (device, stream), bounded, and released with the thread
    assert _lib.lib().redgpu_scratch_entries() <= 64


def test_threads_share_one_stream_for_ragged_dev_calls():
    """the ragged launches are multi-kernel sequences that carry state in a scratch
    buffer (tail pad, length buckets).  Two host threads issuing _dev calls on the SAME stream (the
    default one) with batches of different sizes must not see each other's scratch - it is kept
    per thread - whatever the interleaving; every result against the oracle."""
    import torch
    blob = load_dfa("uri")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    sets = []
    for t, n in enumerate((3000, 70000, 500, 260000)):
        d, o = W.ragged_lines(n, 1, 300 if t % 2 else 90, 31 + t, heads=[W.URI_PLANT], head_every=3)
        exp = cpu.batch("match", 4, 0, d, offsets=o, threads=4)
        sets.append((torch.from_numpy(d).cuda(), torch.from_numpy(o.astype(np.int64)).cuda(), exp))
    torch.cuda.synchronize()
    errors = []

    def worker(t):
        try:
            for rep in range(25):
                dd, do, exp = sets[(t + rep) % len(sets)]
                r, s, e = one_amd.match_batch(exe, dd, 4, 0, offsets=do)   # torch's default stream
                got = (r.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy())
                assert _eq(got, exp), ("thread", t, "rep", rep, one_amd.last_kernel())
        except BaseException as ex:  # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_short_lived_threads_do_not_accumulate_scratch():
    """The shape of tools/thr_red.cpp: workers come and go.  Each one's staging (streams, device
    buffers, the ragged launches' scratch on those streams) goes with it."""
    blob = load_dfa("uri")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    d, o = W.ragged_lines(20000, 1, 200, 9, heads=[W.URI_PLANT], head_every=4)
    exp = cpu.batch("match", 4, 0, d, offsets=o, threads=4)
    base = _lib.lib().redgpu_scratch_entries()
    errors = []

    def worker():
        try:
            assert _eq(one_amd.match_batch(exe, d, 4, 0, offsets=o), exp)
        except BaseException as ex:  # noqa: BLE001
            errors.append(repr(ex))

    for _ in range(12):
        th = threading.Thread(target=worker)
        th.start()
        th.join()
    assert not errors, errors
    assert _lib.lib().redgpu_scratch_entries() <= base + 1


@pytest.mark.parametrize("shape", ["fixed", "ragged"])
def test_host_path_chunk_pipeline_vs_oracle(shape):
    """Host buffers of several chunks (> 64 MiB): chunks alternate between the thread's two
    streams; the offsets of a ragged chunk are rebased on the device to the chunk's own buffer."""
    blob = load_dfa("syn256")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    if shape == "fixed":
        n, stride = (1 << 20) + 777, 128
        data = W.fixed_lines(n, stride, 77, alphabet=False)
        got = one_amd.match_batch(exe, data, 4, 0, stride=stride, n=n)
        exp = cpu.batch("match", 4, 0, data, stride=stride, n=n, threads=8)
        assert _eq(got, exp)
        assert np.array_equal(one_amd.check_batch(exe, data, 5, 0, stride=stride, n=n),
                              cpu.batch("check", 5, 0, data, stride=stride, n=n, threads=8)[0])
    else:
        data, offs = W.ragged_lines(1 << 20, 1, 300, 78, alphabet=False)
        assert int(offs[-1]) > (3 * 32 << 20)
        got = one_amd.match_batch(exe, data, 4, 0, offsets=offs)
        exp = cpu.batch("match", 4, 0, data, offsets=offs, threads=8)
        assert _eq(got, exp)
        # a sub-batch whose first offset is not 0
        sub = offs[1000:900000]
        got = one_amd.match_batch(exe, data, 4, 0, offsets=sub)
        exp = cpu.batch("match", 4, 0, data, offsets=sub, threads=8)
        assert _eq(got, exp)


def test_list_verbs_validate_their_arguments():
    """collect / matchAll host entry points reject what the other verbs reject."""
    blob = load_dfa("uri")
    exe = one_amd.Executable(blob)
    data = np.frombuffer(b"x" * 100, dtype=np.uint8)
    bad = np.array([0, 50, 40, 100], dtype=np.uint64)
    with pytest.raises(one_amd.RedExceptApi):
        one_amd.collect_batch(exe, data, 4, offsets=bad)
    with pytest.raises(one_amd.RedExceptApi):
        one_amd.match_all_batch(exe, data, 4, offsets=bad)
    # n * cap overflow is refused before anything is sized by it
    offs = np.array([0, 50, 100], dtype=np.uint64)
    small = np.zeros(16, dtype=np.uint64)
    rc = _lib.lib().redgpu_collect_batch(exe._h, data.ctypes.data, offs.ctypes.data, 0, 2, 1 << 62,
                                         small.ctypes.data, small.ctypes.data, small.ctypes.data,
                                         small.ctypes.data)
    assert rc == _lib.ELIMIT


@pytest.mark.parametrize("ndev", [1, 3])
def test_group_host_batches_vs_oracle(ndev):
    """redgpu_group_batch: shards by lines (fixed) and by bytes (ragged), one host thread per
    device - the same device named ndev times here - results in the caller's arrays."""
    blob = load_dfa("uri")
    cpu = O.CpuOracle(blob)
    grp = one_amd.Group(blob, [0] * ndev)
    n, stride = 50001, 64
    data = W.fixed_lines(n, stride, 5, plant=W.URI_PLANT)
    cuts = grp.plan(n, stride=stride)
    assert cuts[0] == 0 and cuts[-1] == n and np.all(np.diff(cuts.astype(np.int64)) >= n // ndev)
    for verb, style, lead in (("match", 4, 0), ("match", 5, 1), ("search", 4, 0)):
        got = grp.batch(verb, data, style, lead, stride=stride, n=n)
        exp = cpu.batch(verb, style, lead, data, stride=stride, n=n, threads=8)
        assert _eq(got, exp), (verb, style, lead)
    assert np.array_equal(grp.batch("check", data, 4, 0, stride=stride, n=n)[0],
                          cpu.batch("check", 4, 0, data, stride=stride, n=n, threads=8)[0])
    rd, ro = W.ragged_lines(40000, 1, 400, 6, heads=[W.URI_PLANT], head_every=3)
    cuts = grp.plan(len(ro) - 1, offsets=ro)
    per = [int(ro[int(cuts[g + 1])] - ro[int(cuts[g])]) for g in range(ndev)]
    assert max(per) - min(per) <= 400 * 2  # balanced by BYTES to within a line or two
    got = grp.batch("match", rd, 4, 0, offsets=ro)
    assert _eq(got, cpu.batch("match", 4, 0, rd, offsets=ro, threads=8))
    assert np.array_equal(grp.batch("scan", rd, 1, 0, offsets=ro)[0],
                          cpu.batch("scan", 1, 0, rd, offsets=ro, threads=8)[0])


@pytest.mark.parametrize("ndev,gather", [(1, "peer"), (3, "peer"), (1, "rccl")])
def test_group_device_shards_gathered_on_root(ndev, gather):
    """redgpu_group_batch_dev: device-resident shards scanned on per-device streams, compact
    records (1-byte results, 1/2-byte positions here) moved to the root and widened there."""
    import torch
    for name, stride in (("syn256", 64), ("uri", 4096)):
        blob = load_dfa(name)
        cpu = O.CpuOracle(blob)
        grp = one_amd.Group(blob, [0] * ndev)
        counts = [3000 + 11 * g for g in range(ndev)]
        hosts = [W.fixed_lines(c, stride, 40 + g, alphabet=(name != "syn256"),
                               plant=(W.URI_PLANT if name == "uri" else None))
                 for g, c in enumerate(counts)]
        shards = [torch.from_numpy(h).cuda() for h in hosts]
        for rep in range(3):  # the root record buffer is reused call after call
            r, s, e = grp.batch_dev("match", shards, 4, 0, stride=stride, gather=gather)
            torch.cuda.synchronize()
            exp = [cpu.batch("match", 4, 0, h, stride=stride, n=c, threads=4)
                   for h, c in zip(hosts, counts)]
            cat = [np.concatenate([x[k] for x in exp]) for k in range(3)]
            assert _eq((r.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy()), cat), (name, rep)
        rc = grp.batch_dev("check", shards, 5, 0, stride=stride, gather=gather)[0]
        torch.cuda.synchronize()
        expc = np.concatenate([cpu.batch("check", 5, 0, h, stride=stride, n=c, threads=4)[0]
                               for h, c in zip(hosts, counts)])
        assert np.array_equal(rc.cpu().numpy(), expc)
    # ragged shards: offsets relative to each shard's data
    blob = load_dfa("uri")
    cpu = O.CpuOracle(blob)
    grp = one_amd.Group(blob, [0] * ndev)
    parts = [W.ragged_lines(5000 + g, 1, 700, 60 + g, heads=[W.URI_PLANT], head_every=3)
             for g in range(ndev)]
    shards = [(torch.from_numpy(d).cuda(), torch.from_numpy(o.astype(np.int64)).cuda())
              for d, o in parts]
    r, s, e = grp.batch_dev("match", shards, 4, 0, gather=gather)
    torch.cuda.synchronize()
    exp = [cpu.batch("match", 4, 0, d, offsets=o, threads=4) for d, o in parts]
    cat = [np.concatenate([x[k] for x in exp]) for k in range(3)]
    assert _eq((r.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy()), cat)


def test_group_device_shards_are_stream_ordered():
    """the shards' scans run on the group's private streams.  With shard_streams (the
    Python wrapper passes torch's current stream of each device) the scan waits for the op that
    PRODUCES the shard - queued, not finished, when batch_dev is called - and the producer stream
    waits for the scan before the shard's memory may be reused: no synchronize anywhere between
    producing, scanning and overwriting the shard."""
    import torch
    blob = load_dfa("uri")
    cpu = O.CpuOracle(blob)
    grp = one_amd.Group(blob, [0, 0])
    stride, n = 4096, 6000
    hosts = [W.fixed_lines(n, stride, 90 + g, alphabet=True, plant=W.URI_PLANT) for g in range(3)]
    pinned = [torch.from_numpy(h).pin_memory() for h in hosts]
    exp = [cpu.batch("match", 4, 0, h, stride=stride, n=n, threads=4) for h in hosts]
    bufs = [torch.empty(n * stride, dtype=torch.uint8, device="cuda") for _ in range(2)]
    outs = []
    for rep in range(6):
        a, b = rep % 3, (rep + 1) % 3
        bufs[0].copy_(pinned[a], non_blocking=True)   # queued on the current stream, not waited for
        bufs[1].copy_(pinned[b], non_blocking=True)
        r, s, e = grp.batch_dev("match", bufs, 4, 0, stride=stride)
        outs.append((a, b, r.clone(), s.clone(), e.clone()))
        bufs[0].zero_()                               # overwritten right behind the call
        bufs[1].zero_()
    torch.cuda.synchronize()
    for a, b, r, s, e in outs:
        cat = [np.concatenate([exp[a][k], exp[b][k]]) for k in range(3)]
        assert _eq((r.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy()), cat), (a, b)


@pytest.mark.parametrize("gather", ["peer", "rccl"])
def test_group_two_distinct_devices(gather):
    """The multi-device data path of group.cpp - hipMemcpyPeerAsync / ncclSend + ncclRecv between
    DISTINCT devices: runs wherever two GPUs are visible, skipped on a one-GPU box (where every
    other group test names device 0 several times)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    blob = load_dfa("uri")
    cpu = O.CpuOracle(blob)
    grp = one_amd.Group(blob, [0, 1])
    stride = 4096
    counts = [5000, 5011]
    hosts = [W.fixed_lines(c, stride, 70 + g, alphabet=True, plant=W.URI_PLANT)
             for g, c in enumerate(counts)]
    shards = [torch.from_numpy(h).to("cuda:%d" % g) for g, h in enumerate(hosts)]
    for rep in range(3):
        r, s, e = grp.batch_dev("match", shards, 4, 0, stride=stride, gather=gather)
        torch.cuda.synchronize(0)
        exp = [cpu.batch("match", 4, 0, h, stride=stride, n=c, threads=4) for h, c in zip(hosts, counts)]
        cat = [np.concatenate([x[k] for x in exp]) for k in range(3)]
        assert _eq((r.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy()), cat), (gather, rep)


@pytest.mark.parametrize("rw,pw,with_start", [(1, 1, True), (1, 2, True), (2, 4, False),
                                              (4, 8, True)])
def test_record_planes_pack_and_unpack_roundtrip(rw, pw, with_start):
    """redgpu_records_pack_dev / _unpack_dev: the compact record planes on their own (what one
    process per GPU sends over RCCL), every width, odd counts (planes start 16-byte aligned)."""
    import ctypes as C
    import torch
    l = _lib.lib()
    rng = np.random.default_rng(rw * 10 + pw)
    for n in (1, 63, 4097, 100003):
        res = rng.integers(0, min(2 ** (8 * rw), 2 ** 31), n, dtype=np.int64).astype(np.int32)
        hi = 2 ** (8 * pw) if pw < 8 else 2 ** 63
        st = rng.integers(0, hi, n, dtype=np.uint64 if pw == 8 else np.int64).astype(np.int64)
        en = rng.integers(0, hi, n, dtype=np.uint64 if pw == 8 else np.int64).astype(np.int64)
        d_res, d_st, d_en = (torch.from_numpy(x).cuda() for x in (res, st, en))
        nbytes = l.redgpu_records_bytes(n, rw, pw, int(with_start))
        assert nbytes >= n * (rw + pw * (2 if with_start else 1)) and nbytes % 16 == 0
        rec = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert l.redgpu_records_pack_dev(0, d_res.data_ptr(), d_st.data_ptr() if with_start else None,
                                         d_en.data_ptr(), n, rw, pw, rec.data_ptr(), stream) == 0
        o_res = torch.empty(n, dtype=torch.int32, device="cuda")
        o_st = torch.empty(n, dtype=torch.int64, device="cuda")
        o_en = torch.empty(n, dtype=torch.int64, device="cuda")
        assert l.redgpu_records_unpack_dev(0, rec.data_ptr(), n, rw, pw, o_res.data_ptr(),
                                           o_st.data_ptr() if with_start else None,
                                           o_en.data_ptr(), stream) == 0
        torch.cuda.synchronize()
        assert np.array_equal(o_res.cpu().numpy(), res)
        assert np.array_equal(o_en.cpu().numpy(), en)
        if with_start:
            assert np.array_equal(o_st.cpu().numpy(), st)
        # the plane layout itself: results first, little-endian
        assert np.array_equal(rec[: n * rw].cpu().numpy().view({1: np.uint8, 2: np.uint16, 4: np.uint32}[rw]),
                              res.astype({1: np.uint8, 2: np.uint16, 4: np.uint32}[rw]))
    assert l.redgpu_records_pack_dev(0, None, None, None, 5, 3, 2, None, None) == _lib.EAPI


def test_gather_outcomes_native_records_over_rccl_one_rank(tmp_path):
    """one_amd.sharding.gather_outcomes on DEVICE tensors (the bench's N > 1 path): records
    packed and widened by the library's kernels, moved by torch.distributed's nccl (= RCCL)
    backend - one rank here; the two-rank form of the same function runs over gloo on the CPU."""
    import torch
    import torch.distributed as dist
    from one_amd import sharding
    dist.init_process_group("nccl", init_method="file://%s" % (tmp_path / "rdv"), rank=0,
                            world_size=1)
    try:
        blob = load_dfa("uri")
        cpu = O.CpuOracle(blob)
        exe = one_amd.Executable(blob)
        host = W.fixed_lines(5000, 4096, 9, alphabet=True, plant=W.URI_PLANT)
        data = torch.from_numpy(host).cuda()
        r, s, e = one_amd.match_batch(exe, data, 4, 0, stride=4096, n=5000)  # device tensors
        assert r.is_cuda and s.dtype == torch.int64
        exp = cpu.batch("match", 4, 0, host, stride=4096, n=5000, threads=4)
        for eq in (True, False):
            for with_start in (True, False):
                fin = sharding.gather_outcomes(r, s if with_start else None, e,
                                               max_result=exe.info["max_result"], max_line_len=4096,
                                               equal_counts=eq, async_op=True)
                gr, gs, ge = fin()
                torch.cuda.synchronize()
                assert gr.dtype == torch.int32 and ge.dtype == torch.int64
                assert np.array_equal(gr.cpu().numpy(), exp[0])
                assert np.array_equal(ge.cpu().numpy().astype(np.uint64), exp[2].astype(np.uint64))
                if with_start:
                    assert np.array_equal(gs.cpu().numpy().astype(np.uint64), exp[1].astype(np.uint64))
                else:
                    assert gs is None
    finally:
        dist.destroy_process_group()


def test_dev_entry_point_captured_in_a_hip_graph():
    """The stream-ordered `_dev` entry points on the fixed-stride path neither synchronise nor
    allocate, so a caller may capture them (bench.py's graph_replay leg): four launches
    over two batches captured once, replayed twice, outputs against the oracle."""
    import torch
    blob = load_dfa("syn256")
    cpu = O.CpuOracle(blob)
    exe = one_amd.Executable(blob)
    n, L = 20000, 64
    hosts = [W.fixed_lines(n, L, 70 + k, alphabet=False) for k in range(2)]
    bufs = [torch.from_numpy(h).cuda() for h in hosts]
    outs = [(torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int64, device="cuda"),
             torch.zeros(n, dtype=torch.int64, device="cuda")) for _ in range(2)]
    fn = _lib.lib().redgpu_match_batch_dev

    def call(k, stream):
        r, s, e = outs[k]
        return fn(exe._h, 4, 0, bufs[k].data_ptr(), None, L, n, r.data_ptr(), s.data_ptr(), e.data_ptr(), stream)

    gs = torch.cuda.Stream()
    with torch.cuda.stream(gs):
        for k in range(2):
            assert call(k, gs.cuda_stream) == 0  # warm: every lazy first-use cost happens before the capture
    gs.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=gs):
        for k in (0, 1, 0, 1):
            assert call(k, torch.cuda.current_stream().cuda_stream) == 0
    for o in outs:
        for t in o:
            t.zero_()
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    for k in range(2):
        er, es, ee = cpu.batch("match", 4, 0, hosts[k], stride=L, n=n, threads=4)
        r, s, e = outs[k]
        assert np.array_equal(r.cpu().numpy(), er)
        assert np.array_equal(s.cpu().numpy().astype(np.uint64), es.astype(np.uint64))
        assert np.array_equal(e.cpu().numpy().astype(np.uint64), ee.astype(np.uint64))
