"""GPU parity: the HIP path (through the C-ABI) against the golden vectors and against the CPU
oracle on seeded inputs.  Bit-exact: result codes, start and end offsets are integers."""
import numpy as np
import pytest

import one_amd
import oracle as O
from golden_util import (CONFIG_DFAS, STYLES, expect_of, kat_items, load_dfa, load_omnibus,
                         load_vectors, unb64)
from oracle.reda_writer import random_dfa
from one_amd import workloads as W

pytestmark = pytest.mark.gpu

STY = {"instant": 1, "first": 2, "tangent": 3, "last": 4, "full": 5}


def _gpu_call(exe, verb, style, lead, text):
    if verb == "check":
        return one_amd.check(exe, text, STY[style], lead)
    if verb == "scan":
        return one_amd.scan(exe, text, STY[style], lead)
    if verb == "search":
        return one_amd.search(exe, text, STY[style], lead)
    return one_amd.match(exe, text, STY[style], lead)


def test_native_library_is_loaded():
    from one_amd import _lib
    assert _lib.lib().redgpu_version() >= 100
    exe = one_amd.Executable(load_dfa("err"))
    assert exe.info["device"] >= 0


def test_kat_matcher_cpp_on_gpu():
    """Known answers asserted by the reference's test/matcher.cpp (check / match / scan rows)."""
    n = 0
    for name, fmt, blob, calls in kat_items():
        exe = one_amd.Executable(blob)
        for c in calls:
            got = _gpu_call(exe, c["verb"], c["style"], c["lead"], unb64(c["text"]))
            exp = c["expect"]
            if isinstance(exp, int):
                assert got == exp, (name, fmt, c, got)
            else:
                for g, e in zip(got, exp):
                    assert e is None or g == e, (name, fmt, c, got)
            n += 1
    assert n > 400


def test_omnibus_table_on_gpu():
    """test/omnibus.cpp:506-555: check(styFull) == match(styFull).result_ == expectation."""
    rows, blobs = load_omnibus()
    checked = 0
    for r in rows:
        for fmt, key in r["fmt"].items():
            if key in ("limit", "parse", "err"):
                continue
            exe = one_amd.Executable(blobs[key].tobytes())
            text = unb64(r["text"]).split(b"\0")[0]
            res = one_amd.check(exe, text, one_amd.styFull, True)
            oc = one_amd.match(exe, text, one_amd.styFull, True)
            assert res == oc[0] and (res == 1) == r["match"], (r, fmt, res, oc)
            checked += 1
    assert checked > 500


@pytest.mark.parametrize("mode", ["default", "generic", "global"])
@pytest.mark.parametrize("name", CONFIG_DFAS)
def test_reference_vectors_on_gpu(name, mode):
    """Reference outputs for every verb x style x doLeader on the ragged mixed input set."""
    vec = load_vectors(name)
    exe = one_amd.Executable(load_dfa(name), force_generic=(mode == "generic"),
                             force_global=(mode == "global"))
    data, offsets = vec["data"], vec["offsets"]
    for verb in ("check", "match", "scan", "search"):
        for si in range(1, 6):
            for lead in (0, 1):
                er, es, ee = expect_of(vec, verb, si, lead)
                if verb in ("match", "search"):
                    fn = one_amd.match_batch if verb == "match" else one_amd.search_batch
                    r, s, e = fn(exe, data, si, lead, offsets=offsets)
                    assert np.array_equal(s, es) and np.array_equal(e, ee), (name, verb, si, lead)
                elif verb == "check":
                    r = one_amd.check_batch(exe, data, si, lead, offsets=offsets)
                else:
                    r = one_amd.scan_batch(exe, data, si, lead, offsets=offsets)
                assert np.array_equal(r, er), (name, verb, si, lead)


def _fixed_inputs(name, n, stride, seed):
    if name == "syn256":
        return W.fixed_lines(n, stride, seed, alphabet=False)
    plant = {"uri": W.URI_PLANT, "err": b"error", "num3": b"123abcd ", "newyork": b"New York",
             "aab": b"aab", "dotstar_err": b"an error", "uri_user": W.URI_PLANT}[name]
    buf = W.fixed_lines(n, stride, seed, plant=plant, plant_every=3, plant_at=min(7, stride - 8))
    v = buf.reshape(n, stride)
    v[1::5, :len(plant[:stride])] = np.frombuffer(plant[:stride], dtype=np.uint8)  # at line start
    return buf


@pytest.mark.parametrize("stride,n", [(48, 5000), (80, 4096), (100, 6001), (250, 4500), (1000, 4100),
                                      (33, 7000), (63, 4097)])
@pytest.mark.parametrize("name", ["syn256", "uri", "newyork", "dotstar_err", "uri_user"])
def test_odd_fixed_strides_block_kernel_vs_oracle(name, stride, n):
    """k_style_blocks on fixed strides that are not multiples of 64 (the streaming kernels want
    whole 64-byte blocks): every style of check and match, with and without start, against the
    oracle; the same batch through the per-lane kernels (force_generic) must agree as well."""
    blob = load_dfa(name)
    exe, gen, cpu = one_amd.Executable(blob), one_amd.Executable(blob, force_generic=True), O.CpuOracle(blob)
    data = _fixed_inputs(name, n, stride, seed=3 * stride + n)
    for si in range(1, 6):
        er, es, ee = cpu.batch("match", si, 0, data, stride=stride, n=n, threads=4)
        r, s, e = one_amd.match_batch(exe, data, si, 0, stride=stride, n=n)
        assert one_amd.last_kernel().startswith("k_style_blocks"), (one_amd.last_kernel(), si)
        assert np.array_equal(r, er), (name, "match", si)
        assert np.array_equal(s, es) and np.array_equal(e, ee), (name, "match", si)
        r2, _, e2 = one_amd.match_batch(exe, data, si, 0, stride=stride, n=n, want_start=False)
        assert np.array_equal(r2, er) and np.array_equal(e2, ee)
        rg, sg, eg = one_amd.match_batch(gen, data, si, 0, stride=stride, n=n)
        assert np.array_equal(rg, er) and np.array_equal(sg, es) and np.array_equal(eg, ee)
        cr = cpu.batch("check", si, 0, data, stride=stride, n=n, threads=4)[0]
        assert np.array_equal(one_amd.check_batch(exe, data, si, 0, stride=stride, n=n), cr), (name, si)


@pytest.mark.parametrize("stride,n", [(64, 5000), (16, 1500), (4096, 300), (48, 2049),
                                      (64, 1024 * 4 * 3 + 17), (128, 3000), (192, 1111),
                                      (64, 1), (64, 2047), (64, 2049)])
@pytest.mark.parametrize("name", ["syn256", "uri", "err", "num3", "newyork", "dotstar_err"])
def test_fixed_stride_hot_path_vs_oracle(name, stride, n):
    """The specialised fixed-stride kernels (k_fixed) against the oracle, all styles, check and
    match, with and without the leader, ragged tile counts."""
    blob = load_dfa(name)
    # the anchored DFAs (err, num3) are flagged early_death and would take the early-exit generic
    # kernel by default: force the whole-line kernels so their dead-end handling is what is tested
    exe = one_amd.Executable(blob, force_stream=True)
    cpu = O.CpuOracle(blob)
    data = _fixed_inputs(name, n, stride, seed=stride + n)
    if one_amd.Executable(blob).info["early_death"]:
        dflt = one_amd.Executable(blob)
        r, s, e = one_amd.match_batch(dflt, data, 4, 0, stride=stride, n=n)
        assert one_amd.last_kernel() == "k_generic" and not dflt.info["fast_path"]
        er, es, ee = cpu.batch("match", 4, 0, data, stride=stride, n=n, threads=4)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
    for si in range(1, 6):
        for lead in (0, 1):
            er, es, ee = cpu.batch("match", si, lead, data, stride=stride, n=n, threads=4)
            r, s, e = one_amd.match_batch(exe, data, si, lead, stride=stride, n=n)
            assert np.array_equal(r, er), (name, "match", si, lead)
            assert np.array_equal(s, es) and np.array_equal(e, ee), (name, "match", si, lead)
            r2, _, e2 = one_amd.match_batch(exe, data, si, lead, stride=stride, n=n,
                                            want_start=False)
            assert np.array_equal(r2, er) and np.array_equal(e2, ee)
            cr, _, _ = cpu.batch("check", si, lead, data, stride=stride, n=n, threads=4)
            assert np.array_equal(one_amd.check_batch(exe, data, si, lead, stride=stride, n=n), cr)
    if exe.info["fast_path"]:
        one_amd.match_batch(exe, data, 4, 0, stride=stride, n=n)
        want = "k_stream" if stride % 64 == 0 else "k_fixed"
        assert one_amd.last_kernel().startswith(want), one_amd.last_kernel()
        one_amd.match_batch(exe, data, 1, 0, stride=stride, n=n)
        # the early-exit styles: k_style_blocks from 4096 lines up (anchored DFAs keep k_fixed)
        blocks = n >= 4096 and not exe.info["early_death"]
        assert one_amd.last_kernel().startswith("k_style_blocks" if blocks else "k_fixed"), \
            one_amd.last_kernel()


@pytest.mark.parametrize("chains", [2, 4])
@pytest.mark.parametrize("stride,n", [(64, 1), (64, 255), (64, 257), (64, 256 * 8 * 3 + 65),
                                      (128, 3000), (4096, 700), (64, 70001)])
@pytest.mark.parametrize("name", ["syn256", "uri", "err", "dotstar_err"])
def test_stream_kernels_two_and_four_chains_vs_oracle(name, stride, n, chains):
    """k_stream (2 lines per lane) and k_stream4 (4 lines per lane, wave-tiles handed out by an
    LDS ticket) forced onto batches of any size: styles Last / Full of match and check, with and
    without start, with the leader; tile counts that leave waves without work and tiles cut by
    the end of the batch (Matcher.h:363-495)."""
    blob = load_dfa(name)
    exe = one_amd.Executable(blob, force_stream=True, stream_chains=chains, no_chunking=True)
    cpu = O.CpuOracle(blob)
    data = _fixed_inputs(name, n, stride, seed=3 * stride + n)
    for si in (4, 5):
        for lead in (0, 1):
            er, es, ee = cpu.batch("match", si, lead, data, stride=stride, n=n, threads=4)
            r, s, e = one_amd.match_batch(exe, data, si, lead, stride=stride, n=n)
            k = one_amd.last_kernel()
            assert k.startswith("k_stream4<" if chains == 4 else "k_stream<"), k
            assert np.array_equal(r, er), (name, si, lead, k)
            assert np.array_equal(s, es) and np.array_equal(e, ee), (name, si, lead, k)
            r2, _, e2 = one_amd.match_batch(exe, data, si, lead, stride=stride, n=n,
                                            want_start=False)
            assert np.array_equal(r2, er) and np.array_equal(e2, ee)
            cr, _, _ = cpu.batch("check", si, 0, data, stride=stride, n=n, threads=4)
            assert np.array_equal(one_amd.check_batch(exe, data, si, 0, stride=stride, n=n), cr)


@pytest.mark.parametrize("name", ["log100", "err", "num3", "uri", "newyork", "syn256", "uri_user"])
def test_probe_and_drain_kernel_vs_oracle(name):
    """k_early (probe every line for a few bytes, park the survivors in LDS, drain them densely)
    forced onto every LDS-resident table kind: match, all five styles, with and without the leader,
    ragged lines (empty, shorter than the probe, shorter than 16 B, long) and fixed strides; plus the
    default dispatch, which picks it for early-death DFAs on big batches (Matcher.h:413-495)."""
    blob = load_dfa(name)
    exe = one_amd.Executable(blob, force_early=True)
    if exe.info["table_kind"] not in (1, 2, 3, 7):
        pytest.skip("table not LDS-resident")
    cpu = O.CpuOracle(blob)
    heads = {"log100": W.log100_heads(), "err": [b"error", b"erro", b"error in x"],
             "num3": [b"123abcd ", b"7a", b"55"], "uri": [W.URI_PLANT],
             "newyork": [b"New York", b"New", b"York"], "syn256": None,
             "uri_user": [b"https://me@ab.example.com/x "]}[name]
    n = 20011
    data, offsets = W.ragged_lines(n, 0, 90, 31, alphabet=(name != "syn256"), heads=heads,
                                   head_every=2)
    for si in range(1, 6):
        for lead in (0, 1):
            er, es, ee = cpu.batch("match", si, lead, data, offsets=offsets, threads=4)
            r, s, e = one_amd.match_batch(exe, data, si, lead, offsets=offsets)
            assert one_amd.last_kernel() == "k_early<match>"
            assert np.array_equal(r, er), (name, si, lead)
            assert np.array_equal(s, es) and np.array_equal(e, ee), (name, si, lead)
            cr = one_amd.check_batch(exe, data, si, lead, offsets=offsets)
            want = "k_early<check>" if not (lead and exe.info["leader_len"]) else "k_generic"
            assert one_amd.last_kernel() == want, one_amd.last_kernel()
            assert np.array_equal(cr, cpu.batch("check", si, lead, data, offsets=offsets,
                                                threads=4)[0]), (name, "check", si, lead)
    for stride, m in ((64, 9000), (8, 5000), (24, 3001), (256, 2500)):
        fixed = _fixed_inputs(name, m, stride, seed=stride) if name in (
            "syn256", "uri", "err", "num3", "newyork") else W.fixed_lines(m, stride, stride)
        er, es, ee = cpu.batch("match", 4, 1, fixed, stride=stride, n=m, threads=4)
        r, s, e = one_amd.match_batch(exe, fixed, 4, 1, stride=stride, n=m)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), stride
        r2, _, e2 = one_amd.match_batch(exe, fixed, 5, 0, stride=stride, n=m, want_start=False)
        er2, _, ee2 = cpu.batch("match", 5, 0, fixed, stride=stride, n=m, threads=4)
        assert np.array_equal(r2, er2) and np.array_equal(e2, ee2)
    dflt = one_amd.Executable(blob)
    if dflt.info["early_death"] and n >= 16384:
        r, s, e = one_amd.match_batch(dflt, data, 4, 1, offsets=offsets)
        assert one_amd.last_kernel() == "k_early<match>"
        er, es, ee = cpu.batch("match", 4, 1, data, offsets=offsets, threads=4)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)


def test_probe_and_drain_kernel_accepting_initial_state():
    """The reference reports an accepting INITIAL state only for an empty input (SURVEY 8a-M quirk
    2: `a*` on "" -> {1,0,0}, on "b" -> {0,0,0}); k_early's lean styLast walk and k_generic's must
    keep that on batches with empty lines."""
    blobs = [b for name, fmt, b, _ in kat_items() if name == "quirk_accepting_initial"]
    assert blobs
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 40, 5000)
    lens[::7] = 0
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = rng.choice(np.frombuffer(b"aab", dtype=np.uint8), int(offsets[-1]))
    for blob in blobs:
        cpu = O.CpuOracle(blob)
        for kw in (dict(force_early=True), dict(force_generic=True), dict()):
            exe = one_amd.Executable(blob, **kw)
            for si in range(1, 6):
                er, es, ee = cpu.batch("match", si, 0, data, offsets=offsets, threads=2)
                r, s, e = one_amd.match_batch(exe, data, si, 0, offsets=offsets)
                assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), \
                    (kw, si, one_amd.last_kernel())
            assert int((er > 0).sum()) > 100


@pytest.mark.parametrize("seed", range(6))
def test_random_dfas_vs_oracle(seed):
    """Synthetic DFAs of every table placement (incl. reachable pure dead ends) vs the oracle."""
    n_states = [3, 60, 256, 257, 700, 5000][seed]
    n_cls = [2, 256, 256, 30, 64, 9][seed]
    blob = random_dfa(n_states, n_cls, seed, dead_frac=[0.3, 0.0, 0.02, 0.0, 0.05, 0.1][seed])
    exe = one_amd.Executable(blob)
    cpu = O.CpuOracle(blob)
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 120, 3000)
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = rng.integers(0, 256, int(offsets[-1]), dtype=np.uint8)
    fixed = rng.integers(0, 256, 2000 * 32, dtype=np.uint8)
    for si in range(1, 6):
        for lead in (0, 1):
            er, es, ee = cpu.batch("match", si, lead, data, offsets=offsets, threads=4)
            r, s, e = one_amd.match_batch(exe, data, si, lead, offsets=offsets)
            assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
            assert np.array_equal(one_amd.check_batch(exe, data, si, lead, offsets=offsets),
                                  cpu.batch("check", si, lead, data, offsets=offsets)[0])
            assert np.array_equal(one_amd.scan_batch(exe, data, si, lead, offsets=offsets),
                                  cpu.batch("scan", si, lead, data, offsets=offsets, threads=4)[0])
            er, es, ee = cpu.batch("match", si, lead, fixed, stride=32, n=2000)
            r, s, e = one_amd.match_batch(exe, fixed, si, lead, stride=32, n=2000)
            assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)


@pytest.mark.parametrize("rows", [0, 8, 40, 200])
def test_hot_row_tables_vs_oracle(rows):
    """REDGPU_TAB_HOT_ROWS: the hot states' [hot index][byte] u8 table in LDS, the rest in the class table in L2.
    A random DFA whose walk keeps crossing between hot and cold states (the LDS budget is set
    so that only `rows` rows fit; 0 = default budget), every verb and style, against the
    oracle; then the two reference-compiled big DFAs at several budgets."""
    blob = random_dfa(2500, 48, 77, dead_frac=0.01, accept_frac=0.2)  # 240 KB class table
    # a dense random DFA has no locality: force_hot keeps the hot rows anyway
    exe = one_amd.Executable(blob, force_hot=True, lds_table_max=(rows or 100) * 256)
    cpu = O.CpuOracle(blob)
    assert exe.info["table_kind"] == 6 and exe.info["n_hot"] == (rows or 100)
    rng = np.random.default_rng(rows)
    lens = rng.integers(0, 200, 4000)
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = rng.integers(0, 256, int(offsets[-1]), dtype=np.uint8)
    for si in range(1, 6):
        for lead in (0, 1):
            for verb, fn in (("match", one_amd.match_batch), ("search", one_amd.search_batch)):
                er, es, ee = cpu.batch(verb, si, lead, data, offsets=offsets, threads=4)
                r, s, e = fn(exe, data, si, lead, offsets=offsets)
                assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
            assert np.array_equal(one_amd.check_batch(exe, data, si, lead, offsets=offsets),
                                  cpu.batch("check", si, lead, data, offsets=offsets)[0])
            assert np.array_equal(one_amd.scan_batch(exe, data, si, lead, offsets=offsets),
                                  cpu.batch("scan", si, lead, data, offsets=offsets, threads=4)[0])
    state = np.full(len(lens), one_amd.STATE_INITIAL, dtype=np.uint32)
    ostate = np.full(len(lens), O.STATE_INITIAL, dtype=np.uint32)
    assert np.array_equal(one_amd.advance_batch(exe, data, state, offsets=offsets),
                          cpu.advance_batch(data, ostate, offsets=offsets))
    got = one_amd.match_all_batch(exe, data, 4, True, offsets=offsets)
    exp = cpu.match_all_batch(data, 4, do_leader=True, offsets=offsets)
    assert np.array_equal(got[0], exp[0])
    for name in ("log100", "uri_v6"):
        vec = load_vectors(name)
        exe = one_amd.Executable(load_dfa(name), **({"lds_table_max": rows * 256} if rows else {}))
        # with >= 36 KB of budget the signature set fits LDS in its sparse form (kind 7)
        assert exe.info["table_kind"] == (7 if name == "log100" and rows in (0, 200) else 6)
        for lead in (0, 1):
            r, s, e = one_amd.match_batch(exe, vec["data"], 4, lead, offsets=vec["offsets"])
            key = "match_4_%d_" % lead
            assert np.array_equal(r, vec[key + "res"]) and np.array_equal(s, vec[key + "start"])
            assert np.array_equal(e, vec[key + "end"])


@pytest.mark.parametrize("name", ["uri", "newyork", "dotstar_err", "uri_user"])
def test_loose_start_scan_search_collect_vs_oracle(name):
    """DFAs with L = SIGMA* L (redgpu_info.suffix_closed: patterns added with a loose start): scan,
    search and collect stop at the first attempt that reaches the end of the line without
    accepting - the reference's loop (Matcher.h:511-553, :575-621) walks every later start
    position to the same answer.  Lines with no match, a match at the start / middle / very end,
    several matches, empty lines, every style; fixed strides too."""
    blob = load_dfa(name)
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    assert exe.info["suffix_closed"] == 1
    assert one_amd.Executable(load_dfa("err")).info["suffix_closed"] == 0
    rng = np.random.default_rng(11)
    plants = [W.URI_PLANT, b"New York", b"error", b"York", b"ftp://a.bc/", b"New"]
    lens = list(rng.integers(0, 300, 2500)) + [0, 1, 2, 700, 0, 33]
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = W.alphabet_bytes(int(offsets[-1]), 77)[: int(offsets[-1])].copy()
    for li in range(0, len(lens), 3):  # every third line gets a plant somewhere (or at its end)
        pl = np.frombuffer(plants[(li // 3) % len(plants)], dtype=np.uint8)
        L = lens[li]
        if L < len(pl):
            continue
        at = [0, (L - len(pl)) // 2, L - len(pl)][(li // 3) % 3]
        o = int(offsets[li]) + at
        data[o:o + len(pl)] = pl
    for si in range(1, 6):
        exp = cpu.batch("scan", si, 0, data, offsets=offsets, threads=4)[0]
        assert np.array_equal(one_amd.scan_batch(exe, data, si, 0, offsets=offsets), exp), (name, si)
        er, es, ee = cpu.batch("search", si, 0, data, offsets=offsets, threads=4)
        r, s, e = one_amd.search_batch(exe, data, si, 0, offsets=offsets)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), (name, si)
        assert np.array_equal(one_amd.scan_batch(exe, data, si, 1, offsets=offsets),
                              cpu.batch("scan", si, 1, data, offsets=offsets, threads=4)[0])
    counts, res, st, en = one_amd.collect_batch(exe, data, 3, offsets=offsets)
    ec, er3, es3, ee3 = cpu.collect_batch(data, 3, offsets=offsets)
    assert np.array_equal(counts, ec), name
    mask = np.arange(3)[None, :] < np.minimum(counts, 3).astype(np.int64)[:, None]  # filled slots
    assert np.array_equal(res[mask], np.asarray(er3)[mask]) and np.array_equal(st[mask], np.asarray(es3)[mask])
    assert np.array_equal(en[mask], np.asarray(ee3)[mask]), name
    n, L = 1500, 96
    fixed = W.alphabet_bytes(n * L, 5)[: n * L].copy()
    fixed.reshape(n, L)[::4, 40:40 + len(W.URI_PLANT)] = np.frombuffer(W.URI_PLANT, dtype=np.uint8)
    for si in (1, 4, 5):
        assert np.array_equal(one_amd.scan_batch(exe, fixed, si, 0, stride=L, n=n),
                              cpu.batch("scan", si, 0, fixed, stride=L, n=n, threads=4)[0])
        er, es, ee = cpu.batch("search", si, 0, fixed, stride=L, n=n, threads=4)
        r, s, e = one_amd.search_batch(exe, fixed, si, 0, stride=L, n=n)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)


@pytest.mark.parametrize("name", ["err", "aab", "num3"])  # anchored: scan stays linear on the long line
def test_marked_scan_and_search_vs_oracle(name):
    """k_scan_marked (scan / search; DFAs with <= 4 start bytes through the packed list, num3 -
    ten start bytes, a leading character class - through the flag table): batches that must be halved to
    fit the bitmap, a line longer than the bitmap covers (scanned the old way), empty lines,
    candidates on the first / last byte of a line and straddling the 16-byte pieces, fixed
    strides; every style, with and without the leader; against the oracle and against the
    per-lane kernel (force_generic)."""
    blob = load_dfa(name)
    exe, gen, cpu = one_amd.Executable(blob), one_amd.Executable(blob, force_generic=True), O.CpuOracle(blob)
    rng = np.random.default_rng(5)
    plants = [b"error", b"aab", b"New York", b"eerror", b"aaab", b"erro", b"e", b"a", b"123abcd",
              b"0", b"4567 ", b"99x"]
    def text(total, seed):
        d = W.alphabet_bytes(max(total, 1), seed)[:total].copy()
        for k in range(0, max(total - 12, 0), 53):
            pl = plants[(k // 53) % len(plants)]
            d[k:k + len(pl)] = np.frombuffer(pl, dtype=np.uint8)
        return d
    shapes = [
        list(rng.integers(0, 200, 3000)),
        [1000] * 300,                                   # 256 lines x 1000 B > 64 KB: halved batches
        [5] * 100 + [70000] + [0] * 50 + [17] * 100,    # one line beyond the bitmap
        [0, 0, 0, 5, 0],
        [16] * 700, [15] * 700, [1] * 900,
    ]
    for lens in shapes:
        offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        data = text(int(offsets[-1]), len(lens))
        for lead in (0, 1):
            for si in range(1, 6):
                exp = cpu.batch("scan", si, lead, data, offsets=offsets, threads=4)[0]
                assert np.array_equal(one_amd.scan_batch(exe, data, si, lead, offsets=offsets), exp), \
                    (name, "scan", si, lead, lens[:4])
                assert np.array_equal(one_amd.scan_batch(gen, data, si, lead, offsets=offsets), exp)
                er, es, ee = cpu.batch("search", si, lead, data, offsets=offsets, threads=4)
                r, s, e = one_amd.search_batch(exe, data, si, lead, offsets=offsets)
                assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), \
                    (name, "search", si, lead, lens[:4])
    one_amd.scan_batch(exe, data, 1, 1, offsets=offsets)
    assert one_amd.last_kernel() == "k_scan_marked"
    for L, n in ((48, 2000), (64, 1500), (1000, 300)):
        data = text(L * n, L)
        for lead in (0, 1):
            for si in (1, 4):
                exp = cpu.batch("scan", si, lead, data, stride=L, n=n, threads=4)[0]
                assert np.array_equal(one_amd.scan_batch(exe, data, si, lead, stride=L, n=n), exp)
                er, es, ee = cpu.batch("search", si, lead, data, stride=L, n=n, threads=4)
                r, s, e = one_amd.search_batch(exe, data, si, lead, stride=L, n=n)
                assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)


@pytest.mark.parametrize("seed", [3, 4])
def test_sparse_lds_tables_vs_oracle(seed):
    """REDGPU_TAB_LDS_SPARSE: a class table too big for LDS, mostly dead-state entries, packed in
    row-displacement form.  Random sparse DFAs (88 % of the transitions lead to the error state,
    so walks die early), every verb and style, with and without the leader flag, against the
    oracle; half of the lines follow live transitions, so slots of many rows are read."""
    from oracle.reda_writer import write_reda
    n_st, n_cls = 2600, 56                                   # 291 KB class table
    rng = np.random.default_rng(seed)
    trans = rng.integers(1, n_st, size=(n_st, n_cls), dtype=np.int64)
    trans[rng.random((n_st, n_cls)) < 0.88] = 0              # state 0 = the error state
    trans[0, :] = 0
    results = np.where(rng.random(n_st) < 0.2, rng.integers(1, 6, size=n_st), 0)
    results[0] = results[1] = 0
    equiv = rng.integers(0, n_cls, size=256, dtype=np.int64).astype(np.uint8)
    equiv[:n_cls] = np.arange(n_cls, dtype=np.uint8)
    blob = write_reda(trans, results, equiv=equiv, initial=1)
    exe = one_amd.Executable(blob)
    cpu = O.CpuOracle(blob)
    assert exe.info["table_kind"] == 7 and exe.info["early_death"] == 1, exe.info
    assert one_amd.Executable(blob, force_hot=True).info["table_kind"] == 6
    lens = rng.integers(0, 120, 6000)
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = rng.integers(0, 256, int(offsets[-1]), dtype=np.uint8)
    # every other line starts with bytes that follow live transitions for as long as there are
    # any (random bytes die after 1.1 steps on average: only the first rows would be read)
    byte_of = [np.flatnonzero(equiv == c) for c in range(n_cls)]
    for i in range(0, len(lens), 2):
        st, o = 1, int(offsets[i])
        for k in range(int(lens[i])):
            live = np.flatnonzero(trans[st])
            if live.size == 0:
                break
            c = int(live[rng.integers(0, live.size)])
            data[o + k] = byte_of[c][rng.integers(0, byte_of[c].size)]
            st = int(trans[st, c])
    for si in range(1, 6):
        for lead in (0, 1):
            for verb, fn in (("match", one_amd.match_batch), ("search", one_amd.search_batch)):
                er, es, ee = cpu.batch(verb, si, lead, data, offsets=offsets, threads=4)
                r, s, e = fn(exe, data, si, lead, offsets=offsets)
                assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
            assert np.array_equal(one_amd.check_batch(exe, data, si, lead, offsets=offsets),
                                  cpu.batch("check", si, lead, data, offsets=offsets)[0])
            assert np.array_equal(one_amd.scan_batch(exe, data, si, lead, offsets=offsets),
                                  cpu.batch("scan", si, lead, data, offsets=offsets, threads=4)[0])
    state = np.full(len(lens), one_amd.STATE_INITIAL, dtype=np.uint32)
    ostate = np.full(len(lens), O.STATE_INITIAL, dtype=np.uint32)
    assert np.array_equal(one_amd.advance_batch(exe, data, state, offsets=offsets),
                          cpu.advance_batch(data, ostate, offsets=offsets))
    got = one_amd.match_all_batch(exe, data, 4, True, offsets=offsets)
    exp = cpu.match_all_batch(data, 4, do_leader=True, offsets=offsets)
    assert np.array_equal(got[0], exp[0])
    # fixed-stride lines too (the early-exit generic kernel on every placement)
    n, L = 3000, 64
    d2 = rng.integers(0, 256, n * L, dtype=np.uint8)
    for sty in (4, 5):
        er, es, ee = cpu.batch("match", sty, 0, d2, stride=L, n=n, threads=4)
        r, s, e = one_amd.match_batch(exe, d2, sty, 0, stride=L, n=n)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)


@pytest.mark.parametrize("case", ["uri_v6_random", "uri_v6_text", "log100_text", "random_cold",
                                  "random_cold_dead", "uri_user_text"])
def test_hot_row_streaming_kernel_vs_oracle(case):
    """k_stream<.., hot>: fixed-stride lines over a DFA too big for LDS.  All-hot walks (random
    bytes over a real regex), walks with cold excursions (planted URLs / log signatures), and a
    dense random DFA with 40 forced hot rows where nearly every half-block is re-walked by the
    slow path; every mode of the kernel, 64- and 128-byte blocks, a ragged last tile."""
    if case.startswith("random_cold"):
        # "_dead": a few transitions lead to the (absorbing) error state: hot index 0
        blob = random_dfa(2500, 48, 78, dead_frac=0.004 if case.endswith("dead") else 0.0,
                          accept_frac=0.2)
        exe = one_amd.Executable(blob, force_hot=True, lds_table_max=40 * 256)
        mk = lambda n, L, seed: W.fixed_lines(n, L, seed, alphabet=False)
    else:
        name = ("log100" if case.startswith("log100") else
                "uri_user" if case.startswith("uri_user") else "uri_v6")
        blob = load_dfa(name)
        # (LOG-100 by itself takes the sparse LDS form, kind 7: hot rows only when forced)
        exe = one_amd.Executable(blob, force_hot=(name in ("uri_user", "log100")))
        if case == "uri_user_text":
            # 343 states: the class table fits LDS; hot rows only when forced or after tuning
            mk = lambda n, L, seed: W.fixed_lines(n, L, seed, alphabet=True,
                                                  plant=W.URI_USER_PLANT, plant_every=3,
                                                  plant_at=L // 2 - 9)
        elif case == "uri_v6_random":
            mk = lambda n, L, seed: W.fixed_lines(n, L, seed, alphabet=False)
        elif case == "uri_v6_text":
            mk = lambda n, L, seed: W.fixed_lines(n, L, seed, alphabet=True, plant=W.URI_V6_PLANT,
                                                  plant_every=3, plant_at=L // 2 - 9)
        else:
            heads = W.log100_heads()
            def mk(n, L, seed):
                a = W.fixed_lines(n, L, seed, alphabet=True).reshape(n, L)
                for i in range(0, n, 2):          # every other line starts with a signature
                    h = np.frombuffer(heads[i % len(heads)], dtype=np.uint8)[:L]
                    a[i, :len(h)] = h
                return a.reshape(-1)
    cpu = O.CpuOracle(blob)
    assert exe.info["table_kind"] == 6 and exe.info["n_hot"] <= 254
    anchored = bool(exe.info["early_death"])   # such DFAs keep the early-exit kernels
    for n, L in ((3001, 64), (1500, 128), (700, 192), (257, 4096)):
        data = mk(n, L, 100 + L)
        for sty in (4, 5):
            er, es, ee = cpu.batch("match", sty, 0, data, stride=L, n=n, threads=8)
            r, s, e = one_amd.match_batch(exe, data, sty, 0, stride=L, n=n)
            assert anchored or (one_amd.last_kernel().startswith("k_stream<") and
                                "hot" in one_amd.last_kernel()), one_amd.last_kernel()
            assert np.array_equal(r, er), (case, n, L, sty)
            assert np.array_equal(s, es) and np.array_equal(e, ee), (case, n, L, sty)
            r, _, e = one_amd.match_batch(exe, data, sty, 1, stride=L, n=n, want_start=False)
            assert np.array_equal(r, er) and np.array_equal(e, ee)
            assert np.array_equal(one_amd.check_batch(exe, data, sty, 0, stride=L, n=n),
                                  cpu.batch("check", sty, 0, data, stride=L, n=n, threads=8)[0])
        # StatefulMatcher chunks through the same kernel: two chunks of L/2 when that is whole
        # 64-byte blocks, else one chunk
        state = np.full(n, one_amd.STATE_INITIAL, dtype=np.uint32)
        if L % 128 == 0:
            a = data.reshape(n, L)
            for half in (a[:, :L // 2], a[:, L // 2:]):
                res = one_amd.advance_batch(exe, np.ascontiguousarray(half).reshape(-1), state,
                                            stride=L // 2, n=n)
        else:
            res = one_amd.advance_batch(exe, data, state, stride=L, n=n)
        assert anchored or one_amd.last_kernel() == "k_stream<advance,hot>"
        ostate = np.full(n, O.STATE_INITIAL, dtype=np.uint32)
        assert np.array_equal(res, cpu.advance_batch(data, ostate, stride=L, n=n))


@pytest.mark.parametrize("case", ["uri_v6_text", "uri_v6_random", "random_cold", "random_cold_dead"])
def test_hot_row_ragged_kernel_vs_oracle(case):
    """k_ragged<.., hot>: ragged lines over a DFA too big for LDS - all-hot walks, walks with cold
    excursions (planted URLs), and a dense random DFA with 40 forced hot rows where nearly every
    block is re-walked; every mode; empty lines, lines ending at the very end of the buffer, a
    few long lines; with and without the length-bucketing pass (20000 lines)."""
    rng = np.random.default_rng(21)
    if case.startswith("random_cold"):
        blob = random_dfa(2500, 48, 79, dead_frac=0.004 if case.endswith("dead") else 0.0,
                          accept_frac=0.2)
        kw = dict(force_hot=True, lds_table_max=40 * 256)
        gen = W.random_bytes
    else:
        blob = load_dfa("uri_v6")
        kw = {}
        gen = W.random_bytes if case.endswith("random") else W.alphabet_bytes
    cpu = O.CpuOracle(blob)
    for n in (3000, 20000):
        lens = rng.geometric(1 / 90, n).astype(np.int64) - 1
        lens[rng.integers(0, n, 5)] = rng.integers(2000, 5000, 5)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        data = gen(int(offsets[-1]), 31 + n).copy()
        if case == "uri_v6_text":
            for k in range(0, data.size - 100, 700):
                plant = W.URI_V6_PLANT if (k // 700) % 3 == 0 else W.URI_PLANT
                data[k:k + len(plant)] = np.frombuffer(plant, dtype=np.uint8)
        for extra in ({}, {"no_bucketing": True}):
            exe = one_amd.Executable(blob, **kw, **extra)
            assert exe.info["table_kind"] == 6
            for sty in (4, 5):
                er, es, ee = cpu.batch("match", sty, 0, data, offsets=offsets, threads=8)
                r, s, e = one_amd.match_batch(exe, data, sty, 0, offsets=offsets)
                assert one_amd.last_kernel().startswith("k_ragged<") and \
                    one_amd.last_kernel().endswith("hot>"), one_amd.last_kernel()
                assert np.array_equal(r, er), (case, n, sty)
                assert np.array_equal(s, es) and np.array_equal(e, ee), (case, n, sty)
                r, _, e = one_amd.match_batch(exe, data, sty, 0, offsets=offsets, want_start=False)
                assert np.array_equal(r, er) and np.array_equal(e, ee)
                assert np.array_equal(one_amd.check_batch(exe, data, sty, 0, offsets=offsets),
                                      cpu.batch("check", sty, 0, data, offsets=offsets, threads=8)[0])


@pytest.mark.parametrize("case", ["uri_user", "rnd700x30", "rnd257x100_dead", "rnd1300x25",
                                  "rnd1500x40_big", "rnd900x80_big_dead"])
def test_class_table_streaming_kernel_vs_oracle(case):
    """k_stream<.., cls>: fixed-stride lines over a DFA of more than 256 states whose class table
    (<= 64 KB) sits in LDS in row-offset form - two lookups per byte, no cold path.  Every mode,
    64- and 128-byte blocks, ragged last tile, the leader filter, StatefulMatcher chunks."""
    if case == "uri_user":
        blob = load_dfa("uri_user")
        mk = lambda n, L, seed: W.fixed_lines(n, L, seed, alphabet=True, plant=W.URI_USER_PLANT,
                                              plant_every=3, plant_at=L // 2 - 20)
    else:
        # "_big": class table above 64 KB (120 / 144 KB) - the index form, one workgroup per CU
        n_st, n_cls = {"rnd700x30": (700, 30), "rnd257x100_dead": (258, 100),
                       "rnd1300x25": (1300, 25), "rnd1500x40_big": (1500, 40),
                       "rnd900x80_big_dead": (900, 80)}[case]
        blob = random_dfa(n_st, n_cls, 91, dead_frac=0.01 if case.endswith("dead") else 0.0,
                          accept_frac=0.15)
        mk = lambda n, L, seed: W.fixed_lines(n, L, seed, alphabet=False)
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    assert exe.info["table_kind"] in (2, 3), exe.info
    assert ("big" in case) == (exe.info["table_bytes"] > 65536 and exe.info["table_kind"] == 3)
    for n, L in ((3001, 64), (1500, 128), (700, 192), (130, 4096)):
        data = mk(n, L, 200 + L)
        for sty in (4, 5):
            er, es, ee = cpu.batch("match", sty, 0, data, stride=L, n=n, threads=8)
            r, s, e = one_amd.match_batch(exe, data, sty, 0, stride=L, n=n)
            # (130 x 4 KiB is few enough lines for the chunked form: "k_stream<chunk,cls>+k_chunk")
            assert "cls>" in one_amd.last_kernel(), one_amd.last_kernel()
            assert np.array_equal(r, er), (case, n, L, sty)
            assert np.array_equal(s, es) and np.array_equal(e, ee), (case, n, L, sty)
            r, _, e = one_amd.match_batch(exe, data, sty, 1, stride=L, n=n, want_start=False)
            er1, _, ee1 = cpu.batch("match", sty, 1, data, stride=L, n=n, threads=8)
            assert np.array_equal(r, er1) and np.array_equal(e, ee1)
            assert np.array_equal(one_amd.check_batch(exe, data, sty, 0, stride=L, n=n),
                                  cpu.batch("check", sty, 0, data, stride=L, n=n, threads=8)[0])
        state = np.full(n, one_amd.STATE_INITIAL, dtype=np.uint32)
        if L % 128 == 0:
            a = data.reshape(n, L)
            for half in (a[:, :L // 2], a[:, L // 2:]):
                res = one_amd.advance_batch(exe, np.ascontiguousarray(half).reshape(-1), state,
                                            stride=L // 2, n=n)
        else:
            res = one_amd.advance_batch(exe, data, state, stride=L, n=n)
        assert one_amd.last_kernel() == "k_stream<advance,cls>"
        ostate = np.full(n, O.STATE_INITIAL, dtype=np.uint32)
        assert np.array_equal(res, cpu.advance_batch(data, ostate, stride=L, n=n))
        # the generic kernel must leave the same state tokens behind
        gen = one_amd.Executable(blob, force_generic=True)
        gstate = np.full(n, one_amd.STATE_INITIAL, dtype=np.uint32)
        one_amd.advance_batch(gen, data, gstate, stride=L, n=n)
        assert np.array_equal(gstate, state)


@pytest.mark.parametrize("case", ["uri_user", "rnd700x30", "rnd257x100_dead", "rnd1500x40_big"])
def test_class_table_ragged_kernel_vs_oracle(case):
    """k_ragged<.., cls>: ragged lines over a mid-size DFA (class table <= 64 KB in LDS): every
    mode, empty lines, lines ending at the end of the buffer, long lines, with and without the
    length-bucketing pass."""
    rng = np.random.default_rng(23)
    if case == "uri_user":
        blob, gen = load_dfa("uri_user"), W.alphabet_bytes
    else:
        n_st, n_cls = {"rnd700x30": (700, 30), "rnd257x100_dead": (258, 100),
                       "rnd1500x40_big": (1500, 40)}[case]
        blob = random_dfa(n_st, n_cls, 92, dead_frac=0.01 if case.endswith("dead") else 0.0,
                          accept_frac=0.15)
        gen = W.random_bytes
    cpu = O.CpuOracle(blob)
    for n in (3000, 20000):
        lens = rng.geometric(1 / 90, n).astype(np.int64) - 1
        lens[rng.integers(0, n, 5)] = rng.integers(2000, 5000, 5)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        data = gen(int(offsets[-1]), 41 + n).copy()
        if case == "uri_user":
            for k in range(0, data.size - 100, 600):
                data[k:k + len(W.URI_USER_PLANT)] = np.frombuffer(W.URI_USER_PLANT, dtype=np.uint8)
        for extra in ({}, {"no_bucketing": True}):
            exe = one_amd.Executable(blob, **extra)
            for sty in (4, 5):
                er, es, ee = cpu.batch("match", sty, 0, data, offsets=offsets, threads=8)
                r, s, e = one_amd.match_batch(exe, data, sty, 0, offsets=offsets)
                assert one_amd.last_kernel().startswith("k_ragged<") and \
                    one_amd.last_kernel().endswith("cls>"), one_amd.last_kernel()
                assert np.array_equal(r, er), (case, n, sty)
                assert np.array_equal(s, es) and np.array_equal(e, ee), (case, n, sty)
                r, _, e = one_amd.match_batch(exe, data, sty, 0, offsets=offsets, want_start=False)
                assert np.array_equal(r, er) and np.array_equal(e, ee)
                assert np.array_equal(one_amd.check_batch(exe, data, sty, 0, offsets=offsets),
                                      cpu.batch("check", sty, 0, data, offsets=offsets, threads=8)[0])


@pytest.mark.parametrize("case", ["uri", "dotstar_err", "uri_v6", "syn256_forced", "newyork",
                                  "uri_user"])
def test_speculative_chunking_vs_oracle(case):
    """k_chunk.h: few long lines cut into chunks walked at once from the initial state, wrong
    guesses re-walked round by round.  Loose-start DFAs (mostly right guesses), a dense random
    one with chunking forced (every guess wrong: all rounds, then the serial finish), matches
    that straddle chunk borders, Last and Full, check, the leader filter."""
    name = case.replace("_forced", "")
    blob = load_dfa(name)
    kw = {"force_chunking": True} if case.endswith("forced") else {}
    exe, cpu = one_amd.Executable(blob, **kw), O.CpuOracle(blob)
    rng = np.random.default_rng(51)
    for n, L in ((5, 65536), (64, 16384), (300, 4096), (1, 8192)):
        if name == "syn256":
            data = W.random_bytes(n * L, 61).copy()
        else:
            data = W.alphabet_bytes(n * L, 61).copy()
            plants = {"uri": W.URI_PLANT, "uri_v6": W.URI_PLANT, "dotstar_err": b" an error: x ",
                      "newyork": b"I love New York.", "uri_user": W.URI_USER_PLANT}[name]
            p = np.frombuffer(plants, dtype=np.uint8)
            for k in range(0, data.size - 200, 777):   # 777: lands on and across chunk borders
                data[k:k + len(p)] = p
            # a match ending exactly at a line's last byte and one starting at its first
            data[L - len(p):L] = p
            data[(n - 1) * L:(n - 1) * L + len(p)] = p
        for sty in (4, 5):
            er, es, ee = cpu.batch("match", sty, 0, data, stride=L, n=n, threads=8)
            r, s, e = one_amd.match_batch(exe, data, sty, 0, stride=L, n=n)
            assert "k_chunk" in one_amd.last_kernel(), one_amd.last_kernel()
            assert np.array_equal(r, er), (case, n, L, sty)
            assert np.array_equal(s, es) and np.array_equal(e, ee), (case, n, L, sty)
            r, _, e = one_amd.match_batch(exe, data, sty, 1, stride=L, n=n, want_start=False)
            assert np.array_equal(r, er) and np.array_equal(e, ee)
            assert np.array_equal(one_amd.check_batch(exe, data, sty, 0, stride=L, n=n),
                                  cpu.batch("check", sty, 0, data, stride=L, n=n, threads=8)[0])
        # the same through the whole-line kernels
        plain = one_amd.Executable(blob, no_chunking=True)
        r2, s2, e2 = one_amd.match_batch(plain, data, 4, 0, stride=L, n=n)
        assert "k_chunk" not in one_amd.last_kernel()
        er, es, ee = cpu.batch("match", 4, 0, data, stride=L, n=n, threads=8)
        assert np.array_equal(r2, er) and np.array_equal(s2, es) and np.array_equal(e2, ee)


def test_tune_reranks_hot_rows_results_unchanged():
    """redgpu_dfa_tune: visits counted on a sample of URL-bearing text re-rank the hot rows;
    outputs stay bit-exact, the share of the walk served from LDS goes up (measured on held-out
    text by walking the DFA through the handle's own hot range)."""
    blob = load_dfa("uri_v6")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    n, L = 4096, 256
    sample = W.fixed_lines(n, L, 31, alphabet=True, plant=W.URI_PLANT, plant_every=2, plant_at=40)
    held = W.fixed_lines(n, L, 32, alphabet=True, plant=W.URI_PLANT, plant_every=2, plant_at=17)
    exp = cpu.batch("match", 4, 0, held, stride=L, n=n, threads=8)
    before = exe.info
    got0 = one_amd.match_batch(exe, held, 4, 0, stride=L, n=n)
    after = exe.tune(sample, stride=L, n=n)
    assert after["table_kind"] == 6 and after["n_hot"] == before["n_hot"]
    got1 = one_amd.match_batch(exe, held, 4, 0, stride=L, n=n)
    assert "hot" in one_amd.last_kernel()
    for g0, g1, e in zip(got0, got1, exp):
        assert np.array_equal(g0, e) and np.array_equal(g1, e)
    # every other verb on the re-ranked image
    vec = load_vectors("uri_v6")
    for verb, fn in (("match", one_amd.match_batch), ("search", one_amd.search_batch)):
        r, s, e = fn(exe, vec["data"], 4, 0, offsets=vec["offsets"])
        assert np.array_equal(r, vec[verb + "_4_0_res"]) and np.array_equal(e, vec[verb + "_4_0_end"])
        assert np.array_equal(s, vec[verb + "_4_0_start"])
    # ragged lines take k_ragged<..,hot>, tuned or not
    rng = np.random.default_rng(4)
    lens = rng.integers(0, 300, 5000)
    roff = np.zeros(len(lens) + 1, dtype=np.uint64)
    roff[1:] = np.cumsum(lens)
    rdata = held[: int(roff[-1])]
    rexp = cpu.batch("match", 4, 0, rdata, offsets=roff, threads=8)
    got = one_amd.match_batch(exe, rdata, 4, 0, offsets=roff)
    assert one_amd.last_kernel() == "k_ragged<last,start,end,hot>"
    for g, e in zip(got, rexp):
        assert np.array_equal(g, e)
    got = one_amd.match_batch(one_amd.Executable(blob), rdata, 4, 0, offsets=roff)
    assert one_amd.last_kernel() == "k_ragged<last,start,end,hot>"
    for g, e in zip(got, rexp):
        assert np.array_equal(g, e)
    # StatefulMatcher: after tuning, whole lines in one chunk reproduce check<styFull>
    state = np.full(n, one_amd.STATE_INITIAL, dtype=np.uint32)
    res = one_amd.advance_batch(exe, held, state, stride=L, n=n)
    assert np.array_equal(res, cpu.batch("check", 5, 0, held, stride=L, n=n, threads=8)[0])
    # tuning a DFA whose fused table lives in LDS is a no-op that still validates its arguments
    small = one_amd.Executable(load_dfa("uri"))
    assert small.tune(sample, stride=L, n=n)["table_kind"] == 1
    # a DFA of more than 256 states whose table fits LDS but has no streaming form of its own
    # (280 states x 200 classes: more than 127 classes) moves to hot rows + streaming kernel
    # when the observed walk is practically all hot; a 343-state one (17 KB class table: its
    # own streaming form, k_stream cls) stays where it is
    mid = one_amd.Executable(load_dfa("uri_user"))
    assert mid.info["table_kind"] == 3 and mid.tune(held, stride=L, n=n)["table_kind"] == 3
    one_amd.match_batch(mid, held, 4, 0, stride=L, n=n)
    assert one_amd.last_kernel() == "k_stream<last,start,end,cls>"
    bblob = random_dfa(280, 200, 5, accept_frac=0.1)
    big = one_amd.Executable(bblob)
    assert big.info["table_kind"] == 2
    zeros = np.zeros(n * L, dtype=np.uint8)           # a walk that soon cycles through few states
    after_big = big.tune(zeros, stride=L, n=n)
    assert after_big["table_kind"] == 6, after_big
    bcpu = O.CpuOracle(bblob)
    for data in (zeros, held):
        got = one_amd.match_batch(big, data, 4, 0, stride=L, n=n)
        assert "hot" in one_amd.last_kernel()
        for g, e in zip(got, bcpu.batch("match", 4, 0, data, stride=L, n=n, threads=8)):
            assert np.array_equal(g, e)


def test_split_lines_on_device_then_match():
    """redgpu_split_lines: delimiter scan + prefix sum on the device vs the rule of
    lib/Util.cpp:109-130 (oracle.split_lines), on host and device buffers, aligned and unaligned,
    with runs of empty lines, no trailing newline, cap overflow; then the offsets feed the
    ragged verbs with stride=1 (the delimiter dropped) and must give what the oracle gives on
    the same lines with their delimiters removed."""
    import torch
    blob = load_dfa("uri")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    rng = np.random.default_rng(9)
    for n_bytes, p_nl in ((0, 0.1), (1, 1.0), (15, 0.3), (100000, 0.02), (3_000_001, 0.01),
                          (70000, 0.0), (40000, 1.0)):
        text = W.alphabet_bytes(n_bytes + 3, 77 + n_bytes)[3:].copy()   # unaligned view
        nl = rng.random(n_bytes) < p_nl
        text[nl] = 0x0A
        if n_bytes > 200:   # plant URLs so that some lines match
            for k in range(0, n_bytes - 100, 997):
                text[k:k + len(W.URI_PLANT)] = np.frombuffer(W.URI_PLANT, dtype=np.uint8)
        exp = O.split_lines(text)
        offs, found = one_amd.split_lines(exe, text)
        assert found == len(exp) - 1 and np.array_equal(offs, exp), (n_bytes, p_nl)
        # device form; cap smaller than the number of lines keeps the first cap
        dev = torch.from_numpy(np.ascontiguousarray(text)).cuda()
        cap = max(1, (len(exp) - 1) // 2)
        doffs, dcnt = one_amd.split_lines(exe, dev, cap=cap)
        torch.cuda.synchronize()
        assert int(dcnt.item()) == len(exp) - 1
        k = min(cap, len(exp) - 1)
        assert np.array_equal(doffs[:k + 1].cpu().numpy().astype(np.uint64), exp[:k + 1])
        if len(exp) < 2:
            continue
        # match the lines, delimiter dropped
        n = len(exp) - 1
        keep = text[:int(exp[-1])]
        compact = keep[keep != 0x0A]
        coffs = exp - np.arange(n + 1, dtype=np.uint64)
        er, es, ee = cpu.batch("match", 4, 0, compact, offsets=coffs, threads=4)
        r, s, e = one_amd.match_batch(exe, text, 4, 0, offsets=offs, stride=1)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
        for sty, lead in ((1, 1), (5, 0)):
            assert np.array_equal(one_amd.check_batch(exe, text, sty, lead, offsets=offs, stride=1),
                                  cpu.batch("check", sty, lead, compact, offsets=coffs)[0])
        if n_bytes <= 100000:
            assert np.array_equal(one_amd.scan_batch(exe, text, 1, 1, offsets=offs, stride=1),
                                  cpu.batch("scan", 1, 1, compact, offsets=coffs, threads=4)[0])
            got = one_amd.collect_batch(exe, text, 4, offsets=offs, stride=1)
            expc = cpu.collect_batch(compact, 4, offsets=coffs)
            assert np.array_equal(got[0], expc[0])
    # end to end on the device: split -> match, nothing leaves HBM in between
    big = W.alphabet_bytes(1 << 22, 5).copy()
    big[rng.random(big.size) < 0.01] = 0x0A
    exp = O.split_lines(big)
    dev = torch.from_numpy(big).cuda()
    doffs, dcnt = one_amd.split_lines(exe, dev, cap=len(exp) + 10)
    n = int(dcnt.item())
    assert n == len(exp) - 1
    r, s, e = one_amd.match_batch(exe, dev, 4, 0, offsets=doffs[:n + 1].contiguous(), stride=1)
    keep = big[:int(exp[-1])]
    er, es, ee = cpu.batch("match", 4, 0, keep[keep != 0x0A],
                           offsets=exp - np.arange(n + 1, dtype=np.uint64), threads=8)
    assert np.array_equal(r.cpu().numpy(), er)
    assert np.array_equal(e.cpu().numpy().astype(np.uint64), ee)


@pytest.mark.parametrize("name", ["uri", "syn256", "uri_v6", "log100", "err", "newyork4"])
def test_match_text_one_call(name):
    """redgpu_match_text[_dev]: raw text -> lines (lib/Util.cpp:109-130) -> match / check per line
    (tools/skim_red.cpp:36-46) in one call, against oracle.split_lines + CpuOracle on the lines
    with their delimiters removed.  Buffers: empty; one delimiter; runs of empty lines; no
    trailing delimiter; nothing but delimiters; no delimiter at all; a line far longer than the
    rest; > 1024 lines (the long-lines list of k_ragged is in play).  cap below the number of
    lines keeps the first cap and still reports the count.  DFAs: the k_ragged family reads the
    line count on the device (uri, syn256, uri_v6 hot rows); log100 / err (die early: k_early,
    k_generic) and newyork4 take the count through the host."""
    import torch
    blob = load_dfa(name)
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    rng = np.random.default_rng(31)
    plant = np.frombuffer(W.URI_PLANT, dtype=np.uint8)
    shapes = [(0, 0.1), (1, 1.0), (15, 0.3), (5000, 0.05), (70000, 0.0), (30000, 1.0),
              (2_500_001, 0.008), (1_500_000, 0.02)]
    for k, (n_bytes, p_nl) in enumerate(shapes):
        text = (W.random_bytes if name == "syn256" else W.alphabet_bytes)(n_bytes + 3, 91 + k)[3:].copy()
        text[text == 0x0A] = 0x20
        text[rng.random(n_bytes) < p_nl] = 0x0A
        if n_bytes > 1_000_000:
            text[200000:260000][text[200000:260000] == 0x0A] = 0x2E     # one 60 KB line
        if n_bytes > 200 and name != "syn256":
            for at in range(0, n_bytes - 100, 499):
                text[at:at + len(plant)] = plant
        exp = O.split_lines(text)
        n = len(exp) - 1
        keep = text[:int(exp[-1])]
        compact = keep[keep != 0x0A]
        coffs = exp - np.arange(n + 1, dtype=np.uint64)
        for si, lead in ((4, 0), (5, 0), (4, 1)):
            er, es, ee = cpu.batch("match", si, lead, compact, offsets=coffs, threads=4)
            offs, found, r, s, e = one_amd.match_text(exe, text, si, lead)
            assert found == n and np.array_equal(offs, exp), (name, n_bytes)
            assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), \
                (name, n_bytes, p_nl, si, lead)
            # device form, cap above and below the number of lines; check = no positions
            dev = torch.from_numpy(np.ascontiguousarray(text)).cuda()
            for cap in (n + 7, max(1, n // 3)):
                doffs, dcnt, dr, ds, de = one_amd.match_text(exe, dev, si, lead, cap=cap)
                got = min(n, cap)
                torch.cuda.synchronize()
                assert int(dcnt.item()) == n
                assert np.array_equal(doffs[:got + 1].cpu().numpy().astype(np.uint64), exp[:got + 1])
                assert np.array_equal(dr[:got].cpu().numpy(), er[:got])
                assert np.array_equal(ds[:got].cpu().numpy().astype(np.uint64), es[:got])
                assert np.array_equal(de[:got].cpu().numpy().astype(np.uint64), ee[:got])
                _, _, cr, cs, ce = one_amd.match_text(exe, dev, si, lead, cap=cap, want_start=False,
                                                      want_end=False)
                assert cs is None and ce is None
                ec = cpu.batch("check", si, lead, compact, offsets=coffs, threads=4)[0]
                assert np.array_equal(cr[:got].cpu().numpy(), ec[:got]), (name, n_bytes, si, lead)
    # entries past the lines found are left alone
    dev = torch.from_numpy(np.frombuffer(b"ab\ncd\nxyz", dtype=np.uint8).copy()).cuda()
    doffs, dcnt, dr, ds, de = one_amd.match_text(exe, dev, 4, 0, cap=5)
    torch.cuda.synchronize()
    assert int(dcnt.item()) == 2 and doffs[:3].tolist() == [0, 3, 6]
    if name in ("uri", "syn256", "uri_v6"):
        one_amd.match_text(exe, torch.zeros(1 << 20, dtype=torch.uint8, device="cuda"), 4, 0, cap=10)
        assert one_amd.last_kernel().startswith("k_ragged"), one_amd.last_kernel()


@pytest.mark.parametrize("name", ["syn256", "uri"])
def test_ragged_length_bucketing_vs_oracle(name):
    """k_ragged behind its length-bucketing pass (>= 16384 lines): skewed line lengths - many
    empty and short lines, a geometric tail, a few lines of 64+ blocks - results land on the
    right lines whatever order the kernel walks them in; same answers with bucketing off."""
    blob = load_dfa(name)
    cpu = O.CpuOracle(blob)
    rng = np.random.default_rng(12)
    n = 50000
    lens = rng.geometric(1 / 70, n).astype(np.int64) - 1
    lens[rng.random(n) < 0.05] = 0
    lens[rng.integers(0, n, 20)] = rng.integers(4000, 9000, 20)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = (W.random_bytes if name == "syn256" else W.alphabet_bytes)(int(offsets[-1]), 13).copy()
    if name == "uri":
        for k in range(0, data.size - 100, 1500):
            data[k:k + len(W.URI_PLANT)] = np.frombuffer(W.URI_PLANT, dtype=np.uint8)
    er, es, ee = cpu.batch("match", 4, 0, data, offsets=offsets, threads=8)
    ef = cpu.batch("check", 5, 0, data, offsets=offsets, threads=8)[0]
    for kw in ({}, {"no_bucketing": True}):
        exe = one_amd.Executable(blob, **kw)
        r, s, e = one_amd.match_batch(exe, data, 4, 0, offsets=offsets)
        assert one_amd.last_kernel().startswith("k_ragged")
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
        assert np.array_equal(one_amd.check_batch(exe, data, 5, 0, offsets=offsets), ef)


def test_replace_on_gpu():
    """replace batch form (include/Matcher.h:643-706) vs the reference's known answers
    (test/matcher.cpp:648-691, every format) and its outputs in tests/golden/replace_vectors.npz
    (counts, lengths, FNV-1a-64 per rewritten line; one set byte for byte), every style x
    doLeader, three (replacement, max) cases, LDS and global table placements; the
    delimiter-trimmed ragged form; output capacity too small."""
    import json
    import os
    from golden_util import GOLD
    for k in json.load(open(os.path.join(GOLD, "replace_kat.json"))):
        exe = one_amd.Executable(unb64(k["reda"]))
        sty = {"instant": 1, "first": 2, "tangent": 3, "last": 4, "full": 5}[k["style"]]
        got = one_amd.replace(exe, k["text"].encode(), k["repl"].encode(), k["max"], sty)
        assert got == (k["count"], k["expect"].encode()), k
    vec = np.load(os.path.join(GOLD, "replace_vectors.npz"))
    data, offsets = vec["data"], vec["offsets"]
    for name in ("num3", "newyork", "err", "uri", "abc"):
        for kw in ({}, {"force_global": True}):
            exe = one_amd.Executable(vec[name + "_blob"].tobytes(), **kw)
            for ci in range(3):
                repl, mx = str(vec["case_repl"][ci]).encode(), int(vec["case_max"][ci])
                for si in range(1, 6):
                    for lead in (0, 1):
                        key = "%s_c%d_%d_%d_" % (name, ci, si, lead)
                        counts, ooff, out = one_amd.replace_batch(exe, data, repl, si, lead, mx,
                                                                  offsets=offsets)
                        assert np.array_equal(counts, vec[key + "counts"].astype(np.uint64)), key
                        assert np.array_equal(ooff, vec[key + "ooff"].astype(np.uint64)), key
                        fnv = np.array([O.fnv1a64(out[int(ooff[i]):int(ooff[i + 1])].tobytes())
                                        for i in range(len(ooff) - 1)], dtype=np.uint64)
                        assert np.array_equal(fnv, vec[key + "fnv"]), key
                        if key + "out" in vec:
                            assert np.array_equal(out, vec[key + "out"]), key
    # fixed stride + the oracle; trimmed ragged lines
    blob = load_dfa("num3")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    n, L = 3000, 48
    fixed = W.fixed_lines(n, L, 3, alphabet=True)
    counts, ooff, out = one_amd.replace_batch(exe, fixed, b"#", 4, True, stride=L, n=n)
    for i in (0, 1, 17, n - 1):
        k, o = cpu.replace(fixed[i * L:(i + 1) * L].tobytes(), b"#", 4, True)
        assert int(counts[i]) == k and out[int(ooff[i]):int(ooff[i + 1])].tobytes() == o
    text = b"a1\nb22 c\n\n333\n"
    offs, _ = one_amd.split_lines(exe, text)
    counts, ooff, out = one_amd.replace_batch(exe, text, b"N", 4, True, offsets=offs, stride=1)
    assert counts.tolist() == [1, 1, 0, 1] and out.tobytes() == b"aNbN cN"


def test_loader_cache_on_device_and_tune_detaches():
    """Two handles from one blob share the device image; tuning one gives it a private image and
    leaves the other's results and state tokens untouched."""
    blob = load_dfa("uri_v6")
    a, b = one_amd.Executable(blob), one_amd.Executable(blob)
    assert a.info["image_refs"] == 2
    n, L = 2048, 256
    text = W.fixed_lines(n, L, 41, alphabet=True, plant=W.URI_PLANT, plant_every=2, plant_at=30)
    exp = O.CpuOracle(blob).batch("match", 4, 0, text, stride=L, n=n, threads=4)
    state = np.full(n, one_amd.STATE_INITIAL, dtype=np.uint32)
    one_amd.advance_batch(b, text[:n * L // 2], state[:n // 2].copy(), stride=L, n=n // 2)
    a.tune(text, stride=L, n=n)
    assert a.info["image_refs"] == 1 and b.info["image_refs"] == 1
    for exe in (a, b):
        got = one_amd.match_batch(exe, text, 4, 0, stride=L, n=n)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e)
    c = one_amd.Executable(blob)          # the cache still holds b's (untuned) image
    assert b.info["image_refs"] == 2 and c.info["hot_lo"] == b.info["hot_lo"]


def test_edge_cases():
    exe = one_amd.Executable(load_dfa("err"))
    cpu = O.CpuOracle(load_dfa("err"))
    # empty batch
    assert len(one_amd.check_batch(exe, b"", 4, offsets=[0])) == 0
    assert len(one_amd.match_batch(exe, b"", 4, stride=64, n=0)[0]) == 0
    # all-empty lines, and lines shorter than the 5-byte leader
    offs = np.array([0, 0, 0, 3, 3, 8, 13], dtype=np.uint64)
    data = b"err" + b"error" + b"errox"
    for verb, fn in (("check", one_amd.check_batch), ("scan", one_amd.scan_batch)):
        for si in range(1, 6):
            for lead in (0, 1):
                assert np.array_equal(fn(exe, data, si, lead, offsets=offs),
                                      cpu.batch(verb, si, lead, np.frombuffer(data, np.uint8),
                                                offsets=offs)[0])
    # accepting initial state is reported for empty input only (SURVEY a-M quirk 2)
    import json
    for name, fmt, blob, calls in kat_items():
        if name == "quirk_accepting_initial":
            e2 = one_amd.Executable(blob)
            assert one_amd.match(e2, b"", 4, False) == (1, 0, 0)
            assert one_amd.match(e2, b"b", 4, False) == (0, 0, 0)
    # unsupported style -> RedExceptExec (lib/Matcher.cpp:45)
    with pytest.raises(one_amd.RedExceptExec, match="unsupported style"):
        one_amd.check_batch(exe, b"error", 9, offsets=[0, 5])
    with pytest.raises(one_amd.RedExceptExec):
        one_amd.match_batch(exe, b"error", 0, offsets=[0, 5])


def test_long_single_line_positions():
    """One 3 MiB line: positions beyond 2^16 / 2^21, generic path (ragged) and fixed path."""
    blob = load_dfa("dotstar_err")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    n = 3 * 1024 * 1024
    data = W.alphabet_bytes(n, 77)
    data[n - 100000:n - 100000 + 5] = np.frombuffer(b"error", np.uint8)
    for si in (3, 4, 5):
        er = cpu.match(data.tobytes(), si, False)
        assert one_amd.match_batch(exe, data, si, 0, offsets=[0, n]) [0][0] == er[0]
        r, s, e = one_amd.match_batch(exe, data, si, 0, stride=n, n=1)
        assert (int(r[0]), int(s[0]), int(e[0])) == er


def test_device_resident_api_with_torch():
    import torch
    blob = load_dfa("syn256")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    n, stride = 4096 * 3 + 5, 64
    host = W.fixed_lines(n, stride, 9, alphabet=False)
    dev = torch.from_numpy(host).cuda()
    r, s, e = one_amd.match_batch(exe, dev, 4, 0, stride=stride, n=n)
    torch.cuda.synchronize()
    er, es, ee = cpu.batch("match", 4, 0, host, stride=stride, n=n, threads=4)
    assert np.array_equal(r.cpu().numpy(), er)
    assert np.array_equal(s.cpu().numpy().astype(np.uint64), es)
    assert np.array_equal(e.cpu().numpy().astype(np.uint64), ee)
    offs = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * stride
    r2 = one_amd.check_batch(exe, dev, 5, 1, offsets=offs)
    torch.cuda.synchronize()
    assert np.array_equal(r2.cpu().numpy(), cpu.batch("check", 5, 1, host, stride=stride, n=n)[0])


def test_full_size_config2_bit_exact():
    """BASELINE config 2 at full size: 2^20 lines x 64 B, SYN-256 and URI-D, match<styLast,false>,
    compared line-for-line with the oracle (8 host threads)."""
    n, stride = 1 << 20, 64
    for name in ("syn256", "uri"):
        blob = load_dfa(name)
        exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
        data = (W.fixed_lines(n, stride, 42, alphabet=False) if name == "syn256" else
                W.fixed_lines(n, stride, 2, plant=W.URI_PLANT))
        r, s, e = one_amd.match_batch(exe, data, 4, 0, stride=stride, n=n)
        er, es, ee = cpu.batch("match", 4, 0, data, stride=stride, n=n, threads=8)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
        assert int((r > 0).sum()) > 1000
        # size-independent property: the batch result equals the concatenation of its halves
        h = n // 2
        r1 = one_amd.match_batch(exe, data[:h * stride], 4, 0, stride=stride, n=h)[0]
        r2 = one_amd.match_batch(exe, data[h * stride:], 4, 0, stride=stride, n=h)[0]
        assert np.array_equal(np.concatenate([r1, r2]), r)


def _checksum(*tensors):
    """order-sensitive 64-bit checksum of device int tensors (a checksum of checksums)"""
    import torch
    acc = torch.zeros((), dtype=torch.int64, device=tensors[0].device)
    for k, t in enumerate(tensors):
        v = t.to(torch.int64)
        w = torch.arange(1, v.numel() + 1, dtype=torch.int64, device=v.device)
        acc = acc * 1000003 + (v * (w * (2 * k + 3) + 12345)).sum()
    return int(acc.item())


def test_full_size_config3_shard_properties():
    """BASELINE configs[2] at the FULL per-GPU size: 2^21 lines x 4 KiB = 8 GiB resident, SYN-256,
    match<styLast,false> with the full Outcome.  Too big for a line-for-line oracle pass, so:
    (1) a 2^12-line sample spread over the shard is checked against the oracle bit for bit;
    (2) size-independent properties - the whole shard equals the concatenation of its quarters
    (checksum of checksums), and styFull's result agrees with advance() fed in two chunks."""
    import torch
    n, L = 1 << 21, 4096
    blob = load_dfa("syn256")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    g = torch.Generator(device="cuda").manual_seed(3)
    data = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    for q in range(8):   # generate in slices: randint materialises int64 first
        lo, hi = q * (n * L // 8), (q + 1) * (n * L // 8)
        data[lo:hi] = torch.randint(0, 256, (hi - lo,), generator=g, device="cuda",
                                    dtype=torch.uint8)
    r, s, e = one_amd.match_batch(exe, data, 4, 0, stride=L, n=n)
    assert one_amd.last_kernel() == "k_stream<last,start,end>"
    torch.cuda.synchronize()
    # (1) sample vs oracle
    idx = torch.arange(0, n, n >> 12, device="cuda")
    sample = data.view(n, L)[idx].contiguous().cpu().numpy().reshape(-1)
    er, es, ee = cpu.batch("match", 4, 0, sample, stride=L, n=len(idx), threads=8)
    assert np.array_equal(r[idx].cpu().numpy(), er)
    assert np.array_equal(s[idx].cpu().numpy().astype(np.uint64), es)
    assert np.array_equal(e[idx].cpu().numpy().astype(np.uint64), ee)
    assert int((er > 0).sum()) > len(idx) // 2
    # (2a) quarters
    whole = _checksum(r, s, e)
    parts = []
    for q in range(4):
        lo, hi = q * (n // 4), (q + 1) * (n // 4)
        parts.append(one_amd.match_batch(exe, data[lo * L:hi * L], 4, 0, stride=L, n=n // 4))
    cat = [torch.cat([p[k] for p in parts]) for k in range(3)]
    assert _checksum(*cat) == whole
    del parts, cat
    # (2b) check<styFull> == StatefulMatcher over two 2 KiB chunks per line
    full = one_amd.check_batch(exe, data, 5, 0, stride=L, n=n)
    state = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    v = data.view(n, L)
    for half in (v[:, :L // 2], v[:, L // 2:]):
        chunk = half.contiguous().view(-1)
        adv = one_amd.advance_batch(exe, chunk, state, stride=L // 2, n=n)
        del chunk
    assert torch.equal(adv, full)


def test_full_size_config4_ragged_properties():
    """BASELINE configs[3] at full size: LOG-100, 2^23 ragged lines of 32..256 B (1.1 GiB),
    matchLong = match<styLast,true>; half of the lines start with a signature instance.
    A 2^16-line prefix and a 2^16-line suffix are checked against the oracle; the whole batch
    equals the concatenation of its halves (checksum of checksums)."""
    import torch
    n = 1 << 23
    blob = load_dfa("log100")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    data, offsets = W.ragged_lines(n, 32, 256, 4, heads=W.log100_heads(), head_every=2)
    d = torch.from_numpy(data).cuda()
    o = torch.from_numpy(offsets.astype(np.int64)).cuda()
    r, s, e = one_amd.match_batch(exe, d, 4, 1, offsets=o)
    torch.cuda.synchronize()
    m = 1 << 16
    for lo in (0, n - m):
        sub_off = offsets[lo:lo + m + 1]
        sub = data[int(sub_off[0]):int(sub_off[-1])]
        er, es, ee = cpu.batch("match", 4, 1, sub, offsets=sub_off - sub_off[0], threads=8)
        assert np.array_equal(r[lo:lo + m].cpu().numpy(), er)
        assert np.array_equal(s[lo:lo + m].cpu().numpy().astype(np.uint64), es)
        assert np.array_equal(e[lo:lo + m].cpu().numpy().astype(np.uint64), ee)
        assert len(np.unique(er)) > 50
    whole = _checksum(r, s, e)
    h = n // 2
    cut = int(offsets[h])
    o2 = (o[h:] - cut).contiguous()
    a = one_amd.match_batch(exe, d[:cut], 4, 1, offsets=o[:h + 1].contiguous())
    b = one_amd.match_batch(exe, d[cut:], 4, 1, offsets=o2)
    assert _checksum(*[torch.cat([x, y]) for x, y in zip(a, b)]) == whole


def test_full_size_config5_long_inputs_sample():
    """BASELINE configs[4] at full size: ~4K-state / 256-class DFA (2 MiB table, L2 gather),
    65,536 inputs x 64 KiB = 4 GiB.  128 inputs spread over the batch vs the oracle, and the
    halves property."""
    import torch
    n, L = 1 << 16, 1 << 16
    blob = load_dfa("syn4k")  # SYN-4K as the reference's minimizer + serializer wrote it
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    g = torch.Generator(device="cuda").manual_seed(5)
    data = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    for q in range(8):
        lo, hi = q * (n * L // 8), (q + 1) * (n * L // 8)
        data[lo:hi] = torch.randint(0, 256, (hi - lo,), generator=g, device="cuda",
                                    dtype=torch.uint8)
    r, s, e = one_amd.match_batch(exe, data, 4, 0, stride=L, n=n)
    torch.cuda.synchronize()
    idx = torch.arange(0, n, n >> 7, device="cuda")
    sample = data.view(n, L)[idx].contiguous().cpu().numpy().reshape(-1)
    er, es, ee = cpu.batch("match", 4, 0, sample, stride=L, n=len(idx), threads=8)
    assert np.array_equal(r[idx].cpu().numpy(), er)
    assert np.array_equal(s[idx].cpu().numpy().astype(np.uint64), es)
    assert np.array_equal(e[idx].cpu().numpy().astype(np.uint64), ee)
    h = n // 2
    a = one_amd.match_batch(exe, data[:h * L], 4, 0, stride=L, n=h)
    b = one_amd.match_batch(exe, data[h * L:], 4, 0, stride=L, n=h)
    assert _checksum(*[torch.cat([x, y]) for x, y in zip(a, b)]) == _checksum(r, s, e)


def test_cpp_mirror_through_cabi(tmp_path):
    """include/redgpu.hpp (the C++ mirror of the reference's names) compiled with g++ and run
    against the golden blobs: the reference's tests, re-read through the C-ABI."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "matcher_cabi_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpp", "matcher_cabi_test.cpp"), "-o", exe,
                    "-L", os.path.join(root, "one_amd"), "-lredgpu", "-lpthread",
                    "-Wl,-rpath," + os.path.join(root, "one_amd")], check=True)
    out = subprocess.run([exe, os.path.join(root, "tests", "golden", "dfas")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ mirror checks passed" in out.stdout


def test_config3_shape_4k_lines_vs_oracle():
    """BASELINE configs[2] shape at reduced count: 4 KiB lines (k_stream, 128-byte blocks),
    SYN-256 and URI-D, full Outcome, vs the oracle; plus the split/concatenate property."""
    n, stride = 40000, 4096
    for name in ("syn256", "uri"):
        blob = load_dfa(name)
        exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
        data = (W.fixed_lines(n, stride, 3, alphabet=False) if name == "syn256" else
                W.fixed_lines(n, stride, 3, plant=W.URI_PLANT, plant_at=1000))
        for si in (4, 5):
            r, s, e = one_amd.match_batch(exe, data, si, 0, stride=stride, n=n)
            er, es, ee = cpu.batch("match", si, 0, data, stride=stride, n=n, threads=8)
            assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
        assert one_amd.last_kernel().startswith("k_stream")
        h = 12345
        r1 = one_amd.match_batch(exe, data[:h * stride], 4, 0, stride=stride, n=h)[0]
        r2 = one_amd.match_batch(exe, data[h * stride:], 4, 0, stride=stride, n=n - h)[0]
        assert np.array_equal(np.concatenate([r1, r2]), one_amd.match_batch(
            exe, data, 4, 0, stride=stride, n=n)[0])


def test_config4_shape_log100_ragged_vs_oracle():
    """BASELINE configs[3] shape at reduced count: LOG-100 (3150 states, table in HBM/L2),
    2^17 ragged lines of 32..256 B, matchLong = match<styLast,true>."""
    n = 1 << 17
    blob = load_dfa("log100")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    data, offsets = W.ragged_lines(n, 32, 256, 4, heads=W.log100_heads(), head_every=2)
    r, s, e = one_amd.match_batch(exe, data, 4, 1, offsets=offsets)
    er, es, ee = cpu.batch("match", 4, 1, data, offsets=offsets, threads=8)
    assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)
    assert len(np.unique(er)) > 50  # many different signatures matched
    assert np.array_equal(one_amd.check_batch(exe, data, 5, 1, offsets=offsets),
                          cpu.batch("check", 5, 1, data, offsets=offsets, threads=8)[0])


def test_config5_shape_4k_state_dfa_long_inputs_vs_oracle():
    """BASELINE configs[4] shape at reduced count: ~4K-state / 256-class DFA (2 MiB table,
    L2-resident gather), 64 KiB inputs."""
    for blob in (load_dfa("syn4k"), random_dfa(4097, 256, 5, accept_frac=0.1)):
        _config5_shape(blob)


def _config5_shape(blob):
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    assert exe.info["table_kind"] in (4, 5)
    n, stride = 512, 65536
    data = W.fixed_lines(n, stride, 5, alphabet=False)
    r, s, e = one_amd.match_batch(exe, data, 4, 0, stride=stride, n=n)
    er, es, ee = cpu.batch("match", 4, 0, data, stride=stride, n=n, threads=8)
    assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)


def test_collect_on_gpu():
    """Red::collect batch form vs the reference's known answer (test/red.cpp:190-221) and the
    reference outputs in tests/golden/collect_vectors.npz, for 3 table placements."""
    import json
    import os
    from golden_util import GOLD
    kat = json.load(open(os.path.join(GOLD, "collect_kat.json")))
    exe = one_amd.Executable(unb64(kat["reda"]))
    assert one_amd.collect(exe, unb64(kat["text"])) == [tuple(x) for x in kat["expect"]]
    assert one_amd.collect(exe, b"new york") == [(1, 0, 8)]
    assert one_amd.collect(exe, b"") == []
    vec = np.load(os.path.join(GOLD, "collect_vectors.npz"))
    cap = int(vec["cap"][0])
    for name in ("newyork4", "num3", "newyork_loose"):
        for kw in ({}, {"force_global": True}):
            exe = one_amd.Executable(vec[name + "_blob"].tobytes(), **kw)
            counts, res, st, en = one_amd.collect_batch(exe, vec["data"], cap, offsets=vec["offsets"])
            assert np.array_equal(counts, vec[name + "_counts"]), name
            # slots beyond min(count, cap) are unspecified: compare the filled prefix only
            k = np.minimum(counts, cap).astype(np.int64)
            mask = np.arange(cap)[None, :] < k[:, None]
            assert np.array_equal(res[mask], vec[name + "_res"][mask])
            assert np.array_equal(st[mask], vec[name + "_start"][mask])
            assert np.array_equal(en[mask], vec[name + "_end"][mask])
    # truncation: cap smaller than the number of matches keeps the first cap records
    exe = one_amd.Executable(unb64(kat["reda"]))
    counts, res, st, en = one_amd.collect_batch(exe, unb64(kat["text"]), 2,
                                                offsets=[0, len(unb64(kat["text"]))])
    assert int(counts[0]) == 5 and res[0].tolist() == [1, 4] and en[0].tolist() == [11, 16]


@pytest.mark.parametrize("name", ["syn256", "uri", "dotstar_err", "newyork"])
def test_ragged_stream_kernel_tail_and_shapes(name):
    """k_ragged: lines ending inside the buffer's last 64 bytes (read through the tail pad), runs
    of empty lines (also at the very end), buffers shorter than one block, block-multiple and
    block-multiple+-1 lengths, one very long line among short ones."""
    blob = load_dfa(name)
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    rng = np.random.default_rng(11)
    shapes = [
        [0, 0, 0],                                   # nothing but empty lines
        [5],                                         # total < 64
        [63, 64, 65, 127, 128, 129, 0, 1, 200],      # around block multiples
        [300] + [3] * 40 + [0] * 70,                 # long line first, empties at the end
        list(rng.integers(0, 90, 3000)) + [0, 0, 1, 2, 0],
        [64] * 2050,                                 # whole blocks, > one tile
        [5000] + list(rng.integers(0, 40, 1500)),    # one line far longer than its wave-mates
        # many workgroup ranges (69 of ~1014 lines), each drained through its LDS cursor
        list(rng.integers(0, 200, 70000)),
        [0] * 5000 + [70] + [0] * 5000,              # ranges made of empty lines only
    ]
    for lens in shapes:
        offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        total = int(offsets[-1])
        data = (W.fixed_lines(1, max(total, 1), 21, alphabet=(name != "syn256"),
                              plant=b"x New York http://a.bc/ error " if name != "syn256" else None,
                              plant_every=1, plant_at=0)[:total])
        for si in (4, 5):
            er, es, ee = cpu.batch("match", si, 0, data, offsets=offsets)
            r, s, e = one_amd.match_batch(exe, data, si, 0, offsets=offsets)
            assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), \
                (name, si, lens[:8])
            r2, _, e2 = one_amd.match_batch(exe, data, si, 0, offsets=offsets, want_start=False)
            assert np.array_equal(r2, er) and np.array_equal(e2, ee)
            assert np.array_equal(one_amd.check_batch(exe, data, si, 0, offsets=offsets),
                                  cpu.batch("check", si, 0, data, offsets=offsets)[0])
    one_amd.match_batch(exe, data, 4, 0, offsets=offsets)
    assert one_amd.last_kernel().startswith("k_ragged")


@pytest.mark.parametrize("name", ["syn256", "uri", "uri_v6"])
def test_ragged_long_lines_first(name):
    """k_ragged with k_ragged_outliers' list (batches of >= 1024 lines): lines of at least
    max(512, 4 x mean) bytes are walked first, out of input order, and passed over in the
    workgroup's contiguous range.  Shapes: geometric lengths; a RUN of long lines (every lane of a
    wave finds its current and its next line passed over); long lines at the very start / end; one
    line that holds most of the buffer; lengths right at the threshold; and the same batches with
    REDGPU_F_NO_BUCKETING (plain input order) must give the same arrays."""
    blob = load_dfa(name)
    cpu = O.CpuOracle(blob)
    exe = one_amd.Executable(blob)
    exe_plain = one_amd.Executable(blob, no_bucketing=True)
    rng = np.random.default_rng(23)
    n = 20000
    geo = np.minimum(rng.geometric(1.0 / 90.0, n), 6000)
    run = rng.integers(0, 60, n)
    run[7000:7400] = 2000                      # 400 consecutive long lines, mean ~70 -> T = 512
    ends = rng.integers(0, 100, n)
    ends[:3] = 3000
    ends[-3:] = 2500
    one = rng.integers(0, 30, n)
    one[n // 2] = 600000                       # mean ~45 bytes, T = 512 ... and one huge line
    # mean exactly 128 -> T = 512: lengths 511 / 512 / 513 sit on both sides of the rule
    edge = np.full(n, 128, dtype=np.int64)
    edge[100:100 + 3 * 40:3] = 511
    edge[101:101 + 3 * 40:3] = 512
    edge[102:102 + 3 * 40:3] = 513
    edge[5000:5000 + 960] -= 48                # ... which the 120 lines above exceed by 40 x 1152
    assert edge.sum() == 128 * n
    shapes = {"geometric": geo, "run": run, "ends": ends, "one": one, "edge": edge}
    if name == "uri_v6":
        shapes = {"geometric": geo, "run": run}
    for label, lens in shapes.items():
        lens = np.asarray(lens, dtype=np.int64)
        offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        total = int(offsets[-1])
        data = W.fixed_lines(1, total, 29, alphabet=(name != "syn256"),
                             plant=b" see http://a.bc/x and https://[2001:db8::1]/ " if name != "syn256" else None,
                             plant_every=1, plant_at=0)[:total]
        for si in (4, 5):
            er, es, ee = cpu.batch("match", si, 0, data, offsets=offsets)
            for ex in (exe, exe_plain):
                r, s, e = one_amd.match_batch(ex, data, si, 0, offsets=offsets)
                assert one_amd.last_kernel().startswith("k_ragged"), one_amd.last_kernel()
                assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), \
                    (name, label, si)
            assert np.array_equal(one_amd.check_batch(exe, data, si, 0, offsets=offsets),
                                  cpu.batch("check", si, 0, data, offsets=offsets)[0]), (name, label)
        # delimiter-terminated lines (stride = 1 trailing byte): the rule counts the raw length
        if label == "geometric":
            keep = np.ones(total, dtype=bool)
            keep[(offsets[1:][lens > 0] - np.uint64(1)).astype(np.int64)] = False
            coffs = np.zeros(len(lens) + 1, dtype=np.uint64)
            coffs[1:] = np.cumsum(np.maximum(lens - 1, 0))
            er, es, ee = cpu.batch("match", 4, 0, data[keep], offsets=coffs)
            r, s, e = one_amd.match_batch(exe, data, 4, 0, offsets=offsets, stride=1)
            assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)


@pytest.mark.parametrize("name", ["uri", "newyork", "dotstar_err", "uri_v6", "uri_user", "syn256"])
def test_ragged_huge_lines_in_pieces(name):
    """k_ragged's pieces: a line of >= 8 T bytes (T = max(512, 4 x mean), batches of >= 1024
    lines, a DFA flagged `forgetful`; fused-u8, hot-row (uri_v6) and class (uri_user) tables) is walked as ceil(len / T) pieces at once,
    each entered through a 64-byte lead-in from the initial state; k_ragged_pieces_fold chains the
    records and re-walks pieces whose guess was wrong.  The text is dense with matches, among them
    URLs of ~300 bytes, so that piece borders fall INSIDE matches (wrong guesses, accepts that
    end exactly on a border, records that start in one piece and end in another); lines of 1 KB
    to 600 KB among short ones; delimiter-terminated lines (stride = 1); all of it against the
    oracle and against the same handle with REDGPU_F_NO_BUCKETING (whole lines, no list)."""
    blob = load_dfa(name)
    cpu = O.CpuOracle(blob)
    exe = one_amd.Executable(blob)
    exe_plain = one_amd.Executable(blob, no_bucketing=True)
    rng = np.random.default_rng(41)
    n = 18000
    lens = rng.integers(0, 90, n).astype(np.int64)
    where = rng.choice(n, 60, replace=False)
    lens[where] = rng.integers(1024, 50000, 60)
    lens[where[:8]] = [1024, 1025, 1087, 1088, 1089, 1536, 600000, 2048]
    # T = max(512, 4 x mean) in whole blocks; pieces from 8 T on: lengths on both sides of that
    for _ in range(3):
        t = (max(512, 4 * -(-int(lens.sum()) // n)) + 63) // 64 * 64
        lens[where[8:28]] = 8 * t + np.arange(-10, 10)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    total = int(offsets[-1])
    data = (W.random_bytes if name == "syn256" else W.alphabet_bytes)(total, 43).copy()
    if name != "syn256":
        short = np.frombuffer(W.URI_PLANT + b" New York error ", dtype=np.uint8)
        long_url = np.frombuffer(b" https://www.example.org/" + b"seg-ment_1.2/" * 22 + b"?q=1 New ", dtype=np.uint8)
        for at in range(0, total - 400, 211):
            data[at:at + len(short)] = short
        for at in range(97, total - 400, 1777):
            data[at:at + len(long_url)] = long_url
    info = exe.info
    for si in (4, 5):
        er, es, ee = cpu.batch("match", si, 0, data, offsets=offsets, threads=4)
        for ex in (exe, exe_plain):
            r, s, e = one_amd.match_batch(ex, data, si, 0, offsets=offsets)
            assert one_amd.last_kernel().startswith("k_ragged"), one_amd.last_kernel()
            bad = np.nonzero((r != er) | (s != es) | (e != ee))[0]
            assert bad.size == 0, (name, si, bad[:5], lens[bad[:5]], r[bad[:5]], er[bad[:5]],
                                   s[bad[:5]], es[bad[:5]], e[bad[:5]], ee[bad[:5]], info["forgetful"])
            r2, _, e2 = one_amd.match_batch(ex, data, si, 0, offsets=offsets, want_start=False)
            assert np.array_equal(r2, er) and np.array_equal(e2, ee)
        assert np.array_equal(one_amd.check_batch(exe, data, si, 0, offsets=offsets),
                              cpu.batch("check", si, 0, data, offsets=offsets, threads=4)[0])
    if name != "syn256":
        last = cpu.batch("match", 4, 0, data, offsets=offsets, threads=4)[0].astype(bool)
        assert int(last.sum()) > 1000 and last[where].sum() > 50   # the text does match, huge lines too
    # delimiter-terminated lines: the pieces are cut from the line without its trailing byte
    keep = np.ones(total, dtype=bool)
    keep[(offsets[1:][lens > 0] - np.uint64(1)).astype(np.int64)] = False
    coffs = np.zeros(n + 1, dtype=np.uint64)
    coffs[1:] = np.cumsum(np.maximum(lens - 1, 0))
    er, es, ee = cpu.batch("match", 4, 0, data[keep], offsets=coffs, threads=4)
    r, s, e = one_amd.match_batch(exe, data, 4, 0, offsets=offsets, stride=1)
    assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee)


@pytest.mark.parametrize("case", ["syn256", "rnd700x30", "rnd1500x40_big", "rnd3000_hot"])
def test_ragged_pieces_forced_on_random_dfas(case):
    """REDGPU_F_FORCE_PIECES on DFAs that do NOT forget their past (random transition tables):
    nearly every piece's entry state is guessed wrong, so k_ragged_pieces_fold chains the records
    serially and walks almost every piece again - through the fused table in LDS (syn256), the
    class table in row-offset form (700 x 30) and in index form (1500 x 40, above 64 KB), and a
    hot-row DFA's class table in global memory (3000 states).  Against the oracle and against the
    same DFA without pieces."""
    if case == "syn256":
        blob = load_dfa("syn256")
    else:
        n_st, n_cls = {"rnd700x30": (700, 30), "rnd1500x40_big": (1500, 40),
                       "rnd3000_hot": (3000, 64)}[case]
        blob = random_dfa(n_st, n_cls, 93, accept_frac=0.15)
    cpu = O.CpuOracle(blob)
    hot = case.endswith("_hot")   # (a random table shows no locality: hot rows only when asked for)
    exe = one_amd.Executable(blob, force_pieces=True, force_hot=hot)
    plain = one_amd.Executable(blob, force_hot=hot)
    kind = exe.info["table_kind"]
    assert kind == {"syn256": 1, "rnd700x30": 3, "rnd1500x40_big": 3, "rnd3000_hot": 6}[case], exe.info
    rng = np.random.default_rng(47)
    n = 3000
    lens = rng.integers(0, 80, n).astype(np.int64)
    lens[rng.choice(n, 12, replace=False)] = rng.integers(5000, 30000, 12)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = W.random_bytes(int(offsets[-1]), 49)
    for si in (4, 5):
        er, es, ee = cpu.batch("match", si, 0, data, offsets=offsets, threads=4)
        for ex in (exe, plain):
            r, s, e = one_amd.match_batch(ex, data, si, 0, offsets=offsets)
            assert one_amd.last_kernel().startswith("k_ragged"), one_amd.last_kernel()
            assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), (case, si)
        assert np.array_equal(one_amd.check_batch(exe, data, si, 0, offsets=offsets),
                              cpu.batch("check", si, 0, data, offsets=offsets, threads=4)[0])


def test_match_all_on_gpu():
    """matchAll batch form (include/Matcher.h:711-766) vs the reference's known answers
    (test/matcher.cpp:695-745, every format) and the reference outputs in
    tests/golden/matchall_stateful_vectors.npz, LDS and global table placements, both doLeader."""
    import json
    import os
    from golden_util import GOLD
    for k in json.load(open(os.path.join(GOLD, "matchall_kat.json"))):
        exe = one_amd.Executable(unb64(k["reda"]))
        assert one_amd.match_all(exe, unb64(k["text"])) == [tuple(x) for x in k["expect"]], k["src"]
        assert one_amd.match_all(exe, b"") == []
    vec = np.load(os.path.join(GOLD, "matchall_stateful_vectors.npz"))
    cap = int(vec["cap"][0])
    for name in ("set5", "loose2", "num3", "newyork", "err", "uri", "log100", "syn256"):
        for kw in ({}, {"force_global": True}):
            exe = one_amd.Executable(vec[name + "_blob"].tobytes(), **kw)
            for lead in (1, 0):
                counts, res, st, en = one_amd.match_all_batch(exe, vec["data"], cap, lead,
                                                              offsets=vec["offsets"])
                key = "%s_lead%d_" % (name, lead)
                assert np.array_equal(counts, vec[key + "counts"]), key
                # slots beyond min(count, cap) are unspecified: compare the filled prefix only
                k = np.minimum(counts, cap).astype(np.int64)
                mask = np.arange(cap)[None, :] < k[:, None]
                assert np.array_equal(res[mask], vec[key + "res"][mask]), key
                assert np.array_equal(st[mask], vec[key + "start"][mask]), key
                assert np.array_equal(en[mask], vec[key + "end"][mask]), key
    # fixed-stride form against the oracle
    blob = load_dfa("syn256")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    data = W.fixed_lines(3000, 96, 11, alphabet=False)
    got = one_amd.match_all_batch(exe, data, 4, True, stride=96, n=3000)
    exp = cpu.match_all_batch(data, 4, do_leader=True, stride=96, n=3000)
    assert np.array_equal(got[0], exp[0])
    mask = np.arange(4)[None, :] < np.minimum(exp[0], 4).astype(np.int64)[:, None]
    for g, e in zip(got[1:], exp[1:]):
        assert np.array_equal(g[mask], e[mask])
    # small capacities keep their records in registers (cap <= 4), larger ones store as they go
    rdata, roffs = W.ragged_lines(4000, 0, 120, 12, alphabet=False)
    for name2 in ("syn256", "uri"):
        blob2 = load_dfa(name2)
        exe2, cpu2 = one_amd.Executable(blob2), O.CpuOracle(blob2)
        d2, o2 = (rdata, roffs) if name2 == "syn256" else W.ragged_lines(
            4000, 0, 200, 13, heads=[W.URI_PLANT, b"x " + W.URI_PLANT + W.URI_PLANT], head_every=2)
        for cap2 in (1, 2, 3, 4, 6):
            for lead in (True, False):
                got = one_amd.match_all_batch(exe2, d2, cap2, lead, offsets=o2)
                exp = cpu2.match_all_batch(d2, cap2, do_leader=lead, offsets=o2)
                assert np.array_equal(got[0], exp[0]), (name2, cap2)
                mask = np.arange(cap2)[None, :] < np.minimum(exp[0], cap2).astype(np.int64)[:, None]
                for g, e in zip(got[1:], exp[1:]):
                    assert np.array_equal(g[mask], e[mask]), (name2, cap2, lead)


def test_stateful_matcher_on_gpu():
    """StatefulMatcher (include/Matcher.h:770-792): the reference's charByChar known answer
    (test/matcher.cpp:800-818) byte by byte, then n matchers advanced chunk by chunk - the state
    carried across three launches - against every advance() of the reference recorded in
    tests/golden/matchall_stateful_vectors.npz."""
    import json
    import os
    from golden_util import GOLD
    for k in json.load(open(os.path.join(GOLD, "stateful_kat.json"))):
        exe = one_amd.Executable(unb64(k["reda"]))
        sm = one_amd.StatefulMatcher(exe)
        assert sm.result() == k["initial"]
        got = [sm.advance(b) for b in unb64(k["text"])]
        assert got == k["per_byte"] and sm.result() == k["final"]
    vec = np.load(os.path.join(GOLD, "matchall_stateful_vectors.npz"))
    data, offsets = vec["data"], vec["offsets"]
    lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
    rng = np.random.default_rng(3)
    c1 = (rng.random(len(lens)) * (lens + 1)).astype(np.int64)          # 0..len
    c2 = c1 + (rng.random(len(lens)) * (lens - c1 + 1)).astype(np.int64)  # c1..len
    cuts = [np.zeros_like(lens), c1, c2, lens]
    for name in ("set5", "loose2", "err", "uri", "log100", "syn256"):
        per = vec[name + "_sm_per_byte"]
        ini = int(vec[name + "_sm_initial"][0])
        for kw in ({}, {"force_global": True}):
            exe = one_amd.Executable(vec[name + "_blob"].tobytes(), **kw)
            state = np.full(len(lens), one_amd.STATE_INITIAL, dtype=np.uint32)
            for a, b in zip(cuts[:-1], cuts[1:]):
                # chunk i = bytes [a_i, b_i) of input i, packed back to back
                clen = b - a
                coff = np.zeros(len(lens) + 1, dtype=np.uint64)
                coff[1:] = np.cumsum(clen)
                idx = np.concatenate([np.arange(int(offsets[i]) + int(a[i]),
                                                int(offsets[i]) + int(b[i]))
                                      for i in range(len(lens))]) if clen.sum() else \
                    np.zeros(0, dtype=np.int64)
                chunk = data[idx]
                res = one_amd.advance_batch(exe, chunk, state, offsets=coff)
                # result() after byte b_i - 1 of input i (or the initial result when b_i == 0)
                pos = offsets[:-1].astype(np.int64) + b - 1
                exp = np.where(b > 0, per[np.maximum(pos, 0)], ini)
                assert np.array_equal(res, exp), (name, kw)
            assert np.array_equal(res, vec[name + "_sm_final"])
    # fixed-stride chunks on the device, state resident between launches (torch plumbing)
    import torch
    blob = load_dfa("syn256")
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    n, L = 5000, 192
    host = W.fixed_lines(n, L, 21, alphabet=False)
    dev = torch.from_numpy(host).cuda()
    state = torch.full((n,), -1, dtype=torch.int32, device="cuda")   # 0xffffffff = fresh
    for lo in range(0, L, 64):
        chunk = dev.view(n, L)[:, lo:lo + 64].contiguous().view(-1)
        res = one_amd.advance_batch(exe, chunk, state, stride=64, n=n)
        assert one_amd.last_kernel() == "k_stream<advance>"
    torch.cuda.synchronize()
    exp = cpu.batch("check", "full", 0, host, stride=L, n=n)[0]
    assert np.array_equal(res.cpu().numpy(), exp)
    # the same walk through the generic kernel must leave the same state tokens behind
    state_g = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    exe_g = one_amd.Executable(blob, force_generic=True)
    res_g = one_amd.advance_batch(exe_g, dev, state_g, stride=L, n=n)
    assert one_amd.last_kernel() == "k_advance"
    assert torch.equal(res_g, res) and torch.equal(state_g, state)
    # 128-byte blocks (HALVES = 2) and a ragged last tile
    state2 = torch.full((n - 3,), -1, dtype=torch.int32, device="cuda")
    res2 = one_amd.advance_batch(exe, dev[:(n - 3) * L].view(-1), state2, stride=L, n=n - 3)
    assert one_amd.last_kernel() == "k_stream<advance>"
    # L = 192 is not a multiple of 128: HALVES = 1; 256-byte chunks of the first 128 lines' worth
    assert torch.equal(res2, res[:n - 3]) and torch.equal(state2, state[:n - 3])
    m = (n * L) // 256
    state3 = torch.full((m,), -1, dtype=torch.int32, device="cuda")
    res3 = one_amd.advance_batch(exe, dev[:m * 256], state3, stride=256, n=m)
    exp3 = cpu.batch("check", "full", 0, host[:m * 256], stride=256, n=m)[0]
    assert np.array_equal(res3.cpu().numpy(), exp3)


@pytest.mark.parametrize("mode", ["general", "hot", "cls"])
def test_differential_fuzz_smoke(mode):
    """scripts/fuzz_gpu.py (random DFAs x line shapes x verbs x styles x placement / kernel flags
    vs the oracle), a short fixed-seed run of each bias; the open-ended campaign is run by hand."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "scripts", "fuzz_gpu.py"), "80", "11"]
    if mode != "general":
        cmd.append(mode)
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=root)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
