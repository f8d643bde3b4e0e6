"""CPU-only: the identity behind redgpu_info.suffix_closed, checked on the ORACLE itself.

For a DFA whose language is closed under prepending anything (L = SIGMA* L: every pattern added
with a loose start) the library runs scan as check and search as match.  That is a
claim about the reference's algorithm, so it is pinned here against the CPU restatement of the
reference (oracle/red_oracle.c, itself pinned to the reference's vectors): on such DFAs the
oracle's scan / search must equal its own check / match on every input and style; on DFAs that
are not suffix-closed the two must differ somewhere (the flag is not vacuous)."""
import numpy as np
import pytest

import one_amd
import oracle as O
from golden_util import load_dfa
from one_amd import workloads as W


def _lines(seed, n=3000):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 120, n)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = W.alphabet_bytes(int(offsets[-1]) + 1, seed)[: int(offsets[-1])].copy()
    plants = [W.URI_PLANT, b"New York", b"error", b"York", b"ftp://a.bc/ x", b"123abcd"]
    for li in range(0, n, 4):
        pl = np.frombuffer(plants[(li // 4) % len(plants)], dtype=np.uint8)
        if lens[li] >= len(pl):
            at = int(offsets[li]) + int(rng.integers(0, lens[li] - len(pl) + 1))
            data[at:at + len(pl)] = pl
    return data, offsets


@pytest.mark.parametrize("name", ["uri", "newyork", "dotstar_err", "uri_user"])
def test_scan_is_check_and_search_is_match_on_suffix_closed_dfas(name):
    blob = load_dfa(name)
    assert one_amd.Executable(blob, device="none").info["suffix_closed"] == 1
    cpu = O.CpuOracle(blob)
    data, offsets = _lines(5)
    for style in range(1, 6):
        sc = cpu.batch("scan", style, 0, data, offsets=offsets, threads=4)[0]
        ck = cpu.batch("check", style, 0, data, offsets=offsets, threads=4)[0]
        assert np.array_equal(sc, ck), (name, style)
        sr = cpu.batch("search", style, 0, data, offsets=offsets, threads=4)
        mt = cpu.batch("match", style, 0, data, offsets=offsets, threads=4)
        assert all(np.array_equal(a, b) for a, b in zip(sr, mt)), (name, style)
    assert (cpu.batch("scan", 1, 0, data, offsets=offsets, threads=4)[0] != 0).any()  # some line matches


@pytest.mark.parametrize("name", ["err", "aab", "num3"])
def test_the_identity_fails_without_the_flag(name):
    blob = load_dfa(name)
    assert one_amd.Executable(blob, device="none").info["suffix_closed"] == 0
    cpu = O.CpuOracle(blob)
    data, offsets = _lines(6)
    differs = False
    for style in (1, 4):
        sc = cpu.batch("scan", style, 0, data, offsets=offsets, threads=4)[0]
        ck = cpu.batch("check", style, 0, data, offsets=offsets, threads=4)[0]
        differs = differs or not np.array_equal(sc, ck)
    assert differs, name


@pytest.mark.parametrize("name,uniform", [("uri", True), ("dotstar_err", True), ("uri_user", True),
                                          ("newyork", False), ("num3", False)])
def test_check_styles_agree_on_single_result_dfas(name, uniform):
    """The second identity: when every accepting state reports one result, check
    answers the same for styInstant / styFirst / styTangent / styLast - the library then runs the
    first three as styLast on the streaming kernels.  With several results they differ."""
    cpu = O.CpuOracle(load_dfa(name))
    data, offsets = _lines(7)
    last = cpu.batch("check", 4, 0, data, offsets=offsets, threads=4)[0]
    same = all(np.array_equal(cpu.batch("check", st, 0, data, offsets=offsets, threads=4)[0], last)
               for st in (1, 2, 3))
    assert same == uniform, name
