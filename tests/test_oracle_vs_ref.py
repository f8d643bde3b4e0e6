"""Fuzz: the C restatement against the REAL reference (oracle/_ref), when it is built.
Runs in the build container; on a box without /root/reference and without a prebuilt
oracle/_ref it is skipped (the committed golden vectors still pin the oracle there)."""
import numpy as np
import pytest

import oracle as O
from oracle.reda_writer import random_dfa

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")

STY = ["instant", "first", "tangent", "last", "full"]


def _compare(blob, data, offsets):
    ref, cpu = O.Reference(blob), O.CpuOracle(blob)
    for verb in ("check", "match", "scan", "search"):
        for sty in STY:
            for lead in (0, 1):
                a = ref.batch(verb, sty, lead, data, offsets=offsets)
                b = cpu.batch(verb, sty, lead, data, offsets=offsets)
                for x, y in zip(a, b):
                    assert np.array_equal(x, y), (verb, sty, lead)


def _compare_lists(blob, data, offsets, upto=150):
    """matchAll (public entry: doLeader = true) and StatefulMatcher, input by input"""
    ref, cpu = O.Reference(blob), O.CpuOracle(blob)
    for i in range(min(upto, len(offsets) - 1)):
        t = data[int(offsets[i]):int(offsets[i + 1])].tobytes()
        assert ref.match_all(t, 8) == cpu.match_all(t, True, 8), t
        ini, fin, per = ref.stateful(t)
        fin2, per2 = cpu.stateful(t)
        assert fin == fin2 and np.array_equal(per, per2), t
        assert cpu.stateful(b"")[0] == ini


def _ragged(rng, n, maxlen, hi):
    lens = rng.integers(0, maxlen, n)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    return rng.integers(0, hi, int(offsets[-1]), dtype=np.uint8), offsets


@pytest.mark.parametrize("seed", range(6))
def test_random_regex_dfas(seed):
    rng = np.random.default_rng(seed)
    atoms = ["a", "b", "c", "[ab]", "[^a]", ".", "(ab|c)", "a*", "b+", "c?", "(a|bc)*", "abc"]
    pats = []
    for k in range(int(rng.integers(1, 4))):
        rx = "".join(atoms[i] for i in rng.integers(0, len(atoms), int(rng.integers(1, 6))))
        pats.append((rx, k + 1, int(rng.integers(0, 8))))
    for fmt in (O.FMT_AUTO, O.FMT_2, O.FMT_4):
        blob = O.ref_compile(pats, fmt)
        data, offsets = _ragged(rng, 400, 24, 256)
        data = (data % 4 + ord("a")).astype(np.uint8)
        _compare(blob, data, offsets)
    _compare_lists(blob, data, offsets)


@pytest.mark.parametrize("seed", range(5))
def test_writer_blobs_accepted_by_reference(seed):
    """reda_writer's synthetic blobs are valid REDA to the reference, and both matchers agree
    on them (incl. reachable pure dead ends; up to the 4,097 states of configs[4]'s shape)."""
    n_states = [5, 40, 300, 700, 4097][seed]
    n_cls = [3, 17, 256, 64, 256][seed]
    blob = random_dfa(n_states, n_cls, seed, dead_frac=0.05 if seed % 2 else 0.0)
    assert O.ref_check_header(blob) is None
    rng = np.random.default_rng(seed + 100)
    data, offsets = _ragged(rng, 300, 80, 256)
    _compare(blob, data, offsets)
    _compare_lists(blob, data, offsets)
