"""reda_writer - build a serialized "REDA" DFA blob from explicit tables.  TEST INFRASTRUCTURE.

Restates only the *layout* the reference's writer emits
(/root/reference/quol/red/lib/Serializer.cpp:30-53,158-201 and include/Serializer.h:42-77):
288-byte header, leader bytes (in class space) zero-padded to a multiple of 8, then one
fixed-size row per state: ``Value resultAndDeadEnd; Value offsets[maxChar+1]`` where an offset
is the target row's byte offset from ``base`` divided by ``sizeof(Value)``.  It is used to make
synthetic DFAs (any state count / class count) for property tests; blobs that come from real
regexes are made by the reference itself (oracle/_ref) and committed under tests/golden/.
"""
from __future__ import annotations

import struct

import numpy as np

from . import _orc

_VALUE_DT = {1: np.uint8, 2: np.uint16, 4: np.uint32}
_MAX_OFF = {1: 0xFF, 2: 0xFFFF, 4: 0xFFFFFFFF}
_MAX_RES = {1: 0x7F, 2: 0x7FFF, 4: 0x7FFFFFFF}


def optimal_format(n_states: int, max_char: int, max_result: int) -> int:
    """Proxy.h:152-161 + Serializer.cpp:133-155."""
    tot = n_states * max_char + 2 * n_states
    for fmt in (1, 2, 4):
        if max_result <= _MAX_RES[fmt] and tot <= _MAX_OFF[fmt]:
            return fmt
    raise ValueError("dfa too big for any format")


def write_reda(trans, results, *, equiv=None, dead_end=None, initial=1, leader=b"",
               leader_next=None, fmt=None) -> bytes:
    """trans: int array [n_states, n_classes] of target state ids; results: int [n_states];
    equiv: uint8[256] byte->class (default identity, needs n_classes == 256);
    dead_end: bool [n_states] (default: computed = every transition loops to self, the rule
    of lib/Dfa.cpp:27-37); leader: bytes in CLASS space; leader_next: state after the leader."""
    trans = np.asarray(trans, dtype=np.int64)
    n_states, n_cls = trans.shape
    results = np.asarray(results, dtype=np.int64)
    assert results.shape == (n_states,) and 1 <= n_cls <= 256
    if equiv is None:
        assert n_cls == 256
        equiv = np.arange(256, dtype=np.uint8)
    equiv = np.asarray(equiv, dtype=np.uint8)
    assert equiv.shape == (256,) and int(equiv.max()) < n_cls
    max_char = n_cls - 1
    if dead_end is None:
        dead_end = (trans == np.arange(n_states)[:, None]).all(axis=1)
    dead_end = np.asarray(dead_end, dtype=bool)
    if fmt is None:
        fmt = optimal_format(n_states, max_char, int(results.max(initial=0)))
    assert int(results.max(initial=0)) <= _MAX_RES[fmt]
    assert n_states * max_char + 2 * n_states <= _MAX_OFF[fmt], "offsets do not fit format"
    if leader_next is None:
        leader_next = initial
    assert len(leader) <= 255

    dt = _VALUE_DT[fmt]
    row_vals = n_cls + 1  # in Values
    rows = np.empty((n_states, row_vals), dtype=dt)
    rows[:, 0] = (results & _MAX_RES[fmt]) | (dead_end.astype(np.int64) << (8 * fmt - 1))
    rows[:, 1:] = trans * row_vals  # (target * rowBytes) / sizeof(Value)
    pad = (len(leader) + 7) & ~7
    hdr = bytearray(288)
    hdr[0:4] = b"REDA"
    struct.pack_into("<HH", hdr, 4, 1, 0)
    struct.pack_into("<BBBB", hdr, 12, fmt, max_char, len(leader), 0)
    struct.pack_into("<IIII", hdr, 16, n_states, initial * row_vals * fmt,
                     leader_next * row_vals * fmt, 0)
    hdr[32:288] = equiv.tobytes()
    blob = bytearray(bytes(hdr) + bytes(leader) + b"\0" * (pad - len(leader)) + rows.tobytes())
    csum = _orc().oracle_calc_checksum(bytes(blob), len(blob))
    struct.pack_into("<I", blob, 8, csum)
    return bytes(blob)


def random_dfa(n_states: int, n_cls: int, seed: int, *, accept_frac=0.15, max_result=5,
               with_error_state=True, dead_frac=0.0, fmt=None, equiv=None) -> bytes:
    """A random dense DFA.  State 0 = error (self-loops, result 0, dead end), 1 = initial,
    as the reference numbers them (include/Consts.h:21-22).  With dead_frac > 0 a share of
    transitions target the error state, so pure dead ends are reachable."""
    rng = np.random.default_rng(seed)
    lo = 1 if with_error_state else 0
    trans = rng.integers(lo, n_states, size=(n_states, n_cls), dtype=np.int64)
    if dead_frac > 0 and with_error_state:
        trans[rng.random((n_states, n_cls)) < dead_frac] = 0
    results = np.where(rng.random(n_states) < accept_frac,
                       rng.integers(1, max_result + 1, size=n_states), 0)
    if with_error_state:
        trans[0, :] = 0
        results[0] = 0
    results[1 if with_error_state else 0] = 0
    if equiv is None and n_cls < 256:
        equiv = rng.integers(0, n_cls, size=256, dtype=np.int64).astype(np.uint8)
        equiv[:n_cls] = np.arange(n_cls, dtype=np.uint8)  # every class is used
    return write_reda(trans, results, equiv=equiv, initial=1 if with_error_state else 0, fmt=fmt)
