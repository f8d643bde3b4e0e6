"""oracle - CPU checkers for the RED DFA match-execution path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``one_amd/``, ``include/``) never does.

* :class:`CpuOracle`  - ctypes face of ``oracle/liboracle.so`` (``red_oracle.c``, the plain-C
  restatement of /root/reference/quol/red/include/Matcher.h:333-640).
* :class:`Reference`  - ctypes face of ``oracle/_ref/libredref.so`` (the REAL reference,
  compiled by ``oracle/Makefile`` from the sources under /root/reference; present in the
  build container, travels to the GPU box as a prebuilt .so, never committed).
* :mod:`oracle.reda_writer` - builds REDA blobs from explicit transition tables (numpy) for
  synthetic DFAs and property tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_ORACLE = os.path.join(HERE, "liboracle.so")
LIB_REF = os.path.join(HERE, "_ref", "libredref.so")

STYLES = {"instant": 1, "first": 2, "tangent": 3, "last": 4, "full": 5}
VERBS = {"check": 0, "match": 1, "scan": 2, "search": 3}

_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)


def build(ref: bool = True) -> None:
    """Compile liboracle.so (and, when /root/reference is present, oracle/_ref)."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    if ref:
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


def have_ref() -> bool:
    return os.path.exists(LIB_REF)


class _Dfa(C.Structure):
    _fields_ = [
        ("blob", C.c_void_p), ("len", C.c_size_t), ("equiv", C.c_void_p),
        ("leader", C.c_void_p), ("base", C.c_void_p), ("stateCnt", C.c_uint32),
        ("initialOff", C.c_uint32), ("leaderOff", C.c_uint32), ("fmt", C.c_uint8),
        ("maxChar", C.c_uint8), ("leaderLen", C.c_uint8),
    ]


_liborc = None


def _orc():
    global _liborc
    if _liborc is None:
        if not os.path.exists(LIB_ORACLE):
            build(ref=False)
        lib = C.CDLL(LIB_ORACLE)
        lib.oracle_fnv1a32.restype = C.c_uint32
        lib.oracle_fnv1a32.argtypes = [C.c_void_p, C.c_size_t]
        lib.oracle_fnv1a64.restype = C.c_uint64
        lib.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        lib.oracle_calc_checksum.restype = C.c_uint32
        lib.oracle_calc_checksum.argtypes = [C.c_void_p, C.c_size_t]
        lib.oracle_check_header.restype = C.c_char_p
        lib.oracle_check_header.argtypes = [C.c_void_p, C.c_size_t]
        lib.oracle_dfa_init.restype = C.c_char_p
        lib.oracle_dfa_init.argtypes = [C.POINTER(_Dfa), C.c_void_p, C.c_size_t]
        for name in ("oracle_check", "oracle_scan"):
            f = getattr(lib, name)
            f.restype = C.c_int32
            f.argtypes = [C.POINTER(_Dfa), C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        for name in ("oracle_match", "oracle_search"):
            f = getattr(lib, name)
            f.restype = C.c_int32
            f.argtypes = [C.POINTER(_Dfa), C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                          _u64p, _u64p]
        lib.oracle_collect.restype = C.c_uint64
        lib.oracle_collect.argtypes = [C.POINTER(_Dfa), C.c_void_p, C.c_size_t, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_collect_batch.restype = None
        lib.oracle_collect_batch.argtypes = [C.POINTER(_Dfa), C.c_void_p, C.c_void_p, C.c_uint64,
                                             C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_replace.restype = C.c_uint64
        lib.oracle_replace.argtypes = [C.POINTER(_Dfa), C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                       C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p,
                                       C.c_uint64, C.POINTER(C.c_uint64)]
        lib.oracle_match_all.restype = C.c_uint64
        lib.oracle_match_all.argtypes = [C.POINTER(_Dfa), C.c_void_p, C.c_size_t, C.c_int,
                                         C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_match_all_batch.restype = None
        lib.oracle_match_all_batch.argtypes = [C.POINTER(_Dfa), C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_advance.restype = C.c_int32
        lib.oracle_advance.argtypes = [C.POINTER(_Dfa), C.POINTER(C.c_uint32), C.c_void_p,
                                       C.c_size_t, C.c_void_p]
        lib.oracle_advance_batch.restype = None
        lib.oracle_advance_batch.argtypes = [C.POINTER(_Dfa), C.c_void_p, C.c_void_p, C.c_uint64,
                                             C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        lib.oracle_batch.restype = None
        lib.oracle_batch.argtypes = [C.POINTER(_Dfa), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _liborc = lib
    return _liborc


def fnv1a32(b: bytes) -> int:
    return _orc().oracle_fnv1a32(b, len(b))


def fnv1a64(b: bytes) -> int:
    return _orc().oracle_fnv1a64(b, len(b))


def check_header(blob: bytes):
    """None when the blob is good, else the message (str)."""
    m = _orc().oracle_check_header(blob, len(blob))
    return None if m is None else m.decode()


def _style(s) -> int:
    return STYLES[s] if isinstance(s, str) else int(s)


def _as_u8(data) -> np.ndarray:
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(data), dtype=np.uint8)
    a = np.ascontiguousarray(data)
    assert a.dtype == np.uint8
    return a


class _Batchable:
    """Shared batch plumbing: returns (result int32[n], start uint64[n], end uint64[n])."""

    def _batch_call(self, verb, style, lead, data, offsets, stride, line_len, n, res, st, en,
                    threads):
        raise NotImplementedError

    def batch(self, verb, style, do_leader, data, *, offsets=None, stride=0, line_len=None,
              n=None, threads=1):
        data = _as_u8(data)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = len(offsets) - 1
            off_p = offsets.ctypes.data
            line_len = 0
        else:
            off_p = None
            if line_len is None:
                line_len = stride
            if n is None:
                n = (len(data) // stride) if stride else 0
        res = np.zeros(n, dtype=np.int32)
        st = np.zeros(n, dtype=np.uint64)
        en = np.zeros(n, dtype=np.uint64)
        if n:
            self._batch_call(VERBS[verb] if isinstance(verb, str) else verb, _style(style),
                             int(bool(do_leader)), data.ctypes.data, off_p, int(stride),
                             int(line_len), int(n), res.ctypes.data, st.ctypes.data,
                             en.ctypes.data, int(threads))
        return res, st, en


class CpuOracle(_Batchable):
    """The plain-C restatement (oracle/red_oracle.c) over one REDA blob."""

    def __init__(self, blob: bytes):
        self._blob = bytes(blob)
        self._buf = C.create_string_buffer(self._blob, len(self._blob))
        self._d = _Dfa()
        msg = _orc().oracle_dfa_init(C.byref(self._d), self._buf, len(self._blob))
        if msg is not None:
            raise ValueError(msg.decode())

    @property
    def info(self):
        d = self._d
        return dict(fmt=d.fmt, maxChar=d.maxChar, leaderLen=d.leaderLen, stateCnt=d.stateCnt,
                    initialOff=d.initialOff, leaderOff=d.leaderOff)

    def check(self, text: bytes, style, do_leader) -> int:
        return _orc().oracle_check(C.byref(self._d), text, len(text), _style(style),
                                   int(bool(do_leader)))

    def scan(self, text: bytes, style, do_leader) -> int:
        return _orc().oracle_scan(C.byref(self._d), text, len(text), _style(style),
                                  int(bool(do_leader)))

    def match(self, text: bytes, style, do_leader):
        s, e = C.c_uint64(0), C.c_uint64(0)
        r = _orc().oracle_match(C.byref(self._d), text, len(text), _style(style),
                                int(bool(do_leader)), C.byref(s), C.byref(e))
        return r, s.value, e.value

    def search(self, text: bytes, style, do_leader):
        s, e = C.c_uint64(0), C.c_uint64(0)
        r = _orc().oracle_search(C.byref(self._d), text, len(text), _style(style),
                                 int(bool(do_leader)), C.byref(s), C.byref(e))
        return r, s.value, e.value

    def _batch_call(self, verb, style, lead, data, offsets, stride, line_len, n, res, st, en,
                    threads):
        _orc().oracle_batch(C.byref(self._d), verb, style, lead, data, offsets, stride,
                            line_len, n, res, st, en, threads)

    def collect(self, text: bytes, cap: int = 64):
        """Red::collect: list of (result, start, end), and the number found."""
        res = np.zeros(cap, dtype=np.int32)
        st = np.zeros(cap, dtype=np.uint64)
        en = np.zeros(cap, dtype=np.uint64)
        k = _orc().oracle_collect(C.byref(self._d), text, len(text), cap, res.ctypes.data,
                                  st.ctypes.data, en.ctypes.data)
        m = min(k, cap)
        return [(int(res[i]), int(st[i]), int(en[i])) for i in range(m)], int(k)

    def collect_batch(self, data, cap, *, offsets=None, stride=0, n=None):
        """-> counts uint64[n], result int32[n,cap], start uint64[n,cap], end uint64[n,cap]"""
        data = _as_u8(data)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = len(offsets) - 1
        elif n is None:
            n = len(data) // stride if stride else 0
        counts = np.zeros(n, dtype=np.uint64)
        res = np.zeros((n, cap), dtype=np.int32)
        st = np.zeros((n, cap), dtype=np.uint64)
        en = np.zeros((n, cap), dtype=np.uint64)
        if n:
            _orc().oracle_collect_batch(C.byref(self._d), data.ctypes.data,
                                        offsets.ctypes.data if offsets is not None else None,
                                        int(stride), int(stride), n, cap, counts.ctypes.data,
                                        res.ctypes.data, st.ctypes.data, en.ctypes.data)
        return counts, res, st, en


    def replace(self, text: bytes, repl: bytes, style, do_leader=True, max_count=(1 << 62)):
        """replace<style,doLeader> (Matcher.h:643-706) -> (count, result bytes)"""
        cap = len(text) + (len(text) + 1) * len(repl) + 8
        out = np.zeros(cap, dtype=np.uint8)
        ol = C.c_uint64(0)
        k = _orc().oracle_replace(C.byref(self._d), text, len(text), _style(style),
                                  int(bool(do_leader)), repl, len(repl), int(max_count),
                                  out.ctypes.data, cap, C.byref(ol))
        return int(k), out[:ol.value].tobytes()

    def match_all(self, text: bytes, do_leader=True, cap: int = 64):
        """matchAll (Matcher.h:711-766): list of (result, start, end), and the number found."""
        res = np.zeros(cap, dtype=np.int32)
        st = np.zeros(cap, dtype=np.uint64)
        en = np.zeros(cap, dtype=np.uint64)
        k = _orc().oracle_match_all(C.byref(self._d), text, len(text), int(bool(do_leader)), cap,
                                    res.ctypes.data, st.ctypes.data, en.ctypes.data)
        m = min(k, cap)
        return [(int(res[i]), int(st[i]), int(en[i])) for i in range(m)], int(k)

    def match_all_batch(self, data, cap, *, do_leader=True, offsets=None, stride=0, n=None):
        """-> counts uint64[n], result int32[n,cap], start uint64[n,cap], end uint64[n,cap]"""
        data = _as_u8(data)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = len(offsets) - 1
        elif n is None:
            n = len(data) // stride if stride else 0
        counts = np.zeros(n, dtype=np.uint64)
        res = np.zeros((n, cap), dtype=np.int32)
        st = np.zeros((n, cap), dtype=np.uint64)
        en = np.zeros((n, cap), dtype=np.uint64)
        if n:
            _orc().oracle_match_all_batch(C.byref(self._d), int(bool(do_leader)), data.ctypes.data,
                                          offsets.ctypes.data if offsets is not None else None,
                                          int(stride), int(stride), n, cap, counts.ctypes.data,
                                          res.ctypes.data, st.ctypes.data, en.ctypes.data)
        return counts, res, st, en

    def stateful(self, text: bytes):
        """A fresh StatefulMatcher advanced over text -> (result(), int32[len] of every
        advance()'s return value)."""
        per = np.zeros(max(1, len(text)), dtype=np.int32)
        state = C.c_uint32(STATE_INITIAL)
        r = _orc().oracle_advance(C.byref(self._d), C.byref(state), text, len(text),
                                  per.ctypes.data)
        return int(r), per[:len(text)]

    def advance_batch(self, data, state, *, offsets=None, stride=0, n=None):
        """state: uint32[n] in/out (oracle tokens: row byte offsets; STATE_INITIAL = fresh).
        -> result int32[n]"""
        data = _as_u8(data)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = len(offsets) - 1
        elif n is None:
            n = len(data) // stride if stride else 0
        assert state.dtype == np.uint32 and len(state) == n and state.flags.c_contiguous
        res = np.zeros(n, dtype=np.int32)
        if n:
            _orc().oracle_advance_batch(C.byref(self._d), data.ctypes.data,
                                        offsets.ctypes.data if offsets is not None else None,
                                        int(stride), int(stride), n, state.ctypes.data,
                                        res.ctypes.data)
        return res


STATE_INITIAL = 0xFFFFFFFF


def split_lines(data, delim: int = 0x0A) -> np.ndarray:
    """The line rule of lib/Util.cpp:109-130 (sampleLines: `rv.emplace_back(start, it)` at every
    '\n', `start = it + 1`; what follows the last '\n' is never emitted), as offsets:
    offsets[0] = 0, offsets[k+1] = index just past the k-th delimiter.  Line k is
    data[offsets[k] : offsets[k+1] - 1] (its delimiter excluded)."""
    a = _as_u8(data)
    ends = np.flatnonzero(a == delim).astype(np.uint64) + np.uint64(1)
    return np.concatenate([np.zeros(1, dtype=np.uint64), ends])


def split_lines_loop(data: bytes, delim: int = 0x0A):
    """The same rule written as sampleLines' own loop (small inputs): list of line bytes."""
    out, start = [], 0
    for it, ch in enumerate(data):
        if ch == delim:
            out.append(bytes(data[start:it]))
            start = it + 1
    return out

# ------------------------------------------------------------------------------------------
_libref = None


def _ref():
    global _libref
    if _libref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libredref.so is not built (needs /root/reference)")
        lib = C.CDLL(LIB_REF)
        lib.ref_compile.restype = C.c_int
        lib.ref_compile.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t),
                                    _i32p, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.c_int,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p,
                                    C.c_size_t]
        lib.ref_syn_dfa.restype = C.c_int
        lib.ref_syn_dfa.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p,
                                    C.c_size_t]
        lib.ref_free.argtypes = [C.c_void_p]
        lib.ref_check_header.restype = C.c_char_p
        lib.ref_check_header.argtypes = [C.c_void_p, C.c_size_t]
        lib.ref_exec_create.restype = C.c_void_p
        lib.ref_exec_create.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        lib.ref_exec_destroy.argtypes = [C.c_void_p]
        for name in ("ref_check", "ref_scan"):
            f = getattr(lib, name)
            f.restype = C.c_int32
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        for name in ("ref_match", "ref_search"):
            f = getattr(lib, name)
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, _i32p, _u64p,
                          _u64p]
        lib.ref_collect.restype = C.c_uint64
        lib.ref_collect.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_replace.restype = C.c_uint64
        lib.ref_replace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                    C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_uint64,
                                    C.POINTER(C.c_uint64)]
        lib.ref_match_all.restype = C.c_uint64
        lib.ref_match_all.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        lib.ref_stateful.restype = C.c_int32
        lib.ref_stateful.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, _i32p, C.c_void_p]
        lib.ref_batch.restype = None
        lib.ref_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int]
        _libref = lib
    return _libref


# flags: include/Consts.h:12-16; languages: include/Parser.h:43-49; formats: Serializer.h:34-40
F_IGNORE_CASE, F_LOOSE_START, F_LOOSE_END = 1, 2, 4
LANG_RAW, LANG_AUTO, LANG_GLOB, LANG_EXACT = 1, 2, 3, 4
FMT_1, FMT_2, FMT_4, FMT_AUTO = 1, 2, 4, 255


class RefError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code  # -4 parse, -1 api, -2 exec, -3 limit


def ref_compile(patterns, fmt=FMT_AUTO) -> bytes:
    """patterns: list of (regex: bytes|str, result: int, flags: int[, lang: int])."""
    lib = _ref()
    n = len(patterns)
    pats, lens, res, flg, lng = [], [], [], [], []
    for p in patterns:
        rx = p[0].encode() if isinstance(p[0], str) else bytes(p[0])
        pats.append(rx)
        lens.append(len(rx))
        res.append(p[1])
        flg.append(p[2])
        lng.append(p[3] if len(p) > 3 else LANG_RAW)
    out, outlen = C.c_void_p(), C.c_size_t()
    err = C.create_string_buffer(512)
    rc = lib.ref_compile(n, (C.c_char_p * n)(*pats), (C.c_size_t * n)(*lens),
                         (C.c_int32 * n)(*res), (C.c_uint32 * n)(*flg), (C.c_int * n)(*lng),
                         fmt, C.byref(out), C.byref(outlen), err, 512)
    if rc != 0:
        raise RefError(rc, err.value.decode())
    blob = C.string_at(out.value, outlen.value)
    lib.ref_free(out)
    return blob


def ref_syn_dfa(nstates, seed, accept_every=7, max_result=5, fmt=FMT_AUTO) -> bytes:
    lib = _ref()
    out, outlen = C.c_void_p(), C.c_size_t()
    err = C.create_string_buffer(512)
    rc = lib.ref_syn_dfa(nstates, seed, accept_every, max_result, fmt, C.byref(out),
                         C.byref(outlen), err, 512)
    if rc != 0:
        raise RefError(rc, err.value.decode())
    blob = C.string_at(out.value, outlen.value)
    lib.ref_free(out)
    return blob


def ref_collect(blob: bytes, text: bytes, cap: int = 64):
    """The reference's Red::collect (lib/Red.cpp:103-116) -> ([(result,start,end)...], found)."""
    res = np.zeros(cap, dtype=np.int32)
    st = np.zeros(cap, dtype=np.uint64)
    en = np.zeros(cap, dtype=np.uint64)
    k = _ref().ref_collect(blob, len(blob), text, len(text), cap, res.ctypes.data, st.ctypes.data,
                           en.ctypes.data)
    m = min(k, cap)
    return [(int(res[i]), int(st[i]), int(en[i])) for i in range(m)], int(k)


def ref_check_header(blob: bytes):
    m = _ref().ref_check_header(blob, len(blob))
    return None if m is None else m.decode()


class Reference(_Batchable):
    """The real reference matcher (zezax::red) over one REDA blob."""

    def __init__(self, blob: bytes):
        err = C.create_string_buffer(512)
        self._h = _ref().ref_exec_create(blob, len(blob), err, 512)
        if not self._h:
            raise ValueError(err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None) and _libref is not None:
            _libref.ref_exec_destroy(self._h)
            self._h = None

    def check(self, text: bytes, style, do_leader) -> int:
        return _ref().ref_check(self._h, text, len(text), _style(style), int(bool(do_leader)))

    def scan(self, text: bytes, style, do_leader) -> int:
        return _ref().ref_scan(self._h, text, len(text), _style(style), int(bool(do_leader)))

    def _oc(self, fn, text, style, do_leader):
        r, s, e = C.c_int32(0), C.c_uint64(0), C.c_uint64(0)
        fn(self._h, text, len(text), _style(style), int(bool(do_leader)), C.byref(r),
           C.byref(s), C.byref(e))
        return r.value, s.value, e.value

    def match(self, text: bytes, style, do_leader):
        return self._oc(_ref().ref_match, text, style, do_leader)

    def search(self, text: bytes, style, do_leader):
        return self._oc(_ref().ref_search, text, style, do_leader)

    def replace(self, text: bytes, repl: bytes, style, do_leader=True, max_count=(1 << 62)):
        """The reference's replace<style,doLeader> -> (count, result bytes)"""
        cap = len(text) + (len(text) + 1) * len(repl) + 8
        out = np.zeros(cap, dtype=np.uint8)
        ol = C.c_uint64(0)
        k = _ref().ref_replace(self._h, text, len(text), _style(style), int(bool(do_leader)),
                               repl, len(repl), int(max_count), out.ctypes.data, cap,
                               C.byref(ol))
        return int(k), out[:ol.value].tobytes()

    def match_all(self, text: bytes, cap: int = 64):
        """The reference's matchAll(exec, sv, out) (always doLeader = true)."""
        res = np.zeros(cap, dtype=np.int32)
        st = np.zeros(cap, dtype=np.uint64)
        en = np.zeros(cap, dtype=np.uint64)
        k = _ref().ref_match_all(self._h, text, len(text), cap, res.ctypes.data, st.ctypes.data,
                                 en.ctypes.data)
        m = min(k, cap)
        return [(int(res[i]), int(st[i]), int(en[i])) for i in range(m)], int(k)

    def stateful(self, text: bytes):
        """A fresh reference StatefulMatcher advanced over text ->
        (result() before, result() after, int32[len] of every advance())."""
        per = np.zeros(max(1, len(text)), dtype=np.int32)
        ini = C.c_int32(0)
        r = _ref().ref_stateful(self._h, text, len(text), C.byref(ini), per.ctypes.data)
        return int(ini.value), int(r), per[:len(text)]

    def _batch_call(self, verb, style, lead, data, offsets, stride, line_len, n, res, st, en,
                    threads):
        _ref().ref_batch(self._h, verb, style, lead, data, offsets, stride, line_len, n, res,
                         st, en, threads)
