"""Host-side mirror of the reference's matcher interface for the batch path.

Names follow /root/reference/quol/red: `Executable` (include/Executable.h:28-76), `Style`
(include/Matcher.h:67-74), `Outcome` fields result/start/end (include/Outcome.h:32-35), verbs
`check` / `match` / `scan` (include/Matcher.h:79-92,133-169) and the exception classes of
include/Except.h.  Everything here is plumbing over the C-ABI (include/redgpu.h): inputs may be
host arrays (numpy / bytes -> the library stages them) or torch tensors resident on the GPU
(-> the *_dev entry points, asynchronous on torch's current stream).
"""
from __future__ import annotations

import ctypes as C
import os
import enum

import numpy as np

from . import _lib


class Style(enum.IntEnum):  # include/Matcher.h:67-74
    styInvalid = 0
    styInstant = 1
    styFirst = 2
    styTangent = 3
    styLast = 4
    styFull = 5


styInstant, styFirst, styTangent, styLast, styFull = (Style.styInstant, Style.styFirst,
                                                      Style.styTangent, Style.styLast,
                                                      Style.styFull)


# include/Except.h:9-19
class RedExcept(RuntimeError):
    pass


class RedExceptApi(RedExcept):
    pass


class RedExceptExec(RedExcept):
    pass


class RedExceptLimit(RedExcept):
    pass


class RedExceptHip(RedExceptExec):
    """HIP runtime / device failure (no reference analogue)."""


_EXC = {_lib.EAPI: RedExceptApi, _lib.EEXEC: RedExceptExec, _lib.ELIMIT: RedExceptLimit,
        _lib.EHIP: RedExceptHip}


def _check(rc: int) -> None:
    if rc != _lib.OK:
        msg = _lib.lib().redgpu_last_error().decode()
        raise _EXC.get(rc, RedExcept)(msg)


def check_header(blob: bytes):
    """checkHeader (lib/Serializer.cpp:270-298): None when good, else the message."""
    msg = C.c_char_p()
    rc = _lib.lib().redgpu_reda_check(blob, len(blob), C.byref(msg))
    return None if rc == _lib.OK else msg.value.decode()


class Executable:
    """A validated serialized DFA uploaded to one GPU (mirrors zezax::red::Executable).

    device: HIP ordinal, None = current device, "none" = host-only handle (validate + repack,
    no HIP call; batch verbs on it raise RedExceptApi)."""

    def __init__(self, serialized: bytes, device=None, *, force_generic=False,
                 force_global=False, force_hot=False, no_bucketing=False,
                 force_stream=False, no_chunking=False, force_chunking=False,
                 lds_table_max=0, stream_chains=0, force_early=False, force_lean=False,
                 lean_chains=2, force_pieces=False):
        if serialized is None or len(serialized) == 0:
            raise RedExceptApi("serialized dfa string_view is empty")  # Executable.cpp:66
        o = _lib.Opts()
        o.device = (_lib.DEVICE_NONE if device == "none" else
                    _lib.DEVICE_CURRENT if device is None else int(device))
        o.lds_table_max = lds_table_max
        o.flags = Executable._flags_of(force_generic=force_generic, force_global=force_global,
                                       force_hot=force_hot, no_bucketing=no_bucketing,
                                       force_stream=force_stream, no_chunking=no_chunking,
                                       force_chunking=force_chunking, stream_chains=stream_chains,
                                       force_early=force_early, force_lean=force_lean,
                                       lean_chains=lean_chains, force_pieces=force_pieces)
        self._h = C.c_void_p()
        blob = bytes(serialized)
        _check(_lib.lib().redgpu_dfa_create(blob, len(blob), C.byref(o), C.byref(self._h)))

    @staticmethod
    def _flags_of(force_generic=False, force_global=False, force_hot=False, no_bucketing=False,
                  force_stream=False, no_chunking=False, force_chunking=False, stream_chains=0,
                  force_early=False, force_lean=False, lean_chains=2, force_pieces=False,
                  **_ignored) -> int:
        return ((_lib.F_FORCE_GENERIC if force_generic else 0) |
                (_lib.F_FORCE_GLOBAL if force_global else 0) |
                (_lib.F_FORCE_HOT if force_hot else 0) |
                (_lib.F_NO_BUCKETING if no_bucketing else 0) |
                (_lib.F_FORCE_STREAM if force_stream else 0) |
                (_lib.F_NO_CHUNKING if no_chunking else 0) |
                (_lib.F_FORCE_CHUNKING if force_chunking else 0) |
                (_lib.F_FORCE_EARLY if force_early else 0) |
                (_lib.F_FORCE_LEAN if force_lean else 0) |
                (_lib.F_LEAN_CHAINS_4 if lean_chains == 4 else 0) |
                (_lib.F_FORCE_PIECES if force_pieces else 0) |
                (_lib.F_STREAM_CHAINS_2 if stream_chains == 2 else 0) |
                (_lib.F_STREAM_CHAINS_4 if stream_chains == 4 else 0))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.redgpu_dfa_destroy(h)  # (at interpreter exit the module may be gone)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self) -> dict:
        i = _lib.Info()
        _check(_lib.lib().redgpu_dfa_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in _lib.Info._fields_}

    def tune(self, data, *, offsets=None, stride=0, n=None) -> dict:
        """redgpu_dfa_tune: re-rank the LDS-resident hot rows by the visits a sample of real
        input makes (host bytes / numpy, or a CUDA uint8 tensor).  Returns the new info."""
        l = _lib.lib()
        if _is_torch(data):
            import torch
            if offsets is not None:
                n = offsets.numel() - 1
            elif n is None:
                n = data.numel() // stride if stride else 0
            _check(l.redgpu_dfa_tune_dev(self._h, data.data_ptr(),
                                         offsets.data_ptr() if offsets is not None else None,
                                         stride, n,
                                         torch.cuda.current_stream(data.device).cuda_stream))
            return self.info
        a = _host_u8(data)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = len(offsets) - 1
        elif n is None:
            n = a.size // stride if stride else 0
        _check(l.redgpu_dfa_tune(self._h, a.ctypes.data if a.size else None,
                                 offsets.ctypes.data if offsets is not None else None, stride, n))
        return self.info

    def serialized(self) -> bytes:
        p, n = C.c_void_p(), C.c_size_t()
        _check(_lib.lib().redgpu_dfa_serialized(self._h, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value)


def last_kernel() -> str:
    return _lib.lib().redgpu_last_kernel().decode()


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _host_u8(data) -> np.ndarray:
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(data), dtype=np.uint8)
    a = np.ascontiguousarray(data)
    if a.dtype != np.uint8:
        raise RedExceptApi("input bytes must be uint8")
    return a


_TRACE = bool(os.environ.get("REDGPU_PY_TRACE"))  # debugging aid: host buffer addresses per call


def _run(verb: str, exe: Executable, style, do_leader, data, offsets, stride, n, want_start,
         want_end, out=None):
    """Returns (result, start, end); start/end are None when not requested / not a match verb."""
    l = _lib.lib()
    style = int(style)
    lead = 1 if do_leader else 0
    if _is_torch(data):
        import torch
        if not data.is_cuda or data.dtype != torch.uint8 or not data.is_contiguous():
            raise RedExceptApi("device input must be a contiguous uint8 CUDA tensor")
        dev = data.device
        if offsets is not None:
            if (not _is_torch(offsets) or offsets.dtype not in (torch.int64, torch.uint64)
                    or not offsets.is_cuda or not offsets.is_contiguous()):
                raise RedExceptApi("device offsets must be a contiguous int64 CUDA tensor")
            n = offsets.numel() - 1
            stride = int(stride or 0)  # with offsets: trailing bytes to drop per line
        elif n is None:
            n = data.numel() // stride if stride else 0
        if out is None:
            res = torch.empty(n, dtype=torch.int32, device=dev)
            st = torch.empty(n, dtype=torch.int64, device=dev) if want_start else None
            en = torch.empty(n, dtype=torch.int64, device=dev) if want_end else None
        else:
            res, st, en = out
        stream = torch.cuda.current_stream(dev).cuda_stream
        op = offsets.data_ptr() if offsets is not None else None
        if verb in ("match", "search"):
            fdev = l.redgpu_match_batch_dev if verb == "match" else l.redgpu_search_batch_dev
            rc = fdev(exe._h, style, lead, data.data_ptr(), op, stride, n,
                      res.data_ptr(), st.data_ptr() if st is not None else None,
                      en.data_ptr() if en is not None else None, stream)
        else:
            f = l.redgpu_check_batch_dev if verb == "check" else l.redgpu_scan_batch_dev
            rc = f(exe._h, style, lead, data.data_ptr(), op, stride, n, res.data_ptr(), stream)
        _check(rc)
        return res, st, en

    a = _host_u8(data)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        stride = int(stride or 0)  # with offsets: trailing bytes to drop per line
        if len(offsets) and int(offsets[-1]) > a.size:
            raise RedExceptApi("offsets run past the data buffer")
    else:
        if n is None:
            n = a.size // stride if stride else 0
        if stride * n > a.size:
            raise RedExceptApi("stride * n runs past the data buffer")
    res = np.zeros(n, dtype=np.int32)
    st = np.zeros(n, dtype=np.uint64) if want_start else None
    en = np.zeros(n, dtype=np.uint64) if want_end else None
    op = offsets.ctypes.data if offsets is not None else None
    dp = a.ctypes.data if a.size else None
    if _TRACE:
        print("redgpu host call %s n=%d data=%#x+%d offsets=%s res=%#x start=%s end=%s" % (
            verb, n, dp or 0, a.size, "%#x" % op if op else None, res.ctypes.data,
            "%#x" % st.ctypes.data if st is not None else None,
            "%#x" % en.ctypes.data if en is not None else None), flush=True)
    if verb in ("match", "search"):
        fhost = l.redgpu_match_batch if verb == "match" else l.redgpu_search_batch
        rc = fhost(exe._h, style, lead, dp, op, stride, n, res.ctypes.data,
                   st.ctypes.data if st is not None else None,
                   en.ctypes.data if en is not None else None)
    else:
        f = l.redgpu_check_batch if verb == "check" else l.redgpu_scan_batch
        rc = f(exe._h, style, lead, dp, op, stride, n, res.ctypes.data)
    _check(rc)
    return res, st, en


_VERBS = {"check": _lib.VERB_CHECK, "match": _lib.VERB_MATCH, "scan": _lib.VERB_SCAN,
          "search": _lib.VERB_SEARCH}


class Group:
    """One image of the same serialized DFA on each of several GPUs of a node (the device form
    of tools/thr_red.cpp:84-91: N workers over one shared Red).  devices may repeat a device."""

    def __init__(self, serialized: bytes, devices, **flags):
        o = _lib.Opts()
        o.flags = Executable._flags_of(**flags)
        o.lds_table_max = flags.get("lds_table_max", 0)
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        blob = bytes(serialized)
        _check(_lib.lib().redgpu_group_create(blob, len(blob), C.byref(o), devs, len(devices),
                                              C.byref(self._h)))
        self.devices = [int(d) for d in devices]

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.redgpu_group_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def plan(self, n, *, offsets=None, stride=0):
        """cuts[0..G]: shard g = lines [cuts[g], cuts[g+1]) - equal lines, or equal bytes when
        (host) offsets are given."""
        cuts = np.zeros(len(self.devices) + 1, dtype=np.uint64)
        op = None
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            op = offsets.ctypes.data
        _check(_lib.lib().redgpu_group_plan(self._h, op, stride, n, cuts.ctypes.data))
        return cuts

    def batch(self, verb, data, style, do_leader=True, *, offsets=None, stride=0, n=None):
        """Host buffers: (result, start, end) over all shards, one host thread per device."""
        a = _host_u8(data)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = len(offsets) - 1
        elif n is None:
            n = a.size // stride if stride else 0
        pos = verb in ("match", "search")
        res = np.zeros(n, dtype=np.int32)
        st = np.zeros(n, dtype=np.uint64) if pos else None
        en = np.zeros(n, dtype=np.uint64) if pos else None
        _check(_lib.lib().redgpu_group_batch(
            self._h, _VERBS[verb], int(style), 1 if do_leader else 0,
            a.ctypes.data if a.size else None,
            offsets.ctypes.data if offsets is not None else None, int(stride or 0), n,
            res.ctypes.data, st.ctypes.data if pos else None, en.ctypes.data if pos else None))
        return res, st, en

    def batch_dev(self, verb, shards, style, do_leader=True, *, stride=0, gather="peer"):
        """Device-resident shards: shards[g] = CUDA uint8 tensor on device g, or (data, offsets)
        for ragged lines.  Returns (result, start, end) CUDA tensors on devices[0], complete on
        torch's current stream of that device."""
        import torch
        G = len(self.devices)
        assert len(shards) == G
        datas, offs, ns = [], [], []
        for sh in shards:
            d, o = sh if isinstance(sh, tuple) else (sh, None)
            datas.append(d)
            offs.append(o)
            ns.append((o.numel() - 1) if o is not None else (d.numel() // stride if stride else 0))
        ragged = offs[0] is not None
        total = sum(ns)
        root = torch.device("cuda", self.devices[0])
        pos = verb in ("match", "search")
        res = torch.empty(total, dtype=torch.int32, device=root)
        st = torch.empty(total, dtype=torch.int64, device=root) if pos else None
        en = torch.empty(total, dtype=torch.int64, device=root) if pos else None
        dp = (C.c_void_p * G)(*[d.data_ptr() for d in datas])
        op = (C.c_void_p * G)(*[o.data_ptr() for o in offs]) if ragged else None
        na = (C.c_uint64 * G)(*ns)
        # every shard is ordered against torch's current stream of ITS device: the scan waits for
        # what that stream has queued (the op that produced the shard) and the stream waits for the
        # scan before it may reuse the shard's memory (torch's caching allocator is stream-ordered)
        ss = (C.c_void_p * G)(*[torch.cuda.current_stream(d.device).cuda_stream for d in datas])
        _check(_lib.lib().redgpu_group_batch_dev(
            self._h, _VERBS[verb], int(style), 1 if do_leader else 0, dp, op, int(stride or 0), na,
            res.data_ptr(), st.data_ptr() if pos else None, en.data_ptr() if pos else None,
            _lib.GATHER_RCCL if gather == "rccl" else _lib.GATHER_PEER, ss,
            torch.cuda.current_stream(root).cuda_stream))
        return res, st, en


def check_batch(exe, data, style, do_leader=True, *, offsets=None, stride=0, n=None, out=None):
    """check<style,doLeader> over every line (include/Matcher.h:363-410) -> result int32[n]."""
    return _run("check", exe, style, do_leader, data, offsets, stride, n, False, False,
                None if out is None else (out, None, None))[0]


def scan_batch(exe, data, style, do_leader=True, *, offsets=None, stride=0, n=None, out=None):
    """scan<style,doLeader> over every line (include/Matcher.h:498-554) -> result int32[n]."""
    return _run("scan", exe, style, do_leader, data, offsets, stride, n, False, False,
                None if out is None else (out, None, None))[0]


def match_batch(exe, data, style, do_leader=True, *, offsets=None, stride=0, n=None,
                want_start=True, want_end=True, out=None):
    """match<style,doLeader> over every line (include/Matcher.h:413-495) ->
    (result int32[n], start uint64[n] | None, end uint64[n] | None), the Outcome fields."""
    return _run("match", exe, style, do_leader, data, offsets, stride, n, want_start, want_end,
                out)


def batch_descs(batches, *, stride=0, want_start=True, want_end=True, outs=None, verb="match"):
    """The redgpu_batch array of a *_batches_dev call over CUDA tensors.  batches[k] = data, or
    (data, offsets) for ragged lines; outs[k] = (result, start, end) tensors (allocated when None).
    Returns (descriptor array, outs) - keep both alive until the stream has passed the call."""
    import torch
    pos = verb == "match"
    descs = (_lib.BatchDesc * len(batches))()
    made = []
    for k, sh in enumerate(batches):
        d, o = sh if isinstance(sh, tuple) else (sh, None)
        if not _is_torch(d) or not d.is_cuda or d.dtype != torch.uint8 or not d.is_contiguous():
            raise RedExceptApi("device input must be a contiguous uint8 CUDA tensor")
        n = (o.numel() - 1) if o is not None else (d.numel() // stride if stride else 0)
        if outs is None or outs[k] is None:
            res = torch.empty(n, dtype=torch.int32, device=d.device)
            st = torch.empty(n, dtype=torch.int64, device=d.device) if pos and want_start else None
            en = torch.empty(n, dtype=torch.int64, device=d.device) if pos and want_end else None
        else:
            res, st, en = outs[k]
        made.append((res, st, en))
        descs[k] = _lib.BatchDesc(d.data_ptr(), o.data_ptr() if o is not None else None,
                                  int(stride or 0), n, res.data_ptr(),
                                  st.data_ptr() if st is not None else None,
                                  en.data_ptr() if en is not None else None)
    return descs, made


def match_batches(exe, batches, style, do_leader=True, *, stride=0, want_start=True,
                  want_end=True, outs=None):
    """match<style,doLeader> over every line of SEVERAL device-resident batches in one call
    (redgpu_match_batches_dev: the caller's loop over its inputs, tools/bench.cpp:60-71) ->
    [(result, start, end)] per batch, complete on torch's current stream."""
    import torch
    descs, made = batch_descs(batches, stride=stride, want_start=want_start, want_end=want_end,
                              outs=outs)
    dev = (batches[0][0] if isinstance(batches[0], tuple) else batches[0]).device
    _check(_lib.lib().redgpu_match_batches_dev(exe._h, int(style), 1 if do_leader else 0, descs,
                                               len(batches),
                                               torch.cuda.current_stream(dev).cuda_stream))
    return made


def check_batches(exe, batches, style, do_leader=True, *, stride=0, outs=None):
    """check<style,doLeader> over several device-resident batches in one call -> [result]."""
    import torch
    descs, made = batch_descs(batches, stride=stride, outs=None if outs is None else
                              [(o, None, None) for o in outs], verb="check")
    dev = (batches[0][0] if isinstance(batches[0], tuple) else batches[0]).device
    _check(_lib.lib().redgpu_check_batches_dev(exe._h, int(style), 1 if do_leader else 0, descs,
                                               len(batches),
                                               torch.cuda.current_stream(dev).cuda_stream))
    return [m[0] for m in made]


def search_batch(exe, data, style, do_leader=True, *, offsets=None, stride=0, n=None,
                 want_start=True, want_end=True, out=None):
    """search<style,doLeader> over every line (include/Matcher.h:557-640): the first match found
    sliding over the line -> (result, start, end) like match_batch."""
    return _run("search", exe, style, do_leader, data, offsets, stride, n, want_start, want_end,
                out)


def collect_batch(exe, data, cap, *, offsets=None, stride=0, n=None):
    """Red::collect (lib/Red.cpp:103-116) over every line: all non-overlapping matches in order.
    Host arrays in, host arrays out: (counts uint64[n], result int32[n,cap], start uint64[n,cap],
    end uint64[n,cap]); counts[i] may exceed cap (only the first cap records are kept)."""
    a = _host_u8(data)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        stride = int(stride or 0)  # with offsets: trailing bytes to drop per line
    elif n is None:
        n = a.size // stride if stride else 0
    counts = np.zeros(n, dtype=np.uint64)
    res = np.zeros((n, cap), dtype=np.int32)
    st = np.zeros((n, cap), dtype=np.uint64)
    en = np.zeros((n, cap), dtype=np.uint64)
    _check(_lib.lib().redgpu_collect_batch(
        exe._h, a.ctypes.data if a.size else None,
        offsets.ctypes.data if offsets is not None else None, stride, n, cap, counts.ctypes.data,
        res.ctypes.data, st.ctypes.data, en.ctypes.data))
    return counts, res, st, en


def collect(exe, text: bytes, cap: int = 64):
    """Red::collect on one text -> list of (result, start, end)."""
    counts, res, st, en = collect_batch(exe, text, cap, offsets=[0, len(text)])
    k = int(min(counts[0], cap))
    return [(int(res[0, i]), int(st[0, i]), int(en[0, i])) for i in range(k)]


def match_all_batch(exe, data, cap, do_leader=True, *, offsets=None, stride=0, n=None):
    """matchAll (include/Matcher.h:711-766; the reference's public entry, lib/Matcher.cpp:97-102,
    runs with doLeader = true) over every line: one anchored walk reporting each maximal run of
    one accepted result.  Same return shape as collect_batch."""
    a = _host_u8(data)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        stride = int(stride or 0)  # with offsets: trailing bytes to drop per line
    elif n is None:
        n = a.size // stride if stride else 0
    counts = np.zeros(n, dtype=np.uint64)
    res = np.zeros((n, cap), dtype=np.int32)
    st = np.zeros((n, cap), dtype=np.uint64)
    en = np.zeros((n, cap), dtype=np.uint64)
    _check(_lib.lib().redgpu_match_all_batch(
        exe._h, int(bool(do_leader)), a.ctypes.data if a.size else None,
        offsets.ctypes.data if offsets is not None else None, stride, n, cap, counts.ctypes.data,
        res.ctypes.data, st.ctypes.data, en.ctypes.data))
    return counts, res, st, en


def match_all(exe, text: bytes, cap: int = 64):
    """matchAll(exec, text, out) on one text -> list of (result, start, end)."""
    counts, res, st, en = match_all_batch(exe, text, cap, True, offsets=[0, len(text)])
    k = int(min(counts[0], cap))
    return [(int(res[0, i]), int(st[0, i]), int(en[0, i])) for i in range(k)]


STATE_INITIAL = 0xFFFFFFFF


def advance_batch(exe, data, state, *, offsets=None, stride=0, n=None, out=None):
    """n StatefulMatchers (include/Matcher.h:770-792) advanced by one chunk each.
    `state` (uint32[n], in/out, updated in place) holds each matcher's state token -
    STATE_INITIAL for a fresh matcher; returns result int32[n] = result() after the chunk.
    numpy arrays run through the host entry point, torch CUDA tensors (data uint8, state int32
    viewed as u32) asynchronously on the current stream."""
    l = _lib.lib()
    if _is_torch(data):
        import torch
        if not data.is_cuda or data.dtype != torch.uint8 or not data.is_contiguous():
            raise RedExceptApi("device input must be a contiguous uint8 CUDA tensor")
        dev = data.device
        if offsets is not None:
            if (not _is_torch(offsets) or offsets.dtype not in (torch.int64, torch.uint64)
                    or not offsets.is_cuda or not offsets.is_contiguous()):
                raise RedExceptApi("device offsets must be a contiguous int64 CUDA tensor")
            n = offsets.numel() - 1
            stride = int(stride or 0)  # with offsets: trailing bytes to drop per line
        elif n is None:
            n = data.numel() // stride if stride else 0
        if (not _is_torch(state) or state.numel() != n or state.element_size() != 4 or
                state.device != dev or not state.is_contiguous()):
            raise RedExceptApi("state must be a 4-byte tensor of n elements on the data's device")
        res = out if out is not None else torch.empty(n, dtype=torch.int32, device=dev)
        _check(l.redgpu_advance_batch_dev(
            exe._h, data.data_ptr(), offsets.data_ptr() if offsets is not None else None, stride,
            n, state.data_ptr(), res.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        return res
    a = _host_u8(data)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        stride = int(stride or 0)  # with offsets: trailing bytes to drop per line
    elif n is None:
        n = a.size // stride if stride else 0
    if not (isinstance(state, np.ndarray) and state.dtype == np.uint32 and state.size == n and
            state.flags.c_contiguous):
        raise RedExceptApi("state must be a contiguous uint32 array of n elements")
    res = np.zeros(n, dtype=np.int32)
    _check(l.redgpu_advance_batch(exe._h, a.ctypes.data if a.size else None,
                                  offsets.ctypes.data if offsets is not None else None, stride,
                                  n, state.ctypes.data, res.ctypes.data))
    return res


def replace_batch(exe, data, repl: bytes, style, do_leader=True, max_count=(1 << 62), *,
                  offsets=None, stride=0, n=None):
    """replace<style,doLeader> (include/Matcher.h:643-706) over every line: each match replaced
    by `repl`, at most max_count per line.  Host arrays in and out:
    (counts uint64[n], out_offsets uint64[n+1], out uint8[out_offsets[n]])."""
    a = _host_u8(data)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        stride = int(stride or 0)  # with offsets: trailing bytes to drop per line
    elif n is None:
        n = a.size // stride if stride else 0
    counts = np.zeros(n, dtype=np.uint64)
    ooff = np.zeros(n + 1, dtype=np.uint64)
    r = np.frombuffer(bytes(repl), dtype=np.uint8)
    f = _lib.lib().redgpu_replace_batch
    args = (exe._h, int(style), 1 if do_leader else 0, a.ctypes.data if a.size else None,
            offsets.ctypes.data if offsets is not None else None, stride, n,
            r.ctypes.data if r.size else None, r.size, int(max_count), counts.ctypes.data,
            ooff.ctypes.data)
    # first guess: output about as long as the input; retry once with the exact size
    cap = int(a.size + 64)
    out = np.zeros(cap, dtype=np.uint8)
    _check(f(*args, out.ctypes.data, cap))
    total = int(ooff[n]) if n else 0
    if total > cap:
        out = np.zeros(total, dtype=np.uint8)
        _check(f(*args, out.ctypes.data, total))
    return counts, ooff, out[:total]


def replace(exe, text: bytes, repl: bytes, max_count, style, do_leader=True):
    """replace(exec, text, repl, out, max, style) on one text -> (count, rewritten bytes)."""
    counts, ooff, out = replace_batch(exe, text, repl, style, do_leader, max_count,
                                      offsets=[0, len(text)])
    return int(counts[0]), out.tobytes()


def split_lines(exe, data, delim=b"\n", cap=None):
    """redgpu_split_lines: offsets of the delimiter-terminated lines of a raw text buffer, found
    on the device (the rule of lib/Util.cpp:109-130: bytes after the last delimiter are not a
    line).  Line k = data[offsets[k] : offsets[k+1]] INCLUDING its delimiter - pass the offsets
    to the *_batch verbs with stride=1.  Returns (offsets, n_found): numpy uint64[min(n, cap)+1]
    for host input; for a CUDA uint8 tensor, an int64 tensor of cap+1 entries and a 1-element
    int64 tensor, both on the device, asynchronously on the current stream."""
    l = _lib.lib()
    d = delim[0] if isinstance(delim, (bytes, bytearray)) else int(delim)
    if _is_torch(data):
        import torch
        if not data.is_cuda or data.dtype != torch.uint8 or not data.is_contiguous():
            raise RedExceptApi("device input must be a contiguous uint8 CUDA tensor")
        if cap is None:
            raise RedExceptApi("device split_lines needs cap (room in the offsets tensor)")
        offs = torch.empty(cap + 1, dtype=torch.int64, device=data.device)
        cnt = torch.empty(1, dtype=torch.int64, device=data.device)
        _check(l.redgpu_split_lines_dev(exe._h, data.data_ptr(), data.numel(), d, offs.data_ptr(),
                                        cap, cnt.data_ptr(),
                                        torch.cuda.current_stream(data.device).cuda_stream))
        return offs, cnt
    a = _host_u8(data)
    cnt = C.c_uint64(0)
    dp = a.ctypes.data if a.size else None
    if cap is None:
        # size the result exactly: a first call that keeps no line only counts
        one = np.zeros(1, dtype=np.uint64)
        _check(l.redgpu_split_lines(exe._h, dp, a.size, d, one.ctypes.data, 0, C.byref(cnt)))
        cap = int(cnt.value)
    offs = np.zeros(cap + 1, dtype=np.uint64)
    _check(l.redgpu_split_lines(exe._h, dp, a.size, d, offs.ctypes.data, cap, C.byref(cnt)))
    return offs[: min(cnt.value, cap) + 1], int(cnt.value)


def match_text(exe, data, style, do_leader=True, *, delim=b"\n", cap=None, want_start=True,
               want_end=True):
    """redgpu_match_text[_dev]: the delimiter-terminated lines of a raw text buffer, each
    matched (tools/skim_red.cpp:36-46 over lib/Util.cpp:109-130's lines), in one call.
    Host input -> (offsets uint64[k+1], n_found, result int32[k], start | None, end | None) with
    k = min(n_found, cap); cap=None sizes the arrays with a counting call first.
    A CUDA uint8 tensor -> (offsets int64[cap+1], count int64[1], result int32[cap], start, end)
    tensors, asynchronously on the current stream; entries from min(count, cap) on are untouched.
    want_start = want_end = False is check<style,doLeader>."""
    l = _lib.lib()
    d = delim[0] if isinstance(delim, (bytes, bytearray)) else int(delim)
    if _is_torch(data):
        import torch
        if not data.is_cuda or data.dtype != torch.uint8 or not data.is_contiguous():
            raise RedExceptApi("device input must be a contiguous uint8 CUDA tensor")
        if cap is None:
            raise RedExceptApi("device match_text needs cap (room in the output tensors)")
        dev = data.device
        offs = torch.empty(cap + 1, dtype=torch.int64, device=dev)
        cnt = torch.empty(1, dtype=torch.int64, device=dev)
        res = torch.empty(cap, dtype=torch.int32, device=dev)
        st = torch.empty(cap, dtype=torch.int64, device=dev) if want_start else None
        en = torch.empty(cap, dtype=torch.int64, device=dev) if want_end else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        if st is None and en is None:
            _check(l.redgpu_check_text_dev(exe._h, int(style), 1 if do_leader else 0,
                                           data.data_ptr(), data.numel(), d, offs.data_ptr(), cap,
                                           cnt.data_ptr(), res.data_ptr(), stream))
        else:
            _check(l.redgpu_match_text_dev(exe._h, int(style), 1 if do_leader else 0,
                                           data.data_ptr(), data.numel(), d, offs.data_ptr(), cap,
                                           cnt.data_ptr(), res.data_ptr(),
                                           st.data_ptr() if st is not None else None,
                                           en.data_ptr() if en is not None else None, stream))
        return offs, cnt, res, st, en
    a = _host_u8(data)
    dp = a.ctypes.data if a.size else None
    cnt = C.c_uint64(0)
    if cap is None:
        one = np.zeros(1, dtype=np.uint64)
        _check(l.redgpu_split_lines(exe._h, dp, a.size, d, one.ctypes.data, 0, C.byref(cnt)))
        cap = int(cnt.value)
    offs = np.zeros(cap + 1, dtype=np.uint64)
    res = np.zeros(cap, dtype=np.int32)
    st = np.zeros(cap, dtype=np.uint64) if want_start else None
    en = np.zeros(cap, dtype=np.uint64) if want_end else None
    _check(l.redgpu_match_text(exe._h, int(style), 1 if do_leader else 0, dp, a.size, d,
                               offs.ctypes.data, cap, C.byref(cnt), res.ctypes.data,
                               st.ctypes.data if st is not None else None,
                               en.ctypes.data if en is not None else None))
    k = min(int(cnt.value), cap)
    return (offs[:k + 1], int(cnt.value), res[:k], st[:k] if st is not None else None,
            en[:k] if en is not None else None)


class StatefulMatcher:
    """Mirror of zezax::red::StatefulMatcher (include/Matcher.h:770-792): `advance(byte)` and
    `result()`; `advance_bytes` feeds a whole chunk in one kernel launch.  The executable must
    outlive the matcher, as in the reference."""

    def __init__(self, exe):
        self._exe = exe
        self._state = np.full(1, STATE_INITIAL, dtype=np.uint32)
        self._result = int(advance_batch(exe, b"", self._state, offsets=[0, 0])[0])

    def advance(self, byte) -> int:
        b = bytes([byte]) if isinstance(byte, int) else bytes(byte[:1])
        return self.advance_bytes(b)

    def advance_bytes(self, chunk: bytes) -> int:
        self._result = int(advance_batch(self._exe, chunk, self._state,
                                         offsets=[0, len(chunk)])[0])
        return self._result

    def result(self) -> int:
        return self._result


# single-input forms keep the reference's signatures; they are batches of one ON THE GPU
def check(exe, text: bytes, style, do_leader=True) -> int:
    return int(check_batch(exe, text, style, do_leader, offsets=[0, len(text)])[0])


def scan(exe, text: bytes, style, do_leader=True) -> int:
    return int(scan_batch(exe, text, style, do_leader, offsets=[0, len(text)])[0])


def match(exe, text: bytes, style, do_leader=True):
    r, s, e = match_batch(exe, text, style, do_leader, offsets=[0, len(text)])
    return int(r[0]), int(s[0]), int(e[0])


def search(exe, text: bytes, style, do_leader=True):
    r, s, e = search_batch(exe, text, style, do_leader, offsets=[0, len(text)])
    return int(r[0]), int(s[0]), int(e[0])
