"""GB/s of the "next row" verbs: StatefulMatcher chunks (advance), matchAll,
collect, search - one stream, inputs resident.  Developer tool (bench.py is the contract bench).
usage: bench_lists.py [dfa-name]      (TEXT_ONLY=1: only the raw-text section)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch, one_amd
from one_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "syn256"
blob = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "dfas", name + ".reda"), "rb").read()
exe = one_amd.Executable(blob)
l = _lib.lib()


def timed(fn, it):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


TEXT_ONLY = bool(os.environ.get("TEXT_ONLY"))
for n, L in [] if TEXT_ONLY else [(1 << 20, 64), (1 << 18, 4096), (1 << 20, 96)]:
    total = n * L
    data = torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda")
    state = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    res = torch.empty(n, dtype=torch.int32, device="cuda")
    ms = timed(lambda: one_amd.advance_batch(exe, data, state, stride=L, n=n, out=res), 50)
    print("advance   %8d x %5d B  %8.1f us  %7.1f GB/s  %s" % (n, L, ms * 1e3, total / ms / 1e6, one_amd.last_kernel()), flush=True)
    cap = 4
    cnt = torch.empty(n, dtype=torch.int64, device="cuda")
    r = torch.empty(n * cap, dtype=torch.int32, device="cuda")
    s = torch.empty(n * cap, dtype=torch.int64, device="cuda")
    e = torch.empty(n * cap, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for label, fn in (
            ("matchAll", lambda: l.redgpu_match_all_batch_dev(exe._h, 1, data.data_ptr(), None, L, n, cap, cnt.data_ptr(), r.data_ptr(), s.data_ptr(), e.data_ptr(), stream)),
            ("collect", lambda: l.redgpu_collect_batch_dev(exe._h, data.data_ptr(), None, L, n, cap, cnt.data_ptr(), r.data_ptr(), s.data_ptr(), e.data_ptr(), stream))):
        if label == "collect" and L > 1024:
            continue
        ms = timed(fn, 10)
        print("%-9s %8d x %5d B  %8.1f us  %7.1f GB/s  %s" % (label, n, L, ms * 1e3, total / ms / 1e6, one_amd.last_kernel()), flush=True)
    del data
    torch.cuda.empty_cache()

# line splitting on the device, then matching the split lines (delimiter dropped)
import numpy as np
from one_amd import workloads as W
for total, p_nl in ((1 << 26, 1 / 64), (1 << 28, 1 / 144), (1 << 28, None)):
    host = W.alphabet_bytes(total, 12).copy()
    rng = np.random.default_rng(1)
    if p_nl is None:   # line lengths uniform in 32..256 instead of geometric
        ends = np.cumsum(rng.integers(33, 258, total // 100))
        host[ends[ends < total]] = 0x0A
        label = "uniform 32-256 B lines"
    else:
        host[rng.random(total) < p_nl] = 0x0A
        label = "geometric lines, mean %d B" % round(1 / p_nl)
    dev = torch.from_numpy(host).cuda()
    cap = int(total * (p_nl or 1 / 120) * 1.2) + 16
    ms = timed(lambda: one_amd.split_lines(exe, dev, cap=cap), 20)
    offs, cnt = one_amd.split_lines(exe, dev, cap=cap)
    n = int(cnt.item())
    print("text: %s" % label)
    print("split     %9d bytes -> %8d lines  %8.1f us  %7.1f GB/s" % (total, n, ms * 1e3, total / ms / 1e6), flush=True)
    o = offs[:n + 1].contiguous()
    res = torch.empty(n, dtype=torch.int32, device="cuda")
    st = torch.empty(n, dtype=torch.int64, device="cuda")
    en = torch.empty(n, dtype=torch.int64, device="cuda")
    ms = timed(lambda: one_amd.match_batch(exe, dev, 4, False, offsets=o, stride=1, out=(res, st, en)), 20)
    print("match     same lines (delimiter dropped)      %8.1f us  %7.1f GB/s  %s" % (ms * 1e3, total / ms / 1e6, one_amd.last_kernel()), flush=True)
    ms = timed(lambda: one_amd.match_text(exe, dev, 4, False, cap=cap), 20)
    print("match_text (split + match in one call)        %8.1f us  %7.1f GB/s  %s" % (ms * 1e3, total / ms / 1e6, one_amd.last_kernel()), flush=True)
    del dev

# replace<styLast,false> on device-resident lines (redgpu_replace_batch_dev): count pass, prefix sum
# of the rewritten lengths, write pass - sizes only (out = NULL) and with the rewritten bytes
import ctypes as C
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
from golden_util import load_dfa as _ld
uexe = one_amd.Executable(_ld("uri"))
for n, L in () if TEXT_ONLY else ((1 << 20, 64), (1 << 18, 1024)):
    host = W.fixed_lines(n, L, 5, plant=W.URI_PLANT, plant_every=4, plant_at=8)
    dev = torch.from_numpy(host).cuda()
    repl = torch.from_numpy(np.frombuffer(b"<url>", dtype=np.uint8).copy()).cuda()
    counts = torch.empty(n, dtype=torch.int64, device="cuda")
    ooff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    out = torch.empty(n * L + 64, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for label, optr in (("sizes only", None), ("with bytes", out.data_ptr())):
        fn = lambda: l.redgpu_replace_batch_dev(uexe._h, 4, 0, dev.data_ptr(), None, L, n, repl.data_ptr(), 5,
                                                1 << 62, counts.data_ptr(), ooff.data_ptr(), optr, out.numel(), stream)
        assert fn() == 0, l.redgpu_last_error()
        ms = timed(fn, 10)
        print("replace   %8d x %5d B  URI-D, url in every 4th line, %-10s  %8.1f us  %7.1f GB/s  (%d replacements)" %
              (n, L, label, ms * 1e3, n * L / ms / 1e6, int(counts.sum().item())), flush=True)
    del dev
