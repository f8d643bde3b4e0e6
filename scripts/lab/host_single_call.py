#!/usr/bin/env python3
"""lab: latency of host-buffer calls of one line (the reference's single-input match()) and of
small batches, through the C-ABI."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import one_amd
from one_amd import _lib
from golden_util import load_dfa
exe = one_amd.Executable(load_dfa("newyork"))
l = _lib.lib()
for n, L in ((1, 16), (1, 256), (64, 64), (4096, 64)):
    data = np.frombuffer((b"I love New York. " * (n * L // 17 + 1))[: n * L], dtype=np.uint8).copy()
    res = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.uint64); en = np.zeros(n, dtype=np.uint64)
    f = lambda: l.redgpu_match_batch(exe._h, 4, 1, data.ctypes.data, None, L, n, res.ctypes.data, st.ctypes.data, en.ctypes.data)
    for _ in range(200): f()
    xs = []
    for _ in range(2000):
        a = time.perf_counter(); f(); xs.append((time.perf_counter() - a) * 1e6)
    xs.sort()
    print("%5d x %4d B: median %.1f us  p10 %.1f  p90 %.1f" % (n, L, xs[1000], xs[200], xs[1800]))
