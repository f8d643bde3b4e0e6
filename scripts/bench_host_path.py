"""PCIe-inclusive rate of the HOST-buffer entry points (redgpu_match_batch: copy in, kernel, copy
out, per call), pageable numpy memory as a C++ caller's std::vector would be.  1 thread and 4
threads sharing one handle.  Developer tool; never bench.py's `value`."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, one_amd
from one_amd import workloads as W
from golden_util import load_dfa

exe = one_amd.Executable(load_dfa("syn256"))
for n, L in ((1 << 20, 64), (1 << 18, 1024), (1 << 18, 4096)):
    data = W.fixed_lines(n, L, 1, alphabet=False)
    total = n * L + n * 20
    for _ in range(2):
        one_amd.match_batch(exe, data, 4, 0, stride=L, n=n)
    t0 = time.perf_counter()
    k = 5
    for _ in range(k):
        one_amd.match_batch(exe, data, 4, 0, stride=L, n=n)
    dt = (time.perf_counter() - t0) / k
    print("%8d x %5d B host buffers, 1 thread : %8.2f ms/call  %6.1f GB/s of input  (%5.1f GB/s in+out)" %
          (n, L, dt * 1e3, n * L / dt / 1e9, total / dt / 1e9), flush=True)
    T = 4
    shards = [data[i * (n // T) * L:(i + 1) * (n // T) * L] for i in range(T)]
    def work(i):
        for _ in range(k):
            one_amd.match_batch(exe, shards[i], 4, 0, stride=L, n=n // T)
    for i in range(T): one_amd.match_batch(exe, shards[i], 4, 0, stride=L, n=n // T)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = (time.perf_counter() - t0) / k
    print("%8d x %5d B host buffers, %d threads: %8.2f ms/batch %6.1f GB/s of input" %
          (n, L, T, dt * 1e3, n * L / dt / 1e9), flush=True)

# ---- the same through the C-ABI with caller buffers kept from call to call, pageable and pinned
# (redgpu_host_register): what a C++ caller that reuses its vectors gets
from one_amd import _lib
l = _lib.lib()
for n, L in ((1 << 20, 64), (1 << 18, 1024)):
    for pinned in (False, True):
        T = 4
        sets = []
        for i in range(T + 1):
            m = n if i == 0 else n // T
            d = W.fixed_lines(m, L, 1 + i, alphabet=False)
            r = np.zeros(m, dtype=np.int32); s = np.zeros(m, dtype=np.uint64); e = np.zeros(m, dtype=np.uint64)
            if pinned:
                for a in (d, r, s, e):
                    assert l.redgpu_host_register(a.ctypes.data, a.nbytes) == 0
            sets.append((d, r, s, e, m))
        def call(i):
            d, r, s, e, m = sets[i]
            rc = l.redgpu_match_batch(exe._h, 4, 0, d.ctypes.data, None, L, m, r.ctypes.data, s.ctypes.data, e.ctypes.data)
            assert rc == 0, l.redgpu_last_error()
        for _ in range(2): call(0)
        k = 8
        t0 = time.perf_counter()
        for _ in range(k): call(0)
        dt = (time.perf_counter() - t0) / k
        one = n * L / dt / 1e9
        for i in range(1, T + 1): call(i)
        def work(i):
            for _ in range(k): call(i)
        ths = [threading.Thread(target=work, args=(i,)) for i in range(1, T + 1)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt4 = (time.perf_counter() - t0) / k
        print("%8d x %5d B C-ABI, caller buffers %-8s: 1 thread %6.2f ms/call %6.1f GB/s | 4 threads (a quarter each) %6.2f ms %6.1f GB/s" %
              (n, L, "PINNED" if pinned else "pageable", dt * 1e3, one, dt4 * 1e3, n * L / dt4 / 1e9), flush=True)
        if pinned:
            for d, r, s, e, m in sets:
                for a in (d, r, s, e):
                    l.redgpu_host_unregister(a.ctypes.data)
import ctypes as _C
_rc = (_C.c_uint64 * 4)()
_lib.lib().redgpu_host_route_counts(_rc)
print("transfers by route: caller-pinned %d, pinned arena %d, registered for the call %d, pageable %d" % tuple(_rc))
