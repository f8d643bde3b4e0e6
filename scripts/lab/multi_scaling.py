#!/usr/bin/env python3
"""lab: time of one redgpu_match_batches_dev launch against the number of batches in it, for
several buffer layouts.  python scripts/lab/multi_scaling.py [dfa]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import one_amd
from one_amd import _lib
from golden_util import load_dfa

name = sys.argv[1] if len(sys.argv) > 1 else "syn256"
exe = one_amd.Executable(load_dfa(name))
l = _lib.lib()
n, L = 1 << 20, 64
st = torch.cuda.current_stream().cuda_stream
KMAX = 32


def timeit(descs, k, reps=10):
    win = (_lib.BatchDesc * k).from_buffer(descs, 0)
    for _ in range(2):
        assert l.redgpu_match_batches_dev(exe._h, 4, 0, win, k, st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        l.redgpu_match_batches_dev(exe._h, 4, 0, win, k, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def layout(kind):
    keep = []
    descs = (_lib.BatchDesc * KMAX)()
    if kind == "separate":      # as bench.py: 6 inputs rotating, one output set per batch
        ins = [torch.empty(n * L, dtype=torch.uint8, device="cuda").random_(0, 256) for _ in range(6)]
        for k in range(KMAX):
            r = torch.empty(n, dtype=torch.int32, device="cuda")
            s = torch.empty(n, dtype=torch.int64, device="cuda")
            e = torch.empty(n, dtype=torch.int64, device="cuda")
            keep += [r, s, e]
            descs[k] = _lib.BatchDesc(ins[k % 6].data_ptr(), None, L, n, r.data_ptr(), s.data_ptr(), e.data_ptr())
        keep += ins
    elif kind == "contiguous":  # one big input, one big result / start / end: batch k = slice k
        big = torch.empty(KMAX * n * L, dtype=torch.uint8, device="cuda").random_(0, 256)
        r = torch.empty(KMAX * n, dtype=torch.int32, device="cuda")
        s = torch.empty(KMAX * n, dtype=torch.int64, device="cuda")
        e = torch.empty(KMAX * n, dtype=torch.int64, device="cuda")
        keep += [big, r, s, e]
        for k in range(KMAX):
            descs[k] = _lib.BatchDesc(big.data_ptr() + k * n * L, None, L, n, r.data_ptr() + 4 * k * n,
                                      s.data_ptr() + 8 * k * n, e.data_ptr() + 8 * k * n)
    elif kind == "same":        # every batch the same input and the same outputs (timing only)
        a = torch.empty(n * L, dtype=torch.uint8, device="cuda").random_(0, 256)
        r = torch.empty(n, dtype=torch.int32, device="cuda")
        s = torch.empty(n, dtype=torch.int64, device="cuda")
        e = torch.empty(n, dtype=torch.int64, device="cuda")
        keep += [a, r, s, e]
        for k in range(KMAX):
            descs[k] = _lib.BatchDesc(a.data_ptr(), None, L, n, r.data_ptr(), s.data_ptr(), e.data_ptr())
    return descs, keep


for kind in ("separate", "contiguous", "same"):
    descs, keep = layout(kind)
    row = []
    for k in (2, 4, 8, 16, 20, 32):
        t = timeit(descs, k)
        row.append("K=%d %.1f us (%.2f/batch)" % (k, t, t / k))
    print("%-10s %s  kernel %s" % (kind, "  ".join(row), one_amd.last_kernel()), flush=True)
    del descs, keep
    torch.cuda.empty_cache()
# the single-batch entry point on one 16x batch, for reference
big = torch.empty(16 * n * L, dtype=torch.uint8, device="cuda").random_(0, 256)
r = torch.empty(16 * n, dtype=torch.int32, device="cuda")
s = torch.empty(16 * n, dtype=torch.int64, device="cuda")
e = torch.empty(16 * n, dtype=torch.int64, device="cuda")
args = (exe._h, 4, 0, big.data_ptr(), None, L, 16 * n, r.data_ptr(), s.data_ptr(), e.data_ptr(), st)
for _ in range(2):
    l.redgpu_match_batch_dev(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    l.redgpu_match_batch_dev(*args)
e1.record()
torch.cuda.synchronize()
print("single 16x batch: %.1f us (%.2f per 2^20 lines)  kernel %s" % (e0.elapsed_time(e1) / 5 * 1e3, e0.elapsed_time(e1) / 5 * 1e3 / 16, one_amd.last_kernel()))
