#!/usr/bin/env python3
"""lab: one K-batch launch under bench.py's conditions (idle, short warm-up launch, ONE timed
launch, wall clock and events), repeated; then the same launch sustained back to back.
Variants come from the environment (REDGPU_MULTI_*), one process each."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import one_amd
from one_amd import _lib
from golden_util import load_dfa

tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("REDGPU_")) or "default"
exe = one_amd.Executable(load_dfa(sys.argv[1] if len(sys.argv) > 1 else "syn256"))
l = _lib.lib()
n, L, K = 1 << 20, 64, 20
st = torch.cuda.current_stream().cuda_stream
ins = [torch.empty(n * L, dtype=torch.uint8, device="cuda").random_(0, 256) for _ in range(6)]
descs = (_lib.BatchDesc * 32)()
keep = []
for k in range(32):
    r = torch.empty(n, dtype=torch.int32, device="cuda")
    s = torch.empty(n, dtype=torch.int64, device="cuda")
    e = torch.empty(n, dtype=torch.int64, device="cuda")
    keep += [r, s, e]
    descs[k] = _lib.BatchDesc(ins[k % 6].data_ptr(), None, L, n, r.data_ptr(), s.data_ptr(), e.data_ptr())
w5 = (_lib.BatchDesc * 5).from_buffer(descs, 0)
w20 = (_lib.BatchDesc * K).from_buffer(descs, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
walls, evs, calls = [], [], []
for rep in range(6):
    time.sleep(0.25)
    l.redgpu_match_batches_dev(exe._h, 4, 0, w5, 5, st)
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    l.redgpu_match_batches_dev(exe._h, 4, 0, w20, K, st)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    walls.append((t2 - t0) * 1e6); evs.append(e0.elapsed_time(e1) * 1e3); calls.append((t1 - t0) * 1e6)
print("%-40s bench-like: wall %s | events %s | call returns after %s us" % (
    tag, " ".join("%.0f" % w for w in walls), " ".join("%.0f" % w for w in evs), " ".join("%.0f" % w for w in calls)))
ts = []
for rep in range(40):
    e0.record()
    l.redgpu_match_batches_dev(exe._h, 4, 0, w20, K, st)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("%-40s sustained: first 3 %s | worst %.0f | last 10 mean %.1f us (%.2f/batch, %.0f GB/s)" % (
    tag, " ".join("%.0f" % t for t in ts[:3]), max(ts), sum(ts[-10:]) / 10, sum(ts[-10:]) / 10 / K,
    K * n * L / (sum(ts[-10:]) / 10 * 1e-6) / 1e9))
# the single-batch entry point, 20 launches back to back (REDGPU_MULTI_SINGLE=1 routes them
# through the multi-batch kernel)
a = [(exe._h, 4, 0, ins[k % 6].data_ptr(), None, L, n, keep[3 * k].data_ptr(), keep[3 * k + 1].data_ptr(),
      keep[3 * k + 2].data_ptr(), st) for k in range(20)]
ts = []
for rep in range(8):
    e0.record()
    for k in range(20):
        l.redgpu_match_batch_dev(*a[k])
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 20)
print("%-40s single-batch launches: %s us each  kernel %s" % (tag, " ".join("%.2f" % t for t in ts), one_amd.last_kernel()))
