#!/usr/bin/env python3
"""lab: host time of issuing one redgpu_match_batches_dev call (20 batches) and its pieces."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import one_amd
from one_amd import _lib
from golden_util import load_dfa
exe = one_amd.Executable(load_dfa("syn256"))
l = _lib.lib()
n, L, K = 1 << 20, 64, 20
st = torch.cuda.current_stream().cuda_stream
ins = [torch.empty(n * L, dtype=torch.uint8, device="cuda").random_(0, 256) for _ in range(6)]
descs = (_lib.BatchDesc * K)()
keep = []
for k in range(K):
    r = torch.zeros(n, dtype=torch.int32, device="cuda"); s = torch.zeros(n, dtype=torch.int64, device="cuda"); e = torch.zeros(n, dtype=torch.int64, device="cuda")
    keep += [r, s, e]
    descs[k] = _lib.BatchDesc(ins[k % 6].data_ptr(), None, L, n, r.data_ptr(), s.data_ptr(), e.data_ptr())
def t(fn, reps=200, sync_each=True):
    xs = []
    for _ in range(reps):
        if sync_each: torch.cuda.synchronize()
        a = time.perf_counter(); fn(); b = time.perf_counter()
        xs.append((b - a) * 1e6)
    xs.sort()
    return "min %.1f  median %.1f  p90 %.1f us" % (xs[0], xs[len(xs) // 2], xs[int(len(xs) * 0.9)])
print("ctypes call of redgpu_version          :", t(lambda: l.redgpu_version(), sync_each=False))
print("match_batches_dev, 0 batches           :", t(lambda: l.redgpu_match_batches_dev(exe._h, 4, 0, descs, 0, st), sync_each=False))
print("match_batches_dev, 20 batches (launch) :", t(lambda: l.redgpu_match_batches_dev(exe._h, 4, 0, descs, K, st), reps=50))
print("match_batch_dev, 1 batch (launch)      :", t(lambda: l.redgpu_match_batch_dev(exe._h, 4, 0, ins[0].data_ptr(), None, L, n, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), st), reps=50))
ev = torch.cuda.Event(enable_timing=True)
print("torch event record                     :", t(lambda: ev.record(), reps=50))
print("torch.cuda.synchronize (idle)          :", t(lambda: torch.cuda.synchronize(), sync_each=False))
x = torch.zeros(1, device="cuda")
print("torch tiny kernel launch (x.add_(1))   :", t(lambda: x.add_(1), reps=50))
# the first calls of a process, one by one (bench.py's timed call is the second multi-batch call
# of its process), then calls that follow an idle pause
def one(cnt):
    torch.cuda.synchronize()
    a = time.perf_counter(); l.redgpu_match_batches_dev(exe._h, 4, 0, descs, cnt, st); b = time.perf_counter()
    torch.cuda.synchronize()
    return (b - a) * 1e6
print("fresh shapes, call by call (us): 5 batches %.1f, then 20 batches %s" % (one(5), ["%.1f" % one(20) for _ in range(6)]))
for pause in (0.0, 0.005, 0.05, 0.5):
    xs = []
    for _ in range(5):
        time.sleep(pause)
        xs.append(one(20))
    print("after %.3f s idle: %s" % (pause, ["%.1f" % v for v in xs]))
ev0 = torch.cuda.Event(enable_timing=True)
xs = []
for _ in range(5):
    torch.cuda.synchronize(); ev0.record()
    a = time.perf_counter(); l.redgpu_match_batches_dev(exe._h, 4, 0, descs, K, st); b = time.perf_counter()
    xs.append((b - a) * 1e6)
print("right behind an event record: %s" % ["%.1f" % v for v in xs])
