#!/usr/bin/env python3
"""lab: does the time of the same launch drift with how long the GPU has been busy?"""
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
    r = torch.empty(n, dtype=torch.int32, device="cuda")
    s = torch.empty(n, dtype=torch.int64, device="cuda")
    e = torch.empty(n, dtype=torch.int64, device="cuda")
    keep += [r, s, e]
    descs[k] = _lib.BatchDesc(ins[k % 6].data_ptr(), None, L, n, r.data_ptr(), s.data_ptr(), e.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t_start = time.perf_counter()
for rep in range(40):
    e0.record()
    l.redgpu_match_batches_dev(exe._h, 4, 0, descs, K, st)
    e1.record()
    torch.cuda.synchronize()
    print("launch %2d at %.1f ms: %.1f us (%.2f/batch)" % (rep, (time.perf_counter() - t_start) * 1e3,
                                                          e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / K))
    if rep == 19:
        print("-- 1 s idle --"); time.sleep(1.0)
# single-batch launches, same way
a = (exe._h, 4, 0, ins[0].data_ptr(), None, L, n, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), st)
for rep in range(5):
    e0.record()
    for i in range(20):
        l.redgpu_match_batch_dev(*a)
    e1.record()
    torch.cuda.synchronize()
    print("20 single launches: %.1f us (%.2f/batch)" % (e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / 20))
