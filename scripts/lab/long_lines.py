#!/usr/bin/env python3
"""lab: configs[2]'s shape (2^21 x 4 KiB = 8 GiB) and friends through match<styLast,false>, full
Outcome; REDGPU_LEAN=0 gives the exact step for comparison.  python scripts/lab/long_lines.py [dfa]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import one_amd
from one_amd import _lib, workloads as W
from golden_util import load_dfa

tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("REDGPU_")) or "default"
for name in (sys.argv[1:] or ["syn256", "uri"]):
    exe = one_amd.Executable(load_dfa(name))
    l = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    text = name != "syn256"
    for n, L in ((1 << 21, 4096), (1 << 23, 512), (1 << 19, 16384)):
        g = torch.Generator(device="cuda").manual_seed(5)
        buf = torch.empty(n * L, dtype=torch.uint8, device="cuda")
        alpha = torch.from_numpy(W.ALPHABET47.copy()).cuda()
        for lo in range(0, n * L, 1 << 28):
            hi = min(n * L, lo + (1 << 28))
            v = torch.randint(0, 256, (hi - lo,), generator=g, device="cuda", dtype=torch.uint8)
            buf[lo:hi] = alpha[(v % 47).long()] if text else v
            del v
        if text:
            plant = torch.from_numpy(np.frombuffer(W.URI_PLANT, dtype=np.uint8).copy()).cuda()
            buf.view(n, L)[::8, 100:100 + plant.numel()] = plant
        r = torch.empty(n, dtype=torch.int32, device="cuda")
        s = torch.empty(n, dtype=torch.int64, device="cuda")
        e = torch.empty(n, dtype=torch.int64, device="cuda")
        a = (exe._h, 4, 0, buf.data_ptr(), None, L, n, r.data_ptr(), s.data_ptr(), e.data_ptr(), st)
        for _ in range(2):
            assert l.redgpu_match_batch_dev(*a) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            l.redgpu_match_batch_dev(*a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-28s %-7s %8d x %5d B: %.3f ms  %.0f GB/s  kernel %s  matches %d" % (
            tag, name, n, L, ms, n * L / ms / 1e6, one_amd.last_kernel(), int((r > 0).sum())), flush=True)
        del buf, r, s, e
        torch.cuda.empty_cache()
