"""Anchored DFA (ERR: 'error') on fixed-stride text: streaming kernel vs early-exit generic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, one_amd
from one_amd import workloads as W
from golden_util import load_dfa
for name in ("err", "num3"):
    for L, n in ((64, 1 << 20), (256, 1 << 20), (4096, 1 << 16)):
        t = W.fixed_lines(n, L, 8, alphabet=True, plant=b"error: disk", plant_every=4, plant_at=0)
        d = torch.from_numpy(t).cuda()
        for kw in ({}, {"force_generic": True}):
            exe = one_amd.Executable(load_dfa(name), **kw)
            res = torch.empty(n, dtype=torch.int32, device="cuda"); st = torch.empty(n, dtype=torch.int64, device="cuda"); en = torch.empty(n, dtype=torch.int64, device="cuda")
            f = lambda: one_amd.match_batch(exe, d, 4, False, stride=L, n=n, out=(res, st, en))
            for _ in range(3): f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): f()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 20
            print("%-5s %5d B x %8d  %-22s %8.1f us %8.1f GB/s  %s" % (name, L, n, kw, ms * 1e3, n * L / ms / 1e6, one_amd.last_kernel()), flush=True)
