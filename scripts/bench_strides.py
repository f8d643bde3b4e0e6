"""match<styLast,false> (full Outcome) on fixed strides that are not multiples of 64 - which
kernel serves them.  Developer tool.  usage: bench_strides.py [dfa-name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, one_amd
from golden_util import load_dfa
name = sys.argv[1] if len(sys.argv) > 1 else "syn256"
exe = one_amd.Executable(load_dfa(name))
for L in (16, 48, 64, 80, 96, 100, 120, 128, 200, 250, 1000, 1024):
    n = (1 << 26) // L
    data = torch.randint(0, 256, (n * L,), dtype=torch.uint8, device="cuda")
    f = lambda: one_amd.match_batch(exe, data, 4, 0, stride=L, n=n)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("stride %5d x %8d lines  %8.1f us  %8.1f GB/s  %s" % (L, n, ms * 1e3, n * L / ms / 1e6, one_amd.last_kernel()), flush=True)
