import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, one_amd
from golden_util import load_dfa
n = 1 << 20
def t(label, f, nbytes, it=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / it
    print("%-50s %9.1f us %8.1f GB/s" % (label, ms * 1e3, nbytes / ms / 1e6), flush=True)
exe = one_amd.Executable(load_dfa("err"))
for L in (64, 144, 256):
    d = torch.full((n * L,), ord("z"), dtype=torch.uint8, device="cuda")
    t("all-z fixed %d: scan<Instant,true>" % L, lambda: one_amd.scan_batch(exe, d, 1, 1, stride=L, n=n), n * L)
    t("all-z fixed %d: scan<Instant,false>" % L, lambda: one_amd.scan_batch(exe, d, 1, 0, stride=L, n=n), n * L)
    t("all-z fixed %d: match<Full,false> generic" % L, lambda: one_amd.match_batch(one_amd.Executable(load_dfa("uri"), force_generic=True), d, 5, 0, stride=L, n=n), n * L)
