"""URI-D over 2^20 ragged text lines (32..256 B): match with every style - k_ragged for the
whole-line styles, k_style_blocks for the early-exit ones.  Developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, one_amd
from one_amd import workloads as W
from golden_util import load_dfa
n = int(os.environ.get("LINES", 1 << 20))
data, offsets = W.ragged_lines(n, 32, 256, 4, heads=[W.URI_PLANT], head_every=8)
d = torch.from_numpy(data).cuda(); o = torch.from_numpy(offsets.astype(np.int64)).cuda()
for name in ("uri", "syn256"):
    exe = one_amd.Executable(load_dfa(name))
    for verb, fn in (("match", one_amd.match_batch), ("check", one_amd.check_batch)):
        for sty, sname in ((1, "Instant"), (2, "First"), (3, "Tangent"), (4, "Last"), (5, "Full")):
            f = lambda: fn(exe, d, sty, 0, offsets=o)
            for _ in range(3): f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): f()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            print("%-7s %-6s %-8s %9.1f us %8.1f GB/s  %s" % (name, verb, sname, ms * 1e3, data.size / ms / 1e6, one_amd.last_kernel()), flush=True)
