"""configs[3]'s DFA (LOG-100) on its ragged lines: how the time splits between lines that die at
once and lines that walk a signature.  Developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, one_amd
from one_amd import workloads as W
from golden_util import load_dfa
n = int(os.environ.get("LINES", 1 << 20))
def run(label, exe, d, o, lead=1):
    for _ in range(3): one_amd.match_batch(exe, d, 4, lead, offsets=o)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): one_amd.match_batch(exe, d, 4, lead, offsets=o)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%-58s %8.1f us %8.1f GB/s  %s" % (label, ms * 1e3, d.numel() / ms / 1e6, one_amd.last_kernel()), flush=True)
for every in (0, 8, 2, 1):
    data, offsets = W.ragged_lines(n, 32, 256, 4, heads=W.log100_heads() if every else None, head_every=every or 2)
    d = torch.from_numpy(data).cuda(); o = torch.from_numpy(offsets.astype(np.int64)).cuda()
    for kw, tag in (({}, "sparse rows"), ({"force_hot": True}, "hot rows"), ({"force_global": True}, "L2 only")):
        exe = one_amd.Executable(load_dfa("log100"), **kw)
        run("signature at the head of every %s line, %s" % (every or "no", tag), exe, d, o)
