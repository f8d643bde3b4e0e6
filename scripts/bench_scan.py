"""Where does scan's time go?  ERR on ragged text lines: scan vs check vs match, leader on/off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, one_amd
from one_amd import workloads as W
from golden_util import load_dfa
n = 1 << 20
data, offsets = W.ragged_lines(n, 32, 256, 4, heads=W.log100_heads(), head_every=2)
d = torch.from_numpy(data).cuda(); o = torch.from_numpy(offsets.astype(np.int64)).cuda()
def t(label, f, it=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / it
    print("%-44s %9.1f us %8.1f GB/s  %s" % (label, ms * 1e3, data.size / ms / 1e6, one_amd.last_kernel()), flush=True)
for name in ("err", "num3", "aab"):
    exe = one_amd.Executable(load_dfa(name))
    print(name, {k: exe.info[k] for k in ("leader_len", "states_used", "n_pure_dead", "table_kind")})
    t(name + " scan<Instant,true>", lambda: one_amd.scan_batch(exe, d, 1, 1, offsets=o))
    t(name + " scan<Instant,false>", lambda: one_amd.scan_batch(exe, d, 1, 0, offsets=o))
    t(name + " scan<Last,false>", lambda: one_amd.scan_batch(exe, d, 4, 0, offsets=o))
    t(name + " search<Last,false>", lambda: one_amd.search_batch(exe, d, 4, 0, offsets=o))
    t(name + " check<Instant,true>", lambda: one_amd.check_batch(exe, d, 1, 1, offsets=o))
    t(name + " match<Last,false>", lambda: one_amd.match_batch(exe, d, 4, 0, offsets=o))
