"""Host time per ragged call (tiny GPU work): wrapper + scratch allocation + launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, one_amd
from one_amd import _lib
from golden_util import load_dfa
n = 20000
lens = np.full(n, 8, dtype=np.int64)
off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
d = torch.randint(0, 256, (int(off[-1]),), dtype=torch.uint8, device="cuda")
o = torch.from_numpy(off).cuda()
res = torch.empty(n, dtype=torch.int32, device="cuda"); st = torch.empty(n, dtype=torch.int64, device="cuda"); en = torch.empty(n, dtype=torch.int64, device="cuda")
for kw in ({}, {"no_bucketing": True}, {"force_generic": True}):
    exe = one_amd.Executable(load_dfa("uri"), **kw)
    f = _lib.lib().redgpu_match_batch_dev
    args = (exe._h, 4, 0, d.data_ptr(), o.data_ptr(), 0, n, res.data_ptr(), st.data_ptr(), en.data_ptr(), None)
    for _ in range(20): f(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): f(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(kw, "host issue %.1f us/call, incl. drain %.1f us/call" % ((t1 - t0) / 500 * 1e6, (t2 - t0) / 500 * 1e6), one_amd.last_kernel())
