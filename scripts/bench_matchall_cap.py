import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, one_amd
from one_amd import _lib
from golden_util import load_dfa
l = _lib.lib()
for name in ("syn256", "uri"):
    exe = one_amd.Executable(load_dfa(name))
    n, L = 1 << 20, 64
    data = torch.randint(0, 256, (n * L,), dtype=torch.uint8, device="cuda")
    for cap in (0, 1, 4, 8):
        cnt = torch.empty(n, dtype=torch.int64, device="cuda")
        r = torch.empty(max(1, n * cap), dtype=torch.int32, device="cuda")
        s = torch.empty(max(1, n * cap), dtype=torch.int64, device="cuda")
        e = torch.empty(max(1, n * cap), dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        f = lambda: l.redgpu_match_all_batch_dev(exe._h, 1, data.data_ptr(), None, L, n, cap, cnt.data_ptr(), r.data_ptr(), s.data_ptr(), e.data_ptr(), st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print("%s matchAll cap=%d: %.1f us  %.1f GB/s  mean records/line %.2f" % (name, cap, ms * 1e3, n * L / ms / 1e6, cnt.float().mean().item()))
