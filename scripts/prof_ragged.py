"""rocprofv3 target: ragged match on uniform 32..256 B lines and on geometric lines (URI-D)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, one_amd
from one_amd import workloads as W
from golden_util import load_dfa
exe = one_amd.Executable(load_dfa(sys.argv[1] if len(sys.argv) > 1 else "uri"),
                         no_bucketing=("nob" in sys.argv[2:]), force_generic=("generic" in sys.argv[2:]))
print("#", " ".join(sys.argv[1:]) or "uri")
n = 1 << 20
data, offsets = W.ragged_lines(n, 32, 256, 4)
d = torch.from_numpy(data).cuda(); o = torch.from_numpy(offsets.astype(np.int64)).cuda()
rng = np.random.default_rng(1)
lens = rng.geometric(1 / 144, n).astype(np.int64)
off2 = np.zeros(n + 1, dtype=np.int64); off2[1:] = np.cumsum(lens)
d2 = torch.from_numpy(W.alphabet_bytes(int(off2[-1]), 9)).cuda(); o2 = torch.from_numpy(off2).cuda()
for dd, oo, label in ((d, o, "uniform"), (d2, o2, "geometric")):
    for _ in range(3): one_amd.match_batch(exe, dd, 4, 0, offsets=oo)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): one_amd.match_batch(exe, dd, 4, 0, offsets=oo)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(label, "%.1f us  %.1f GB/s" % (ms * 1e3, dd.numel() / ms / 1e6), one_amd.last_kernel(), flush=True)
# big real-regex DFA (hot rows) on the same geometric text lines, untuned and tuned
if len(sys.argv) > 1 and sys.argv[1] in ("uri_v6", "uri_user"):
    t = W.alphabet_bytes(int(off2[-1]), 19).copy()
    for k in range(0, t.size - 100, 1200):
        t[k:k + len(W.URI_PLANT)] = np.frombuffer(W.URI_PLANT, dtype=np.uint8)
    d3 = torch.from_numpy(t).cuda()
    for label in ("untuned", "tuned"):
        if label == "tuned":
            exe.tune(d3[: int(off2[1 << 14])], offsets=o2[: (1 << 14) + 1].contiguous())
        for _ in range(3): one_amd.match_batch(exe, d3, 4, 0, offsets=o2)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): one_amd.match_batch(exe, d3, 4, 0, offsets=o2)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print("geometric text + https URLs,", label, "%.1f us  %.1f GB/s" % (ms * 1e3, d3.numel() / ms / 1e6), one_amd.last_kernel(), flush=True)
