"""GB/s of match<styLast,false> (full Outcome) for several (lines, line length) shapes, one
stream, inputs resident, rotating buffers.  Developer tool (bench.py is the contract bench)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, one_amd
blob = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "dfas", sys.argv[1] + ".reda"), "rb").read() if len(sys.argv) > 1 else None
blob = blob or open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "dfas", "syn256.reda"), "rb").read()
exe = one_amd.Executable(blob)
shapes = [(1 << 20, 64), (1 << 24, 64), (1 << 22, 128), (1 << 22, 256), (1 << 18, 4096), (1 << 21, 4096), (1 << 14, 65536)]
if os.environ.get("SHAPES"):  # e.g. SHAPES=1048576x64,2097152x4096
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ["SHAPES"].split(",")]
# TEXT=1: bytes from the 47-character alphabet of the text workloads instead of uniform bytes
alpha = torch.tensor(list(b"abcdefghijklmnopqrstuvwxyz0123456789 ./:-_=&?%"), dtype=torch.uint8, device="cuda")
def make(total):
    if os.environ.get("TEXT"):
        return alpha[torch.randint(0, alpha.numel(), (total,), device="cuda")]
    return torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda")
for n, L in shapes:
    total = n * L
    nb = max(2, min(6, (3 << 30) // total))
    bufs = [make(total) for _ in range(nb)]
    res = torch.empty(n, dtype=torch.int32, device="cuda")
    st = torch.empty(n, dtype=torch.int64, device="cuda")
    en = torch.empty(n, dtype=torch.int64, device="cuda")
    it = max(4, min(100, int(2e10 // total)))
    for i in range(3):
        one_amd.match_batch(exe, bufs[i % nb], 4, False, stride=L, n=n, out=(res, st, en))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(it):
        one_amd.match_batch(exe, bufs[i % nb], 4, False, stride=L, n=n, out=(res, st, en))
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / it
    print("%9d lines x %6d B  %8.1f us  %7.1f GB/s  %s" % (n, L, ms * 1e3, total / ms / 1e6, one_amd.last_kernel()), flush=True)
    del bufs
    torch.cuda.empty_cache()
