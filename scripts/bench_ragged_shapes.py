"""Ragged match<styLast,false> on synthetic length distributions: isolates the step cost (all
lines equal and aligned), the unaligned-read cost (equal, odd length) and the idle-lane cost
(uniform / geometric lengths).  Developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, one_amd
from golden_util import load_dfa
exe = one_amd.Executable(load_dfa(sys.argv[1] if len(sys.argv) > 1 else "syn256"))
n = int(os.environ.get("LINES", 1 << 20))
rng = np.random.default_rng(3)
cases = [("all 256 B", np.full(n, 256)), ("all 250 B", np.full(n, 250)), ("all 64 B", np.full(n, 64)),
         ("all 60 B", np.full(n, 60)), ("uniform 32-256", rng.integers(32, 257, n)),
         ("geometric mean 144", rng.geometric(1 / 144, n)), ("uniform 1-2048", rng.integers(1, 2049, n)),
         ("sorted uniform 32-256", np.sort(rng.integers(32, 257, n))),
         ("blocks of 128 equal, 32-256", np.repeat(rng.integers(32, 257, n // 128), 128)),
         ("uniform 193-256", rng.integers(193, 257, n)), ("half 64 half 256", np.where(rng.integers(0, 2, n) == 1, 64, 256)),
         ("X-A 64*r(1..4)", 64 * rng.integers(1, 5, n)), ("X-B 64*r(0..3)+37", 64 * rng.integers(0, 4, n) + 37),
         ("X-C 16*r(2..16)", 16 * rng.integers(2, 17, n)), ("X-D 64*r(0..3)+r(1..64)", 64 * rng.integers(0, 4, n) + rng.integers(1, 65, n)),
         ("X-E 64*r(0..3)+4*r(1..16)", 64 * rng.integers(0, 4, n) + 4 * rng.integers(1, 17, n)),
         ("one 1 MB line among 32-256", np.concatenate([rng.integers(32, 257, n - 1), [1 << 20]])),
         ("a 64 KB line per 4096", np.where(np.arange(n) % 4096 == 7, 65536, rng.integers(32, 257, n)))]
only = os.environ.get("CASES")
for label, lens in cases:
    if only and not any(k in label for k in only.split(",")): continue
    off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
    total = int(off[-1])
    d = torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda")
    o = torch.from_numpy(off).cuda()
    for _ in range(3): one_amd.match_batch(exe, d, 4, 0, offsets=o)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): one_amd.match_batch(exe, d, 4, 0, offsets=o)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%-22s %9.1f us  %7.1f GB/s  %s" % (label, ms * 1e3, total / ms / 1e6, one_amd.last_kernel()), flush=True)
    del d, o
