"""Few long lines (configs[4] shape: 65,536 x 64 KiB): throughput is bounded by how many lines there
are to spread over the CUs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, one_amd
from golden_util import load_dfa
from oracle.reda_writer import random_dfa
def t(label, exe, d, L, n, it=3):
    res = torch.empty(n, dtype=torch.int32, device="cuda"); st = torch.empty(n, dtype=torch.int64, device="cuda"); en = torch.empty(n, dtype=torch.int64, device="cuda")
    f = lambda: one_amd.match_batch(exe, d, 4, False, stride=L, n=n, out=(res, st, en))
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / it
    print("%-46s %10.1f us %8.1f GB/s  %s" % (label, ms * 1e3, n * L / ms / 1e6, one_amd.last_kernel()), flush=True)
n, L = 1 << 16, 1 << 16
d = torch.randint(0, 256, (n * L,), dtype=torch.uint8, device="cuda")
if "syn4k" in sys.argv:
    t("configs[4] full: SYN-4K 65536 x 64 KiB", one_amd.Executable(random_dfa(4097, 256, 5, accept_frac=0.1)), d, L, n, it=2)
t("configs[4] alt: URI-V6 65536 x 64 KiB", one_amd.Executable(load_dfa("uri_v6")), d, L, n)
t("  same, no chunking", one_amd.Executable(load_dfa("uri_v6"), no_chunking=True), d, L, n)
t("URI-V6 4096 x 64 KiB", one_amd.Executable(load_dfa("uri_v6")), d, L, 4096)
t("  same, no chunking", one_amd.Executable(load_dfa("uri_v6"), no_chunking=True), d, L, 4096)
t("URI-D 16384 x 64 KiB", one_amd.Executable(load_dfa("uri")), d, L, 16384)
t("  same, no chunking", one_amd.Executable(load_dfa("uri"), no_chunking=True), d, L, 16384)
t("SYN-256 65536 x 64 KiB", one_amd.Executable(load_dfa("syn256")), d, L, n)
t("SYN-256 16384 x 64 KiB", one_amd.Executable(load_dfa("syn256")), d, L, n // 4)
t("SYN-256 same bytes as 2^20 x 4 KiB", one_amd.Executable(load_dfa("syn256")), d, 4096, 1 << 20)
