"""Throughput of the not-yet-tuned paths: ragged lines (k_generic), big DFAs (table in HBM/L2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, one_amd
from one_amd import workloads as W
from golden_util import load_dfa
from oracle.reda_writer import random_dfa

def timeit(fn, total_bytes, label, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / it
    print("%-62s %9.1f us %8.1f GB/s  %s" % (label, ms * 1e3, total_bytes / ms / 1e6, one_amd.last_kernel()), flush=True)

# config 4: LOG-100, ragged 32..256 B
n = 1 << 20
data, offsets = W.ragged_lines(n, 32, 256, 4, heads=W.log100_heads(), head_every=2)
d = torch.from_numpy(data).cuda(); o = torch.from_numpy(offsets.astype(np.int64)).cuda()
exe = one_amd.Executable(load_dfa("log100"))
timeit(lambda: one_amd.match_batch(exe, d, 4, 1, offsets=o), data.size, "config4: LOG-100 (3150 st) 2^20 ragged lines match<Last,true>")
exe_u = one_amd.Executable(load_dfa("uri"))
timeit(lambda: one_amd.match_batch(exe_u, d, 4, 0, offsets=o), data.size, "URI-D (LDS table) same ragged lines match<Last,false>")
timeit(lambda: one_amd.scan_batch(exe_u, d, 1, 1, offsets=o), data.size, "URI-D scan<Instant,true> same ragged lines")
exe_e = one_amd.Executable(load_dfa("err"))
timeit(lambda: one_amd.scan_batch(exe_e, d, 1, 1, offsets=o), data.size, "ERR scan<Instant,true> same ragged lines (anchored, leader)")
# config 5: 4K-state DFA, 64 KiB inputs
blob = random_dfa(4097, 256, 5, accept_frac=0.1)
exe5 = one_amd.Executable(blob)
n5, L5 = 4096, 65536
d5 = torch.randint(0, 256, (n5 * L5,), dtype=torch.uint8, device="cuda")
timeit(lambda: one_amd.match_batch(exe5, d5, 4, 0, stride=L5, n=n5), n5 * L5, "config5: SYN-4K (2 MiB table in L2) 4096 x 64 KiB", it=3)
n6, L6 = 1 << 18, 1024
d6 = torch.randint(0, 256, (n6 * L6,), dtype=torch.uint8, device="cuda")
timeit(lambda: one_amd.match_batch(exe5, d6, 4, 0, stride=L6, n=n6), n6 * L6, "SYN-4K 2^18 x 1 KiB", it=3)
# real-regex big DFA (class table larger than LDS): random bytes and text with planted URLs
exe7 = one_amd.Executable(load_dfa("uri_v6"))
print("uri_v6:", exe7.info)
timeit(lambda: one_amd.match_batch(exe7, d6, 4, 0, stride=L6, n=n6), n6 * L6, "URI-V6 (3254 st / 35 cls) 2^18 x 1 KiB random bytes", it=3)
n8, L8 = 1 << 20, 256
t8 = W.fixed_lines(n8, L8, 8, alphabet=True, plant=W.URI_V6_PLANT, plant_every=8, plant_at=32)
d8 = torch.from_numpy(t8).cuda()
timeit(lambda: one_amd.match_batch(exe7, d8, 4, 0, stride=L8, n=n8), n8 * L8, "URI-V6 2^20 x 256 B text, IPv6 URL planted every 8th line", it=3)
t9 = W.fixed_lines(n8, L8, 9, alphabet=True, plant=W.URI_PLANT, plant_every=8, plant_at=32)
d9 = torch.from_numpy(t9).cuda()
timeit(lambda: one_amd.match_batch(exe7, d9, 4, 0, stride=L8, n=n8), n8 * L8, "URI-V6 2^20 x 256 B text, https://name URL every 8th line", it=3)
t10 = W.fixed_lines(n8, L8, 10, alphabet=True)
d10 = torch.from_numpy(t10).cuda()
timeit(lambda: one_amd.match_batch(exe7, d10, 4, 0, stride=L8, n=n8), n8 * L8, "URI-V6 2^20 x 256 B text, no URL", it=3)
t11 = W.fixed_lines(1 << 14, L8, 11, alphabet=True, plant=W.URI_PLANT, plant_every=8, plant_at=90)
print("tune on a 4 MiB sample of other text with https URLs:", exe7.tune(t11, stride=L8, n=1 << 14))
timeit(lambda: one_amd.match_batch(exe7, d9, 4, 0, stride=L8, n=n8), n8 * L8, "URI-V6 tuned: text, https://name URL every 8th line", it=3)
timeit(lambda: one_amd.match_batch(exe7, d8, 4, 0, stride=L8, n=n8), n8 * L8, "URI-V6 tuned: text, IPv6 URL every 8th line (not in the sample)", it=3)
timeit(lambda: one_amd.match_batch(exe7, d6, 4, 0, stride=L6, n=n6), n6 * L6, "URI-V6 tuned: 2^18 x 1 KiB random bytes", it=3)
exe7g = one_amd.Executable(load_dfa("uri_v6"), force_generic=True)
timeit(lambda: one_amd.match_batch(exe7g, d9, 4, 0, stride=L8, n=n8), n8 * L8, "  same (https URL), generic kernel + hot rows", it=3)
exe7gg = one_amd.Executable(load_dfa("uri_v6"), force_generic=True, force_global=True)
timeit(lambda: one_amd.match_batch(exe7gg, d9, 4, 0, stride=L8, n=n8), n8 * L8, "  same (https URL), generic kernel, table in L2 only", it=3)
exe9 = one_amd.Executable(load_dfa("log100"))
timeit(lambda: one_amd.match_batch(exe9, d8, 4, 0, stride=L8, n=n8), n8 * L8, "LOG-100 match<Last,false> same text (dies early)", it=3)

# SURVEY's 343-state "userinfo" URI DFA: class table in LDS + k_generic by default; hot rows +
# streaming kernel when forced / after tuning on representative input
tt = W.fixed_lines(1 << 20, 64, 2, alphabet=True, plant=W.URI_USER_PLANT, plant_every=8, plant_at=8)
dd = torch.from_numpy(tt).cuda()
for kw, label in (({}, "default"), ({"force_hot": True}, "hot rows forced (untuned)")):
    exe_u2 = one_amd.Executable(load_dfa("uri_user"), **kw)
    timeit(lambda: one_amd.match_batch(exe_u2, dd, 4, 0, stride=64, n=1 << 20), tt.size,
           "URI-USER (343 st) 2^20 x 64 B text+URLs, %s [kind %d]" % (label, exe_u2.info["table_kind"]), it=5)
exe_u3 = one_amd.Executable(load_dfa("uri_user"))
exe_u3.tune(dd[: 64 << 14], stride=64, n=1 << 14)
timeit(lambda: one_amd.match_batch(exe_u3, dd, 4, 0, stride=64, n=1 << 20), tt.size,
       "URI-USER (343 st) same, after tune [kind %d]" % exe_u3.info["table_kind"], it=5)
# class table above 64 KB (1500 states x 40 classes = 120 KB): index-form streaming kernel vs generic
blob_big = random_dfa(1500, 40, 5, accept_frac=0.1)
for kw, label in (({}, "default"), ({"force_generic": True}, "k_generic")):
    exe_b = one_amd.Executable(blob_big, **kw)
    timeit(lambda: one_amd.match_batch(exe_b, d6, 4, 0, stride=L6, n=n6), n6 * L6,
           "RND-1500x40 (120 KB class table) 2^18 x 1 KiB, %s" % label, it=3)
