"""GB/s of every (verb, style) on fixed-stride lines - which kernel serves it.  Developer tool.
usage: bench_styles.py [dfa-name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, one_amd
from golden_util import load_dfa
from one_amd import workloads as W
name = sys.argv[1] if len(sys.argv) > 1 else "syn256"
flags = {"force_generic": True} if os.environ.get("FORCE_GENERIC") == "1" else {}
exe = one_amd.Executable(load_dfa(name), **flags)
only = os.environ.get("VERBS", "check,match,scan,search").split(",")
text = name not in ("syn256", "syn4k")


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


for n, L in ((1 << 20, 64), (1 << 18, 4096)):
    if text:
        host = W.fixed_lines(n, L, 7, plant=W.URI_PLANT)
        data = torch.from_numpy(host).cuda()
    else:
        data = torch.randint(0, 256, (n * L,), dtype=torch.uint8, device="cuda")
    for verb, fn in (("check", one_amd.check_batch), ("match", one_amd.match_batch),
                     ("scan", one_amd.scan_batch), ("search", one_amd.search_batch)):
        if verb not in only:
            continue
        for sty, sname in ((1, "Instant"), (2, "First"), (3, "Tangent"), (4, "Last"), (5, "Full")):
            if verb in ("scan", "search") and sty == 5 and L > 1024 and not exe.info["suffix_closed"]:
                # sliding styFull over a DFA that is not suffix-closed is O(n^2) per line by the
                # reference's own definition: 9.2 s per call on 2^18 x 4 KiB (measured once)
                print("%-6s %-8s %8d x %5d B  skipped: quadratic by definition (9.2 s per call)" % (verb, sname, n, L))
                continue
            ms = timed(lambda: fn(exe, data, sty, 0, stride=L, n=n), 10)
            print("%-6s %-8s %8d x %5d B  %9.1f us  %8.1f GB/s  %s" %
                  (verb, sname, n, L, ms * 1e3, n * L / ms / 1e6, one_amd.last_kernel()), flush=True)
