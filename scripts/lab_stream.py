"""Timing lab for the fixed-stride hot path: one line per (shape, chains) with the single-stream
launch time, the 3-stream step time and an output checksum (compare across REDGPU_STREAM_CHAINS
values: 0 = k_stream, 3 / 4 = k_stream4).  Developer tool; bench.py is the contract bench.

    REDGPU_STREAM_CHAINS=4 python3 scripts/lab_stream.py [syn256|uri] [--check]
    SHAPES=1048576x64,2097152x4096 ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import one_amd
from one_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "syn256"
root = os.path.join(os.path.dirname(__file__), "..")
blob = open(os.path.join(root, "tests", "golden", "dfas", name + ".reda"), "rb").read()
exe = one_amd.Executable(blob)
shapes = [(1 << 20, 64), (1 << 24, 64), (1 << 22, 256), (1 << 18, 4096), (1 << 21, 4096)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ["SHAPES"].split(",")]
chains = os.environ.get("REDGPU_STREAM_CHAINS", "default")
fn = _lib.lib().redgpu_match_batch_dev
gen = torch.Generator(device="cuda")
for n, L in shapes:
    total = n * L
    nb = max(2, min(6, (6 << 30) // total))
    gen.manual_seed(1234 + L)
    bufs = [torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda", generator=gen)
            for _ in range(nb)]
    outs = [(torch.empty(n, dtype=torch.int32, device="cuda"),
             torch.empty(n, dtype=torch.int64, device="cuda"),
             torch.empty(n, dtype=torch.int64, device="cuda")) for _ in range(6)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    cur = torch.cuda.current_stream().cuda_stream

    def call(i, st):
        r, s, e = outs[i % len(outs)]
        rc = fn(exe._h, 4, 0, bufs[i % nb].data_ptr(), None, L, n, r.data_ptr(), s.data_ptr(),
                e.data_ptr(), st)
        assert rc == 0, _lib.lib().redgpu_last_error().decode()

    it = max(6, min(200, int(4e10 // total)))
    for i in range(4):
        call(i, cur)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(it):
        call(i, cur)
    b.record()
    torch.cuda.synchronize()
    one = a.elapsed_time(b) / it
    # checksum of the outputs of buffer 0
    call(0, cur)
    torch.cuda.synchronize()
    r, s, e = outs[0]
    ck = (int(r.to(torch.int64).sum().item()), int(s.sum().item()), int(e.sum().item()))
    if "--check" in sys.argv and total <= (1 << 28):
        import oracle
        er, es, ee = oracle.CpuOracle(blob).batch("match", "last", 0, bufs[0].cpu().numpy(),
                                                  stride=L, n=n, threads=16)
        ok = (np.array_equal(r.cpu().numpy(), er) and
              np.array_equal(s.cpu().numpy().astype(np.uint64), es) and
              np.array_equal(e.cpu().numpy().astype(np.uint64), ee))
    else:
        ok = None
    # 3 streams round-robin
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    a.record()
    for i in range(it):
        call(i, streams[i % 3].cuda_stream)
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    b.record()
    torch.cuda.synchronize()
    three = a.elapsed_time(b) / it
    print("chains=%s %9d x %6d B  1 stream %8.2f us %7.1f GB/s | 3 streams %8.2f us %7.1f GB/s | "
          "%s ck=%x oracle=%s" % (chains, n, L, one * 1e3, total / one / 1e6, three * 1e3,
                                  total / three / 1e6, one_amd.last_kernel(),
                                  hash(ck) & 0xffffffffffff, ok), flush=True)
    del bufs, outs
    torch.cuda.empty_cache()
