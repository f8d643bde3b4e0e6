#!/bin/bash
set -u
mkdir -p gpurun_out
: > gpurun_out/r2_modes.log
for extra in "" "--no-start"; do
for cfg in 1 2; do
timeout -k 10 500 python3 bench.py --config $cfg $extra --no-cpu-baseline --no-calibration > gpurun_out/r2_tmp.log 2>&1 || { tail -20 gpurun_out/r2_tmp.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_tmp.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $cfg $extra:', j['value'], 'GB/s', j['roofline']['kernel_ms'], 'ms', j['kernel'], j['bit_exact'])" | tee -a gpurun_out/r2_modes.log
done
done
python3 - <<'PY' | tee -a gpurun_out/r2_modes.log
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, one_amd
from golden_util import load_dfa
exe = one_amd.Executable(load_dfa("syn256"))
n, L = 1 << 21, 4096
bufs = [torch.randint(0, 256, (n * L,), dtype=torch.uint8, device="cuda") for _ in range(2)]
res = torch.empty(n, dtype=torch.int32, device="cuda")
for name, fn in (("check<styFull>", lambda b: one_amd.check_batch(exe, b, 5, 0, stride=L, n=n, out=res)),
                 ("check<styLast>", lambda b: one_amd.check_batch(exe, b, 4, 0, stride=L, n=n, out=res))):
    for i in range(3): fn(bufs[i % 2])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(10): fn(bufs[i % 2])
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("2^21 x 4 KiB %s: %.3f ms %.1f GB/s %s" % (name, ms, n * L / ms / 1e6, one_amd.last_kernel()))
PY
