#!/bin/bash
set -u
mkdir -p gpurun_out
: > gpurun_out/r2_spread.log
for sp in 1 2 4 8; do
REDGPU_GENERIC_SPREAD=$sp timeout -k 10 500 python3 bench.py --config 4 --steps 3 --no-cpu-baseline --no-calibration > gpurun_out/r2_tmp.log 2>&1 || { tail -20 gpurun_out/r2_tmp.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_tmp.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SYN-4K 65536 x 64 KiB, one line per $sp lanes:', j['value'], 'GB/s', j['roofline']['kernel_ms'], 'ms', j['kernel'], 'bit_exact', j['bit_exact'])" | tee -a gpurun_out/r2_spread.log
done
