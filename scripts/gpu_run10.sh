#!/bin/bash
set -u
mkdir -p gpurun_out
for v in 0 1 2 3 4 5; do
REDGPU_EARLY_LAB=$v timeout -k 10 500 python3 bench.py --config 3 --no-cpu-baseline --no-calibration > gpurun_out/r2_tmp.log 2>&1 || { tail -20 gpurun_out/r2_tmp.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_tmp.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k_early lab $v:', j['value'], 'GB/s', j['roofline']['kernel_ms'], 'ms', j['kernel'], j['bit_exact'])"
done
