#!/bin/bash
# k_early deep-probe variants (REDGPU_EARLY_VARIANT: lines per lane / waves per SIMD / 16-byte
# pieces of each line held in registers): parity first, then configs[3]
set -u
mkdir -p gpurun_out
for v in 0 2 3 4 5 6; do
REDGPU_EARLY_VARIANT=$v timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -k "probe_and_drain or config3 or ragged" > gpurun_out/r2_tmp.log 2>&1 || { tail -30 gpurun_out/r2_tmp.log; exit 1; }
tail -1 gpurun_out/r2_tmp.log
REDGPU_EARLY_VARIANT=$v timeout -k 10 500 python3 bench.py --config 3 --no-cpu-baseline --no-calibration > gpurun_out/r2_tmp.log 2>&1 || { tail -20 gpurun_out/r2_tmp.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_tmp.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k_early variant $v:', j['value'], 'GB/s', j['roofline']['kernel_ms'], 'ms', j['kernel'], j['bit_exact'])"
done
