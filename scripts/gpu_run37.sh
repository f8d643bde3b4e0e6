#!/bin/bash
# k_early's drain without a branch per byte (REDGPU_EARLY_LEAN=1, the default) against the
# reference loop body through walkBytes (=0): parity, then configs[3]
set -u
mkdir -p gpurun_out
for v in 1 0; do
REDGPU_EARLY_LEAN=$v timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -k "probe_and_drain or config3 or ragged or early" > gpurun_out/r2_tmp.log 2>&1 || { tail -30 gpurun_out/r2_tmp.log; exit 1; }
tail -1 gpurun_out/r2_tmp.log
REDGPU_EARLY_LEAN=$v timeout -k 10 500 python3 bench.py --config 3 --no-cpu-baseline --no-calibration > gpurun_out/r2_tmp.log 2>&1 || { tail -20 gpurun_out/r2_tmp.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_tmp.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean drain $v:', j['value'], 'GB/s', j['roofline']['kernel_ms'], 'ms', j['kernel'], j['bit_exact'])"
done
REDGPU_EARLY_LEAN=1 timeout -k 10 500 python3 scripts/fuzz_gpu.py 300 71 > gpurun_out/r2_fuzz.log 2>&1 || { tail -30 gpurun_out/r2_fuzz.log; exit 1; }
tail -1 gpurun_out/r2_fuzz.log | cut -c1-120
