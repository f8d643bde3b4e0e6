#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/r2_tmp.log 2>&1 || { tail -40 gpurun_out/r2_tmp.log; exit 1; }
tail -1 gpurun_out/r2_tmp.log
timeout -k 10 500 python3 scripts/fuzz_gpu.py 400 111 > gpurun_out/r2_fuzz.log 2>&1 || { tail -30 gpurun_out/r2_fuzz.log; exit 1; }
tail -1 gpurun_out/r2_fuzz.log | cut -c1-100
timeout -k 10 300 python3 scripts/bench_ragged_styles.py 2>&1 | grep -v amdgpu | grep "Instant\|Last"
