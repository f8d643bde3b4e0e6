#!/bin/bash
# the block kernels after a change: parity, the block-biased fuzz, the stride / style benchmarks
set -u
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/r2_tmp.log 2>&1 || { tail -40 gpurun_out/r2_tmp.log; exit 1; }
tail -1 gpurun_out/r2_tmp.log
for sd in 21 22; do timeout -k 10 400 python3 scripts/fuzz_gpu.py 1500 $sd blocks 2>&1 | grep -v amdgpu | tail -1 | cut -c1-40; done
timeout -k 10 400 python3 scripts/fuzz_gpu.py 300 23 lists 2>&1 | grep -v amdgpu | tail -1 | cut -c1-40
timeout -k 10 250 python3 scripts/bench_strides.py syn256 2>&1 | grep -v amdgpu
timeout -k 10 250 python3 scripts/bench_matchall_cap.py 2>&1 | grep -v amdgpu | grep syn256
