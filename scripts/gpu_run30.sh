#!/bin/bash
# scan / search / collect on DFAs with L = SIGMA* L (loose start): parity, fuzz, the benchmarks
set -u
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "scan or search or collect or kat or omnibus or all_verbs or vectors or loose or cpp" > gpurun_out/r2_tmp.log 2>&1 || { tail -40 gpurun_out/r2_tmp.log; exit 1; }
tail -1 gpurun_out/r2_tmp.log
timeout -k 10 500 python3 scripts/fuzz_gpu.py 400 41 > gpurun_out/r2_fuzz.log 2>&1 || { tail -30 gpurun_out/r2_fuzz.log; exit 1; }
tail -1 gpurun_out/r2_fuzz.log | cut -c1-200
timeout -k 10 300 python3 scripts/bench_generic.py 2>&1 | grep -v amdgpu.ids | grep "scan"
timeout -k 10 300 python3 scripts/bench_lists.py uri 2>&1 | grep -v amdgpu.ids | grep "collect\|matchAll"
