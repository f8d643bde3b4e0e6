#!/bin/bash
# lean collect: parity (tests + list fuzz), then the list benchmarks
set -u
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "collect or all_verbs or replace" > gpurun_out/r2_tmp.log 2>&1 || { tail -40 gpurun_out/r2_tmp.log; exit 1; }
tail -1 gpurun_out/r2_tmp.log
timeout -k 10 500 python3 scripts/fuzz_gpu.py 300 13 lists > gpurun_out/r2_fuzz.log 2>&1 || { tail -30 gpurun_out/r2_fuzz.log; exit 1; }
tail -1 gpurun_out/r2_fuzz.log
timeout -k 10 300 python3 scripts/bench_lists.py 2>&1 | grep -v amdgpu.ids | grep "collect\|matchAll"
timeout -k 10 300 python3 scripts/bench_lists.py uri 2>&1 | grep -v amdgpu.ids | grep "collect\|matchAll"
