#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "match_all or collect" > gpurun_out/r2_pytest_ma.log 2>&1 || { tail -40 gpurun_out/r2_pytest_ma.log; exit 1; }
tail -1 gpurun_out/r2_pytest_ma.log
python3 scripts/bench_lists.py syn256 2>&1 | grep -v amdgpu | grep -E "matchAll|collect" | head -4
python3 scripts/bench_lists.py uri 2>&1 | grep -v amdgpu | grep -E "matchAll" | head -3
timeout -k 10 400 python3 scripts/fuzz_gpu.py 400 21 > gpurun_out/r2_fuzz_a.log 2>&1 || { tail -5 gpurun_out/r2_fuzz_a.log; exit 1; }
tail -1 gpurun_out/r2_fuzz_a.log | cut -c1-600
timeout -k 10 300 python3 scripts/fuzz_gpu.py 150 22 hot > gpurun_out/r2_fuzz_b.log 2>&1 || { tail -5 gpurun_out/r2_fuzz_b.log; exit 1; }
tail -1 gpurun_out/r2_fuzz_b.log | cut -c1-400
timeout -k 10 300 python3 scripts/fuzz_gpu.py 150 23 cls > gpurun_out/r2_fuzz_c.log 2>&1 || { tail -5 gpurun_out/r2_fuzz_c.log; exit 1; }
tail -1 gpurun_out/r2_fuzz_c.log | cut -c1-400
