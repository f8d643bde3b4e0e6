#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "probe_and_drain or config4_shape or full_size_config4 or reference_vectors" > gpurun_out/r2_pytest_early.log 2>&1 || { tail -40 gpurun_out/r2_pytest_early.log; exit 1; }
tail -2 gpurun_out/r2_pytest_early.log
timeout -k 10 500 python3 bench.py --config 3 --no-cpu-baseline > gpurun_out/r2_bench_c3.log 2>&1 || { tail -20 gpurun_out/r2_bench_c3.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_bench_c3.log | cut -c1-1500
