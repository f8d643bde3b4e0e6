#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_boundary.py -q -x > gpurun_out/r2_tmp.log 2>&1 || { tail -40 gpurun_out/r2_tmp.log; exit 1; }
tail -1 gpurun_out/r2_tmp.log
BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --lines 65536 --steps 6 --warmup 2 2>&1 | grep -v amdgpu | tail -1 | cut -c1-600
