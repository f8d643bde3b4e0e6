#!/bin/bash
# first GPU call of round 2: parity of k_stream4, then timing of 2 / 3 / 4 chains
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_and_four_chains or fixed_stride_hot_path" > gpurun_out/r2_pytest_s4.log 2>&1 || { tail -30 gpurun_out/r2_pytest_s4.log; exit 1; }
tail -2 gpurun_out/r2_pytest_s4.log
for c in 0 4 3; do
  REDGPU_STREAM_CHAINS=$c timeout -k 10 300 python3 scripts/lab_stream.py syn256 --check >> gpurun_out/r2_lab1.log 2>&1 || { tail -20 gpurun_out/r2_lab1.log; exit 1; }
done
cat gpurun_out/r2_lab1.log
