#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_and_four_chains" > gpurun_out/r2_pytest_s4b.log 2>&1 || { tail -30 gpurun_out/r2_pytest_s4b.log; exit 1; }
tail -2 gpurun_out/r2_pytest_s4b.log
: > gpurun_out/r2_lab2.log
for c in 0 3 4; do
  SHAPES=4194304x256,262144x4096,2097152x4096,1048576x128 REDGPU_STREAM_CHAINS=$c timeout -k 10 300 python3 scripts/lab_stream.py syn256 >> gpurun_out/r2_lab2.log 2>&1 || { tail -20 gpurun_out/r2_lab2.log; exit 1; }
done
SHAPES=2097152x4096 REDGPU_STREAM_CHAINS=0 timeout -k 10 300 python3 scripts/lab_stream.py uri >> gpurun_out/r2_lab2.log 2>&1
SHAPES=2097152x4096 REDGPU_STREAM_CHAINS=4 timeout -k 10 300 python3 scripts/lab_stream.py uri >> gpurun_out/r2_lab2.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_lab2.log
