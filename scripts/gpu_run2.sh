#!/bin/bash
# boundary tests + whole GPU suite
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r2_boundary.log 2>&1 || { tail -40 gpurun_out/r2_boundary.log; exit 1; }
tail -3 gpurun_out/r2_boundary.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2_pytest_all.log 2>&1 || { tail -40 gpurun_out/r2_pytest_all.log; exit 1; }
tail -3 gpurun_out/r2_pytest_all.log
