#!/bin/bash
set -u
bash scripts/gpu_run12.sh || exit 1
bash scripts/profile_r02.sh 4 uri_v6 5 || exit 1
timeout -k 10 500 python3 scripts/fuzz_gpu.py 300 31 hot > gpurun_out/r2_fuzz_hot.log 2>&1 || { tail -30 gpurun_out/r2_fuzz_hot.log; exit 1; }
tail -1 gpurun_out/r2_fuzz_hot.log | cut -c1-300
bash scripts/secondary_benchmarks.sh
