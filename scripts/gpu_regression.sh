#!/bin/bash
set -u
bash scripts/gpu_verify_all.sh || exit 1
for mode in "" hot cls lists; do
timeout -k 10 500 python3 scripts/fuzz_gpu.py 300 61 $mode > gpurun_out/r2_fuzz_$mode.log 2>&1 || { tail -30 gpurun_out/r2_fuzz_$mode.log; exit 1; }
echo "fuzz[$mode]: $(tail -1 gpurun_out/r2_fuzz_$mode.log | cut -c1-60)"
done
bash scripts/secondary_benchmarks.sh
