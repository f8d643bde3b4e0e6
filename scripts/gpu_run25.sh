#!/bin/bash
# scan / search with more than 4 start bytes through k_scan_marked's flag table: parity, benches
set -u
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "scan or search or kat or omnibus or all_verbs or vectors" > gpurun_out/r2_tmp.log 2>&1 || { tail -40 gpurun_out/r2_tmp.log; exit 1; }
tail -1 gpurun_out/r2_tmp.log
timeout -k 10 500 python3 scripts/fuzz_gpu.py 400 21 > gpurun_out/r2_fuzz.log 2>&1 || { tail -30 gpurun_out/r2_fuzz.log; exit 1; }
tail -1 gpurun_out/r2_fuzz.log
timeout -k 10 300 python3 scripts/bench_scan.py 2>&1 | grep -v amdgpu.ids
