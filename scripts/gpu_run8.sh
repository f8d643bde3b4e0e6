#!/bin/bash
set -u
mkdir -p gpurun_out
python3 scripts/bench_host_path.py > gpurun_out/r2_host_path.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_host_path.log
bash scripts/profile_r02.sh 1 syn256 200 || exit 1
bash scripts/profile_r02.sh 1 uri 200 || exit 1
bash scripts/profile_r02.sh 2 syn256 12 || exit 1
bash scripts/profile_r02.sh 3 log100 20 || exit 1
bash scripts/profile_r02.sh 4 syn4k 3 || exit 1
bash scripts/profile_r02.sh 4 uri_v6 8 || exit 1
python3 scripts/summarize_r02.py
