#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/r2_bench_all.log
: > $L
run() { echo "### $*" >> $L; timeout -k 10 500 "$@" >> $L 2>&1 || { echo "FAILED: $*"; tail -30 $L; exit 1; }; }
run python3 bench.py --steps 100 --warmup 10
run python3 bench.py --dfa uri --steps 100 --warmup 10 --cpu-seconds 6
run python3 bench.py --config 2 --cpu-seconds 6
run python3 bench.py --config 2 --dfa uri --cpu-seconds 6
run python3 bench.py --config 3 --cpu-seconds 6
run python3 bench.py --config 4 --cpu-seconds 6
run python3 bench.py --config 4 --dfa uri_v6 --cpu-seconds 6
BENCH_BACKEND=gloo run python3 bench.py --gpus 2 --config 2 --lines 65536 --steps 6 --warmup 2
grep -v "amdgpu.ids" $L | cut -c1-1800
