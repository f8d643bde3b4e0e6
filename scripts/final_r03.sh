#!/bin/bash
# Runs ON THE GPU BOX: the round's closing measurements with the build as it stands -
#   gpurun_out/r03_bench_all_configs.log  bench.py as the driver runs it (three times) and every
#                                         other config once, one "## name:" + JSON line each
#   gpurun_out/secondary_benchmarks.log   scripts/secondary_benchmarks.sh
# (copied into profiles/ by hand afterwards)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
L=gpurun_out/r03_bench_all_configs.log
: > $L
one() { name=$1; shift; echo "## $name: " >> $L; timeout -k 10 500 python3 bench.py --gpus 1 "$@" 2>gpurun_out/final_$name.err | tail -1 >> $L || echo "FAILED $name" >> $L; }
one default1 --steps 20 --warmup 5
one default2 --steps 20 --warmup 5
one default3 --steps 20 --warmup 5
one c1_uri --steps 20 --warmup 5 --dfa uri --no-cpu-baseline
one c2_syn256 --config 2 --steps 6 --warmup 2 --no-cpu-baseline
one c2_uri --config 2 --steps 6 --warmup 2 --dfa uri --no-cpu-baseline
one c3_log100 --config 3 --steps 20 --warmup 3 --no-cpu-baseline
one c4_syn4k --config 4 --steps 3 --warmup 1 --no-cpu-baseline
one c4_uri_v6 --config 4 --steps 5 --warmup 1 --dfa uri_v6 --no-cpu-baseline
python3 - $L <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); r = d["roofline"]
        print("%-60s value %8.1f ms/step %.5f kernel_ms %.5f frac %.4f exact %s" % (
            d["config"]["workload"][:60], d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], d["bit_exact"]))
    elif "FAILED" in line:
        print(line.strip())
PY
bash scripts/secondary_benchmarks.sh > /dev/null 2>&1
tail -3 gpurun_out/secondary_benchmarks.log
