#!/bin/bash
# the root's share of an N > 1 step, priced on one GPU: configs[2] with and without the per-step
# record gather (one-rank RCCL group)
set -u
mkdir -p gpurun_out
L=gpurun_out/r2_rehearse.log
: > $L
timeout -k 10 400 python3 bench.py --config 2 --no-cpu-baseline --no-calibration >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 400 python3 bench.py --rehearse-gather --no-cpu-baseline --no-calibration >> $L 2>&1 || { tail -20 $L; exit 1; }
python3 - <<'PY'
import json
for line in open('gpurun_out/r2_rehearse.log'):
    if line.startswith('{'):
        j=json.loads(line)
        print(j['config']['sharding'][:60], '| value', j['value'], '| ms/step', j['ms_per_step'], '| solo', j.get('single_gpu_same_workload_GBps'), '| gathered', j.get('gathered_steps'), '| exact', j['bit_exact'])
PY
