#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/r2_rehearse.log
: > $L
BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --lines 65536 --steps 6 --warmup 2 >> $L 2>&1 || { tail -30 $L; exit 1; }
timeout -k 10 400 python3 bench.py --rehearse-gather --no-cpu-baseline --no-calibration >> $L 2>&1 || { tail -20 $L; exit 1; }
python3 - <<'PY'
import json
for line in open('gpurun_out/r2_rehearse.log'):
    if line.startswith('{'):
        j=json.loads(line)
        print('n', j['n_gpus'], '| value', j['value'], '| ms/step', j['ms_per_step'], '| solo', j.get('single_gpu_same_workload_GBps'), '| gathered', j.get('gathered_steps'), j.get('gathered_matches_ranks'), '| exact', j['bit_exact'])
PY
