#!/bin/bash
# configs[1]: how many independent batches in flight (HIP streams) before the gain flattens
set -u
for s in 1 2 3 4 6 8; do
timeout -k 10 300 python3 bench.py --streams $s --no-cpu-baseline --no-calibration --steps 300 --warmup 30 2>&1 | grep -v amdgpu | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s:', j['value'], 'GB/s', j['ms_per_step'], 'ms/step', 'buffers', j['config']['rotating_input_buffers'], j['bit_exact'])"
done
