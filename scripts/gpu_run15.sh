#!/bin/bash
set -u
for rep in 1 2; do
for st in 2 3 4 6; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --streams $st --no-cpu-baseline --no-calibration 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st rep $rep:', j['value'], 'GB/s', j['ms_per_step'], 'ms/step', j['bit_exact'])"
done
done
