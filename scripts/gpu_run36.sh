#!/bin/bash
# configs[3]'s lines under DFAs that die at once: what the probe alone (offsets, first piece,
# Outcome stores) costs on 2^23 ragged lines
set -u
for dfa in err aab log100; do
timeout -k 10 300 python3 bench.py --config 3 --dfa $dfa --no-cpu-baseline --no-calibration 2>&1 | grep -v amdgpu | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$dfa:', j['value'], 'GB/s | kernel_ms', r['kernel_ms'], j['kernel'], j['bit_exact'])"
done
