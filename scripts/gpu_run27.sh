#!/bin/bash
set -u
for t in 0 1; do
BENCH_TUNE=$t timeout -k 10 300 python3 bench.py --config 4 --dfa uri_v6 --no-cpu-baseline --no-calibration 2>&1 | grep -v amdgpu | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tune $t:', j['value'], 'GB/s', j['roofline']['kernel_ms'], 'ms', j['kernel'], j['bit_exact'])"
done
timeout -k 10 300 python3 scripts/bench_generic.py 2>&1 | grep -v amdgpu.ids | grep "URI-V6\|URI-USER"
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -k "hot or tune or cold" 2>&1 | tail -1
