#!/bin/bash
set -u
timeout -k 10 300 python3 bench.py --config 4 --dfa uri_v6 --cpu-seconds 6 > gpurun_out/r2_c4uri.log 2>&1 || { tail gpurun_out/r2_c4uri.log; exit 1; }
grep -v amdgpu gpurun_out/r2_c4uri.log | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 4 uri_v6:', j['value'], 'GB/s', j['roofline']['kernel_ms'], 'ms', j['config']['hot_rows'], j['bit_exact'])"
