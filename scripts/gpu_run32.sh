#!/bin/bash
set -u
for c in 1 2; do
timeout -k 10 300 python3 bench.py --config $c --dfa uri --no-cpu-baseline 2>&1 | grep -v amdgpu | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('config $c uri:', j['value'], 'GB/s | scan_instant', r.get('scan_instant_GBps'), r.get('scan_instant_kernel'), r.get('scan_instant_agrees_with_match'), '| result_end_only', r.get('result_end_only_GBps'), '| total traffic', r.get('total_traffic_GBps'), j['bit_exact'])"
done
