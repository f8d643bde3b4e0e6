#!/bin/bash
set -u
timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('config 1:', j['value'], 'GB/s | kernel_ms', r['kernel_ms'], '| graph', r.get('graph_replay_GBps'), r.get('graph_replay_error'), '| exact', j['bit_exact'])"
