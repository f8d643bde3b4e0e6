#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2_pytest_all3.log 2>&1 || { tail -40 gpurun_out/r2_pytest_all3.log; exit 1; }
tail -2 gpurun_out/r2_pytest_all3.log
python3 scripts/bench_lists.py syn256 > gpurun_out/r2_lists_syn256.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_lists_syn256.log
python3 scripts/bench_lists.py uri > gpurun_out/r2_lists_uri.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_lists_uri.log
python3 scripts/bench_scan.py > gpurun_out/r2_scan.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_scan.log | tail -8
for args in "--config 3" "--config 4 --steps 3"; do
timeout -k 10 500 python3 bench.py $args --no-cpu-baseline --no-calibration > gpurun_out/r2_tmp.log 2>&1 || { tail -20 gpurun_out/r2_tmp.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_tmp.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', j['value'], j['roofline']['kernel_ms'], j['kernel'], j['bit_exact'])"
done
