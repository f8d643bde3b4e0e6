#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2_pytest_all4.log 2>&1 || { tail -40 gpurun_out/r2_pytest_all4.log; exit 1; }
tail -1 gpurun_out/r2_pytest_all4.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu
L=gpurun_out/r2_bench_final.log
: > $L
run() { echo "### $*" >> $L; timeout -k 10 500 "$@" >> $L 2>&1 || { echo "FAILED: $*"; tail -30 $L; exit 1; }; }
run python3 bench.py --steps 20 --warmup 5
run python3 bench.py --steps 300 --warmup 30
run python3 bench.py --dfa uri --steps 300 --warmup 30 --cpu-seconds 6
run python3 bench.py --config 2 --cpu-seconds 6
run python3 bench.py --config 2 --dfa uri --cpu-seconds 6
run python3 bench.py --config 3 --cpu-seconds 6
run python3 bench.py --config 4 --cpu-seconds 6
run python3 bench.py --config 4 --dfa uri_v6 --cpu-seconds 6
BENCH_BACKEND=gloo run python3 bench.py --gpus 2 --lines 65536 --steps 6 --warmup 2
python3 - <<'PY'
import json
for line in open('gpurun_out/r2_bench_final.log'):
    if line.startswith('{'):
        j=json.loads(line)
        print(j['config']['workload'][:52], '| steps', j['steps'], '| value', j['value'], '| kernel_ms', j['roofline']['kernel_ms'], '| frac', j['roofline']['frac'], '| cpu', j.get('cpu_baseline',{}).get('value'), '| bit_exact', j['bit_exact'], '| n', j['n_gpus'])
PY
