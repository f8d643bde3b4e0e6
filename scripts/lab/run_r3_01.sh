#!/bin/bash
# round 3, GPU call 1: multi-batch parity + bench + lab knobs
set -o pipefail
mkdir -p gpurun_out/r3_01
O=gpurun_out/r3_01
./scripts/lab/d16_probe > $O/d16.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_multibatch.py -x -q > $O/pytest_multi.log 2>&1 || { tail -30 $O/pytest_multi.log; exit 1; }
tail -3 $O/pytest_multi.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.log 2>&1 || { tail -30 $O/bench_default.log; exit 1; }
tail -c 3000 $O/bench_default.log
for v in "REDGPU_MULTI_NT=1" "REDGPU_MULTI_NT=2" "REDGPU_MULTI_NT=3" "REDGPU_MULTI_WGS=2" ; do
  env $v timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration > $O/bench_$v.log 2>&1 || { tail -30 $O/bench_$v.log; exit 1; }
  echo "$v: $(python -c "import json,sys; d=json.loads(open('$O/bench_$v.log').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])")"
done
timeout -k 10 200 python bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline --no-calibration > $O/bench_300.log 2>&1
python -c "import json,sys; d=json.loads(open('$O/bench_300.log').read().strip().split('\n')[-1]); print('steps300', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout -k 10 200 python bench.py --gpus 1 --steps 300 --warmup 30 --per-call 1 --streams 3 --no-cpu-baseline --no-calibration > $O/bench_3streams.log 2>&1
python -c "import json,sys; d=json.loads(open('$O/bench_3streams.log').read().strip().split('\n')[-1]); print('3streams', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --dfa uri --no-cpu-baseline --no-calibration > $O/bench_uri.log 2>&1
python -c "import json,sys; d=json.loads(open('$O/bench_uri.log').read().strip().split('\n')[-1]); print('uri', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
