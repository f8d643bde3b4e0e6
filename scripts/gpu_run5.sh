#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "probe_and_drain or config4_shape or full_size_config4" > gpurun_out/r2_pytest_early2.log 2>&1 || { tail -40 gpurun_out/r2_pytest_early2.log; exit 1; }
tail -2 gpurun_out/r2_pytest_early2.log
timeout -k 10 500 python3 bench.py --config 3 --no-cpu-baseline --no-calibration > gpurun_out/r2_bench_c3b.log 2>&1 || { tail -20 gpurun_out/r2_bench_c3b.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_bench_c3b.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3', j['value'], j['roofline']['kernel_ms'], j['kernel'], j['bit_exact'])"
for nt in 0 1; do
REDGPU_GATHER_NT=$nt timeout -k 10 500 python3 bench.py --config 4 --no-cpu-baseline --no-calibration --steps 3 > gpurun_out/r2_bench_c4_nt$nt.log 2>&1 || { tail -20 gpurun_out/r2_bench_c4_nt$nt.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_bench_c4_nt$nt.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4 nt=$nt', j['value'], j['roofline']['kernel_ms'], j['kernel'], j['bit_exact'])"
done
bash scripts/pmc_quick.sh c3c "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" --config 3 --steps 6 --warmup 1
