#!/bin/bash
# lab: match_text parity; wave priority for long lines on / off
set -e
mkdir -p gpurun_out
L=gpurun_out/r3_ragged_long2.log
: > $L
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ragged or text or split" >> $L 2>&1
for n in 1048576 2097152 8388608; do
  for p in 0 1; do
    echo "## LINES=$n REDGPU_RAGGED_PRIO=$p" >> $L
    LINES=$n REDGPU_RAGGED_PRIO=$p CASES="geometric,uniform 32-256,uniform 1-2048" \
      timeout -k 10 300 python scripts/bench_ragged_shapes.py uri >> $L 2>&1
  done
done
tail -5 $L
