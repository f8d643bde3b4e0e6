#!/bin/bash
# lab: k_ragged with the long lines of a batch first (k_ragged_outliers) - parity, then the
# threshold factor X (REDGPU_RAGGED_LONG_X; 0 = plain input order) on skewed and even lengths
set -e
mkdir -p gpurun_out
L=gpurun_out/r3_ragged_long.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ragged" >> $L 2>&1
for n in 1048576 2097152 8388608; do
  for x in 0 2 3 4 8; do
    echo "## LINES=$n REDGPU_RAGGED_LONG_X=$x" >> $L
    LINES=$n REDGPU_RAGGED_LONG_X=$x CASES="geometric,uniform 32-256,uniform 1-2048,all 256" \
      timeout -k 10 300 python scripts/bench_ragged_shapes.py uri >> $L 2>&1
  done
done
echo "## text: split + match (bench_lists tail)" >> $L
for x in 0 4; do
  echo "## REDGPU_RAGGED_LONG_X=$x" >> $L
  REDGPU_RAGGED_LONG_X=$x timeout -k 10 300 python scripts/bench_lists.py 2>&1 | grep -E "^split|^match" >> $L
done
tail -5 $L
