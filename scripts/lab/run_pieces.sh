#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/r3_pieces_bench.log
: > $L
for n in 1048576 2097152 8388608; do
  for p in 0 1; do
    echo "## LINES=$n REDGPU_RAGGED_PIECES=$p" >> $L
    LINES=$n REDGPU_RAGGED_PIECES=$p CASES="geometric,uniform 32-256,uniform 1-2048,one 1 MB,a 64 KB" \
      timeout -k 10 300 python scripts/bench_ragged_shapes.py uri 2>&1 | grep -v amdgpu.ids >> $L
  done
done
for p in 0 1; do
  echo "## text REDGPU_RAGGED_PIECES=$p" >> $L
  TEXT_ONLY=1 REDGPU_RAGGED_PIECES=$p timeout -k 10 300 python scripts/bench_lists.py uri 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
