#!/bin/bash
# lab: what the long-lines list costs a SMALL ragged batch, and what it saves when one line is huge
set -e
mkdir -p gpurun_out
L=gpurun_out/r3_small.log
: > $L
for n in 1024 4096 16384; do
  for m in 1000000000 64; do
    echo "## LINES=$n REDGPU_RAGGED_LONG_MIN=$m" >> $L
    LINES=$n REDGPU_RAGGED_LONG_MIN=$m CASES="uniform 32-256,geometric,one 1 MB" \
      timeout -k 10 300 python scripts/bench_ragged_shapes.py uri 2>&1 | grep -v amdgpu.ids >> $L
  done
done
cat $L
