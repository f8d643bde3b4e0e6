#!/bin/bash
set -u
mkdir -p gpurun_out
: > gpurun_out/r2_stagger.log
for st in 0 1 2 4 8 16; do
REDGPU_STREAM_STAGGER=$st SHAPES=1048576x64,262144x4096 timeout -k 10 300 python3 scripts/lab_stream.py syn256 2>&1 | grep chains | sed "s/^/stagger=$st /" | tee -a gpurun_out/r2_stagger.log
done
