#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/r3_text.log
: > $L
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "text or split" >> $L 2>&1
cat $L
bash scripts/lab/run_text_prof.sh
