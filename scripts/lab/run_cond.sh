#!/bin/bash
O=gpurun_out/$1; shift
mkdir -p $O
for v in "$@"; do
  [ "$v" == "-" ] && v="X_=1"
  env ${v//,/ } timeout -k 10 120 python scripts/lab/multi_cond.py 2>&1 | grep -v amdgpu.ids | tee -a $O/cond.log
done
