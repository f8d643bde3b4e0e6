#!/bin/bash
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TEXT_ONLY=1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/text_prof -o tp -- python3 $R/scripts/bench_lists.py uri > $R/gpurun_out/text_prof.log 2>&1
grep -v "simple_timer\|amdgpu.ids" $R/gpurun_out/text_prof.log | tail -12
