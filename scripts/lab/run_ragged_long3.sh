#!/bin/bash
# lab: where the long-lines-first launch spends its extra microseconds on even lengths
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for x in 0 4; do
  export REDGPU_RAGGED_LONG_X=$x LINES=1048576 CASES="uniform 32-256,geometric"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rl3_x$x -o rl3 -- python3 $R/scripts/bench_ragged_shapes.py uri > $R/gpurun_out/rl3_x$x.log 2>&1
done
cd $R
for x in 0 4; do
  echo "## X=$x"; cat gpurun_out/rl3_x$x.log | grep -v amdgpu.ids
  f=$(find gpurun_out/rl3_x$x -name "*kernel_stats.csv" | head -1)
  head -8 "$f"
done > gpurun_out/r3_ragged_long3.log
cat gpurun_out/r3_ragged_long3.log
