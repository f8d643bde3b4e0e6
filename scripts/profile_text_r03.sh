#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the raw-text path (bench_lists.py uri, TEXT_ONLY)
# and of ragged batches with huge lines (bench_ragged_shapes.py uri) -> gpurun_out/prof_text_r03/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/prof_text_r03
mkdir -p $O
TEXT_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/text -- python3 $R/scripts/bench_lists.py uri > $O/text.log 2>&1
export LINES=1048576 CASES="geometric,uniform 32-256,one 1 MB,a 64 KB"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ragged -- python3 $R/scripts/bench_ragged_shapes.py uri > $O/ragged.log 2>&1
for d in text ragged; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r03_kernel_stats_${d}_uri.csv
done
grep -v "simple_timer\|amdgpu.ids\|^W2\|^E2" $O/text.log | tail -9
head -8 $R/gpurun_out/r03_kernel_stats_text_uri.csv
