#!/bin/bash
# lab: per-kernel times (rocprofv3) of bench_ragged_shapes.py, older library vs today's
set -e
R=$(pwd)
export TMPDIR=/tmp LINES=8388608 CASES="all 256"
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_new -o n -- python3 $R/scripts/bench_ragged_shapes.py uri > $R/gpurun_out/ab_new.log 2>&1
cd $R/scripts/lab/ab_old
cat > /tmp/old_bench.py <<PY
import os, sys
sys.path.insert(0, "$R/scripts/lab/ab_old"); sys.path.insert(0, "$R/tests")
sys.argv = ["bench_ragged_shapes.py", "uri"]
src = open("$R/scripts/bench_ragged_shapes.py").read()
src = src.replace("sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))", "")
src = src.replace('sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))', "")
exec(compile(src, "bench_ragged_shapes.py", "exec"))
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_old -o o -- python3 /tmp/old_bench.py > $R/gpurun_out/ab_old.log 2>&1
grep "all 256" $R/gpurun_out/ab_new.log $R/gpurun_out/ab_old.log
