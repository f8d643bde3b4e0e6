#!/bin/bash
# lab: bench_ragged_shapes.py with the library of an older commit (a copy of its one_amd/ under
# scripts/lab/ab_old/, not kept in the tree) against today's, same box, alternating
set -e
R=$(pwd)
mkdir -p gpurun_out
L=gpurun_out/ab_ragged.log
: > $L
for rep in 1 2; do
  for which in old new; do
    echo "## $which (rep $rep) LINES=${LINES:-8388608}" >> $L
    if [ $which = old ]; then
      (cd scripts/lab/ab_old && LINES=${LINES:-8388608} CASES="uniform 32-256,all 256,geometric" PYTHONPATH=$R/scripts/lab/ab_old:$R/tests python - <<'PY' >> $R/$L 2>&1
import os, sys, runpy
sys.path.insert(0, os.path.join(os.getcwd()))
sys.argv = ["bench_ragged_shapes.py", "uri"]
src = open(os.path.join(os.environ.get("R", "../../.."), "scripts", "bench_ragged_shapes.py")).read()
src = src.replace("sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))", "")
src = src.replace('sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))', "")
exec(compile(src, "bench_ragged_shapes.py", "exec"))
PY
      )
    else
      LINES=${LINES:-8388608} CASES="uniform 32-256,all 256,geometric" python scripts/bench_ragged_shapes.py uri >> $L 2>&1
    fi
  done
done
grep -v amdgpu $L
