#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of scripts/bench_scan.py (scan / search over ERR,
# NUM3, AAB) and of scripts/bench_generic.py (ragged lines incl. the loose-start scan)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $R
for S in bench_scan bench_generic; do
  OUT=$R/gpurun_out/prof_r02_$S
  mkdir -p $OUT
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/$S.py > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
done
echo scan_profile_done
