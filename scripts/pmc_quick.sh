#!/bin/bash
# bash scripts/pmc_quick.sh <tag> "<counters>" <bench args...>   (runs ON THE GPU BOX)
set -u
TAG=$1; P=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --no-calibration --streams 1 "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if any(x in k for x in ("rocclr", "at::", "elementwise", "distribution")): continue
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
