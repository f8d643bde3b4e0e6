#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over scripts/bench_ragged_shapes.py (set LINES / CASES).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/pmc_ragged
rm -rf $OUT; mkdir -p $OUT
for P in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$N -- python3 $R/scripts/bench_ragged_shapes.py > $OUT/$N.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*/*_counter_collection.csv")):
    agg = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        if "k_refill" in row["Kernel_Name"] or "k_ragged" in row["Kernel_Name"]:
            agg.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in agg.items():
        g = [v[i:i + 13] for i in range(0, len(v), 13)]   # 13 launches per case
        print(k, " | ".join("%.0f" % (sum(x) / len(x)) for x in g))
PY
