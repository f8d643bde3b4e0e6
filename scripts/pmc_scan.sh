#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over scripts/bench_scan.py (k_scan_marked).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/pmc_scan
rm -rf $OUT; mkdir -p $OUT
for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$N -- python3 $R/scripts/bench_scan.py > $OUT/$N.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*/*_counter_collection.csv")):
    agg = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        if "k_scan_marked" in row["Kernel_Name"]:
            agg.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(k, "n", len(v), "first8avg %.0f" % (sum(v[:8]) / max(1, len(v[:8]))), "all avg %.0f" % (sum(v) / len(v)))
PY
