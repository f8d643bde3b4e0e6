#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench command,
# then PMC passes (own runs, --kernel-trace only, as gpurun requires) for HBM traffic and LDS.
# Outputs under gpurun_out/prof_<tag>/ ; summarise with scripts/summarize_profiles.py.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 $R/bench.py --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- $BENCH --steps 300 --warmup 30 > $OUT/stats_default.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_1stream -- $BENCH --steps 300 --warmup 30 --streams 1 > $OUT/stats_1stream.log 2>&1 || exit 1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$N -- $BENCH --steps 60 --warmup 10 --streams 1 > $OUT/pmc_$N.log 2>&1 || exit 1
done
echo profile_done
