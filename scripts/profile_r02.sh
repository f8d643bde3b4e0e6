#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  bash scripts/profile_r02.sh <config> <dfa> [steps] [extra bench args]
# rocprofv3 kernel-trace stats of `bench.py --config C --dfa D` (one stream, and the default
# streams for config 1), then PMC passes (own runs, --kernel-trace only, as gpurun requires) for
# HBM traffic, LDS and L2.  Outputs under gpurun_out/prof_r02_c<C>_<D>/ ; summarise with
# scripts/summarize_r02.py (copies what is judged into profiles/).
set -u
C=${1:-1}
D=${2:-syn256}
STEPS=${3:-100}
shift 3 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_r02_c${C}_${D}
mkdir -p $OUT
WARM=$(( STEPS / 10 + 1 ))
BENCH="python3 $R/bench.py --config $C --dfa $D --no-cpu-baseline --no-calibration --steps $STEPS --warmup $WARM $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_1stream -- $BENCH --streams 1 > $OUT/stats_1stream.log 2>&1 || { tail -5 $OUT/stats_1stream.log; exit 1; }
if [ "$C" = "1" ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- $BENCH > $OUT/stats_default.log 2>&1 || { tail -5 $OUT/stats_default.log; exit 1; }
fi
PSTEPS=$(( STEPS / 4 + 2 ))
PB="python3 $R/bench.py --config $C --dfa $D --no-cpu-baseline --no-calibration --steps $PSTEPS --warmup 2 --streams 1 $*"
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$N -- $PB > $OUT/pmc_$N.log 2>&1 || { tail -5 $OUT/pmc_$N.log; exit 1; }
done
echo profile_done $C $D
