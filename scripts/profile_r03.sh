#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  bash scripts/profile_r03.sh <config> <dfa> [steps] [extra bench args]
# rocprofv3 kernel-trace stats of `bench.py --config C --dfa D` as bench.py issues it by default
# (config 1: 20 steps per redgpu_match_batches_dev call) and, for config 1, one step per call;
# then PMC passes (own runs, --kernel-trace only, as gpurun requires) for HBM traffic, LDS, VALU
# and L2.  Outputs under gpurun_out/prof_r03_c<C>_<D>/ ; summarise with scripts/summarize_r03.py
# (copies what is judged into profiles/).
set -u
C=${1:-1}
D=${2:-syn256}
STEPS=${3:-200}
shift 3 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_r03_c${C}_${D}
mkdir -p $OUT
WARM=$(( STEPS / 10 + 1 ))
PER=""
[ "$C" = "1" ] && PER="--per-call 20"
BENCH="python3 $R/bench.py --config $C --dfa $D --no-cpu-baseline --no-calibration --steps $STEPS --warmup $WARM $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- $BENCH $PER > $OUT/stats_default.log 2>&1 || { tail -5 $OUT/stats_default.log; exit 1; }
if [ "$C" = "1" ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_1percall -- $BENCH --per-call 1 > $OUT/stats_1percall.log 2>&1 || { tail -5 $OUT/stats_1percall.log; exit 1; }
fi
PSTEPS=$(( STEPS / 4 + 2 ))
[ "$C" = "1" ] && PSTEPS=60
PB="python3 $R/bench.py --config $C --dfa $D --no-cpu-baseline --no-calibration --steps $PSTEPS --warmup 2 $PER $*"
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$N -- $PB > $OUT/pmc_$N.log 2>&1 || { tail -5 $OUT/pmc_$N.log; exit 1; }
done
echo profile_done $C $D
