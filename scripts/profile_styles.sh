#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of scripts/bench_styles.py uri (check / match, every
# style: k_stream, k_style_blocks) and a PMC pass (L2 requests, LDS) of the same command
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export VERBS=check,match
cd $R
OUT=$R/gpurun_out/prof_r02_styles
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/bench_styles.py uri > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$N -- python3 $R/scripts/bench_styles.py uri > $OUT/pmc_$N.log 2>&1 || { tail -5 $OUT/pmc_$N.log; exit 1; }
done
echo styles_profile_done
