#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats + PMC passes (own runs) of scripts/prof_lists.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $R
for CAP in 4 8; do
  export CAP
  OUT=$R/gpurun_out/prof_r02_lists_cap$CAP
  mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/prof_lists.py > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
  for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES"; do
    N=$(echo $P | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$N -- python3 $R/scripts/prof_lists.py > $OUT/pmc_$N.log 2>&1 || { tail -5 $OUT/pmc_$N.log; exit 1; }
  done
done
echo lists_profile_done
