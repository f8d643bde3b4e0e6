#!/bin/bash
set -u
for v in 0 1; do
export REDGPU_EARLY_VARIANT=$v
echo "== variant $v"
bash scripts/pmc_quick.sh early${v}a "FETCH_SIZE WRITE_SIZE" --config 3 --steps 5 --warmup 1 | grep k_early
bash scripts/pmc_quick.sh early${v}b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --config 3 --steps 5 --warmup 1 | grep k_early
bash scripts/pmc_quick.sh early${v}c "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" --config 3 --steps 5 --warmup 1 | grep k_early
bash scripts/pmc_quick.sh early${v}d "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" --config 3 --steps 5 --warmup 1 | grep k_early
done
