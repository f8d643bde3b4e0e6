#!/bin/bash
# lab: bench.py under a list of "ENV=val[,ENV=val] extra args" variants; one summary line each
# usage: run_bench_variants.sh OUTDIR "variant1" "variant2" ...   (variant: "ENV=..;args")
set -o pipefail
O=$1; shift
mkdir -p $O
i=0
for v in "$@"; do
  envs="${v%%;*}"; args="${v#*;}"
  [ "$envs" == "-" ] && envs="X_=1"
  i=$((i+1))
  env ${envs//,/ } timeout -k 10 300 python bench.py --gpus 1 --no-cpu-baseline --no-calibration $args > $O/v$i.log 2>&1 || { echo "FAILED $v"; tail -20 $O/v$i.log; exit 1; }
  python - "$O/v$i.log" "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
r=d['roofline']
print("%-50s value %7.1f  ms/step %.5f  region/step %.5f  launch_ms %.5f (x%d) frac %.4f  single %.5f  exact %s" % (
    sys.argv[2], d['value'], d['ms_per_step'], r['timed_region_ms_per_step'], r['kernel_ms'], r['batches_per_launch'], r['frac'],
    r.get('single_batch_launch',{}).get('kernel_ms', r['kernel_ms']), d['bit_exact']))
PY
done
