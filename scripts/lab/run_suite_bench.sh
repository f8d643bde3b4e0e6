#!/bin/bash
# full GPU test suite, then bench.py three times (variance) - logs under gpurun_out/$1
set -o pipefail
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $2 > $O/bench$i.log 2>&1 || { tail -20 $O/bench$i.log; exit 1; }
  python - $O/bench$i.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
r=d['roofline']
print("value %.1f ms/step %.5f region/step %.5f issue_ms %.4f launch_ms %.5f (x%d) frac %.4f single %.5f (%.4f) exact %s" % (
    d['value'], d['ms_per_step'], r['timed_region_ms_per_step'], r.get('timed_region_host_issue_ms',0), r['kernel_ms'], r['batches_per_launch'], r['frac'],
    r.get('single_batch_launch',{}).get('kernel_ms',0), r.get('single_batch_launch',{}).get('frac',0), d['bit_exact']))
PY
done
