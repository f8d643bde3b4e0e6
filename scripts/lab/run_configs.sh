#!/bin/bash
# every bench config once (short), logs under gpurun_out/$1
O=gpurun_out/$1; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py --gpus 1 --no-cpu-baseline "$@" > $O/$name.log 2>&1 || { echo "FAILED $name"; tail -5 $O/$name.log; }
  python - $O/$name.log $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
except Exception as e:
    print(sys.argv[2], "no json", e); sys.exit(0)
r=d['roofline']
keys=('read_ceiling_GBps','lds_gather_roof_GBps','lines64_memory_roof_GBps','long_line_pattern_roof_GBps','l2_gather_roof_Glookups','result_end_only_GBps','scan_instant_GBps','scanned_GBps')
print("%-14s value %8.1f ms/step %.5f kernel %s launch_ms %.5f achieved %.1f %s frac %.4f exact %s | %s" % (sys.argv[2], d['value'], d['ms_per_step'], d['kernel'], r['kernel_ms'], r['achieved'], r['unit'], r['frac'], d['bit_exact'], " ".join("%s=%s" % (k.replace('_GBps','').replace('_Glookups',''), r[k]) for k in keys if k in r)))
PY
}
run c1_syn256 --steps 20 --warmup 5
run c1_uri --steps 20 --warmup 5 --dfa uri
run c2_syn256 --config 2 --steps 6 --warmup 2
run c2_uri --config 2 --steps 6 --warmup 2 --dfa uri
run c3_log100 --config 3 --steps 20 --warmup 3
run c4_syn4k --config 4 --steps 3 --warmup 1
run c4_uri_v6 --config 4 --steps 5 --warmup 1 --dfa uri_v6
