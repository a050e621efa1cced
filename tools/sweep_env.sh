#!/bin/bash
# sweep of one environment switch on ONE box: usage tools/sweep_env.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/sweep_env.json 2> gpurun_out/sweep_env.err || { tail -5 gpurun_out/sweep_env.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/sweep_env.json')); k=d['kernels_ms']
print('$var=$v', round(d['ms_per_step'],3), {n: round(x,3) for n,x in k.items() if n in ('linear:label_stage_a','label_at_heads','label_reduce','mst')})"
done
