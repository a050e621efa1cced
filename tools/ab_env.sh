#!/bin/bash
# A/B of one environment switch on ONE box: usage tools/ab_env.sh VAR valA valB [reps]
var=$1; a=$2; b=$3; reps=${4:-2}
for rep in $(seq $reps); do
  for v in $a $b; do
    env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_env.json 2> gpurun_out/ab_env.err || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/ab_env.json')); k=d['kernels_ms']
print('$var=$v', round(d['ms_per_step'],3), {n: round(x,3) for n,x in k.items()})"
  done
done
