#!/bin/bash
# A/B on ONE box: T in the rows-by-4 layout (default) vs row-major, same bench, back to back, twice
for rep in 1 2; do
  for r4 in 1 0; do
    PHONLP_ROWS4=$r4 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_r4_$r4.json 2> gpurun_out/ab_r4_$r4.err || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/ab_r4_$r4.json')); k=d['kernels_ms']
print('rows4=$r4 rep=$rep', round(d['ms_per_step'],3), 'stageA', k['linear:label_stage_a'], 'at_heads', k['label_at_heads'], 'mst', k['mst'])"
  done
done
