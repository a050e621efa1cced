#!/bin/bash
# sweep of the MST waiter cap on ONE box (bench step + MST alone)
for w in 0 16 48 96 256; do
  PHONLP_MST_WAITERS=$w timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_w.json 2> gpurun_out/ab_w.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/ab_w.json')); k=d['kernels_ms']
print('waiters=$w', round(d['ms_per_step'],3), 'stageA', k['linear:label_stage_a'], 'at_heads', k['label_at_heads'], 'mst', k['mst'])"
  PHONLP_MST_WAITERS=$w timeout -k 10 100 python tools/mst_probe.py 2>&1 | tail -1
done
