#!/bin/bash
# timing-only ablations of the split256 steady-state loop (results are garbage for ABL != 0)
for abl in 1 2 3 0; do
  for shape in "32768 13464 408" "4096 4096 4096"; do
    PHONLP_X6_ABLATION=$abl timeout -k 10 60 python tools/gemm_probe.py $shape 10 "abl=$abl" || exit 1
  done
done
