#!/bin/bash
# timing-only ablations of the f16x3 DMA kernel's epilogue (results are garbage for ABL != 0)
for abl in 0 1 2; do
  PHONLP_X3_ABLATION=$abl timeout -k 10 100 python tools/f16x3_probe.py 10 2>&1 | grep "f16x3" | sed "s/^/abl=$abl /" || exit 1
done
