#!/bin/bash
# Diagnostic build of the fused arc kernel with cycle stamps of block 0 (-DARC_STAMPS; tools/arc_probe.py reads them):
#   bash tools/build_arc_stamps.sh [extra -D flags]   ->  tools/libarc_stamps.so  (the other objects are the shipped ones)
set -e
cd "$(dirname "$0")/.."
python -m phonlp_amd.build
O=phonlp_amd/lib/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -Wno-inline-asm -DARC_STAMPS "$@" -c -o /tmp/arc_stamps.o phonlp_amd/csrc/arc_f16x3.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libarc_stamps.so /tmp/arc_stamps.o $(ls $O/*.o | grep -v arc_f16x3.o)
