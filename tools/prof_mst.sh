#!/bin/bash
# kernel trace of the MST probe: per-kernel durations of the round-based and the reference-order decoders
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $OUT/prof -o mst --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/mst_probe.py 10 > $OUT/probe.log 2> $OUT/prof.err || { tail -5 $OUT/prof.err; exit 1; }
tail -4 $OUT/probe.log
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv
grep -i "mst" $OUT/kernel_stats.csv | cut -c1-220
