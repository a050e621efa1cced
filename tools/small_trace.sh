#!/bin/bash
# bash tools/small_trace.sh B L eager|graph : kernel trace of one small-shape step, printed as a timeline
B=$1; L=$2; M=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/st_${B}_${L}_$M
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/small_trace.py $B $L $M 200 > $OUT/plain.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/small_trace.py $B $L $M > $OUT/run.log 2> $OUT/err.log || exit 1
cat $OUT/plain.log
python3 $GRAFT_REPO_ROOT/tools/print_step.py $OUT 8 > $OUT/timeline.txt && cat $OUT/timeline.txt
