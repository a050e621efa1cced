#!/bin/bash
# Kernel trace of the bench under the current environment, printed as one step's timeline: bash tools/trace_step.sh <tag>
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log || exit 1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $OUT $OUT/step_timeline.json
