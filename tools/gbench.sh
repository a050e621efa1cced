#!/bin/bash
# On the GPU box: one bench.py run into gpurun_out/r3/<name>.json and a short digest of it.  usage: bash tools/gbench.sh <name> [bench args]
N=$1; shift
mkdir -p gpurun_out/r3
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r3/$N.json 2> gpurun_out/r3/$N.err || { tail -20 gpurun_out/r3/$N.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r3/$N.json"))
print("step", round(d["ms_per_step"], 4), "stageA", round(d["roofline"]["mean_ms"], 4), "frac", round(d["roofline"]["frac"], 4),
      "streaming", round(d.get("streaming", {}).get("ms_per_step", 0), 4), "graph", round(d.get("graph_replay", {}).get("ms_per_step", 0), 4))
print(d["kernels_ms"])
for k, v in d.get("other_configs", {}).items(): print(k, v["ms_per_step"], v.get("graph_replay_ms", ""), v.get("graph_equals_eager", ""), v.get("kernels_ms", ""))
PY
