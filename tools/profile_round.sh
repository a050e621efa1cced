#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench, then two PMC passes (HBM read / write bytes).
# usage (from the repo root, under gpurun): bash tools/profile_round.sh <tag>
# Writes gpurun_out/prof_<tag>/{stats,fetch,write}/... and gpurun_out/prof_<tag>/summary.json
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
export PHONLP_BENCH_EXTRAS=0   # only the timed steps + the kernels_ms pass: per-kernel averages are those of the step
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $B --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 $B --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 $B --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/mfma -o mfma --output-format csv -- python3 $B --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/mfma.err || exit 1
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
def pmc(name):
    f = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = pmc("fetch"), pmc("write")
per = []
for k in sorted(set(fetch) | set(write)):
    if "phonlp" not in k: continue
    fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
    per.append({"kernel": k.replace("void ", ""), "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024})
# matrix-pipe utilisation per kernel: busy cycles summed over the 1024 SIMDs / (1024 * elapsed GPU cycles)
import csv as _csv
f = glob.glob(f"{out}/mfma/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in _csv.DictReader(open(f)):
    acc[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
mf = []
for k, c in sorted(acc.items()):
    if "phonlp" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in c: continue
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
    if busy <= 0: continue
    mf.append({"kernel": k, "mfma_busy_cycles_per_launch": busy, "gpu_cycles_per_launch": gui,
               "mfma_pipe_util": busy / (1024.0 * gui / 8.0),
               "mops_bf16": sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0])) / max(len(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0])), 1),
               "mops_f16": sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", [0])) / max(len(c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", [0])), 1),
               "mops_f32": sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", [0])) / max(len(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", [0])), 1)})
stats = glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True)
summary = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over python3 bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline; per-launch mean; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md: FETCH_SIZE "
                   "reports half of a wide coalesced read on gfx950, WRITE_SIZE is exact)", "per_kernel": per,
           "_mfma_how": "separate pass --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...; util = busy / (1024 SIMDs * GRBM_GUI_ACTIVE / 8): "
                        "GRBM_GUI_ACTIVE comes summed over the 8 XCCs; counter collection serialises the launches, so every kernel is alone "
                        "on the chip in this pass",
           "mfma": mf}
json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)
print(open(stats[0]).read()[:3000])
for m in mf: print(m["kernel"][:70], "mfma util %.3f" % m["mfma_pipe_util"])
for p in per: print(p["kernel"][:70], round(p["hbm_bytes_per_launch_corrected"] / 1e6, 1), "MB")
PY
