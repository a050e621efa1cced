#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh run (gpurun_out/prof_<tag>/) into profiles/ (tracked) and refreshes
profiles/traffic.json, which bench.py quotes as roofline.traffic.   usage: python tools/save_profile.py <tag>"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_bench.csv"))
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{tag}_pmc_traffic_mfma.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
summ = json.load(open(os.path.join(src, "summary.json")))
tpath = os.path.join(dst, "traffic.json")
t = json.load(open(tpath))
bench = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
fused = "labels-at-heads epilogue" in bench["roofline"]["kernel"]
for k in summ["per_kernel"]:
    if "linear_f16x3_wide16_heads_kernel" in k["kernel"] or "linear_f16x3_wide_kernel<1>" in k["kernel"] or "linear_f16x3_dma_kernel<1, false" in k["kernel"]:
        key = "linear:label_stage_a@engine2" + ("+fused" if fused else "")
        t[key] = k["hbm_bytes_per_launch_corrected"]
        t[key + "_detail"] = {"FETCH_SIZE_KB": k["FETCH_SIZE_KB_per_launch"], "WRITE_SIZE_KB": k["WRITE_SIZE_KB_per_launch"]}
        print(key, round(t[key] / 1e6, 1), "MB per launch", t[key + "_detail"])
t["source"] = f"profiles/{tag}_pmc_traffic_mfma.json"
t[f"per_kernel_{tag}"] = summ["per_kernel"]
json.dump(t, open(tpath, "w"), indent=1)
