"""Prints one mid-run step of a rocprofv3 --kernel-trace CSV as a timeline: python3 tools/print_step.py <dir> [step]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "label_reduce" in r["Kernel_Name"]]
t0 = int(rows[idx[k]]["End_Timestamp"])
for r in rows[idx[k] + 1:idx[k + 1] + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f %7.1f  q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-45:]))
