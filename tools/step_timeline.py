"""One bench step as a timeline, from a rocprofv3 --kernel-trace CSV.

    python3 tools/step_timeline.py <dir with *kernel_trace.csv> [out.json]

Picks the last complete step (the kernels between the last two labels-at-heads launches) and prints every kernel's
start / end in microseconds from the end of the previous step's labels-at-heads kernel, with its queue."""
import csv
import glob
import json
import sys


def main():
    src = sys.argv[1]
    f = glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "label_at_heads" in r[3] or "label_reduce" in r[3]]
    if len(marks) < 3:
        raise SystemExit("need at least three steps in the trace")
    a, b = marks[-3], marks[-2]  # the last step may be cut by the end of the timed region
    t0 = rows[a][1]
    queues = {}
    ks = []
    for s, e, q, n in rows[a + 1:b + 1]:
        queues.setdefault(q, str(len(queues) + 1))
        ks.append({"start_us": round((s - t0) / 1e3, 3), "end_us": round((e - t0) / 1e3, 3), "queue": queues[q], "kernel": n})
    out = {"_what": "kernel trace (rocprofv3 --kernel-trace) of ONE bench step, microseconds from the end of the previous "
                    "step's labels-at-heads kernel", "kernels": ks}
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k in ks:
        print(f"{k['start_us']:9.1f} {k['end_us']:9.1f} {k['end_us'] - k['start_us']:8.1f}  q{k['queue']}  {k['kernel'][:80]}")


if __name__ == "__main__":
    main()
