#!/usr/bin/env python3
"""Two whole vector steps out of the middle of a rocprofv3 --kernel-trace CSV, anchored on a kernel name: start, duration,
end, gap to the latest end so far, queue.  usage: trace_steps.py <kernel_trace.csv> [anchor substring] [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "rainbow_act_kernel"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[len(idx) - 40], idx[len(idx) - 40 + steps]
t0 = int(rows[a]["Start_Timestamp"])
pe = t0
for r in rows[a - 8:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
    print(f"{(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f} end {(e - t0) / 1e3:8.1f} gap {(s - pe) / 1e3:6.1f} q{r['Queue_Id']:>2} {name}")
    pe = max(pe, e)
