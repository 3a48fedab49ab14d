#!/usr/bin/env python3
"""One steady-state cfg1 step as a timeline (developer tool): kernel, start and end in us relative to the step's first kernel.
usage: step_timeline.py KERNEL_TRACE.csv"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("scr::", "")[:40]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("zero_list")]
a = idx[len(idx) // 2]
b = idx[len(idx) // 2 + 1]
t0 = rows[a][0]
prev_end = None
for s, e, k in rows[a:b + 1]:
    gap = "" if prev_end is None else f"  (idle {max(0, s - prev_end) / 1e3:6.2f} us)"
    print(f"{k:40s} {(s - t0) / 1e3:9.2f} -> {(e - t0) / 1e3:9.2f} us{gap}")
    prev_end = max(prev_end or 0, e)
