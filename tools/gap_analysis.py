#!/usr/bin/env python3
"""Idle time between the kernels of a cfg1 step, from a rocprofv3 --kernel-trace csv (developer tool).
usage: gap_analysis.py KERNEL_TRACE.csv   -> per kernel: mean duration, mean idle gap BEFORE it (to the previous kernel's end)"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("scr::", "")))
rows.sort()
gap, dur, cnt = collections.Counter(), collections.Counter(), collections.Counter()
last_end = None
# steady state: the last 60 % of the launches
start = int(len(rows) * 0.4)
for s, e, k in rows[start:]:
    if last_end is not None:
        gap[k] += max(0, s - last_end)
    dur[k] += e - s
    cnt[k] += 1
    last_end = max(last_end or 0, e)
tot_gap = tot_dur = 0
for k in sorted(cnt, key=lambda k: -dur[k]):
    print(f"{k[:44]:44s} n={cnt[k]:4d} mean {dur[k] / cnt[k] / 1e3:8.2f} us   idle before it {gap[k] / cnt[k] / 1e3:7.2f} us")
steps = max(cnt.get("blend_backward_kernel", 1), 1)
print(f"per step: kernels {sum(dur.values()) / steps / 1e3:.1f} us + idle {sum(gap.values()) / steps / 1e3:.1f} us")
