#!/usr/bin/env python3
"""One steady-state step of an anchor configuration (bench.py --config cfg2..4) as a list of the idle gaps on the GPU:
kernel-trace csv of rocprofv3; a step = from one filter_kernel launch to the next.  Developer tool.
usage: step_gaps_cfg.py KERNEL_TRACE.csv [min_gap_us]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("scr::", "")[:48]))
rows.sort()
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
idx = [i for i, r in enumerate(rows) if r[2].startswith("filter_kernel")]
for which in (len(idx) - 3, len(idx) - 2):
    a, b = idx[which], idx[which + 1]
    t0, prev_end, busy, idle = rows[a][0], None, 0, 0
    print(f"--- step from launch {a} to {b}: {(rows[b][0] - t0) / 1e3:.1f} us")
    for s, e, k in rows[a:b]:
        if prev_end is not None:
            g = max(0, s - prev_end) / 1e3
            idle += g
            if g >= min_gap:
                print(f"  idle {g:8.1f} us before {k:48s} at {(s - t0) / 1e3:9.1f} us")
        busy += (e - max(s, prev_end or 0)) / 1e3 if (prev_end is None or e > prev_end) else 0
        prev_end = max(prev_end or 0, e)
    print(f"  kernels busy {busy:.1f} us, idle {idle:.1f} us ({len(rows[a:b])} launches)")
