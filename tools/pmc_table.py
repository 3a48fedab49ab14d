#!/usr/bin/env python3
"""Per-kernel averages of any rocprofv3 --pmc counters (one directory per pass), next to the kernel-trace durations.

usage: pmc_table.py OUT.txt DIR [DIR ...]
Every *counter_collection.csv under the directories is read; counters are averaged per launch of a kernel (name without
template arguments); durations come from the *kernel_trace.csv files of the same passes (average over all passes)."""
import collections
import csv
import glob
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("scr::", "").split("<")[0]


def main(out, dirs):
    val = collections.defaultdict(collections.Counter)
    cnt = collections.defaultdict(collections.Counter)
    dur, nd = collections.Counter(), collections.Counter()
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                val[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
        for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
                nd[k] += 1
    counters = sorted({c for k in val for c in val[k]})
    with open(out, "w") as f:
        f.write("# per launch; us = kernel-trace duration under the counter passes\n")
        f.write(f"{'kernel':34s} {'n':>5s} {'us':>9s} " + " ".join(f"{c[-22:]:>22s}" for c in counters) + "\n")
        for k in sorted(val, key=lambda k: -dur[k]):
            if not nd[k]:
                continue
            f.write(f"{k[:34]:34s} {nd[k]:5d} {dur[k] / nd[k]:9.1f} " +
                    " ".join(f"{(val[k][c] / cnt[k][c] if cnt[k][c] else float('nan')):22.4g}" for c in counters) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
