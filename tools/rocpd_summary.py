#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite database (rocprofv3 --kernel-trace --stats) into the compact text
summary committed under profiles/ (the .db files are scratch under gpurun_out/)."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    unit = 1.0  # the top_kernels view reports microseconds
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({db_path.split('/')[-1]})\n")
        if note:
            f.write(f"# {note}\n")
        f.write(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}\n")
        for name, calls, total, avg, pct in rows:
            short = name.split("(")[0].replace("void ", "")
            if len(short) > 70:
                short = short[:67] + "..."
            f.write(f"{short:70s} {calls:6d} {total / unit:12.1f} {avg / unit:10.2f} {pct:6.2f}\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
