#!/usr/bin/env python
"""rocprofv3 results database (--kernel-trace) -> per-kernel stats CSV in the layout of
rocprofv3's own kernel_stats.csv (used when the run wrote only the .db)."""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc"))
    tot = float(sum(r[2] for r in rows)) or 1.0
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, n, total, avg, mn, mx in rows:
            w.writerow([name, n, total, round(avg, 3), round(100.0 * total / tot, 2), mn, mx])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
