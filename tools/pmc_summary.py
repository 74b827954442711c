#!/usr/bin/env python
"""Per-kernel averages of the counters collected by tools/pmc_passes.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if "bnpvi" not in k:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = []
    for k in sorted(acc):
        short = k.split("(")[0][:70]
        lines.append(short)
        for c in sorted(acc[k]):
            v = acc[k][c]
            lines.append("    %-28s avg %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
    text = "\n".join(lines)
    open(os.path.join(root, "summary.txt"), "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1])
