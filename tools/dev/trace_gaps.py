#!/usr/bin/env python
"""Per-kernel durations and inter-kernel gaps of the steady-state steps in a rocprofv3 --kernel-trace database.
usage: tools/dev/trace_gaps.py <results.db> [marker-kernel-substring] [steps-from-the-end]"""
import sqlite3, sys, collections
db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "leaf_stats"
last = int(sys.argv[3]) if len(sys.argv) > 3 else 100
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = list(c.execute("select name, start, end from kernels order by start"))
# steps start at each marker kernel
idx = [i for i, r in enumerate(rows) if marker in r[0]]
idx = idx[-last - 1:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); step = []
for a, b in zip(idx[:-1], idx[1:]):
    step.append(rows[b][1] - rows[a][1])
    for i in range(a, b):
        n = rows[i][0].split("(")[0][-60:]
        dur[(i - a, n)].append(rows[i][2] - rows[i][1])
        gap[(i - a, n)].append(rows[i + 1][1] - rows[i][2])
med = lambda x: sorted(x)[len(x) // 2]
print("steps %d  median step %.1f us" % (len(step), med(step) / 1e3))
tot_d = tot_g = 0
for k in sorted(dur):
    d, g = med(dur[k]) / 1e3, med(gap[k]) / 1e3
    tot_d += d; tot_g += g
    print("  %2d %-62s dur %7.1f us   gap after %6.1f us   (n=%d)" % (k[0], k[1], d, g, len(dur[k])))
print("  sum of durations %.1f us, sum of gaps %.1f us" % (tot_d, tot_g))
