#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/c5l -o p -- python3 tools/dropin_profile.py config5_large > gpurun_out/r3prof/c5l.txt 2> gpurun_out/r3prof/c5l.err || { tail -5 gpurun_out/r3prof/c5l.err; exit 1; }
tail -1 gpurun_out/r3prof/c5l.txt | cut -c1-250
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r3prof/c5l/p_kernel_stats.csv")))
for r in rows[:22]:
    print("%-70s calls %5s total %9.1f us avg %9.1f us %5s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
