#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3prof
timeout -k 10 300 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread -k "remap or regime or random_traj or dynamic or kmeans or seeding or golden" > gpurun_out/t3.log 2>&1 || { tail -20 gpurun_out/t3.log; exit 1; }
tail -2 gpurun_out/t3.log
rm -rf gpurun_out/r3prof/c5
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/c5 -o p -- python3 tools/dropin_profile.py config5 > gpurun_out/r3prof/c5.txt 2> gpurun_out/r3prof/c5.err || { tail -5 gpurun_out/r3prof/c5.err; exit 1; }
tail -1 gpurun_out/r3prof/c5.txt | cut -c1-250
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r3prof/c5/p_kernel_stats.csv")))
for r in rows[:24]:
    print("%-70s calls %5s total %9.1f us avg %9.1f us %5s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
