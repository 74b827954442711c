#!/bin/bash
# rocprofv3 kernel trace of one bench.py run on the GPU box:  tools/prof_run.sh <name> <bench.py args...>
# writes gpurun_out/r3prof/<name>/ (kernel_stats.csv, kernel_trace.csv), <name>.json (the bench line) and prints
# the ten longest kernels with their names cut short.
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3prof
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/$name -o p -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/r3prof/$name.json 2> gpurun_out/r3prof/$name.err
f=$(find gpurun_out/r3prof/$name -name "*kernel_stats.csv" | head -1)
echo "== $name"
python3 - "$f" gpurun_out/r3prof/$name.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print("%-62s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
print("ms/step %.4f  estep %.4f  mstep %.4f  bound %s frac %.3f  iteration_frac %.3f" % (d["ms_per_step"], r["estep_ms"], r["mstep_stats_ms"], r["bound"], r["frac"], r["iteration_frac"]))
PY
