#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3prof
rm -rf gpurun_out/r3prof/c5l
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/c5l -o p -- python3 tools/dropin_profile.py config5_large > gpurun_out/r3prof/c5l.txt 2> gpurun_out/r3prof/c5l.err || { tail -5 gpurun_out/r3prof/c5l.err; exit 1; }
tail -1 gpurun_out/r3prof/c5l.txt | cut -c1-250
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r3prof/c5l/p_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find call boundaries: the tool's latents generation starts each call with the gather kernel
names=[r["Kernel_Name"] for r in rows]
starts=[i for i,n in enumerate(names) if "vectorized_gather_kernel" in n]
print("calls found", len(starts))
import collections
for ci in range(1, min(4, len(starts)-1)):
    seg=rows[starts[ci]:starts[ci+1]]
    t0=int(seg[0]["Start_Timestamp"]); t1=int(seg[-1]["End_Timestamp"])
    agg=collections.defaultdict(lambda:[0,0.0])
    busy=0
    for r in seg:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        agg[r["Kernel_Name"][:64]][0]+=1; agg[r["Kernel_Name"][:64]][1]+=d; busy+=d
    print("call %d: span %.1f us, kernels busy %.1f us, %d launches" % (ci, (t1-t0)/1e3, busy, len(seg)))
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]:
        print("   %-64s x%3d %9.1f us" % (k, v[0], v[1]))
PY
