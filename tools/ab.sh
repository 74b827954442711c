#!/bin/bash
# Development A/B: the default library and every bnp_vae_amd/variants/libbnpvi_*.so through the
# config-2 iteration (tools/kbench.py per-call medians + bench.py), same box, back to back.
# usage: tools/ab.sh [extra kbench args]
set -e
mkdir -p gpurun_out
out=gpurun_out/ab.log
: > $out
for lib in default bnp_vae_amd/variants/libbnpvi_*.so; do
  if [ "$lib" = default ]; then unset BNPVI_LIB; else export BNPVI_LIB=$PWD/$lib; fi
  echo "== $lib" >> $out
  timeout -k 10 120 python tools/kbench.py --only leaf_stats,estep,estep_stats,iteration --iters 30 "$@" >> $out 2>&1
  timeout -k 10 120 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-config3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('bench ms/step %.4f  estep %.4f mstep %.4f iter_frac %.3f' % (d['ms_per_step'], r['estep_ms'], r['mstep_stats_ms'], r['iteration_frac']))
" >> $out
done
cat $out
