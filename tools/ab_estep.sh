#!/bin/bash
# Development A/B of the E-step only: default library and every bnp_vae_amd/variants/libbnpvi_*.so, on the
# bench mixture and on tight far-from-origin clusters (--far 32: the exact-distance path on every frame).
# usage: tools/ab_estep.sh [extra kbench args]
set -e
mkdir -p gpurun_out
out=gpurun_out/ab_estep.log
: > $out
for lib in default bnp_vae_amd/variants/libbnpvi_*.so; do
  if [ "$lib" = default ]; then unset BNPVI_LIB; else export BNPVI_LIB=$PWD/$lib; fi
  echo "== $lib" >> $out
  timeout -k 10 120 python tools/kbench.py --only estep,estep_stats --iters 30 "$@" >> $out 2>&1
  timeout -k 10 120 python tools/kbench.py --only estep,estep_stats --iters 10 --far 32 --tag far32 "$@" >> $out 2>&1
done
cat $out
