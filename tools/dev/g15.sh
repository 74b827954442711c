#!/bin/bash
set -o pipefail
for v in "" prio0 prio1 ""; do
  if [ -n "$v" ]; then export BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_$v.so; else unset BNPVI_LIB; fi
  timeout -k 10 200 python -u tools/kbench.py --n 999936 --iters 40 --only estep_stats,iteration --tag "config2 ${v:-shipped}" || exit 1
done 2>&1 | grep -v amdgpu.ids
