#!/bin/bash
set -o pipefail
for v in "" nopk; do
  if [ -n "$v" ]; then export BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_$v.so; fi
  timeout -k 10 200 python -u tools/kbench.py --n 999936 --iters 30 --only estep,estep_stats --tag "config2 ${v:-shipped}" || exit 1
  timeout -k 10 200 python -u tools/kbench.py --n 2000128 --levels 3 --iters 10 --only estep_stats --tag "config3 ${v:-shipped}" || exit 1
done 2>&1 | grep -v amdgpu.ids
