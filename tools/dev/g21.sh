#!/bin/bash
set -o pipefail
export BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_knobs.so
for shape in "128 4 5" "128 4 6" "256 4 4" "256 8 3" "256 6 4" "200 5 5"; do
  set -- $shape
  for m in off on; do
    BNPVI_WIDE_F16=$m timeout -k 10 200 python -u tools/kbench.py --n 500000 --d $1 --bf $2 --levels $3 --iters 5 --only leaf_stats --tag "D=$1 f16=$m" 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
