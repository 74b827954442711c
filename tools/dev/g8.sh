#!/bin/bash
set -o pipefail
for v in "" B0_NOLEVELS B0_NOPACKALL; do
  if [ -n "$v" ]; then export BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_$v.so; fi
  timeout -k 10 120 python -u tools/kbench.py --n 999936 --iters 50 --only params_prior --tag "${v:-shipped}" || exit 1
done 2>&1 | grep -v amdgpu.ids
