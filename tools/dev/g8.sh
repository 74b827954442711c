#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in "" NODIGAMMA NOPRIORPASS NOBLOCK0; do
  if [ -n "$v" ]; then export BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_$v.so; fi
  timeout -k 10 120 python -u tools/kbench.py --n 999936 --iters 30 --only params_prior,gamma_prior,node_params,prior_table,gamma_update --tag "${v:-shipped}" || exit 1
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g8.log
