#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in A_NOPRIOR A_NOEPI A_NOSTORE ALL; do
  export BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_a256_$v.so
  timeout -k 10 200 python -u tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only estep --tag "config4-131k $v" || exit 1
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g11.log
