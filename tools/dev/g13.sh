#!/bin/bash
set -o pipefail
for v in deph40 deph80; do
  BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_$v.so timeout -k 10 200 python -u tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only estep --tag "config4-131k $v" || exit 1
done 2>&1 | grep -v amdgpu.ids
