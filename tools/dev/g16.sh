#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_knobs.so timeout -k 10 300 python -u tools/dev/wide_f16_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g16_check.log || exit 1
for m in on off; do
BNPVI_WIDE_F16=$m BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_knobs.so timeout -k 10 200 python -u tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only leaf_stats,iteration --tag "config4-131k wide f16=$m" 2>&1 | grep -v amdgpu.ids || exit 1
done
