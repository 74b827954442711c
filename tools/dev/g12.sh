#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_knobs.so timeout -k 10 300 python -u tools/dev/stream2_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g12_check.log || exit 1
timeout -k 10 200 python -u tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only estep,estep_stats,iteration --tag "config4-131k stream2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g12.log || exit 1
BNPVI_ESTEP_STREAM2=off BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_knobs.so timeout -k 10 200 python -u tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only estep,estep_stats,iteration --tag "config4-131k old" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/g12.log || exit 1
