#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in "" st64; do
  if [ -n "$v" ]; then export BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_$v.so; fi
  timeout -k 10 200 python -u tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only estep,estep_stats,leaf_stats --tag "config4-131k ${v:-shipped}" || exit 1
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g10.log
unset BNPVI_LIB
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "oracle or full_size or streamed or config4 or golden" > gpurun_out/g10_tests.log 2>&1
rc=$?; tail -4 gpurun_out/g10_tests.log; exit $rc
