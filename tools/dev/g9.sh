#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/refine_fraction.py > gpurun_out/g9_refine.txt 2>&1 || { tail -20 gpurun_out/g9_refine.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/g9_refine.txt
for far in 0 32; do
  timeout -k 10 120 python -u tools/kbench.py --n 999936 --iters 20 --far $far --only estep,estep_stats --tag "config2 far=$far" || exit 1
  timeout -k 10 120 python -u tools/kbench.py --n 2000128 --levels 3 --iters 10 --far $far --only estep,estep_stats --tag "config3 far=$far" || exit 1
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g9_kbench.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/g9_tests.log 2>&1
rc=$?; tail -5 gpurun_out/g9_tests.log; exit $rc
