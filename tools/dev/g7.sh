#!/bin/bash
# round-3 check: tightened sigma / near-tie tests, S0 A/B (shipped vs unpinned build), fp64 cross-check debug build
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/g7_tests.log 2>&1
rc=$?; tail -5 gpurun_out/g7_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python -u tools/dev/s0_ab.py config2 999936 10000 > gpurun_out/g7_s0_shipped.log 2>&1 || exit 1
cat gpurun_out/g7_s0_shipped.log
BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_nopin.so timeout -k 10 120 python -u tools/dev/s0_ab.py config2 999936 10000 > gpurun_out/g7_s0_nopin.log 2>&1 || exit 1
cat gpurun_out/g7_s0_nopin.log
BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_s0chk.so timeout -k 10 300 python -u -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "s0_is_exact or phases or full_size or mstep" > gpurun_out/g7_s0chk.log 2>&1
rc=$?; tail -3 gpurun_out/g7_s0chk.log; echo "cross-check failures printed: $(grep -c 'cross-check FAILED' gpurun_out/g7_s0chk.log)"
exit $rc
