#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/g14_tests.log 2>&1
rc=$?; tail -4 gpurun_out/g14_tests.log; [ $rc -ne 0 ] && exit $rc
bash tools/prof_run.sh config4_4M --config config4 --steps 3 --warmup 1 || exit 1
bash tools/prof_run.sh config4_131k --config config4 --frames 131072 --steps 5 --warmup 2 || exit 1
