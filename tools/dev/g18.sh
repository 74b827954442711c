#!/bin/bash
# round-3 closing measurements: the driver's command, then kernel traces of configs 2, 3 (2M and 16M)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3final
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3final/bench_driver_form.json 2> gpurun_out/r3final/bench.err || { tail -5 gpurun_out/r3final/bench.err; exit 1; }
tail -c 3000 gpurun_out/r3final/bench_driver_form.json
bash tools/prof_run.sh config2 --steps 50 --warmup 10 --no-config3 || exit 1
bash tools/prof_run.sh config3 --config config3 --frames 2000128 --steps 10 --warmup 3 --no-config3 || exit 1
bash tools/prof_run.sh config3_16M --config config3 --steps 5 --warmup 2 --no-config3 || exit 1
