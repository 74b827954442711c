set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/v8
timeout -k 10 300 python bench.py > gpurun_out/v8/bench.json 2> gpurun_out/v8/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v8/prof -o p -- python bench.py --no-cpu-baseline > gpurun_out/v8/prof_bench.json 2> gpurun_out/v8/prof.err
ls -R gpurun_out/v8/prof | head -20
