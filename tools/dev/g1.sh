set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/dev/rows1_check.py > gpurun_out/rows1_check.log 2>&1 || { tail -30 gpurun_out/rows1_check.log; exit 1; }
tail -20 gpurun_out/rows1_check.log
timeout -k 10 200 python -u tools/kbench.py --n 2000128 --bf 8 --levels 3 --iters 10 --only leaf_stats,estep,estep_stats,estep_stats_s2,gated_s2,iteration > gpurun_out/kb3.log 2>&1
cat gpurun_out/kb3.log
