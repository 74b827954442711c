set -e
mkdir -p gpurun_out
: > gpurun_out/kb_shapes.log
for args in "--d 64 --bf 128 --levels 1" "--d 64 --bf 16 --levels 2" "--d 64 --bf 320 --levels 1" "--d 32 --bf 8 --levels 3" "--d 128 --bf 16 --levels 2" "--d 256 --bf 128 --levels 1"; do
  timeout -k 10 200 python -u tools/kbench.py --n 2000128 $args --iters 10 --only leaf_stats,estep,estep_stats,estep_stats_s2 >> gpurun_out/kb_shapes.log 2>&1
done
cat gpurun_out/kb_shapes.log
