# development A/B of the M-step data pass: per-wave DMA ring (BNPVI_MSTEP_PATH=dma) vs workgroup-wide tiles
set -e
for path in dma ""; do
  export BNPVI_MSTEP_PATH=$path
  timeout -k 10 120 python tools/kbench.py --n 1000000 --d 256 --bf 16 --levels 2 --iters 10 --only leaf_stats --tag "path=$path"
  timeout -k 10 120 python tools/kbench.py --n 1000000 --d 256 --bf 7 --levels 3 --iters 10 --only leaf_stats --tag "path=$path"
done
