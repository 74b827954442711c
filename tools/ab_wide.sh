# development A/B of the wide M-step: rounds of workgroups per CU
set -e
for r in 1 2 4; do
  export BNPVI_WIDE_ROUNDS=$r
  timeout -k 10 120 python tools/kbench.py --n 2000128 --d 64 --bf 8 --levels 3 --iters 10 --only leaf_stats --tag "rounds=$r"
  timeout -k 10 120 python tools/kbench.py --n 1000000 --d 128 --bf 4 --levels 5 --iters 10 --only leaf_stats --tag "rounds=$r"
  timeout -k 10 120 python tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only leaf_stats --tag "rounds=$r"
done
