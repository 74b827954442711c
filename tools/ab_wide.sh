# development timing of the streamed E-step (config-4 shape and a mid-size streamed shape)
set -e
timeout -k 10 120 python tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only estep,estep_stats --tag "cfg4"
timeout -k 10 120 python tools/kbench.py --n 500000 --d 256 --bf 7 --levels 3 --iters 5 --only estep,estep_stats --tag "L343"
timeout -k 10 120 python tools/kbench.py --n 500000 --d 128 --bf 4 --levels 5 --iters 5 --only estep,estep_stats --tag "L1024-D128"
