#!/bin/bash
# which kernel the dispatchers pick against the forced alternatives, over (D, L); needs the knobs variant:
#   make -C bnp_vae_amd/csrc variant NAME=knobs DEFS=-DBNPVI_DEV_KNOBS SRCS="estep_dp32 estep_dp64 estep_dp128 estep_dp256 mstep mstep_dma mstep_wide"
export BNPVI_LIB=bnp_vae_amd/variants/libbnpvi_knobs.so
N=${N:-1000000}
for d in 32 64 128 256; do
  for tree in "4 2" "6 2" "8 2" "9 2" "11 2" "12 2" "14 2" "16 2" "20 2" "8 3"; do
    set -- $tree
    timeout -k 10 100 python tools/kbench.py --n $N --d $d --bf $1 --levels $2 --only estep,estep_stats,leaf_stats --tag default || exit 1
    BNPVI_ESTEP_PATH=fused BNPVI_MSTEP_PATH=dma timeout -k 10 100 python tools/kbench.py --n $N --d $d --bf $1 --levels $2 --only estep,estep_stats,leaf_stats --tag fused/dma || exit 1
    BNPVI_ESTEP_PATH=rows BNPVI_MSTEP_PATH=reg timeout -k 10 100 python tools/kbench.py --n $N --d $d --bf $1 --levels $2 --only estep,estep_stats,leaf_stats --tag rows/reg || exit 1
  done
done
