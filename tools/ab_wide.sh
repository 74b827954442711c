# development A/B of the wide M-step (variants built by hand into bnp_vae_amd/variants/)
set -e
for v in $AB_VARIANTS ""; do
  if [ -n "$v" ]; then export BNPVI_LIB=$PWD/bnp_vae_amd/variants/libbnpvi_$v.so; else unset BNPVI_LIB; fi
  timeout -k 10 120 python tools/kbench.py --n 131072 --d 256 --bf 6 --levels 5 --iters 5 --only leaf_stats --tag "v=$v"
  timeout -k 10 120 python tools/kbench.py --n 262144 --d 128 --bf 4 --levels 5 --iters 5 --only leaf_stats --tag "v=$v"
done
