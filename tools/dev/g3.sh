set -e
mkdir -p gpurun_out
out=gpurun_out/ab_r1.log
: > $out
for lib in default bnp_vae_amd/variants/libbnpvi_r1*.so; do
  if [ "$lib" = default ]; then unset BNPVI_LIB; else export BNPVI_LIB=$PWD/$lib; fi
  timeout -k 10 120 python tools/kbench.py --n 2000128 --bf 8 --levels 3 --iters 20 --only estep,estep_stats >> $out 2>&1
done
cat $out
