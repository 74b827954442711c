set -e
mkdir -p gpurun_out
out=gpurun_out/ab_f32.log
: > $out
for lib in default bnp_vae_amd/variants/libbnpvi_f16split.so default bnp_vae_amd/variants/libbnpvi_f16split.so; do
  if [ "$lib" = default ]; then unset BNPVI_LIB; else export BNPVI_LIB=$PWD/$lib; fi
  timeout -k 10 120 python tools/kbench.py --iters 30 --only estep,estep_stats,iteration >> $out 2>&1
done
unset BNPVI_LIB
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread 2>&1 | tail -3 >> $out
cat $out
