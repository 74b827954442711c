set -e
for c in "config1" "config2" "config3 --frames 2000128 --steps 30 --warmup 3" "config4 --frames 131072 --steps 10 --warmup 2"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --config $c | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['workload'][:60], '| ms/step %.4f  value %.4g  estep %.4f  mstep %.4f  iteration_frac %.3f' % (d['ms_per_step'], d['value'], r['estep_ms'], r['mstep_stats_ms'], r['iteration_frac']))"
done
