set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3prof
for spec in "c3_16M:--config config3 --steps 6 --warmup 2" "c3_2M:--config config3 --frames 2000128 --steps 30 --warmup 3" "c2:--config config2 --no-config3 --steps 100 --warmup 10"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/$name -o p -- python3 bench.py --no-cpu-baseline $args > gpurun_out/r3prof/$name.json 2> gpurun_out/r3prof/$name.err
  f=$(find gpurun_out/r3prof/$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; head -12 "$f"
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3prof/$name.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.4f estep %.4f mstep %.4f iter_frac %.3f' % (d['ms_per_step'], r['estep_ms'], r['mstep_stats_ms'], r['iteration_frac']))"
done
