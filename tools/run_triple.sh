# threshold sweep of the three-rounds-per-pass schedule: gpurun -- 'bash tools/run_triple.sh'
set -o pipefail
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for n in 22 23 24; do
for cfg in "--deg2-triple-from 64" "--deg2-triple-from 21" "--deg2-triple-from 22" "--deg2-triple-from 23" "--deg2-triple-from 24"; do
  echo "== 4x2^$n $cfg"
  timeout -k 10 120 $B --log-n $n $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prove_ms', round(d['ms_per_step'],4), {k:(v['launches'], round(v['total_ms']/5,4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()})
" || exit 1
done; done
