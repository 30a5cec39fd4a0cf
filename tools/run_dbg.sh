set -o pipefail
B="python -u bench.py --workload sum2x2 --log-n 26 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for d in 0 1 2 3; do
  echo "== dbg $d"
  ZK_CROSS_DBG=$d timeout -k 10 120 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prove_ms', round(d['ms_per_step'],4), {k:(v['launches'], round(v['total_ms']/5,4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()})
" || exit 1
done
