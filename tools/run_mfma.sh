set -o pipefail
B="python -u bench.py --no-extra --no-absorb --no-cpu-baseline --steps 10 --warmup 3 --wide-mfma-wg 1 --wide-mfma-depth 2"
for n in 21 22 23 24; do
for cfg in "--wide-mfma 64" "--wide-mfma 12 --wide-lds-below 12 --wide-batch 8" "--wide-mfma 12 --wide-lds-below 12 --wide-batch 8 --wide-mfma-wg 2"; do
  echo "== 2^$n $cfg"
  timeout -k 10 120 $B --log-n $n $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prove_ms', round(d['ms_per_step'],4), {k:(round(v['avg_ms'],4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()})
" || exit 1
done; done
