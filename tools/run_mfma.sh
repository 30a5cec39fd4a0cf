set -o pipefail
B="python -u bench.py --no-extra --no-absorb --no-cpu-baseline --steps 5 --warmup 2"
for cfg in "--wide-mfma 17 --wide-mfma-wg 1 --wide-mfma-depth 2" "--wide-mfma 17 --wide-mfma-wg 1 --wide-mfma-depth 3" "--wide-mfma 17 --wide-mfma-wg 1 --wide-mfma-depth 4" "--wide-mfma 17 --wide-mfma-wg 2 --wide-mfma-depth 2" "--wide-mfma 17 --wide-mfma-wg 2 --wide-mfma-depth 3"; do
  echo "== $cfg"
  timeout -k 10 120 $B $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prove_ms', round(d['ms_per_step'],4), {k:(round(v['avg_ms'],4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()})
" || exit 1
done
