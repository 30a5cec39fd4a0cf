set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_gkr.py -x -q -k "fused or sum2x2 or gkr" --timeout 300 --timeout-method thread || exit 1
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for n in 22 20 18 16; do
for cfg in "--deg2-fused-below 0" "--deg2-fused-below 16" "--deg2-fused-below 18" "--deg2-fused-below 20" "--deg2-fused-below 22"; do
  echo "== 4x2^$n $cfg"
  timeout -k 10 120 $B --log-n $n $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prove_ms', round(d['ms_per_step'],4), {k:(v['launches'], round(v['total_ms']/5,4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()})
" || exit 1
done; done
for cfg in "--deg2-fused-below 0" "--deg2-fused-below 18" "--deg2-fused-below 20"; do
  echo "== gkr $cfg"
  timeout -k 10 200 python -u bench.py --workload gkr --steps 3 --warmup 1 $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('prove_ms','layers_ms_total','ms_per_round','circuit_evaluate_ms','output_absorb_ms')})" || exit 1
done
