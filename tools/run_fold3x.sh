# parity + timing of the fused three-variable fold + cross-sum sweep: gpurun -- 'bash tools/run_fold3x.sh'
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "fold_fused_with_next" --timeout 120 --timeout-method thread || exit 1
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for n in 26 25 24; do
for cfg in "--deg2-fold3-fused 0" "--deg2-fold3-fused 3" "--deg2-fold3-fused 4" "--deg2-fold3-fused 5"; do
  echo "== 4x2^$n $cfg"
  timeout -k 10 120 $B --log-n $n $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prove_ms', round(d['ms_per_step'],4), {k:(v['launches'], round(v['total_ms']/5,4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()})
" || exit 1
done; done
