#!/bin/bash
# threshold of the fused fold + nine-sum kernel of the small degree-2 passes: gpurun -- 'bash tools/run_fused.sh'
set -o pipefail
LABEL=run_fused
. tools/lib.sh
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_gkr.py -x -q -k "fused or sum2x2 or gkr" --timeout 300 --timeout-method thread 2>> $OUT/$LABEL.err || exit 1
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for n in 22 20 18 16; do
for cfg in "--deg2-fused-below 0" "--deg2-fused-below 16" "--deg2-fused-below 18" "--deg2-fused-below 20" "--deg2-fused-below 22"; do
  echo "== 4x2^$n $cfg"
  bench_json 120 $B --log-n $n $cfg | summ_deg2 || exit 1
done; done
for cfg in "--deg2-fused-below 0" "--deg2-fused-below 18" "--deg2-fused-below 20"; do
  echo "== gkr $cfg"
  bench_json 200 python -u bench.py --workload gkr --steps 3 --warmup 1 --no-cpu-baseline $cfg | summ_gkr || exit 1
done
