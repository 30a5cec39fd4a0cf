#!/bin/bash
# parity + timing of the fused three-variable fold + cross-sum sweep: gpurun -- 'bash tools/run_fold3x.sh'
set -o pipefail
LABEL=run_fold3x
. tools/lib.sh
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "fold_fused_with_next" --timeout 120 --timeout-method thread 2>> $OUT/$LABEL.err || exit 1
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for n in 26 25 24; do
for cfg in "--deg2-fold3-fused 0" "--deg2-fold3-fused 3" "--deg2-fold3-fused 4" "--deg2-fold3-fused 5" "--deg2-fold3-fused 8"; do
  echo "== 4x2^$n $cfg"
  bench_json 120 $B --log-n $n $cfg | summ_deg2 || exit 1
done; done
