#!/bin/bash
# threshold of the matrix-core nine-sum sweep: gpurun -- 'bash tools/run_mfma.sh'
set -o pipefail
LABEL=run_mfma
. tools/lib.sh
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for n in 26 24 22; do
for cfg in "--deg2-mfma-from 64" "--deg2-mfma-from 24" "--deg2-mfma-from 22" "--deg2-mfma-from 20" "--deg2-mfma-from 18"; do
  echo "== 4x2^$n $cfg"
  bench_json 120 $B --log-n $n $cfg | summ_deg2 || exit 1
done; done
