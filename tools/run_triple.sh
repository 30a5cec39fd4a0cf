#!/bin/bash
# threshold sweep of the three-rounds-per-pass schedule: gpurun -- 'bash tools/run_triple.sh'
set -o pipefail
LABEL=run_triple
. tools/lib.sh
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for n in 22 23 24; do
for cfg in "--deg2-triple-from 64" "--deg2-triple-from 21" "--deg2-triple-from 22" "--deg2-triple-from 23" "--deg2-triple-from 24"; do
  echo "== 4x2^$n $cfg"
  bench_json 120 $B --log-n $n $cfg | summ_deg2 || exit 1
done; done
