#!/bin/bash
# which passes take the eight-round wide form: gpurun -- 'bash tools/wide_sweep.sh'
set -o pipefail
LABEL=wide_sweep
. tools/lib.sh
B="python bench.py --no-absorb --no-cpu-baseline --no-extra --steps 5 --warmup 2"
for cfg in "28 -1 17" "28 0 17" "28 7 17" "26 -1 17" "26 0 17" "25 8 17" "25 0 17" "25 7 17" "24 8 16" "24 7 17" "24 6 17" "24 0 17" "22 8 14" "22 6 16" "22 5 17" "22 0 17" "20 6 14" "20 0 17"; do
  set -- $cfg
  echo -n "log_n $1 wide $2 min_log_out $3: "
  bench_json 120 $B --log-n $1 --wide-batch $2 --wide-min-log-out $3 | summ_mle || exit 1
done
