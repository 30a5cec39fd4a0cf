#!/bin/bash
# tail of the prove: eight-round passes with one wave per output (zk_ctx_set_tail_fold) vs m = 4 passes
set -o pipefail
LABEL=tail_sweep
. tools/lib.sh
B="python bench.py --no-absorb --no-cpu-baseline --no-extra --steps 10 --warmup 3"
for n in 16 20 22 24 26 28; do
  for cfg in "0 13" "20 13" "20 10" "20 16" "16 13" "22 13"; do
    set -- $cfg
    echo -n "log_n $n tail_max_log $1 out_below $2: "
    bench_json 120 $B --log-n $n --tail-max-log $1 --tail-out-below $2 | summ_mle || exit 1
  done
done
