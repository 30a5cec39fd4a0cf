#!/bin/bash
# three rounds per pass (matrix-core cross sums) down to 2^k-entry tables, with the small-table 27-sum passes below: k = 21 (default) vs 20, 19
set -o pipefail
LABEL=run_triple_from2
. tools/lib.sh
: > $OUT/$LABEL.txt
for n in 26 24 23 22 21; do
for k in 21 20 19 21 20; do
    echo "== 4x2^$n --deg2-triple-from $k" | tee -a $OUT/$LABEL.txt
    bench_json 150 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --no-live-traffic --steps 10 --warmup 2 --log-n $n --deg2-triple-from $k | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
