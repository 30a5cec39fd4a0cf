#!/bin/bash
# finer threshold sweep of k_fold_sum27 around its break-even: gpurun -- 'bash tools/run_small_triple2.sh'
set -o pipefail
LABEL=run_small_triple2
. tools/lib.sh
: > $OUT/$LABEL.txt
for rep in 1 2; do
for k in 0 15 16 17; do
    echo "== gkr cfg5 --deg2-small-triple $k (rep $rep)" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload gkr --steps 5 --warmup 1 --no-cpu-baseline --deg2-small-triple $k | summ_gkr | tee -a $OUT/$LABEL.txt || exit 1
done; done
for n in 20 17; do
for k in 0 15 16 17; do
    echo "== 4x2^$n --deg2-small-triple $k" | tee -a $OUT/$LABEL.txt
    bench_json 120 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 20 --warmup 3 --log-n $n --deg2-small-triple $k | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
