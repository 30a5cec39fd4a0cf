#!/bin/bash
# GKR config 5 with the phases in their product-plus-linear form (k_fold_sum27<M, PWQ>) against the general four-table kernels, and
# the general prover's first pass as 27 sums: gpurun -- 'bash tools/run_pwq.sh'
set -o pipefail
LABEL=run_pwq
. tools/lib.sh
: > $OUT/$LABEL.txt
for rep in 1 2; do
for lin in 0 21 20 18; do
    echo "== gkr cfg5 --deg2-first-triple 0 $lin (rep $rep)" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload gkr --steps 5 --warmup 1 --no-cpu-baseline --deg2-first-triple 0 $lin | summ_gkr | tee -a $OUT/$LABEL.txt || exit 1
done; done
for n in 20 18 16 14; do
for k in 0 15 17 19 21; do
    echo "== 4x2^$n --deg2-first-triple $k 21" | tee -a $OUT/$LABEL.txt
    bench_json 120 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --no-live-traffic --steps 20 --warmup 3 --log-n $n --deg2-first-triple $k 21 | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
