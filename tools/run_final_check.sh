#!/bin/bash
# the three workloads at the library defaults, a few repetitions each: gpurun -- 'bash tools/run_final_check.sh'
set -o pipefail
LABEL=run_final_check
. tools/lib.sh
: > $OUT/$LABEL.txt
for rep in 1 2 3; do
    echo "== gkr cfg5 (rep $rep)" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload gkr --steps 5 --warmup 1 --no-cpu-baseline | summ_gkr | tee -a $OUT/$LABEL.txt || exit 1
done
for n in 26 23 20; do
for rep in 1 2 3; do
    echo "== 4x2^$n (rep $rep)" | tee -a $OUT/$LABEL.txt
    bench_json 150 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --no-live-traffic --steps 10 --warmup 2 --log-n $n | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
