#!/bin/bash
# k_fold_sum27's upper threshold on the mid-size and large degree-2 proves: gpurun -- 'bash tools/run_small_triple3.sh'
set -o pipefail
LABEL=run_small_triple3
. tools/lib.sh
: > $OUT/$LABEL.txt
for n in 21 22 23 24 26; do
for k in 0 18 19 20 21; do
    echo "== 4x2^$n --deg2-small-triple $k" | tee -a $OUT/$LABEL.txt
    bench_json 150 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 10 --warmup 2 --log-n $n --deg2-small-triple $k | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
