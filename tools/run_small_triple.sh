#!/bin/bash
# three rounds per pass on small tables (k_fold_sum27): threshold sweep on GKR config 5 and the small degree-2 proves
#   gpurun -- 'bash tools/run_small_triple.sh'   -> gpurun_out/run_small_triple.txt (+ .err)
set -o pipefail
LABEL=run_small_triple
. tools/lib.sh
: > $OUT/$LABEL.txt
for k in 0 10 12 14 16 18 19 20 21; do
    echo "== gkr cfg5 --deg2-small-triple $k" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload gkr --steps 3 --warmup 1 --no-cpu-baseline --deg2-small-triple $k | summ_gkr | tee -a $OUT/$LABEL.txt || exit 1
done
for n in 20 18 16; do
for k in 0 12 16 19 20 21; do
    echo "== 4x2^$n --deg2-small-triple $k" | tee -a $OUT/$LABEL.txt
    bench_json 120 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 20 --warmup 3 --log-n $n --deg2-small-triple $k | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
