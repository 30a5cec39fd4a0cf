#!/bin/bash
# with the raw-digit second stage the three-round sweeps have a smaller fixed cost: is 2^20 (GKR phases, 4 x 2^20..2^22 proves) now theirs?
set -o pipefail
LABEL=run_triple_from3
. tools/lib.sh
: > $OUT/$LABEL.txt
for k in 21 20 21 20; do
    echo "== gkr cfg5 --deg2-triple-from $k" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload gkr --steps 3 --warmup 1 --no-cpu-baseline --deg2-triple-from $k | summ_gkr | tee -a $OUT/$LABEL.txt || exit 1
done
for n in 20 21 22; do
for k in 21 20 19 21 20 19; do
    echo "== 4x2^$n --deg2-triple-from $k" | tee -a $OUT/$LABEL.txt
    bench_json 120 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --no-live-traffic --steps 20 --warmup 3 --log-n $n --deg2-triple-from $k | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
