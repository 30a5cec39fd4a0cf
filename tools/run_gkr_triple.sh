#!/bin/bash
# three rounds per pass on the SMALL tables too (GKR config 5: four tables of 2^20 entries per phase; the 4 x 2^20 degree-2 prove)
#   gpurun -- 'bash tools/run_gkr_triple.sh'   -> gpurun_out/run_gkr_triple.txt (+ .err)
set -o pipefail
LABEL=run_gkr_triple
. tools/lib.sh
: > $OUT/$LABEL.txt
for k in 21 20 18 16 14 12 10; do
    echo "== gkr cfg5 --deg2-triple-from $k" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload gkr --steps 3 --warmup 1 --no-cpu-baseline --deg2-triple-from $k | summ_gkr | tee -a $OUT/$LABEL.txt || exit 1
done
for n in 20 18; do
for k in 21 20 18 16 14 12 10; do
    echo "== 4x2^$n --deg2-triple-from $k" | tee -a $OUT/$LABEL.txt
    bench_json 120 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 20 --warmup 3 --log-n $n --deg2-triple-from $k | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
