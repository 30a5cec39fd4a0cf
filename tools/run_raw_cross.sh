#!/bin/bash
# three-round sweeps: raw integer digits + one product per digit in the second stage (1) against field elements per workgroup (0)
set -o pipefail
LABEL=run_raw_cross
. tools/lib.sh
: > $OUT/$LABEL.txt
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "three_rounds or fold_fused or cross" 2>&1 | tail -3 | tee -a $OUT/$LABEL.txt || exit 1
for n in 26 23; do
for r in 0 1 0 1 0 1; do
    echo "== 4x2^$n --deg2-raw-cross $r" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload sum2x2 --log-n $n --no-extra --no-cpu-baseline --no-absorb --no-live-traffic --steps 10 --warmup 2 --deg2-raw-cross $r | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done; done
