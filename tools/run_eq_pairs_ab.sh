#!/bin/bash
# GKR: eq half tables from host-built two-variable tables (k_eq_small_pairs, new) against one factor per variable (k_eq_small, old); libraries swapped on one box
set -o pipefail
LABEL=run_eq_pairs_ab
. tools/lib.sh
: > $OUT/$LABEL.txt
cp tools/exp/libzkmi_new.so zk_amd/libzkmi.so
timeout -k 10 900 python -u -m pytest tests/test_gpu_gkr.py tests/test_gpu_cpp_host.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 | tee -a $OUT/$LABEL.txt || exit 1
for v in old new old new old new; do
    cp tools/exp/libzkmi_$v.so zk_amd/libzkmi.so
    echo "== $v gkr cfg5" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload gkr --steps 3 --warmup 1 --no-cpu-baseline | summ_gkr | tee -a $OUT/$LABEL.txt || exit 1
done
cp tools/exp/libzkmi_new.so zk_amd/libzkmi.so
