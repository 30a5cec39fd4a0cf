#!/bin/bash
# GKR phases' first sweep: k_sum9_lazy<PWQ> (linear term as additions, ones never read) against the general sweep; libraries swapped on one box
set -o pipefail
LABEL=run_pwq_sum9_ab
. tools/lib.sh
: > $OUT/$LABEL.txt
cp tools/exp/libzkmi_new.so zk_amd/libzkmi.so
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_gkr.py -x -q --timeout 200 --timeout-method thread -k "prod_plus or gkr or circuit" 2>&1 | tail -3 | tee -a $OUT/$LABEL.txt || exit 1
for v in old new old new old new; do
    cp tools/exp/libzkmi_$v.so zk_amd/libzkmi.so
    echo "== $v gkr cfg5" | tee -a $OUT/$LABEL.txt
    bench_json 200 python -u bench.py --workload gkr --steps 3 --warmup 1 --no-cpu-baseline | summ_gkr | tee -a $OUT/$LABEL.txt || exit 1
done
cp tools/exp/libzkmi_new.so zk_amd/libzkmi.so
