#!/bin/bash
# eq weights from host-built half trees (new) against the in-kernel trees (old = commit 1211819): same box, libraries swapped
set -o pipefail
LABEL=run_eq_halves_ab
. tools/lib.sh
: > $OUT/$LABEL.txt
cp tools/exp/libzkmi_new.so zk_amd/libzkmi.so
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread -k "not sum2x2 and not prod_plus" 2>&1 | tail -3 | tee -a $OUT/$LABEL.txt || exit 1
for n in 20 24 25 28; do
for v in old new old new; do
    cp tools/exp/libzkmi_$v.so zk_amd/libzkmi.so
    echo "== $v 2^$n" | tee -a $OUT/$LABEL.txt
    bench_json 150 python -u bench.py --no-extra --no-cpu-baseline --no-absorb --no-live-traffic --steps 20 --warmup 3 --log-n $n | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), d['proof_check']['verified'], {k:(v['launches'], round(v['avg_ms'],4)) for k,v in d['kernels'].items()})" | tee -a $OUT/$LABEL.txt || exit 1
done; done
for v in old new old new; do
    cp tools/exp/libzkmi_$v.so zk_amd/libzkmi.so
    echo "== $v 4x2^22" | tee -a $OUT/$LABEL.txt
    bench_json 120 python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --no-live-traffic --steps 20 --warmup 3 --log-n 22 | summ_deg2 | tee -a $OUT/$LABEL.txt || exit 1
done
cp tools/exp/libzkmi_new.so zk_amd/libzkmi.so
