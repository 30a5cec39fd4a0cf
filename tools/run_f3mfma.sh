#!/bin/bash
# A/B of the fused three-variable fold + cross sums: VALU fold (k_fold3_cross64, slots 4) against the matrix-core fold
# (k_fold3_cross64_mfma, slots 8).  Parity first; every step under its own time limit; stderr and exit codes kept under gpurun_out/.
#   gpurun --timeout 900 -- 'bash tools/run_f3mfma.sh r03'
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -k "fold_fused or three_rounds" --timeout 120 --timeout-method thread > $OUT/${TAG}_f3_pytest.log 2>&1 || { tail -40 $OUT/${TAG}_f3_pytest.log; exit 1; }
tail -2 $OUT/${TAG}_f3_pytest.log
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
: > $OUT/${TAG}_f3_ab.txt
for n in 24 25 26; do
for cfg in "--deg2-fold3-fused 4" "--deg2-fold3-fused 8" "--deg2-fold3-fused 0"; do
  label="4x2^$n $cfg"
  timeout -k 10 150 $B --log-n $n $cfg > $OUT/${TAG}_f3_line.json 2> $OUT/${TAG}_f3_last.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "== $label: exit $rc" | tee -a $OUT/${TAG}_f3_ab.txt; tail -20 $OUT/${TAG}_f3_last.err; exit 1; fi
  python3 -c "
import json,sys
d=json.loads(open('$OUT/${TAG}_f3_line.json').read())
print('== $label: prove_ms', round(d['ms_per_step'],4), 'verified', d['proof_check']['verified'], {k:(v['launches'], round(v['total_ms']/5,4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()}, d['kernel_forms'])
" | tee -a $OUT/${TAG}_f3_ab.txt || exit 1
done; done
