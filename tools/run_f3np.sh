#!/bin/bash
# ring-depth sweep of k_fold3_cross64_mfma (4..7 steps per wave) at 4 x 2^26; stderr kept.  gpurun -- 'bash tools/run_f3np.sh r03'
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2 --log-n 26"
: > $OUT/${TAG}_f3np.txt
for f in 12 13 14 4 0; do
  timeout -k 10 150 $B --deg2-fold3-fused $f > $OUT/${TAG}_f3np_line.json 2> $OUT/${TAG}_f3np_last.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "== fused $f: exit $rc" | tee -a $OUT/${TAG}_f3np.txt; tail -20 $OUT/${TAG}_f3np_last.err; exit 1; fi
  python3 -c "
import json
d=json.loads(open('$OUT/${TAG}_f3np_line.json').read())
print('== fused $f: prove_ms', round(d['ms_per_step'],4), 'verified', d['proof_check']['verified'], {k:(v['launches'], round(v['total_ms']/5,4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()})
" | tee -a $OUT/${TAG}_f3np.txt || exit 1
done
