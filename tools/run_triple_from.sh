#!/bin/bash
# threshold sweep: three rounds per pass down to tables of 2^k entries (k_cross64_mfma / k_fold3_cross64_mfma) at 4 x 2^26, 2^24, 2^22
# gpurun -- 'bash tools/run_triple_from.sh r03'   (stderr and exit codes kept under gpurun_out/)
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2"
: > $OUT/${TAG}_triple_from.txt
for n in 26 24 22; do
for k in 21 20 19 18 17; do
  timeout -k 10 150 $B --log-n $n --deg2-triple-from $k > $OUT/${TAG}_tf_line.json 2> $OUT/${TAG}_tf_last.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "== 4x2^$n from $k: exit $rc" | tee -a $OUT/${TAG}_triple_from.txt; tail -20 $OUT/${TAG}_tf_last.err; exit 1; fi
  python3 -c "
import json
d=json.loads(open('$OUT/${TAG}_tf_line.json').read())
print('== 4x2^$n triple-from $k: prove_ms', round(d['ms_per_step'],4), 'verified', d['proof_check']['verified'], {k:(v['launches'], round(v['total_ms']/5,4)) for k,v in d['kernels'].items()})
" | tee -a $OUT/${TAG}_triple_from.txt || exit 1
done; done
