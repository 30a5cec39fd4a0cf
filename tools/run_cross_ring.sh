#!/bin/bash
# LDS ring depth of k_cross64_mfma (3..5 tiles of 32 KiB) at 4 x 2^26: gpurun -- 'bash tools/run_cross_ring.sh r03'
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
B="python -u bench.py --workload sum2x2 --no-extra --no-cpu-baseline --steps 5 --warmup 2 --log-n 26"
: > $OUT/${TAG}_cross_ring.txt
for r in 3 4 5 4 5; do
  timeout -k 10 150 $B --deg2-triple-ring $r > $OUT/${TAG}_cr_line.json 2> $OUT/${TAG}_cr_last.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "== ring $r: exit $rc" | tee -a $OUT/${TAG}_cross_ring.txt; tail -20 $OUT/${TAG}_cr_last.err; exit 1; fi
  python3 -c "
import json
d=json.loads(open('$OUT/${TAG}_cr_line.json').read())
print('== ring $r: prove_ms', round(d['ms_per_step'],4), 'verified', d['proof_check']['verified'], {k:(v['launches'], round(v['total_ms']/5,4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items()})
" | tee -a $OUT/${TAG}_cross_ring.txt || exit 1
done
