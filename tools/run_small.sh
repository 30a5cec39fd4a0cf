#!/bin/bash
# small-table latency: the single-MLE prove at 2^20, 2^24 and 2^25 entries (the per-rank table of an 8-way sharded 2^28 prove)
#   gpurun -- 'bash tools/run_small.sh r03'      -> gpurun_out/<tag>_small.txt (+ .err: stderr and exit codes)
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
: > $OUT/${TAG}_small.txt
: > $OUT/${TAG}_small.err
for LOGN in 20 24 25; do
    timeout -k 10 200 python3 -u bench.py --log-n $LOGN --no-absorb --no-cpu-baseline --no-extra --steps 50 --warmup 5 >> $OUT/${TAG}_small.txt 2>> $OUT/${TAG}_small.err
    rc=$?
    echo "log_n $LOGN exit $rc" >> $OUT/${TAG}_small.err
    [ $rc -eq 0 ] || { tail -20 $OUT/${TAG}_small.err; exit 1; }
done
python3 - <<PY
import json
for l in open('$OUT/${TAG}_small.txt'):
    if l.startswith('{'):
        d = json.loads(l); print(d['config'].get('log_n'), d['ms_per_step'], 'ms', d.get('proof_check'))
PY
