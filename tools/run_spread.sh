#!/bin/bash
# matrix-core wide fold: the default LDS-DMA ring (wg 1, depth 2) vs the table operand by register loads with the weights shared through
# LDS (depth 22 / 24 = 2 / 4 steps per block) at several workgroups per CU, 2^28 .. 2^24
set -o pipefail
LABEL=run_spread
. tools/lib.sh
: > $OUT/$LABEL.txt
for n in 28 26 24; do
for cfg in "1 2" "2 22" "3 22" "4 22" "1 24" "2 24" "1 2" "2 22"; do
    set -- $cfg
    echo "== 2^$n --wide-mfma 15 --wide-mfma-wg $1 --wide-mfma-depth $2" | tee -a $OUT/$LABEL.txt
    bench_json 150 python -u bench.py --no-extra --no-cpu-baseline --no-absorb --no-live-traffic --steps 10 --warmup 2 --log-n $n --wide-mfma 15 --wide-mfma-wg $1 --wide-mfma-depth $2 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), d['proof_check']['verified'], {k:(v['launches'], round(v['avg_ms'],4), round(v['achieved_GBs'])) for k,v in d['kernels'].items()})" | tee -a $OUT/$LABEL.txt || exit 1
done; done
