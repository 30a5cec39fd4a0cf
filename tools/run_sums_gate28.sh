#!/bin/bash
# block sums at 2^28 and 2^27: LDS-DMA form (default from 2^27) vs the register-load form (gate 64 = never DMA)
set -o pipefail
LABEL=run_sums_gate28
. tools/lib.sh
: > $OUT/$LABEL.txt
for n in 28 27; do
for k in 27 64 27 64; do
    echo "== 2^$n --dma-gates $k 22 22" | tee -a $OUT/$LABEL.txt
    bench_json 150 python -u bench.py --no-extra --no-cpu-baseline --no-absorb --no-live-traffic --steps 10 --warmup 2 --log-n $n --dma-gates $k 22 22 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), d['proof_check']['verified'], d['kernel_forms'], {k:(v['launches'], round(v['avg_ms'],4), round(v['achieved_GBs'])) for k,v in d['kernels'].items()})" | tee -a $OUT/$LABEL.txt || exit 1
done; done
