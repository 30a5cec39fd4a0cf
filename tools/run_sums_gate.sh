#!/bin/bash
# block sums by LDS-DMA (k_block_sums_dma) from 2^k entries: the default 27 against lower thresholds at 2^26 .. 2^23
set -o pipefail
LABEL=run_sums_gate
. tools/lib.sh
: > $OUT/$LABEL.txt
for n in 26 25 24 23; do
for k in 27 $n 27 $n; do
    echo "== 2^$n --dma-gates $k 22 22" | tee -a $OUT/$LABEL.txt
    bench_json 150 python -u bench.py --no-extra --no-cpu-baseline --no-absorb --no-live-traffic --steps 20 --warmup 3 --log-n $n --dma-gates $k 22 22 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), d['proof_check']['verified'], d['kernel_forms'], {k:(v['launches'], round(v['avg_ms'],4), round(v['achieved_GBs'])) for k,v in d['kernels'].items()})" | tee -a $OUT/$LABEL.txt || exit 1
done; done
