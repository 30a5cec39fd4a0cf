#!/bin/bash
# register-load block sums: k_block_sums (0) against k_block_sums_lazy with 2, 4, 8 entries in flight per lane, at 2^24 .. 2^28
# (DMA sums gated off so the register form runs at every size; the last line of each size is the default gate for comparison)
set -o pipefail
LABEL=run_sums_inflight
. tools/lib.sh
: > $OUT/$LABEL.txt
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "block_sums_unreduced or dma_gate_arguments" 2>&1 | tail -3 | tee -a $OUT/$LABEL.txt || exit 1
for n in 24 25 26 28; do
for u in 0 2 4 8 0 2 4 8 dma; do
    if [ $u = dma ]; then g="--dma-gates 0 22 22"; else g="--dma-gates 64 22 22 --sums-inflight $u"; fi
    echo "== 2^$n $g" | tee -a $OUT/$LABEL.txt
    bench_json 150 python -u bench.py --no-extra --no-cpu-baseline --no-absorb --no-live-traffic --steps 20 --warmup 3 --log-n $n $g | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), d['proof_check']['verified'], {k:(v['launches'], round(v['avg_ms'],4), round(v['achieved_GBs'])) for k,v in d['kernels'].items() if 'sums' in k})" | tee -a $OUT/$LABEL.txt || exit 1
done; done
