#!/bin/bash
# kernel trace of the GKR config-5 prove (bench.py --workload gkr): gpurun -- 'bash tools/prof_gkr.sh r02'
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/${TAG}_gkr_trace
timeout -k 10 300 python3 -u bench.py --workload gkr --steps 3 --warmup 1 > $OUT/${TAG}_gkr_cfg5.json 2> $OUT/${TAG}_gkr.err || { tail -20 $OUT/${TAG}_gkr.err; exit 1; }
timeout -s KILL 240 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/${TAG}_gkr_trace -- python3 bench.py --no-live-traffic --workload gkr --steps 3 --warmup 1 > $OUT/${TAG}_gkr_trace.log 2>&1 || { tail -30 $OUT/${TAG}_gkr_trace.log; exit 1; }
python3 tools/prof_summary.py stats $OUT/${TAG}_gkr_trace > $OUT/${TAG}_gkr_kernel_stats.txt && cat $OUT/${TAG}_gkr_kernel_stats.txt
python3 -c "
import json; d=json.load(open('$OUT/${TAG}_gkr_cfg5.json')); print({k:d[k] for k in ('prove_ms','layers_ms_total','ms_per_round','circuit_evaluate_ms','output_absorb_ms')})"
