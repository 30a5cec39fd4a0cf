#!/bin/bash
# kernel trace of the degree-2 4 x 2^26 prove (bench.py --workload sum2x2): gpurun -- 'bash tools/prof_deg2.sh r02'
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/${TAG}_deg2_trace
timeout -k 10 300 python3 -u bench.py --workload sum2x2 --log-n 26 --no-extra --no-cpu-baseline --steps 5 --warmup 2 > $OUT/${TAG}_sum2x2_log26.json 2> $OUT/${TAG}_deg2.err || { tail -20 $OUT/${TAG}_deg2.err; exit 1; }
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_deg2_trace -- python3 bench.py --no-live-traffic --workload sum2x2 --log-n 26 --no-extra --no-cpu-baseline --steps 3 --warmup 1 > $OUT/${TAG}_deg2_trace.log 2>&1 || { tail -30 $OUT/${TAG}_deg2_trace.log; exit 1; }
python3 tools/prof_summary.py stats $OUT/${TAG}_deg2_trace > $OUT/${TAG}_deg2_kernel_stats.txt && cat $OUT/${TAG}_deg2_kernel_stats.txt
python3 -c "
import json; d=json.load(open('$OUT/${TAG}_sum2x2_log26.json')); print(d['ms_per_step'], d['roofline'])"
# HBM traffic of the same command (two separate counter passes, as MI355X_MICROARCH.md prescribes)
B2="bench.py --no-live-traffic --workload sum2x2 --log-n 26 --no-extra --no-cpu-baseline --steps 3 --warmup 1"
rm -rf $OUT/${TAG}_deg2_pmc_rd $OUT/${TAG}_deg2_pmc_wr
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_deg2_pmc_rd -- python3 $B2 > $OUT/${TAG}_deg2_pmc_rd.log 2>&1 || { tail -30 $OUT/${TAG}_deg2_pmc_rd.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_deg2_pmc_wr -- python3 $B2 > $OUT/${TAG}_deg2_pmc_wr.log 2>&1 || { tail -30 $OUT/${TAG}_deg2_pmc_wr.log; exit 1; }
{ python3 tools/prof_summary.py pmc $OUT/${TAG}_deg2_pmc_rd; python3 tools/prof_summary.py pmc $OUT/${TAG}_deg2_pmc_wr; } > $OUT/${TAG}_deg2_hbm_traffic.txt && cat $OUT/${TAG}_deg2_hbm_traffic.txt
