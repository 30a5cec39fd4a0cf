#!/bin/bash
# SQ counters of the 2^28 MLE bench and of the degree-2 4 x 2^26 bench (one --pmc pass each, 8 SQ counters): gpurun -- 'bash tools/prof_sq.sh r02'
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
CNT="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
rm -rf $OUT/${TAG}_sq_mle $OUT/${TAG}_sq_deg2
timeout -s KILL 240 rocprofv3 --pmc $CNT --output-format csv -d $OUT/${TAG}_sq_mle -- python3 bench.py --no-live-traffic --no-absorb --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $OUT/${TAG}_sq_mle.log 2>&1 || { tail -30 $OUT/${TAG}_sq_mle.log; exit 1; }
{ echo "# rocprofv3 --pmc $CNT (one pass) -- python3 bench.py --no-live-traffic --no-absorb --no-cpu-baseline --no-extra --steps 3 --warmup 1   (2^28 entries)"; python3 tools/prof_summary.py pmc $OUT/${TAG}_sq_mle; } > $OUT/${TAG}_sq_counters_bench28.txt
timeout -s KILL 240 rocprofv3 --pmc $CNT --output-format csv -d $OUT/${TAG}_sq_deg2 -- python3 bench.py --no-live-traffic --workload sum2x2 --log-n 26 --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $OUT/${TAG}_sq_deg2.log 2>&1 || { tail -30 $OUT/${TAG}_sq_deg2.log; exit 1; }
{ echo "# rocprofv3 --pmc $CNT (one pass) -- python3 bench.py --no-live-traffic --workload sum2x2 --log-n 26 --no-cpu-baseline --no-extra --steps 3 --warmup 1"; python3 tools/prof_summary.py pmc $OUT/${TAG}_sq_deg2; } > $OUT/${TAG}_sq_counters_deg2.txt
grep -E "k_multifold_wide_mfma|k_block_sums_dma" $OUT/${TAG}_sq_counters_bench28.txt
grep -E "k_cross64_mfma|k_fold3_cross64" $OUT/${TAG}_sq_counters_deg2.txt
