#!/bin/bash
# round-3 evidence batch: per-rank floor, rank rehearsal, GKR A/B with kept stderr, small-table timelines
set -o pipefail
export PYTHONPATH=$PWD:$PYTHONPATH
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 120 python3 tools/sharded_overhead.py 25 > $OUT/r03c_sharded_overhead.txt 2> $OUT/r03c_sharded_overhead.err || { echo "overhead 25 exit $?"; tail -20 $OUT/r03c_sharded_overhead.err; exit 1; }
timeout -k 10 120 python3 tools/sharded_overhead.py 28 >> $OUT/r03c_sharded_overhead.txt 2>> $OUT/r03c_sharded_overhead.err || { echo "overhead 28 exit $?"; tail -20 $OUT/r03c_sharded_overhead.err; exit 1; }
cat $OUT/r03c_sharded_overhead.txt
bash tools/rehearse_ranks.sh > $OUT/r03c_rehearse_ranks.txt 2>&1 || { echo "rehearse exit $?"; tail -30 $OUT/r03c_rehearse_ranks.txt; exit 1; }
cat $OUT/r03c_rehearse_ranks.txt
bash tools/run_gkr_ab.sh > $OUT/r03c_gkr_ab.txt 2>&1 || { echo "gkr_ab exit $?"; tail -30 $OUT/r03c_gkr_ab.txt; exit 1; }
cat $OUT/r03c_gkr_ab.txt
bash tools/trace_gaps.sh r03c 20 > /dev/null && bash tools/trace_gaps.sh r03c 24 > /dev/null && bash tools/trace_gaps.sh r03c 25 > /dev/null && cat $OUT/r03c_gaps_20.txt $OUT/r03c_gaps_24.txt $OUT/r03c_gaps_25.txt
