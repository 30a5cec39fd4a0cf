#!/bin/bash
# SQ counters of the degree-2 4 x 2^26 bench for a given fused-fold form: gpurun -- 'bash tools/prof_sq_deg2.sh r03 8'
# Two passes of 8 SQ counters each (issue mix; wait/busy breakdown).  stderr and exit codes kept under gpurun_out/.
set -o pipefail
TAG=${1:-r03}
FORM=${2:-8}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py --no-live-traffic --workload sum2x2 --log-n 26 --no-cpu-baseline --no-extra --steps 3 --warmup 1 --deg2-fold3-fused $FORM"
C1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
C2="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
for p in 1 2; do
  eval CNT=\$C$p
  rm -rf $OUT/${TAG}_sqd_$p
  timeout -s KILL 200 rocprofv3 --pmc $CNT --output-format csv -d $OUT/${TAG}_sqd_$p -- python3 $B > $OUT/${TAG}_sqd_$p.log 2>&1 || { echo "pass $p: exit $?"; tail -30 $OUT/${TAG}_sqd_$p.log; exit 1; }
done
{ echo "# rocprofv3 --pmc (two passes) -- python3 $B"; python3 tools/prof_summary.py pmc $OUT/${TAG}_sqd_1; python3 tools/prof_summary.py pmc $OUT/${TAG}_sqd_2; } > $OUT/${TAG}_sq_deg2_form$FORM.txt
grep -E "k_cross64_mfma|k_fold3_cross64" $OUT/${TAG}_sq_deg2_form$FORM.txt
