#!/bin/bash
# One GPU-box pass: parity tests, the default bench line, a kernel-trace profile and the two HBM
# counter passes of the same bench command.  Run through gpurun from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/gpu_round.sh r02'        (PROFILE_ONLY=1: skip the tests, smoke and the full bench)
# Each GPU step has its own time limit and the chain stops at the first failure.
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="bench.py --no-live-traffic --no-absorb --no-cpu-baseline --no-extra --steps 3 --warmup 1"

if [ -z "$PROFILE_ONLY" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/${TAG}_pytest_gpu.log 2>&1 || { tail -30 $OUT/${TAG}_pytest_gpu.log; exit 1; }
tail -3 $OUT/${TAG}_pytest_gpu.log
timeout -k 10 300 python -u __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1 || { tail -30 $OUT/${TAG}_smoke.log; exit 1; }
tail -1 $OUT/${TAG}_smoke.log
timeout -k 10 600 python -u bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -30 $OUT/${TAG}_bench.err; exit 1; }
cat $OUT/${TAG}_bench.json
fi

rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_rd $OUT/${TAG}_pmc_wr
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $BENCH > $OUT/${TAG}_trace.log 2>&1 || { tail -30 $OUT/${TAG}_trace.log; exit 1; }
python3 tools/prof_summary.py stats $OUT/${TAG}_trace > $OUT/${TAG}_kernel_stats.txt && cat $OUT/${TAG}_kernel_stats.txt
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_rd -- python3 $BENCH > $OUT/${TAG}_pmc_rd.log 2>&1 || { tail -30 $OUT/${TAG}_pmc_rd.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_wr -- python3 $BENCH > $OUT/${TAG}_pmc_wr.log 2>&1 || { tail -30 $OUT/${TAG}_pmc_wr.log; exit 1; }
{ python3 tools/prof_summary.py pmc $OUT/${TAG}_pmc_rd; python3 tools/prof_summary.py pmc $OUT/${TAG}_pmc_wr; } > $OUT/${TAG}_hbm_traffic.txt && cat $OUT/${TAG}_hbm_traffic.txt
python3 tools/prof_summary.py traffic $OUT/${TAG}_pmc_rd $OUT/${TAG}_pmc_wr 28 > $OUT/${TAG}_hbm_traffic.json
