#!/bin/bash
# A/B of the GKR layer prover's schedule knobs on config 5: gpurun -- 'bash tools/run_gkr_ab.sh'
# (results on stdout, stderr and exit codes in gpurun_out/run_gkr_ab.err)
set -o pipefail
LABEL=run_gkr_ab
. tools/lib.sh
B="python -u bench.py --workload gkr --steps 3 --warmup 1 --no-cpu-baseline"
for cfg in "" "--deg2-lazy-from -1" "--deg2-mfma-from 18" "--deg2-fused-below 0" "--deg2-lazy-from 64"; do
    echo "== gkr $cfg"
    # the last case is the one that printed nothing in round 2: 64 is outside the setter's range (never = a negative value),
    # the library answers ZK_E_ARG and bench.py ends with a ZkError on stderr -- expected to FAIL here, visibly
    if [ "$cfg" = "--deg2-lazy-from 64" ]; then
        bench_json 200 $B $cfg | summ_gkr && { echo "expected a refusal"; exit 1; }
        echo "(refused as expected: see the tail above)"
    else
        bench_json 200 $B $cfg | summ_gkr || exit 1
    fi
done
