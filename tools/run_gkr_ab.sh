set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_gkr.py tests/test_gpu_parity.py -x -q -k "gkr or sum2x2" --timeout 300 --timeout-method thread || exit 1
run() {
  echo "== gkr $*"
  timeout -k 10 200 env "$@" python -u bench.py --workload gkr --steps 3 --warmup 1 $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('prove_ms','layers_ms_total','ms_per_round','circuit_evaluate_ms','output_absorb_ms')})" || exit 1
}
ARGS="" run ZK_LAZY_CAP=1024
ARGS="--spin-wait-us 0" run ZK_LAZY_CAP=1024
ARGS="" run ZK_LAZY_CAP=2048
ARGS="" run ZK_LAZY_CAP=4096
ARGS="--deg2-mfma-from 18" run ZK_LAZY_CAP=1024
ARGS="--deg2-lazy-from 64" run ZK_LAZY_CAP=1024
