set -o pipefail
run() {
  echo "== gkr $* $ARGS"
  timeout -k 10 200 env "$@" python -u bench.py --workload gkr --steps 3 --warmup 1 $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('prove_ms','layers_ms_total','ms_per_round','circuit_evaluate_ms','output_absorb_ms')})" || exit 1
}
ARGS="" run ZK_F2S9_CAP=1024
ARGS="" run ZK_F2S9_CAP=512
ARGS="" run ZK_F2S9_CAP=256
ARGS="" run ZK_F2S9_CAP=128
