set -o pipefail
B="python -u bench.py --no-extra --no-absorb --no-cpu-baseline --steps 10 --warmup 3"
run() {
  echo "== $*"
  timeout -k 10 120 env "$@" $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prove_ms', round(d['ms_per_step'],4), {k:(round(v['avg_ms'],4), round(v['achieved_GBs'],0)) for k,v in d['kernels'].items() if v['launches']})
" || exit 1
}
run ZK_BS_ROT=0
run ZK_BS_ROT=64
run ZK_BS_ROT=67
run ZK_BS_ROT=1
run ZK_BS_ROT=8
run ZK_BS_ROT=0 ZK_BS_GPB=4
run ZK_BS_ROT=64 ZK_BS_GPB=4
run ZK_BS_ROT=0 ZK_BS_GPB=16
run ZK_BS_ROT=64 ZK_BS_GPB=16
run ZK_BS_ROT=16 ZK_BS_GPB=16
