# tail of the prove: eight-round passes with one wave per output (zk_ctx_set_tail_fold) vs m = 4 passes
set -o pipefail
B="python bench.py --no-absorb --no-cpu-baseline --steps 10 --warmup 3"
for n in 16 20 22 24 26 28; do
  for cfg in "0 13" "20 13" "20 10" "20 16" "16 13" "22 13"; do
    set -- $cfg
    echo "log_n $n tail_max_log $1 out_below $2: $(timeout -k 10 120 $B --log-n $n --tail-max-log $1 --tail-out-below $2 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:(v["launches"], round(v["total_ms"],3)) for k,v in d.get("kernels",{}).items()})')" || exit 1
  done
done
