# Rehearsal of the multi-rank bench path on ONE GPU: gloo ranks sharing cuda:0 (RCCL cannot put two ranks on one device).
set -o pipefail
for w in 2 4 8; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus $w --steps 3 --warmup 1 --log-n 26 --backend gloo --shared-device --no-absorb --no-cpu-baseline 2>/dev/null | tail -1 |
    python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["n_gpus"], round(d["ms_per_step"],3), d["scaling"], d["proof_check"], d["config"]["exchange"][:60])' || exit 1
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 \
  bench.py --gpus 2 --steps 3 --warmup 1 --log-n 24 --workload sum2x2 --backend gloo --shared-device --no-cpu-baseline 2>/dev/null | tail -1 |
  python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["n_gpus"], round(d["ms_per_step"],3), d["proof_check"], d["roofline"]["kernel"])'
