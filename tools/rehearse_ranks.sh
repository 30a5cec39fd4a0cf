#!/bin/bash
# Rehearsal of the multi-rank bench path on ONE GPU: gloo ranks sharing cuda:0 (RCCL cannot put two ranks on one device).
#   gpurun -- 'bash tools/rehearse_ranks.sh'   (stderr of every launch and its exit code: gpurun_out/rehearse_ranks.err)
set -o pipefail
LABEL=rehearse_ranks
. tools/lib.sh
TR="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
for w in 2 4 8; do
  bench_json 300 $TR --nproc-per-node $w --master-port $((29610 + w)) bench.py --gpus $w --steps 3 --warmup 1 --log-n 26 --backend gloo --shared-device --no-absorb --no-cpu-baseline |
    python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["n_gpus"], round(d["ms_per_step"],3), d["scaling"], d["proof_check"], d["config"]["exchange"][:60])' || exit 1
done
bench_json 300 $TR --nproc-per-node 2 --master-port 29619 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 24 --workload sum2x2 --backend gloo --shared-device --no-cpu-baseline |
  python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["n_gpus"], round(d["ms_per_step"],3), d["proof_check"], d["roofline"]["kernel"])' || exit 1
bench_json 300 $TR --nproc-per-node 2 --master-port 29620 bench.py --gpus 2 --steps 2 --warmup 1 --workload gkr --backend gloo --shared-device --no-cpu-baseline |
  python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["n_gpus"], round(d["ms_per_step"],3), d["proof_check"], d.get("rccl_note"))' || exit 1
