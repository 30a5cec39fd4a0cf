#!/usr/bin/env python3
"""Where the time of one sharded prove goes on one rank (RCCL backend, world 1): per-pass wall-clock of the
device launch calls, the D2H of the block sums and the host rounds.  python3 tools/sharded_latency.py [log_n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import zk_amd
from zk_amd import sharded

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ctx = sharded.torch_stream_context(0)
poly = zk_amd.MultiLinearPolynomial.generate(7, 1 << log_n, ctx=ctx)
base = zk_amd.Transcript()
base.append(b"x")

acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


orig_gather, orig_rounds = sharded.gather_sums, sharded.rounds_from_block_sums
sharded.gather_sums = timed("gather_sums (D2H + sync: includes waiting for the kernels)", orig_gather)
sharded.rounds_from_block_sums = timed("host rounds", orig_rounds)
REPS = 20
for rep in range(REPS + 3):
    if rep == 3:
        acc.clear()
        torch.cuda.synchronize()
        t_all = time.perf_counter()
    sh = sharded.DeviceShard(poly)
    sh.block_sums = timed("block_sums launch", sh.block_sums)
    sh.fold = timed("fold launch", sh.fold)
    sharded.generate_round_polys_sharded(sh, base.clone(), 1, dist, None)
    sh.close()
total = (time.perf_counter() - t_all) / REPS
print(f"2^{log_n}: {total * 1e3:.3f} ms per prove")
for k, v in acc.items():
    print(f"  {k:62s} {v / REPS * 1e6:8.1f} us per prove")
print(f"  {'unaccounted (python loop, numpy, shard handle, transcript clone)':62s} {(total - sum(acc.values()) / REPS) * 1e6:8.1f} us per prove")
dist.destroy_process_group()
