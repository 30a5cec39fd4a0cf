#!/usr/bin/env python3
"""Per-prove overhead of the in-library sharded provers with ONE rank (RCCL communicator of size 1, or no transport) against the
plain single-GPU prover on the same table: what the exchange path itself costs at the per-GPU size of an 8-GPU run.
    gpurun -- 'python3 tools/sharded_overhead.py 25'"""
import sys
import time

import numpy as np

import zk_amd
from zk_amd import sharded

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
reps = 10
ctx = zk_amd.Context(0)
poly = zk_amd.MultiLinearPolynomial.generate(0x5EED, 1 << log_n, ctx=ctx)


def timed(fn):
    fn()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def plain():
    tr = zk_amd.Transcript()
    tr.append(b"x")
    zk_amd.SumcheckProver.generate_round_polys(poly, tr)


base = timed(plain)
out = {"log_n": log_n, "plain_ms": round(base, 4)}
for transport in ("rccl", "host"):
    sharded.comm_init(ctx, None, None, transport)

    def lib():
        tr = zk_amd.Transcript()
        tr.append(b"x")
        sharded.prove_mle_sharded(poly, tr, False, 13)

    out[f"sharded_{transport}_world1_ms"] = round(timed(lib), 4)
    sharded.comm_destroy(ctx)
print(out)
