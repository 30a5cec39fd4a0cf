"""The sharded prover with the real DeviceShard (zk_shard_* of libzkmi.so) against the oracle's proof
of the unsharded table, bit for bit.

A GPU box for tests has ONE MI355X, and RCCL refuses two ranks on one device, so the multi-rank cases
put every rank on cuda:0 and exchange through gloo; the single-rank case goes through the nccl (RCCL)
backend so the device-tensor collective path that bench.py --gpus N uses is executed too."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import zk_amd
from zk_amd import sharded

pytestmark = pytest.mark.gpu
SEED = 0x5EED0000


def _free_port():
    """A rendezvous for one spawn: a FileStore URL, not a TCP port -- a port picked here can be taken by another process before
    the ranks bind it (seen once on a GPU box: EADDRINUSE in TCPStore), a fresh file name cannot."""
    import tempfile
    import uuid
    return "file://" + os.path.join(tempfile.gettempdir(), f"zk_rdv_{os.getpid()}_{uuid.uuid4().hex}")


def _worker(rank, world, port, backend, log_n, fold_batch, generated, outdir, gather_log=None):
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=port, rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, init_method=port, rank=rank, world_size=world)
    try:
        ctx = sharded.torch_stream_context(0)
        if fold_batch is None:  # the library's own schedule, with the wide passes switched on for tables this small
            ctx.set_wide_fold(8, 2, 10)
        n = 1 << log_n
        if generated:  # made on the device, strided
            local = zk_amd.MultiLinearPolynomial.generate(SEED, n // world, first=rank, step=world, ctx=ctx)
        else:          # scattered from a host table
            local = zk_amd.MultiLinearPolynomial.upload_shard(O.gen_table(SEED, 0, n), rank, world, ctx=ctx)
        tr = zk_amd.Transcript()
        sharded.absorb_table_sharded(local, tr, world, dist, None, chunk=1 << 8)
        sh = sharded.DeviceShard(local)
        claim, coeffs, ch = sharded.generate_round_polys_sharded(sh, tr, world, dist, None, fold_batch, gather_log)
        sh.close()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), claim=claim, coeffs=coeffs, ch=ch)
    finally:
        dist.destroy_process_group()


def _check(tmp_path, world, log_n):
    table = O.gen_table(SEED, 0, 1 << log_n)
    oclaim, ocoeffs, och = O.sumcheck_prove(table)
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(d["claim"], oclaim), f"rank {r}: claim"
        assert np.array_equal(d["coeffs"], ocoeffs), f"rank {r}: round polynomials"
        assert np.array_equal(d["ch"], och), f"rank {r}: challenges"


@pytest.mark.parametrize("world,log_n,fold_batch,generated", [(2, 14, 4, True), (4, 13, 3, False), (2, 3, 4, False), (8, 12, 4, True),
                                                             (2, 14, None, True), (2, 13, 7, False)])
def test_sharded_ranks_on_one_gpu_gloo(tmp_path, world, log_n, fold_batch, generated):
    mp.spawn(_worker, args=(world, _free_port(), "gloo", log_n, fold_batch, generated, str(tmp_path)), nprocs=world, join=True)
    _check(tmp_path, world, log_n)


@pytest.mark.parametrize("world,log_n,fold_batch,gather_log", [(2, 14, 4, 6), (4, 13, None, 13), (8, 12, 4, 3), (2, 15, None, 9)])
def test_sharded_ranks_gather_small_tables_and_finish_replicated(tmp_path, world, log_n, fold_batch, gather_log):
    mp.spawn(_worker, args=(world, _free_port(), "gloo", log_n, fold_batch, True, str(tmp_path), gather_log), nprocs=world, join=True)
    _check(tmp_path, world, log_n)


def test_single_rank_rccl_with_table_gather(tmp_path):
    mp.spawn(_worker, args=(1, _free_port(), "nccl", 12, None, True, str(tmp_path), 7), nprocs=1, join=True)
    _check(tmp_path, 1, 12)


def test_single_rank_through_rccl_backend(tmp_path):
    mp.spawn(_worker, args=(1, _free_port(), "nccl", 12, 4, True, str(tmp_path)), nprocs=1, join=True)
    _check(tmp_path, 1, 12)


def test_device_shard_steps_match_oracle():
    """zk_shard_block_sums / zk_shard_fold one by one, no process group."""
    ctx = zk_amd.default_context()
    n = 1 << 11
    table = O.gen_table(SEED + 5, 0, n)
    m = zk_amd.MultiLinearPolynomial(table, ctx=ctx)
    sh = sharded.DeviceShard(m)
    buf = sh.new_sums(16)
    sh.block_sums(3, buf)
    ctx.synchronize()
    got = buf[:8].cpu().numpy().view(np.uint64)
    for b in range(8):
        assert np.array_equal(got[b], O.mle_sum(table[b * (n // 8):(b + 1) * (n // 8)]))
    r = O.gen_table(99, 0, 3)
    sh.fold(3, r, 2, buf)
    ctx.synchronize()
    t = table
    for l in range(3):
        t = O.mle_partially_evaluate(t, 0, r[l])
    got = buf[:4].cpu().numpy().view(np.uint64)
    q = t.shape[0] // 4
    for b in range(4):
        assert np.array_equal(got[b], O.mle_sum(t[b * q:(b + 1) * q]))
    assert sh.vars_left() == 8
    with pytest.raises(zk_amd.ZkError):
        sh.fold(4, O.gen_table(1, 0, 4), 5, buf)  # 4 + 5 > 8 variables left (and buf holds 16 sums)
    big = sh.new_sums(512)
    with pytest.raises(zk_amd.ZkError):
        sh.block_sums(9, big)  # more rounds than one pass can bind (8)
    assert sh.next_batch() == 8 and sh.next_batch(6) == 2 and sh.next_batch(8) == 0  # the default schedule on a small table: one batch
    ctx.set_tail_fold(0)
    assert sh.next_batch() == 4 and sh.next_batch(5) == 3
    ctx.set_tail_fold()
    assert np.array_equal(sh.table().cpu().numpy().view(np.uint64), t)               # the folded local table, as gathered
    sh.fold(3, r, 4, buf)  # 8 -> 5 variables left
    assert sh.vars_left() == 5
    with pytest.raises(zk_amd.ZkError):
        sh.fold(3, r, 3, buf)  # 3 + 3 > 5 variables left
    ctx.synchronize()
    sh.close()
    assert np.array_equal(m.get_evaluation_points(), table)  # the caller's table is left unmodified


# ---------------------------------------------------------------- degree-2 branch, sharded
def _worker4(rank, world, port, backend, log_n, gather_log, outdir):
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=port, rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, init_method=port, rank=rank, world_size=world)
    try:
        ctx = sharded.torch_stream_context(0)
        n = 1 << log_n
        tabs = [zk_amd.MultiLinearPolynomial.generate(SEED + 10 + k, n // world, first=rank, step=world, ctx=ctx) for k in range(4)]
        tr = zk_amd.Transcript()
        tr.append(b"degree-2 sharded")
        sh = sharded.DeviceShard4(tabs)
        coeffs, ch = sharded.generate_round_polys_sum2x2_sharded(sh, tr, world, dist, None, gather_log)
        sh.close()
        np.savez(os.path.join(outdir, f"q{rank}.npz"), coeffs=coeffs, ch=ch)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend,log_n,gather_log", [
    (2, "gloo", 10, None), (4, "gloo", 9, None), (2, "gloo", 2, None), (1, "nccl", 8, None),
    # two rounds per exchange, then gathered tables finished on every rank
    (2, "gloo", 12, 4), (4, "gloo", 11, 1), (8, "gloo", 12, 3), (2, "gloo", 9, 13), (1, "nccl", 10, 5), (2, "gloo", 13, 7)])
def test_sharded_degree2_on_one_gpu(tmp_path, world, backend, log_n, gather_log):
    mp.spawn(_worker4, args=(world, _free_port(), backend, log_n, gather_log, str(tmp_path)), nprocs=world, join=True)
    tabs = [O.gen_table(SEED + 10 + k, 0, 1 << log_n) for k in range(4)]
    otr = O.OTranscript()
    otr.append(b"degree-2 sharded")
    oc, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(0), otr)
    for r in range(world):
        d = np.load(tmp_path / f"q{r}.npz")
        assert np.array_equal(d["coeffs"], oc) and np.array_equal(d["ch"], och), f"rank {r}"


# ---------------------------------------------------------------- a context that owns its stream
@pytest.mark.parametrize("gather_log", [None, 6, 13])
def test_sharded_loops_on_a_context_with_its_own_stream(gather_log):
    """DeviceShard / DeviceShard4 on a plain Context (its own non-blocking stream, not torch's current one): the
    gathers and D2H copies of zk_amd.sharded run on torch's stream, so the shards order the two themselves."""
    ctx = zk_amd.Context(0)
    assert ctx.stream != torch.cuda.current_stream().cuda_stream
    for log_n in (9, 15):
        table = O.gen_table(SEED + log_n, 0, 1 << log_n)
        local = zk_amd.MultiLinearPolynomial(table, ctx=ctx)
        sh = sharded.DeviceShard(local)
        tr = zk_amd.Transcript()
        tr.append(local.to_bytes())
        claim, coeffs, ch = sharded.generate_round_polys_sharded(sh, tr, 1, None, None, None, gather_log)
        sh.close()
        oclaim, ocoeffs, och = O.sumcheck_prove(table)
        assert list(claim) == list(oclaim) and np.array_equal(coeffs, ocoeffs) and np.array_equal(ch, och)
    for log_n in (6, 12):
        tabs = [O.gen_table(SEED + 50 + k, 0, 1 << log_n) for k in range(4)]
        m = [zk_amd.MultiLinearPolynomial(t, ctx=ctx) for t in tabs]
        sh4 = sharded.DeviceShard4(m)
        tr, otr = zk_amd.Transcript(), O.OTranscript()
        coeffs, ch = sharded.generate_round_polys_sum2x2_sharded(sh4, tr, 1, None, None, gather_log)
        sh4.close()
        ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(0), otr)
        assert np.array_equal(coeffs, ocoeffs) and np.array_equal(ch, och) and tr.peek_digest() == otr.peek_digest()


# ---------------------------------------------------------------- the loops inside the library (sharded.cpp)
def _worker_lib(rank, world, port, backend, transport, log_n, gather_log, absorb, knobs, outdir):
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=port, rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, init_method=port, rank=rank, world_size=world)
    try:
        ctx = zk_amd.Context(0)  # its own stream: the library orders its kernels, copies and collectives itself
        if knobs is True:
            ctx.set_wide_fold(8, 2, 10)
            ctx.set_dma_gates(0, 0, 0)
        elif knobs in ("triple", "triple_unfused", "triple_mfma"):  # degree 2: three rounds per exchange on these small tables
            ctx.set_deg2_triple(0, 4)
            ctx.set_deg2_fold3_fused(8 if knobs == "triple_mfma" else 4 if knobs == "triple" else 0)
        assert sharded.comm_init(ctx, dist, None, transport) == transport
        assert zk_amd.lib().zk_comm_rank(ctx.h) == rank and zk_amd.lib().zk_comm_world(ctx.h) == world
        n = 1 << log_n
        local = zk_amd.MultiLinearPolynomial.generate(SEED, n // world, first=rank, step=world, ctx=ctx)
        if absorb:      # generate_sumcheck_proof: fresh transcript, the WHOLE table's bytes absorbed (gathered chunk by chunk)
            claim, coeffs, ch = sharded.prove_mle_sharded(local, None, True, gather_log)
        else:
            tr = zk_amd.Transcript()
            tr.append(O.mle_to_bytes(O.gen_table(SEED, 0, n)))
            claim, coeffs, ch = sharded.prove_mle_sharded(local, tr, False, gather_log)
        tabs = [zk_amd.MultiLinearPolynomial.generate(SEED + 10 + k, n // world, first=rank, step=world, ctx=ctx) for k in range(4)]
        tr2 = zk_amd.Transcript()
        tr2.append(b"degree-2 sharded")
        ctx.forms_used()
        c2, ch2 = sharded.prove_sum2x2_sharded(tabs, tr2, None, gather_log)
        if knobs in ("triple", "triple_unfused", "triple_mfma") and gather_log is not None and n // world >= 512 and (n // world).bit_length() - 1 > max(gather_log, 1):
            forms = ctx.forms_used()
            assert "cross64" in forms and ("fold3_cross64" in forms or "fold3_cross64_mfma" in forms or "fold3_dma" in forms), forms
        np.savez(os.path.join(outdir, f"L{rank}.npz"), claim=claim, coeffs=coeffs, ch=ch, c2=c2, ch2=ch2, dg=np.frombuffer(tr2.peek_digest(), dtype=np.uint8))
        sharded.comm_destroy(ctx)
        assert zk_amd.lib().zk_comm_world(ctx.h) == 0
    finally:
        dist.destroy_process_group()


def _check_lib(tmp_path, world, log_n):
    table = O.gen_table(SEED, 0, 1 << log_n)
    oclaim, ocoeffs, och = O.sumcheck_prove(table)
    tabs = [O.gen_table(SEED + 10 + k, 0, 1 << log_n) for k in range(4)]
    otr = O.OTranscript()
    otr.append(b"degree-2 sharded")
    oc2, och2 = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(0), otr)
    for r in range(world):
        d = np.load(tmp_path / f"L{r}.npz")
        assert np.array_equal(d["claim"], oclaim), f"rank {r}: claim"
        assert np.array_equal(d["coeffs"], ocoeffs), f"rank {r}: round polynomials"
        assert np.array_equal(d["ch"], och), f"rank {r}: challenges"
        assert np.array_equal(d["c2"], oc2) and np.array_equal(d["ch2"], och2), f"rank {r}: degree 2"
        assert d["dg"].tobytes() == otr.peek_digest(), f"rank {r}: transcript state"


@pytest.mark.parametrize("world,log_n,gather_log,absorb,knobs", [
    (2, 14, 13, True, False), (4, 13, 6, False, False), (8, 12, None, True, False), (2, 3, None, False, False), (2, 15, 9, False, True),
    (4, 14, None, False, True), (2, 12, 1, False, False), (8, 9, 13, True, False), (2, 1, None, False, False), (4, 2, 0, False, False),
    (2, 15, 3, False, "triple"), (2, 15, 3, False, "triple_mfma"), (4, 16, 6, False, "triple_mfma"), (4, 15, 6, False, "triple"), (2, 14, 1, False, "triple_unfused"), (8, 16, 4, False, "triple"), (2, 16, 9, False, "triple_unfused")])
def test_library_sharded_provers_ranks_on_one_gpu(tmp_path, world, log_n, gather_log, absorb, knobs):
    """zk_sumcheck_prove_mle_sharded / _sum2x2_sharded (the loops in C++, exchange through the library's host all-gather
    callback, here over gloo: RCCL refuses several ranks on one GPU) against the oracle's proof of the unsharded table."""
    mp.spawn(_worker_lib, args=(world, _free_port(), "gloo", "host", log_n, gather_log, absorb, knobs, str(tmp_path)), nprocs=world, join=True)
    _check_lib(tmp_path, world, log_n)


@pytest.mark.parametrize("log_n,gather_log,absorb", [(12, 7, True), (13, None, False), (10, 13, False)])
def test_library_sharded_provers_single_rank_rccl(tmp_path, log_n, gather_log, absorb):
    """the RCCL transport (ncclCommInitRank + ncclAllGather inside libzkmi.so) with the one rank a test box can hold"""
    mp.spawn(_worker_lib, args=(1, _free_port(), "nccl", "rccl", log_n, gather_log, absorb, False, str(tmp_path)), nprocs=1, join=True)
    _check_lib(tmp_path, 1, log_n)


def test_library_sharded_prover_without_transport_and_errors():
    ctx = zk_amd.Context(0)
    L = zk_amd.lib()
    table = O.gen_table(SEED, 0, 1 << 10)
    local = zk_amd.MultiLinearPolynomial(table, ctx=ctx)
    with pytest.raises(RuntimeError):
        sharded.prove_mle_sharded(local)            # no communicator yet
    claim = np.empty(4, dtype=np.uint64)
    coeffs = np.empty((10, 2, 4), dtype=np.uint64)
    rc = L.zk_sumcheck_prove_mle_sharded(ctx.h, local.h, 1, None, 13, claim.ctypes.data_as(zk_amd._lib.u64p), coeffs.ctypes.data_as(zk_amd._lib.u64p), None)
    assert rc == 10 and b"communicator" in L.zk_last_error(ctx.h)   # ZK_E_COMM
    assert L.zk_comm_init_host(ctx.h, 0, 3, None, None) == 8        # not a power of two / no callback
    assert L.zk_comm_init_host(ctx.h, 0, 1, None, None) == 0        # a single rank that exchanges nothing
    assert L.zk_comm_init_host(ctx.h, 0, 1, None, None) == 10       # one communicator per context
    for gather_log in (None, 4, 13):
        c, co, ch = sharded.prove_mle_sharded(local, None, True, gather_log)
        oclaim, ocoeffs, och = O.sumcheck_prove(table)
        assert list(c) == list(oclaim) and np.array_equal(co, ocoeffs) and np.array_equal(ch, och)
    sharded.comm_destroy(ctx)


# ---------------------------------------------------------------- the GKR layer prover over ranks (zk_gkr_prove_sharded)
def _gkr_case(case):
    import random
    from test_gpu_gkr import CIRCUIT_GKR, random_circuit
    if case == "reference":        # gkr_protocol.rs:67-94
        return CIRCUIT_GKR, O.ints_to_mont([1, 2, 3, 4, 5, 6, 7, 8])
    if case == "random":
        return random_circuit([32, 32, 16, 32, 7], 11), O.gen_table(SEED + 11, 0, 32)
    W, depth = 1 << 10, 3          # wide: beyond the dense oracle, compared with the single-GPU prover
    rng = random.Random(9)
    return [[((2 * g + 3 * d) % W, (5 * g + 1) % W, rng.randrange(2)) for g in range(W)] for d in range(depth)], O.gen_table(SEED + 99, 0, W)


def _worker_gkr(rank, world, port, case, gather_log, outdir):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        from test_gpu_gkr import mk, proof_bytes
        ctx = zk_amd.Context(0)
        assert sharded.comm_init(ctx, dist, None, "host") == "host"
        layers, inputs = _gkr_case(case)
        tr = zk_amd.Transcript()
        proof = zk_amd.GKRProver.generate_proof(mk(layers, ctx), tr, zk_amd.MultiLinearPolynomial(inputs, ctx=ctx), sharded_gather_log=gather_log)
        with open(os.path.join(outdir, f"G{rank}.bin"), "wb") as f:
            f.write(tr.peek_digest() + proof_bytes(proof, ctx))
        sharded.comm_destroy(ctx)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case,gather_log", [(2, "reference", 0), (2, "random", 1), (4, "random", 2), (2, "wide", 5), (4, "wide", 3), (8, "wide", 13)])
def test_library_sharded_gkr_prover_ranks_on_one_gpu(tmp_path, world, case, gather_log):
    """zk_gkr_prove_sharded: every layer's sumcheck over tables sharded on their low index bits (gate list partitioned by input
    index), exchange through the host all-gather callback over gloo.  Every rank's proof and final transcript state equal the
    single-GPU prover's, which test_gpu_gkr.py pins to the dense oracle; the small cases are compared with the oracle here too."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_gkr import mk, proof_bytes
    mp.spawn(_worker_gkr, args=(world, _free_port(), case, gather_log, str(tmp_path)), nprocs=world, join=True)
    layers, inputs = _gkr_case(case)
    ctx = zk_amd.default_context()
    tr = zk_amd.Transcript()
    proof = zk_amd.GKRProver.generate_proof(mk(layers, ctx), tr, inputs)
    want = tr.peek_digest() + proof_bytes(proof, ctx)
    if case != "wide":
        oc = O.OCircuit(layers)
        otr = O.OTranscript()
        oblob, _ = O.gkr_prove(oc, otr, inputs).serialize()
        assert want == otr.peek_digest() + oblob
    for r in range(world):
        assert (tmp_path / f"G{r}.bin").read_bytes() == want, f"rank {r}"


def test_sharded_gkr_needs_a_communicator():
    from test_gpu_gkr import CIRCUIT_GKR, mk
    ctx = zk_amd.Context(0)
    with pytest.raises(zk_amd.ZkError) as ei:
        zk_amd.GKRProver.generate_proof(mk(CIRCUIT_GKR, ctx), zk_amd.Transcript(), O.ints_to_mont([1, 2, 3, 4, 5, 6, 7, 8]), sharded_gather_log=3)
    assert ei.value.status == 10
