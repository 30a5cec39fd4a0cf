"""Multi-rank logic of the sharded prover on CPU ranks (gloo, world_size 2 and 4).

No GPU here, so each rank's LOCAL table work (block sums, fold of the top variables) is done by a
stand-in shard backed by the CPU oracle -- test infrastructure standing in for DeviceShard.  What is
under test is the product code around it: the low-bit sharding convention, the all-gather of the
per-rank block sums, the host round function of libzkmi.so (zk_sumcheck_rounds_from_block_sums, which
needs no GPU) and the host tail over the last log2(world) variables.  The result must equal the
oracle's proof of the UNSHARDED table bit for bit.  tests/test_gpu_sharded.py runs the same loop with
the real DeviceShard."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import zk_amd
from zk_amd import sharded

SEED = 0x5EED0000


class OracleShard:
    """Stand-in for zk_amd.sharded.DeviceShard: the rank's local table as a numpy array, local passes
    computed with the CPU oracle."""

    def __init__(self, local_table: np.ndarray):
        self.t = np.ascontiguousarray(local_table, dtype=np.uint64).reshape(-1, 4)

    def vars_left(self):
        return self.t.shape[0].bit_length() - 1

    def new_sums(self, count):
        return torch.zeros((count, 4), dtype=torch.int64)

    def _sums(self, m, out):
        nb = 1 << m
        lb = self.t.shape[0] // nb
        for b in range(nb):
            out[b] = torch.from_numpy(O.mle_sum(self.t[b * lb:(b + 1) * lb]).view(np.int64))

    def block_sums(self, m, out):
        self._sums(m, out)

    def fold(self, m, r, m_next, out):
        for l in range(m):
            self.t = O.mle_partially_evaluate(self.t, 0, r[l])
        if m_next:
            self._sums(m_next, out)

    def table(self):
        return torch.from_numpy(self.t.view(np.int64).copy())

    def finish(self, table, transcript):
        """the remaining table on one rank: libzkmi's host round function on the table itself (world 1, sums =
        entries) is the whole round loop"""
        t = table.numpy().view(np.uint64).reshape(1, -1, 4)
        coeffs, ch, _, total = sharded.rounds_from_block_sums(transcript, t, want_total=True)
        return total, coeffs, ch


def _free_port():
    """A rendezvous for one spawn: a FileStore URL, not a TCP port -- a port picked here can be taken by another process before
    the ranks bind it (seen once on a GPU box: EADDRINUSE in TCPStore), a fresh file name cannot."""
    import tempfile
    import uuid
    return "file://" + os.path.join(tempfile.gettempdir(), f"zk_rdv_{os.getpid()}_{uuid.uuid4().hex}")


def _worker(rank, world, port, log_n, fold_batch, gather_log, outdir):
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        n = 1 << log_n
        table = O.gen_table(SEED, 0, n)
        local = table[rank::world]  # entry j of the shard = global entry j*world + rank
        tr = zk_amd.Transcript()
        tr.append(O.mle_to_bytes(table))  # generate_sumcheck_proof absorbs the whole table first
        claim, coeffs, ch = sharded.generate_round_polys_sharded(OracleShard(local), tr, world, dist, None, fold_batch, gather_log)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), claim=claim, coeffs=coeffs, ch=ch, digest=np.frombuffer(tr.peek_digest(), dtype=np.uint8))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,fold_batch,gather_log", [
    (2, 10, 4, None), (2, 6, 3, None), (4, 9, 4, None), (2, 2, 4, None), (4, 3, 1, None), (2, 10, 8, None), (2, 9, None, None),
    # switch to replicated passes once the local tables are down to 2^gather_log entries
    (2, 10, 4, 5), (4, 9, 3, 2), (2, 8, 4, 0), (2, 6, 4, 13), (4, 7, 8, 4), (2, 1, 4, 13)])
def test_sharded_round_loop_matches_unsharded_oracle(tmp_path, world, log_n, fold_batch, gather_log):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, log_n, fold_batch, gather_log, str(tmp_path)), nprocs=world, join=True)
    table = O.gen_table(SEED, 0, 1 << log_n)
    oclaim, ocoeffs, och = O.sumcheck_prove(table)
    digests = []
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(d["claim"], oclaim), f"rank {r}: claim"
        assert np.array_equal(d["coeffs"], ocoeffs), f"rank {r}: round polynomials"
        assert np.array_equal(d["ch"], och), f"rank {r}: challenges"
        digests.append(d["digest"].tobytes())
    assert len(set(digests)) == 1  # every rank's transcript ended in the same state


def test_rounds_from_block_sums_is_the_round_loop_on_a_small_table():
    """world = 1, sums = the table itself: the host function alone is the whole MLE-branch round loop."""
    for m in (1, 2, 3, 5):
        table = O.gen_table(SEED + m, 0, 1 << m)
        tr = zk_amd.Transcript()
        tr.append(O.mle_to_bytes(table))
        coeffs, ch, folded, total = sharded.rounds_from_block_sums(tr, table[None], want_folded=True, want_total=True)
        oclaim, ocoeffs, och = O.sumcheck_prove(table)
        assert np.array_equal(total, oclaim) and np.array_equal(coeffs, ocoeffs) and np.array_equal(ch, och)
        pts = [och[i] for i in range(m)]
        assert np.array_equal(folded[0], O.mle_evaluate(table, pts)[0])


def test_rounds_from_block_sums_rejects_bad_shapes():
    tr = zk_amd.Transcript()
    with pytest.raises(ValueError):
        sharded.rounds_from_block_sums(tr, np.zeros((1, 3, 4), dtype=np.uint64))
    with pytest.raises(ValueError):
        sharded.rounds_from_block_sums(tr, np.zeros((1, 1, 4), dtype=np.uint64))


# ---------------------------------------------------------------- degree-2 branch
P = O.P


class OracleShard4:
    """Stand-in for DeviceShard4: four local tables as Python ints; evaluations and folds in big ints."""

    def __init__(self, tables):
        self.t = [O.mont_to_ints(t) for t in tables]

    def local_len(self):
        return len(self.t[0])

    def new_buf(self):
        return torch.zeros((4, 4), dtype=torch.int64)

    def _put(self, vals, out):
        a = O.ints_to_mont(vals)
        out[: len(vals)] = torch.from_numpy(a.view(np.int64))

    def eval(self, out):
        h = len(self.t[0]) // 2
        ev = [0, 0, 0]
        for j in range(h):
            lo = [t[j] for t in self.t]
            hi = [t[j + h] for t in self.t]
            for x in range(3):
                v = [(lo[q] + x * (hi[q] - lo[q])) % P for q in range(4)]
                ev[x] = (ev[x] + v[0] * v[1] + v[2] * v[3]) % P
        self._put(ev, out)

    def fold(self, r, out):
        rr = O.mont_to_ints(np.asarray(r).reshape(1, 4))[0]
        h = len(self.t[0]) // 2
        self.t = [[(t[j] + (t[j + h] - t[j]) * rr) % P for j in range(h)] for t in self.t]
        if h >= 2:
            self.eval(out)
        else:
            self._put([t[0] for t in self.t], out)


    # two rounds per exchange
    def new_buf9(self):
        return torch.zeros((9, 4), dtype=torch.int64)

    def sum9(self, out):
        """S[3Y + t]: the table index is (b1, b2, j); F(b1, Y) = F[b1,0] + Y (F[b1,1] - F[b1,0]); f0 = F(0,Y), f1 = F(1,Y),
        df = f1 - f0; t = 0: f0 g0, 1: f1 g1, 2: df dg; summed over j and both product terms (deg2_kernels.hip, k_sum9)"""
        q = len(self.t[0]) // 4
        S = [0] * 9
        for j in range(q):
            for Y in range(3):
                at = lambda t, b1: (t[b1 * 2 * q + j] + Y * (t[b1 * 2 * q + q + j] - t[b1 * 2 * q + j])) % P
                for F, G in ((self.t[0], self.t[1]), (self.t[2], self.t[3])):
                    f0, f1, g0, g1 = at(F, 0), at(F, 1), at(G, 0), at(G, 1)
                    S[3 * Y] = (S[3 * Y] + f0 * g0) % P
                    S[3 * Y + 1] = (S[3 * Y + 1] + f1 * g1) % P
                    S[3 * Y + 2] = (S[3 * Y + 2] + (f1 - f0) * (g1 - g0)) % P
        self._put(S, out)

    def fold2(self, r2, out):
        for r in np.asarray(r2).reshape(2, 4):
            rr = O.mont_to_ints(r.reshape(1, 4))[0]
            h = len(self.t[0]) // 2
            self.t = [[(t[j] + (t[j + h] - t[j]) * rr) % P for j in range(h)] for t in self.t]
        if out is not None and len(self.t[0]) >= 4:
            self.sum9(out)

    def tables_tensor(self):
        return torch.from_numpy(np.stack([O.ints_to_mont(t) for t in self.t]).view(np.int64))

    def finish(self, tables, transcript):
        """the remaining tables on one rank: libzkmi's host degree-2 prover"""
        return sharded.prove_sum2x2_host(transcript, tables.numpy().view(np.uint64))


def _worker4(rank, world, port, log_n, gather_log, outdir):
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        n = 1 << log_n
        tabs = [O.gen_table(SEED + 10 + k, 0, n) for k in range(4)]
        tr = zk_amd.Transcript()
        tr.append(b"degree-2 sharded")
        coeffs, ch = sharded.generate_round_polys_sum2x2_sharded(OracleShard4([t[rank::world] for t in tabs]), tr, world, dist, None, gather_log)
        np.savez(os.path.join(outdir, f"q{rank}.npz"), coeffs=coeffs, ch=ch, digest=np.frombuffer(tr.peek_digest(), dtype=np.uint8))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,gather_log", [(2, 6, None), (4, 5, None), (2, 2, None), (4, 3, None),
                                                    (2, 7, 2), (4, 7, 1), (2, 6, 3), (2, 4, 13), (4, 2, 1), (2, 8, 0)])
def test_sharded_degree2_matches_unsharded_oracle(tmp_path, world, log_n, gather_log):
    mp.spawn(_worker4, args=(world, _free_port(), log_n, gather_log, str(tmp_path)), nprocs=world, join=True)
    tabs = [O.gen_table(SEED + 10 + k, 0, 1 << log_n) for k in range(4)]
    otr = O.OTranscript()
    otr.append(b"degree-2 sharded")
    oc, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(0), otr)
    for r in range(world):
        d = np.load(tmp_path / f"q{r}.npz")
        assert np.array_equal(d["coeffs"], oc) and np.array_equal(d["ch"], och), f"rank {r}"
        assert d["digest"].tobytes() == otr.peek_digest()


def test_host_degree2_prover_matches_oracle():
    for log_n in (1, 2, 4):
        tabs = [O.gen_table(SEED + 20 + k, 0, 1 << log_n) for k in range(4)]
        tr, otr = zk_amd.Transcript(), O.OTranscript()
        c, ch = sharded.prove_sum2x2_host(tr, np.stack(tabs))
        oc, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(0), otr)
        assert np.array_equal(c, oc) and np.array_equal(ch, och) and tr.peek_digest() == otr.peek_digest()


@pytest.mark.parametrize("world,log_n", [(1, 3), (1, 6), (2, 5), (4, 7)])
def test_three_rounds_from_cross_sums_match_oracle(world, log_n):
    """zk_sumcheck_rounds_from_cross64 (host only): from the ranks' shares of the 2 x 64 cross sums of the three top variables
    the first three rounds' coefficients and challenges equal the oracle's (sumcheck/src/prover.rs:89-135 three times)."""
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 30 + k, 0, n) for k in range(4)]
    ints = [O.mont_to_ints(t) for t in tabs]
    q = n // 8
    cross = np.empty((world, 2, 8, 8, 4), dtype=np.uint64)
    for rank in range(world):           # rank r holds the entries whose index is r mod world
        for term in range(2):
            F, G = ints[2 * term], ints[2 * term + 1]
            vals = [sum(F[e * q + j] * G[e2 * q + j] for j in range(rank, q, world)) % P for e in range(8) for e2 in range(8)]
            cross[rank, term] = O.ints_to_mont(vals).reshape(8, 8, 4)
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    tr.append(b"cross")
    otr.append(b"cross")
    coeffs, ch = sharded.rounds_from_cross64(tr, cross)
    oc, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(0), otr)
    assert np.array_equal(coeffs, oc[:3]) and np.array_equal(ch, och[:3])
