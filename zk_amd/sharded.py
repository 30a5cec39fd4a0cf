"""Sumcheck provers over tables sharded across the GPUs of one node.

One process per GPU.  Rank q of `world` (a power of two, g = log2 world) holds the entries whose
global index is q mod world, stored at index >> g: the table is sharded on the g variables that are
bound LAST, so for the first n-g rounds every fold pair (j, j + N/2) of evaluation_form.rs:61-79
lives on one GPU and the folds are rank-local.  The only exchange per pass is the all-gather of every
rank's small sums (2^m block sums for the MLE branch, nine sums per two rounds for the degree-2 branch;
RCCL over xGMI through torch.distributed, backend "nccl"); every rank then runs the identical host part
(Keccak transcript, round polynomials, challenges), so nothing is broadcast back.

Small passes are bound by the latency of that exchange, not by their work: once a rank's table is down to
2^gather_log entries (default 2^13 = 256 KiB) the ranks all-gather the TABLES (entry j of rank q is entry
j*world + q of what remains) and every rank finishes the remaining rounds alone with the single-GPU prover
on the running transcript -- replicated work, no further exchange.  With gather_log=None the loops stay
sharded to the end: when the local tables are used up, the remaining g-variable table (one entry per rank)
is already on every host and the last g rounds run there.

The loops are written against a small "shard" interface so the multi-rank logic can be exercised on CPU
ranks (gloo) in tests with a stand-in shard; the product shards are DeviceShard / DeviceShard4 = zk_shard_* /
zk_shard4_* of libzkmi.so.  There is no CPU compute path in this package.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, u64p
from .context import Context
from .polynomials import MultiLinearPolynomial
from .transcript import Transcript

MAX_FOLD_BATCH = 8      # rounds one device pass can bind (zkk::kMaxFoldBatch)
DEFAULT_FOLD_BATCH = 4  # schedule of shards that do not bring their own (zk_shard_next_batch)
DEFAULT_GATHER_LOG = 13 # local tables of <= 2^13 entries (256 KiB) are all-gathered and finished on every rank


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(u64p)


def rounds_from_block_sums(transcript: Transcript, sums: np.ndarray, want_folded: bool = False, want_total: bool = False):
    """Host part of one batch of m rounds (zk_sumcheck_rounds_from_block_sums; no GPU needed).
    sums: (world, 2^m, 4) uint64.  -> (coeffs (m,2,4), challenges (m,4), folded (world,4) | None, total (4,) | None)"""
    sums = np.ascontiguousarray(sums, dtype=np.uint64)
    world, nb = sums.shape[0], sums.shape[1]
    m = nb.bit_length() - 1
    if nb != 1 << m or m < 1:
        raise ValueError("block-sum count must be a power of two >= 2")
    coeffs = np.empty((m, 2, 4), dtype=np.uint64)
    ch = np.empty((m, 4), dtype=np.uint64)
    folded = np.empty((world, 4), dtype=np.uint64) if want_folded else None
    total = np.empty(4, dtype=np.uint64) if want_total else None
    check(lib().zk_sumcheck_rounds_from_block_sums(
        transcript.h, _ptr(sums), world, m, _ptr(coeffs), _ptr(ch),
        _ptr(folded) if want_folded else None, _ptr(total) if want_total else None))
    return coeffs, ch, folded, total


class DeviceShard:
    """This rank's local table on its GPU (zk_shard in include/zkmi.h).  Sums land in torch device
    tensors so the collective can take them without a host round trip."""

    def __init__(self, local: MultiLinearPolynomial):
        import torch

        self._torch = torch
        self.local = local
        self.ctx = local.ctx
        h = C.c_void_p()
        check(lib().zk_shard_begin(self.ctx.h, local.h, C.byref(h)), self.ctx.h)
        self.h = h
        self._foreign = _foreign_stream(self.ctx)

    def _order(self) -> None:
        """The gathers and D2H copies of this module run on torch's current stream.  A context created on that stream
        (torch_stream_context) orders with them by itself; one that owns its stream is synchronized after every enqueue."""
        if self._foreign:
            self.ctx.synchronize()

    def vars_left(self) -> int:
        return lib().zk_shard_vars_left(self.h)

    def next_batch(self, after_binding: int = 0) -> int:
        """Rounds a pass should bind once `after_binding` more variables are bound (the context's schedule:
        wide passes while the table is large)."""
        return lib().zk_shard_next_batch(self.h, after_binding)

    def new_sums(self, count: int):
        return self._torch.empty((count, 4), dtype=self._torch.int64, device=self._torch.device("cuda", self.ctx.device_id))

    def block_sums(self, m: int, out) -> None:
        _wait_torch(self)
        check(lib().zk_shard_block_sums(self.h, m, C.c_void_p(out.data_ptr())), self.ctx.h)
        self._order()

    def fold(self, m: int, r: np.ndarray, m_next: int, out) -> None:
        r = np.ascontiguousarray(r, dtype=np.uint64)
        _wait_torch(self)
        check(lib().zk_shard_fold(self.h, m, _ptr(r), m_next, C.c_void_p(out.data_ptr())), self.ctx.h)
        self._order()

    def table(self):
        """The current local table as a (2^vars_left, 4) int64 device tensor (a copy, on the context stream)."""
        out = self._torch.empty((1 << self.vars_left(), 4), dtype=self._torch.int64,
                                device=self._torch.device("cuda", self.ctx.device_id))
        check(lib().zk_shard_copy_table(self.h, C.c_void_p(out.data_ptr())), self.ctx.h)
        self._order()
        return out

    def finish(self, table, transcript: Transcript):
        """All remaining rounds on this GPU alone: `table` is the whole remaining table (device tensor)."""
        from .sumcheck import SumcheckProver

        table = table.contiguous()
        _wait_torch(self)  # the gathered table was produced on torch's stream
        poly = MultiLinearPolynomial.wrap_device(table.data_ptr(), table.shape[0], ctx=self.ctx)
        return SumcheckProver.generate_round_polys(poly, transcript)

    def close(self):
        if getattr(self, "h", None):
            lib().zk_shard_end(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _foreign_stream(ctx: Context) -> bool:
    """True when the context's kernels do not run on torch's current stream of its device."""
    import torch

    return ctx.stream != torch.cuda.current_stream(torch.device("cuda", ctx.device_id)).cuda_stream


def _wait_torch(shard) -> None:
    """Work torch enqueued on its stream (a gathered table, a freshly allocated buffer's producer) before a context that
    owns its stream reads or overwrites it."""
    if shard._foreign:
        shard._torch.cuda.current_stream(shard._torch.device("cuda", shard.ctx.device_id)).synchronize()


_pinned = {}


def _to_host(t) -> np.ndarray:
    """Device tensor -> host uint64 view through a cached pinned buffer (one async copy + one stream sync; the
    pageable path of Tensor.cpu() stages and blocks twice).  The view is valid until the next call with the same
    shape: callers consume it at once."""
    import torch

    if not t.is_cuda:
        return t.contiguous().numpy().view(np.uint64)
    key = (tuple(t.shape), t.dtype, t.device.index)
    h = _pinned.get(key)
    if h is None:
        h = _pinned[key] = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h.numpy().view(np.uint64)


def gather_sums(local, world: int, dist=None, group=None) -> np.ndarray:
    """all-gather of one (k,4) int64 tensor per rank -> (world, k, 4) uint64 on the host.
    nccl: the collective runs on the device tensors (RCCL), one D2H of the gathered block;
    other backends (gloo in tests): D2H first, collective on host tensors."""
    import torch

    if world == 1 or dist is None:
        return _to_host(local)[None]
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return _to_host(out)
    host = local.cpu().contiguous()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=group)
    return torch.stack(parts).numpy().view(np.uint64)


def gather_tables(local, world: int, dist=None, group=None):
    """all-gather of the ranks' local tables (2^l, 4) -> the remaining table (world * 2^l, 4), entry j*world + q =
    rank q's entry j, on the device the local table lives on."""
    import torch

    if world == 1 or dist is None:
        return local
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    else:
        host = local.cpu().contiguous()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        out = torch.stack(parts).to(local.device)
    return out.permute(1, 0, 2).contiguous().reshape(-1, 4)


def generate_round_polys_sharded(shard, transcript: Transcript, world: int, dist=None, group=None,
                                 fold_batch: int | None = None, gather_log: int | None = DEFAULT_GATHER_LOG):
    """generate_round_polys (sumcheck/src/prover.rs:45-155, MLE branch) for a sharded table.
    Every rank returns the same (claim (4,), round_polys (n,2,4), challenges (n,4)).
    fold_batch: rounds per device pass; None = the shard's own schedule (DeviceShard.next_batch), which only
    depends on the local table size and the context settings, i.e. is the same on every rank.
    gather_log: once a rank's table is down to 2^gather_log entries the ranks all-gather their tables and every
    rank finishes the remaining rounds alone (replicated, no further exchange): small passes are bound by the
    latency of the exchange, not by the work.  None: sharded to the end."""
    g = world.bit_length() - 1
    if world != 1 << g:
        raise ValueError("world size must be a power of two")
    ell = shard.vars_left()
    n = ell + g
    own = fold_batch is None and hasattr(shard, "next_batch")
    M = max(1, min(fold_batch or DEFAULT_FOLD_BATCH, MAX_FOLD_BATCH))
    T = gather_log if (gather_log is not None and hasattr(shard, "finish")) else -1

    def batch(left: int, after_binding: int) -> int:
        return shard.next_batch(after_binding) if own else min(M, left)

    coeffs = np.empty((n, 2, 4), dtype=np.uint64)
    ch = np.empty((n, 4), dtype=np.uint64)

    def finish_gathered(k: int, claim):
        total, c, r = shard.finish(gather_tables(shard.table(), world, dist, group), transcript)
        coeffs[k:], ch[k:] = c, r
        return (total if k == 0 else claim), coeffs, ch

    if ell <= T:
        return finish_gathered(0, None)
    if ell < 1:
        raise ValueError("each rank needs at least 2 local entries")
    claim = None
    buf = shard.new_sums(1 << MAX_FOLD_BATCH)
    m, k = batch(ell, 0), 0
    shard.block_sums(m, buf)
    while True:
        sums = gather_sums(buf[: 1 << m], world, dist, group)
        last = ell == m
        c, r, folded, total = rounds_from_block_sums(transcript, sums, want_folded=last, want_total=k == 0)
        coeffs[k:k + m], ch[k:k + m] = c, r
        if k == 0:
            claim = total
        k += m
        if last:
            break
        if ell - m <= T:
            shard.fold(m, r, 0, buf)
            return finish_gathered(k, claim)
        m_next = batch(ell - m, m)
        shard.fold(m, r, m_next, buf)
        ell -= m
        m = m_next
    if g:  # the remaining g-variable table, entry q = rank q's fully folded local table
        c, r, _, _ = rounds_from_block_sums(transcript, folded.reshape(1, world, 4))
        coeffs[k:], ch[k:] = c, r
    return claim, coeffs, ch


def absorb_table_sharded(local: MultiLinearPolynomial, transcript: Transcript, world: int, dist=None, group=None,
                         chunk: int = 1 << 16) -> None:
    """transcript.append(&init_polynomial.to_bytes()) (prover.rs:162) for a sharded table: the byte
    stream is in GLOBAL index order, so chunks of every shard are gathered and interleaved.  Serial
    Keccak over 32*N bytes on every rank -- meant for parity tests, not for speed."""
    import torch

    n_local = len(local)
    base = lib().zk_mle_device_ptr(local.h)
    for off in range(0, n_local, chunk):
        cnt = min(chunk, n_local - off)
        if cnt & (cnt - 1):
            raise ValueError("chunk and local length must be powers of two")
        part = MultiLinearPolynomial.wrap_device(base + 32 * off, cnt, ctx=local.ctx)
        mine = torch.from_numpy(np.frombuffer(part.to_bytes(), dtype=np.uint8).reshape(cnt, 32).copy())
        if world == 1 or dist is None:
            allp = mine[None]
        else:
            if dist.get_backend(group) == "nccl":
                dev = mine.to(torch.device("cuda", local.ctx.device_id))
                outd = torch.empty((world, cnt, 32), dtype=torch.uint8, device=dev.device)
                dist.all_gather_into_tensor(outd, dev, group=group)
                allp = outd.cpu()
            else:
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine, group=group)
                allp = torch.stack(parts)
        transcript.append(allp.permute(1, 0, 2).contiguous().numpy().tobytes())


def torch_stream_context(local_rank: int) -> Context:
    """A context whose kernels run on torch's current stream of cuda:local_rank, so that they order
    with torch.distributed's collectives without extra synchronisation."""
    import torch

    torch.cuda.set_device(local_rank)
    return Context.on_stream(local_rank, torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------- degree-2 (SumPolynomial branch)
class DeviceShard4:
    """This rank's shards of the four factor tables of Sum[Prod[t0,t1], Prod[t2,t3]] (zk_shard4)."""

    def __init__(self, tables):
        import torch

        self._torch = torch
        self.tables = list(tables)
        self.ctx = self.tables[0].ctx
        arr = (C.c_void_p * 4)(*[t.h for t in self.tables])
        h = C.c_void_p()
        check(lib().zk_shard4_begin(self.ctx.h, arr, C.byref(h)), self.ctx.h)
        self.h = h
        self._foreign = _foreign_stream(self.ctx)

    def _order(self) -> None:  # as DeviceShard._order
        if self._foreign:
            self.ctx.synchronize()

    def local_len(self) -> int:
        return lib().zk_shard4_len(self.h)

    def new_buf(self):
        return self._torch.empty((4, 4), dtype=self._torch.int64, device=self._torch.device("cuda", self.ctx.device_id))

    def eval(self, out) -> None:
        _wait_torch(self)
        check(lib().zk_shard4_eval(self.h, C.c_void_p(out.data_ptr())), self.ctx.h)
        self._order()

    def fold(self, r: np.ndarray, out) -> None:
        r = np.ascontiguousarray(r, dtype=np.uint64)
        _wait_torch(self)
        check(lib().zk_shard4_fold(self.h, _ptr(r), C.c_void_p(out.data_ptr())), self.ctx.h)
        self._order()

    # two rounds per exchange
    def new_buf9(self):
        return self._torch.empty((9, 4), dtype=self._torch.int64, device=self._torch.device("cuda", self.ctx.device_id))

    def sum9(self, out) -> None:
        _wait_torch(self)
        check(lib().zk_shard4_sum9(self.h, C.c_void_p(out.data_ptr())), self.ctx.h)
        self._order()

    def fold2(self, r2: np.ndarray, out) -> None:
        r2 = np.ascontiguousarray(r2, dtype=np.uint64)
        _wait_torch(self)
        check(lib().zk_shard4_fold2(self.h, _ptr(r2), C.c_void_p(out.data_ptr()) if out is not None else None), self.ctx.h)
        self._order()

    def tables_tensor(self):
        """The four current local tables as one (4, len, 4) int64 device tensor (a copy, on the context stream)."""
        out = self._torch.empty((4, self.local_len(), 4), dtype=self._torch.int64,
                                device=self._torch.device("cuda", self.ctx.device_id))
        check(lib().zk_shard4_copy_tables(self.h, C.c_void_p(out.data_ptr())), self.ctx.h)
        self._order()
        return out

    def finish(self, tables, transcript: Transcript):
        """All remaining rounds on this GPU alone: `tables` (4, L, 4) are the whole remaining tables."""
        from .polynomials import ProductPolynomial, SumPolynomial
        from .sumcheck import SumcheckProver

        tables = tables.contiguous()
        _wait_torch(self)
        t = [MultiLinearPolynomial.wrap_device(tables[k].data_ptr(), tables.shape[1], ctx=self.ctx) for k in range(4)]
        f = SumPolynomial([ProductPolynomial(t[:2]), ProductPolynomial(t[2:])])
        proof, ch = SumcheckProver.generate_proof_for_partial_verify(np.zeros(4, dtype=np.uint64), f, transcript)
        return proof.round_polys, ch

    def close(self):
        if getattr(self, "h", None):
            lib().zk_shard4_end(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def round_from_evals(transcript: Transcript, evals: np.ndarray):
    """evals (world, 3, 4) -> (coefficients (3,4), challenge (4,)); zk_sumcheck_round_from_evals, host only."""
    evals = np.ascontiguousarray(evals, dtype=np.uint64)
    coeffs = np.empty((3, 4), dtype=np.uint64)
    r = np.empty(4, dtype=np.uint64)
    check(lib().zk_sumcheck_round_from_evals(transcript.h, _ptr(evals), evals.shape[0], _ptr(coeffs), _ptr(r)))
    return coeffs, r


def prove_sum2x2_host(transcript: Transcript, tables: np.ndarray):
    """tables (4, len, 4) on the host -> (coeffs (log2 len, 3, 4), challenges (log2 len, 4)); host only."""
    tables = np.ascontiguousarray(tables, dtype=np.uint64)
    ln = tables.shape[1]
    n = ln.bit_length() - 1
    coeffs = np.empty((max(n, 1), 3, 4), dtype=np.uint64)
    ch = np.empty((max(n, 1), 4), dtype=np.uint64)
    check(lib().zk_sumcheck_prove_sum2x2_host(transcript.h, _ptr(tables), ln, _ptr(coeffs), _ptr(ch)))
    return coeffs[:n], ch[:n]


def round_pair_from_sum9(transcript: Transcript, sums: np.ndarray):
    """sums (world, 9, 4) -> (coefficients (2,3,4), challenges (2,4)); zk_sumcheck_round_pair_from_sum9, host only."""
    sums = np.ascontiguousarray(sums, dtype=np.uint64)
    coeffs = np.empty((2, 3, 4), dtype=np.uint64)
    r = np.empty((2, 4), dtype=np.uint64)
    check(lib().zk_sumcheck_round_pair_from_sum9(transcript.h, _ptr(sums), sums.shape[0], _ptr(coeffs), _ptr(r)))
    return coeffs, r


def rounds_from_cross64(transcript: Transcript, cross: np.ndarray):
    """cross (world, 2, 8, 8, 4): the ranks' shares of P[term][e][e'] = sum_j F_e[j] G_e'[j] over the three top variables ->
    (coefficients (3,3,4), challenges (3,4)); zk_sumcheck_rounds_from_cross64, host only."""
    cross = np.ascontiguousarray(cross, dtype=np.uint64)
    coeffs = np.empty((3, 3, 4), dtype=np.uint64)
    r = np.empty((3, 4), dtype=np.uint64)
    check(lib().zk_sumcheck_rounds_from_cross64(transcript.h, _ptr(cross), cross.shape[0], _ptr(coeffs), _ptr(r)))
    return coeffs, r


def gather_tables4(local, world: int, dist=None, group=None):
    """all-gather of the ranks' four local tables (4, L, 4) -> the remaining tables (4, world * L, 4), entry
    j*world + q = rank q's entry j."""
    import torch

    if world == 1 or dist is None:
        return local
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    else:
        host = local.cpu().contiguous()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        out = torch.stack(parts).to(local.device)
    return out.permute(1, 2, 0, 3).contiguous().reshape(4, -1, 4)


def generate_round_polys_sum2x2_sharded(shard4, transcript: Transcript, world: int, dist=None, group=None,
                                        gather_log: int | None = DEFAULT_GATHER_LOG):
    """generate_round_polys (sumcheck/src/prover.rs:45-155, SumPolynomial branch) for four tables sharded on
    the variables bound last.  Every rank returns the same (round_polys (n,3,4), challenges (n,4)).
    gather_log (>= 1): two rounds per exchange (nine sums per rank) while a rank's tables have more than
    2^gather_log entries, then the ranks all-gather their tables and every rank finishes alone (replicated).
    None: one round per exchange (three evaluation shares), sharded to the end, host tail over the rank bits."""
    g = world.bit_length() - 1
    if world != 1 << g:
        raise ValueError("world size must be a power of two")
    ln = shard4.local_len()
    ell = ln.bit_length() - 1
    n = ell + g
    coeffs = np.empty((n, 3, 4), dtype=np.uint64)
    ch = np.empty((n, 4), dtype=np.uint64)
    if gather_log is not None and hasattr(shard4, "finish"):
        T = max(1, gather_log)
        k = 0
        if ell > T:
            buf = shard4.new_buf9()
            shard4.sum9(buf)
            while ell > T:  # ell >= 2
                sums = gather_sums(buf, world, dist, group)
                coeffs[k:k + 2], ch[k:k + 2] = round_pair_from_sum9(transcript, sums)
                ell -= 2
                shard4.fold2(ch[k:k + 2], buf if ell > T else None)
                k += 2
        c, r = shard4.finish(gather_tables4(shard4.tables_tensor(), world, dist, group), transcript)
        coeffs[k:], ch[k:] = c, r
        return coeffs, ch
    if ell < 1:
        raise ValueError("each rank needs at least 2 local entries per table")
    buf = shard4.new_buf()
    shard4.eval(buf)
    for k in range(ell):
        evals = gather_sums(buf[:3], world, dist, group)
        coeffs[k], ch[k] = round_from_evals(transcript, evals)
        shard4.fold(ch[k], buf)
    if g:
        rows = gather_sums(buf[:4], world, dist, group)           # (world, 4 tables, 4 limbs)
        c, r = prove_sum2x2_host(transcript, rows.transpose(1, 0, 2))
        coeffs[ell:], ch[ell:] = c, r
    return coeffs, ch


# ---------------------------------------------------------------- the same loops INSIDE the library (sharded.cpp)
# zk_sumcheck_prove_mle_sharded / zk_sumcheck_prove_sum2x2_sharded run the loops above in C++ with the exchange done by
# the library itself: RCCL (ncclAllGather on the context stream) or a host all-gather callback.  Python only hands over
# the unique id (or the callback); a Rust caller does the same through the C ABI and needs no torch.
_HOST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


def comm_init(ctx: Context, dist=None, group=None, transport: str | None = None) -> str:
    """Make `ctx` this process's rank of the in-library sharded provers.  `dist`: an initialised torch.distributed
    (None: a single rank).  transport "rccl": rank 0's ncclUniqueId is broadcast through `dist`, every rank joins an
    RCCL communicator owned by the library (default with the nccl backend, and for a single rank); "host": the
    library's host all-gather callback goes through `dist` on CPU tensors (default with gloo).  Returns the transport."""
    import torch

    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    backend = dist.get_backend(group) if dist is not None else None
    if transport is None:
        transport = "rccl" if backend in (None, "nccl") else "host"
    if transport == "rccl":
        # Every rank probes RCCL locally first (zk_comm_unique_id resolves librccl.so.1 and makes an id; only rank 0's is used) and
        # the ranks vote BEFORE any id exchange: a rank that cannot load RCCL must not leave its peers waiting in a broadcast or
        # inside ncclCommInitRank's rendezvous.  All ranks raise together or go on together.
        ident = np.zeros(128, dtype=np.uint8)
        rc = lib().zk_comm_unique_id(ident.ctypes.data_as(C.POINTER(C.c_uint8)))
        if world > 1:
            dev = torch.device("cuda", ctx.device_id) if backend == "nccl" else torch.device("cpu")
            flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0:
                check(rc, ctx.h)       # this rank's own failure, with the library's text
                raise RuntimeError("zk_comm_unique_id failed on another rank: no RCCL communicator on any rank")
            t = torch.from_numpy(ident).to(dev)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = t.cpu().numpy()
        else:
            check(rc, ctx.h)
        check(lib().zk_comm_init_rccl(ctx.h, ident.ctypes.data_as(C.POINTER(C.c_uint8)), rank, world), ctx.h)
    elif transport == "host":
        if backend == "nccl":
            raise ValueError("the host transport needs a backend that gathers CPU tensors (gloo)")

        def gather(_user, send, recv, nbytes):
            try:
                src = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(send))).clone()
                dst = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * (nbytes * world)).from_address(recv)))
                if world == 1:
                    dst.copy_(src)
                else:
                    dist.all_gather(list(dst.view(world, nbytes).unbind(0)), src, group=group)
                return 0
            except Exception as e:  # an exception must not unwind through the C caller
                import sys
                print(f"zk_amd.sharded: host all-gather failed: {e!r}", file=sys.stderr)
                return 1

        cb = _HOST_FN(gather)
        ctx._comm_cb = cb  # the library keeps the pointer: keep the object alive as long as the context
        check(lib().zk_comm_init_host(ctx.h, rank, world, C.cast(cb, C.c_void_p), None), ctx.h)
    else:
        raise ValueError("transport: 'rccl' or 'host'")
    return transport


def comm_destroy(ctx: Context) -> None:
    lib().zk_comm_destroy(ctx.h)
    ctx._comm_cb = None


def _gather_arg(gather_log):
    return -1 if gather_log is None else int(gather_log)


def prove_mle_sharded(local: MultiLinearPolynomial, transcript: Transcript | None = None, absorb_table: bool = False,
                      gather_log: int | None = DEFAULT_GATHER_LOG):
    """zk_sumcheck_prove_mle_sharded on this rank's shard: (claim (4,), round_polys (n,2,4), challenges (n,4)), the same on
    every rank.  transcript None = generate_sumcheck_proof (fresh transcript + absorb of the whole table's bytes)."""
    ctx = local.ctx
    world = lib().zk_comm_world(ctx.h)
    if world < 1:
        raise RuntimeError("comm_init(ctx, ...) first")
    n = len(local).bit_length() - 1 + world.bit_length() - 1
    claim = np.empty(4, dtype=np.uint64)
    coeffs = np.empty((max(n, 1), 2, 4), dtype=np.uint64)
    ch = np.empty((max(n, 1), 4), dtype=np.uint64)
    check(lib().zk_sumcheck_prove_mle_sharded(ctx.h, local.h, int(absorb_table), transcript.h if transcript is not None else None,
                                              _gather_arg(gather_log), _ptr(claim), _ptr(coeffs), _ptr(ch)), ctx.h)
    return claim, coeffs[:n], ch[:n]


def prove_sum2x2_sharded(tables, transcript: Transcript, claim=None, gather_log: int | None = DEFAULT_GATHER_LOG):
    """zk_sumcheck_prove_sum2x2_sharded on this rank's shards of the four factor tables: (round_polys (n,3,4), challenges (n,4))."""
    ctx = tables[0].ctx
    world = lib().zk_comm_world(ctx.h)
    if world < 1:
        raise RuntimeError("comm_init(ctx, ...) first")
    n = len(tables[0]).bit_length() - 1 + world.bit_length() - 1
    claim = np.zeros(4, dtype=np.uint64) if claim is None else np.ascontiguousarray(claim, dtype=np.uint64)
    coeffs = np.empty((max(n, 1), 3, 4), dtype=np.uint64)
    ch = np.empty((max(n, 1), 4), dtype=np.uint64)
    arr = (C.c_void_p * 4)(*[t.h for t in tables])
    check(lib().zk_sumcheck_prove_sum2x2_sharded(ctx.h, arr, _ptr(claim), transcript.h, _gather_arg(gather_log), _ptr(coeffs), _ptr(ch)), ctx.h)
    return coeffs[:n], ch[:n]
