"""zk_fold_block_sums_noalloc -- the entry point that allocates nothing (static __device__ scratch) -- against
zk_shard_fold, the existing entry point it is a variant of, through the C ABI: folded table and block sums bit-exact,
on calls in a row that shrink (the array keeps the previous call's contents), and against the oracle on a small table."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import zk_amd
from zk_amd._lib import check, lib, u64p

pytestmark = pytest.mark.gpu
SEED = 0x5EED0000


def _dl(ctx, ptr, n):
    m = zk_amd.MultiLinearPolynomial.wrap_device(ptr, n, ctx=ctx)
    return m.get_evaluation_points()


def _shard_fold(ctx, table, m, r, m_next):
    """the existing entry point: (folded table, block sums)"""
    L = lib()
    sh = C.c_void_p()
    check(L.zk_shard_begin(ctx.h, table.h, C.byref(sh)), ctx.h)
    try:
        n_out = len(table) >> m
        sums = zk_amd.MultiLinearPolynomial.generate(1, 1 << m_next, ctx=ctx) if m_next else None
        check(L.zk_shard_fold(sh, m, r.ctypes.data_as(u64p), m_next, C.c_void_p(L.zk_mle_device_ptr(sums.h)) if sums else None), ctx.h)
        dst = zk_amd.MultiLinearPolynomial.generate(2, n_out, ctx=ctx)
        check(L.zk_shard_copy_table(sh, C.c_void_p(L.zk_mle_device_ptr(dst.h))), ctx.h)
        ctx.synchronize()
        return dst.get_evaluation_points(), (sums.get_evaluation_points() if sums else None)
    finally:
        L.zk_shard_end(sh)


def _noalloc_fold(ctx, table, m, r, m_next):
    L = lib()
    n_out = len(table) >> m
    dst = zk_amd.MultiLinearPolynomial.generate(3, n_out, ctx=ctx)            # caller-owned output buffers, filled with junk
    sums = zk_amd.MultiLinearPolynomial.generate(4, 1 << m_next, ctx=ctx) if m_next else None
    ctx.synchronize()
    rc = L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), C.c_void_p(L.zk_mle_device_ptr(table.h)), len(table), m, r.ctypes.data_as(u64p),
                                      C.c_void_p(L.zk_mle_device_ptr(dst.h)), m_next, C.c_void_p(L.zk_mle_device_ptr(sums.h)) if sums else None)
    check(rc)
    ctx.synchronize()
    return dst.get_evaluation_points(), (sums.get_evaluation_points() if sums else None)


def test_noalloc_fold_matches_shard_fold_on_shrinking_calls():
    ctx = zk_amd.Context(0)
    # (log_len, m, m_next): a wide LDS-DMA pass with scalar-load weights from the scratch array, then smaller inputs through the
    # other kernel forms (weights in kernel arguments; one wave per output; weights per workgroup in LDS), each after a larger call
    for log_len, m, m_next in ((25, 8, 4), (20, 4, 4), (16, 8, 8), (23, 6, 0), (12, 3, 2), (9, 8, 1), (5, 1, 0), (1, 1, 0)):
        table = zk_amd.MultiLinearPolynomial.generate(SEED + log_len, 1 << log_len, ctx=ctx)
        r = O.gen_table(SEED + 77 + log_len, 0, m)
        want_t, want_s = _shard_fold(ctx, table, m, r, m_next)
        got_t, got_s = _noalloc_fold(ctx, table, m, r, m_next)
        assert np.array_equal(got_t, want_t), (log_len, m, m_next)
        if m_next:
            assert np.array_equal(got_s, want_s), (log_len, m, m_next)
        del table


def test_noalloc_fold_vs_oracle_and_errors():
    ctx = zk_amd.Context(0)
    L = lib()
    t = O.gen_table(SEED + 5, 0, 1 << 14)
    table = zk_amd.MultiLinearPolynomial(t, ctx=ctx)
    r = O.gen_table(SEED + 6, 0, 5)
    got_t, got_s = _noalloc_fold(ctx, table, 5, r, 3)
    want = t
    for k in range(5):
        want = O.mle_partially_evaluate(want, 0, r[k])
    assert np.array_equal(got_t, want)
    blk = len(want) // 8
    for b in range(8):
        assert list(got_s[b]) == list(O.mle_sum(want[b * blk:(b + 1) * blk]))
    assert np.array_equal(table.get_evaluation_points(), t)  # the input is left unmodified
    # sizes beyond the stated maximum are refused before anything is launched
    p = C.c_void_p(L.zk_mle_device_ptr(table.h))
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 1 << 31, 8, r.ctypes.data_as(u64p), p, 0, None) == 11
    assert b"maximum" in L.zk_strerror(11)
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 3 << 10, 2, r.ctypes.data_as(u64p), p, 0, None) == 1   # not a power of two
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 1 << 14, 9, r.ctypes.data_as(u64p), p, 0, None) == 8
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 1 << 14, 4, r.ctypes.data_as(u64p), p, 2, None) == 8   # sums asked for, no buffer
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), None, 1 << 14, 4, r.ctypes.data_as(u64p), p, 0, None) == 8
    # ranges that intersect are refused (the kernels stream src while storing dst): dst inside src, dst == src, sums inside dst
    base = L.zk_mle_device_ptr(table.h)
    other = zk_amd.MultiLinearPolynomial(t, ctx=ctx)
    q = C.c_void_p(L.zk_mle_device_ptr(other.h))
    rp = r.ctypes.data_as(u64p)
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 1 << 14, 4, rp, p, 0, None) == 8
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 1 << 14, 4, rp, C.c_void_p(base + (1 << 13) * 32), 0, None) == 8
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 1 << 14, 4, rp, q, 2, C.c_void_p(L.zk_mle_device_ptr(other.h) + 64)) == 8
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 1 << 14, 4, rp, q, 2, C.c_void_p(base + 32)) == 8
    # dst directly behind src (adjacent, not intersecting) is fine
    assert L.zk_fold_block_sums_noalloc(C.c_void_p(ctx.stream), p, 1 << 13, 4, rp, C.c_void_p(base + (1 << 13) * 32), 0, None) == 0
    ctx.synchronize()
