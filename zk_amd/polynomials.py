"""Host-side mirror of the reference's `polynomials` crate for the sumcheck path: the
evaluation table lives in HBM in Montgomery form (zk_mle), every method is one C-ABI call.

  MultiLinearPolynomial  polynomials/src/multilinear_polynomial/evaluation_form.rs:17-313
  ProductPolynomial      polynomials/src/product_polynomial.rs:7-78
  SumPolynomial          polynomials/src/sum_polynomial.rs:7-82

Field elements are numpy uint64 arrays of 4 little-endian Montgomery limbs (ark-ff's
in-memory form).  Where the reference panics, ZkError is raised with the matching status.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import ZkError, check, lib, u64p, u8p
from .context import Context, default_context


def _limbs(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


class MultiLinearPolynomial:
    def __init__(self, evaluation_points=None, ctx: Context | None = None, _handle=None):
        """MultiLinearPolynomial::new (:22-32): copies (n,4) Montgomery limbs to the device."""
        self.ctx = ctx or default_context()
        if _handle is not None:
            self.h = _handle
            return
        pts = _limbs(evaluation_points).reshape(-1, 4)
        h = C.c_void_p()
        if pts.shape[0] == 0:
            raise ZkError(1)
        check(lib().zk_mle_upload_mont(self.ctx.h, pts.ctypes.data_as(u64p), pts.shape[0], C.byref(h)), self.ctx.h)
        self.h = h

    # ---- constructors without a reference counterpart
    @classmethod
    def from_le32(cls, data: bytes, ctx: Context | None = None):
        ctx = ctx or default_context()
        if len(data) == 0 or len(data) % 32:
            raise ZkError(1)
        buf = np.frombuffer(data, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().zk_mle_upload_le32(ctx.h, buf.ctypes.data_as(u8p), len(data) // 32, C.byref(h)), ctx.h)
        return cls(ctx=ctx, _handle=h)

    @classmethod
    def generate(cls, seed: int, length: int, first: int = 0, ctx: Context | None = None, step: int = 1):
        """Synthetic table made on the device; local entry j = global entry first + j*step."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib().zk_mle_generate_strided(ctx.h, seed, first, step, length, C.byref(h)), ctx.h)
        return cls(ctx=ctx, _handle=h)

    @classmethod
    def upload_shard(cls, evaluation_points, rank: int, world: int, ctx: Context | None = None):
        """Shard `rank` of `world` of a host table: local entry j = global entry j*world + rank."""
        ctx = ctx or default_context()
        pts = _limbs(evaluation_points).reshape(-1, 4)
        h = C.c_void_p()
        check(lib().zk_mle_upload_mont_shard(ctx.h, pts.ctypes.data_as(u64p), pts.shape[0], rank, world, C.byref(h)), ctx.h)
        return cls(ctx=ctx, _handle=h)

    @classmethod
    def wrap_device(cls, device_ptr: int, length: int, ctx: Context | None = None):
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib().zk_mle_wrap_device(ctx.h, C.c_void_p(device_ptr), length, C.byref(h)), ctx.h)
        return cls(ctx=ctx, _handle=h)

    # ---- reference API
    def __len__(self):
        return lib().zk_mle_len(self.h)

    def number_of_variables(self) -> int:      # :159-161
        return len(self).bit_length() - 1

    def get_evaluation_points(self) -> np.ndarray:  # :155-157
        out = np.empty((len(self), 4), dtype=np.uint64)
        check(lib().zk_mle_download_mont(self.h, out.ctypes.data_as(u64p)), self.ctx.h)
        return out

    def clone(self):
        h = C.c_void_p()
        check(lib().zk_mle_clone(self.h, C.byref(h)), self.ctx.h)
        return MultiLinearPolynomial(ctx=self.ctx, _handle=h)

    def partially_evaluate(self, variable):    # :163-192, variable = (index, value)
        idx, r = variable
        h = C.c_void_p()
        check(lib().zk_mle_partially_evaluate(self.h, idx, _limbs(r).ctypes.data_as(u64p), C.byref(h)), self.ctx.h)
        return MultiLinearPolynomial(ctx=self.ctx, _handle=h)

    def compute_quotient_remainder(self, divisor, variable_index: int):   # :280-301 -> (quotient values, remainder)
        q, r = C.c_void_p(), C.c_void_p()
        check(lib().zk_mle_quotient_remainder(self.h, variable_index, _limbs(divisor).ctypes.data_as(u64p), C.byref(q), C.byref(r)), self.ctx.h)
        quotient = MultiLinearPolynomial(ctx=self.ctx, _handle=q)
        return quotient.get_evaluation_points(), MultiLinearPolynomial(ctx=self.ctx, _handle=r)

    def minus(self, other):                    # :260-272: F(x) - V
        h = C.c_void_p()
        check(lib().zk_mle_minus(self.h, _limbs(other).ctypes.data_as(u64p), C.byref(h)), self.ctx.h)
        return MultiLinearPolynomial(ctx=self.ctx, _handle=h)

    def evaluate(self, points):                # :194-223, points: list of None | limbs
        has = np.array([0 if p is None else 1 for p in points] or [0], dtype=np.uint8)
        vals = np.zeros((max(len(points), 1), 4), dtype=np.uint64)
        for i, p in enumerate(points):
            if p is not None:
                vals[i] = _limbs(p)
        h = C.c_void_p()
        check(lib().zk_mle_evaluate(self.h, has.ctypes.data_as(u8p), vals.ctypes.data_as(u64p), len(points),
                                    C.byref(h)), self.ctx.h)
        return MultiLinearPolynomial(ctx=self.ctx, _handle=h)

    def to_bytes(self) -> bytes:               # :225-234
        out = np.empty(32 * len(self), dtype=np.uint8)
        check(lib().zk_mle_to_le32(self.h, out.ctypes.data_as(u8p)), self.ctx.h)
        return out.tobytes()

    def evaluation_sum(self) -> np.ndarray:    # :236-238
        out = np.empty(4, dtype=np.uint64)
        check(lib().zk_mle_evaluation_sum(self.h, out.ctypes.data_as(u64p)), self.ctx.h)
        return out

    def _binary(self, fn, other):
        h = C.c_void_p()
        check(fn(self.h, other.h, C.byref(h)), self.ctx.h)
        return MultiLinearPolynomial(ctx=self.ctx, _handle=h)

    @staticmethod
    def w_add(w_b, w_c):                       # :303-308
        return w_b._binary(lib().zk_mle_w_add, w_c)

    def w_mul(self, other):                    # :310-312 -- operate_w(other, self, Mul)
        return self._binary(lib().zk_mle_w_mul, other)

    def scalar_mul(self, scalar):              # :151-153
        h = C.c_void_p()
        check(lib().zk_mle_scalar_mul(self.h, _limbs(scalar).ctypes.data_as(u64p), C.byref(h)), self.ctx.h)
        return MultiLinearPolynomial(ctx=self.ctx, _handle=h)

    def __add__(self, other):                  # _add / impl Add :241-258, :303-313
        return self._binary(lib().zk_mle_add, other)

    def round_sums(self) -> np.ndarray:
        """(e0, e1) of sumcheck/src/prover.rs:108-116 as a (2,4) array."""
        out = np.empty((2, 4), dtype=np.uint64)
        check(lib().zk_mle_round_sums(self.h, out.ctypes.data_as(u64p)), self.ctx.h)
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().zk_mle_free(self.h)
                self.h = None
        except Exception:
            pass


class ProductPolynomial:
    def __init__(self, polys):                 # product_polynomial.rs:12-22
        polys = list(polys)
        if not polys:
            raise ZkError(3)
        n = len(polys[0])
        for p in polys:
            if len(p) != n:
                raise ZkError(3)               # "All polynomials must have the same length"
        self.polys = polys

    def length(self) -> int:                   # :27-29
        return len(self.polys[0])


class SumPolynomial:
    def __init__(self, prod_polys):            # sum_polynomial.rs:12-22
        prod_polys = list(prod_polys)
        if not prod_polys:
            raise ZkError(3)
        n = prod_polys[0].length()
        for p in prod_polys:
            if p.length() != n:
                raise ZkError(3)
        self.prod_polys = prod_polys

    def length(self) -> int:                   # :71-73
        return self.prod_polys[0].length()

    def number_of_variables(self) -> int:      # :75-77
        return self.length().bit_length() - 1

    def degree(self) -> int:                   # :79-81 -- the number of product terms
        return len(self.prod_polys)
