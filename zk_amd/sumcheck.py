"""Host-side mirror of the reference's `sumcheck` crate (prover + verifier + proof).

  SumCheckProof     sumcheck/src/sumcheck_protocol.rs:5-9
  SumcheckProver    sumcheck/src/prover.rs:17-191
  SumcheckVerifier  sumcheck/src/verifier.rs:9-95
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import ZkError, check, lib, u64p
from .polynomials import MultiLinearPolynomial, SumPolynomial
from .transcript import Transcript


@dataclass
class SumCheckProof:
    initial_claim_sum: np.ndarray   # (4,) Montgomery limbs
    round_polys: np.ndarray         # (n_rounds, n_coeffs, 4): UnivariatePolynomial coefficients, ascending

    def to_bytes_le32(self, to_le32) -> bytes:
        """struct order, LE32 per element (to_le32: limbs -> 32 bytes)."""
        out = bytearray(to_le32(self.initial_claim_sum))
        for g in self.round_polys:
            for c in g:
                out += to_le32(c)
        return bytes(out)


class SumcheckProver:
    @staticmethod
    def generate_sumcheck_proof(init_polynomial: MultiLinearPolynomial, return_challenges: bool = False):
        """prover.rs:158-173.  The table is absorbed into a fresh transcript first."""
        n = init_polynomial.number_of_variables()
        claim = np.empty(4, dtype=np.uint64)
        coeffs = np.empty((max(n, 1), 2, 4), dtype=np.uint64)
        ch = np.empty((max(n, 1), 4), dtype=np.uint64)
        ctx = init_polynomial.ctx
        check(lib().zk_sumcheck_prove_mle(ctx.h, init_polynomial.h, 1, None, claim.ctypes.data_as(u64p),
                                          coeffs.ctypes.data_as(u64p), ch.ctypes.data_as(u64p)), ctx.h)
        proof = SumCheckProof(claim, coeffs[:n])
        return (proof, ch[:n]) if return_challenges else proof

    @staticmethod
    def generate_round_polys(polynomial: MultiLinearPolynomial, transcript: Transcript):
        """prover.rs:45-155, MultilinearPolynomial branch, on the caller's transcript
        (no table absorb).  -> (claim, round_polys, challenges)"""
        n = polynomial.number_of_variables()
        claim = np.empty(4, dtype=np.uint64)
        coeffs = np.empty((max(n, 1), 2, 4), dtype=np.uint64)
        ch = np.empty((max(n, 1), 4), dtype=np.uint64)
        ctx = polynomial.ctx
        check(lib().zk_sumcheck_prove_mle(ctx.h, polynomial.h, 0, transcript.h, claim.ctypes.data_as(u64p),
                                          coeffs.ctypes.data_as(u64p), ch.ctypes.data_as(u64p)), ctx.h)
        return claim, coeffs[:n], ch[:n]

    @staticmethod
    def generate_proof_for_partial_verify(initial_claim_sum, init_poly: SumPolynomial, transcript: Transcript):
        """prover.rs:175-190 -> (SumCheckProof, random_points).  Only the shape the reference's
        degree-2 branch can express: 2 product terms (F9) of 2 factors each."""
        if init_poly.degree() != 2 or any(len(p.polys) != 2 for p in init_poly.prod_polys):
            raise ZkError(5)
        tabs = [t for p in init_poly.prod_polys for t in p.polys]
        ctx = tabs[0].ctx
        n = init_poly.number_of_variables()
        coeffs = np.empty((max(n, 1), 3, 4), dtype=np.uint64)
        ch = np.empty((max(n, 1), 4), dtype=np.uint64)
        claim = np.ascontiguousarray(initial_claim_sum, dtype=np.uint64)
        arr = (C.c_void_p * 4)(*[t.h for t in tabs])
        check(lib().zk_sumcheck_prove_sum2x2(ctx.h, arr, claim.ctypes.data_as(u64p), transcript.h,
                                             coeffs.ctypes.data_as(u64p), ch.ctypes.data_as(u64p)), ctx.h)
        return SumCheckProof(claim.copy(), coeffs[:n]), ch[:n]


    @staticmethod
    def generate_proof_product_plus_linear(initial_claim_sum, f: MultiLinearPolynomial, g: MultiLinearPolynomial, h: MultiLinearPolynomial,
                                           transcript: Transcript):
        """generate_proof_for_partial_verify (prover.rs:175-190) for SumPolynomial[Product[f, g], Product[h, ones]] -- the relation
        f g + h of a GKR phase -- without the table of ones: same proof bytes, the constant factor is never read."""
        ctx = f.ctx
        n = f.number_of_variables()
        coeffs = np.empty((max(n, 1), 3, 4), dtype=np.uint64)
        ch = np.empty((max(n, 1), 4), dtype=np.uint64)
        claim = np.ascontiguousarray(initial_claim_sum, dtype=np.uint64)
        arr = (C.c_void_p * 3)(f.h, g.h, h.h)
        check(lib().zk_sumcheck_prove_prod_plus_linear(ctx.h, arr, claim.ctypes.data_as(u64p), transcript.h,
                                                       coeffs.ctypes.data_as(u64p), ch.ctypes.data_as(u64p)), ctx.h)
        return SumCheckProof(claim.copy(), coeffs[:n]), ch[:n]


class SumcheckVerifier:
    @staticmethod
    def partial_verify(proof: SumCheckProof, transcript: Transcript):
        """verifier.rs:29-62 -> (ok, final_claim_sum, challenges)"""
        rp = np.ascontiguousarray(proof.round_polys, dtype=np.uint64)
        n_rounds = rp.shape[0]
        ncoef = rp.shape[1] if n_rounds else 2
        ok = C.c_int(0)
        final = np.empty(4, dtype=np.uint64)
        ch = np.empty((max(n_rounds, 1), 4), dtype=np.uint64)
        claim = np.ascontiguousarray(proof.initial_claim_sum, dtype=np.uint64)
        rpp = rp if n_rounds else np.zeros((1, 2, 4), dtype=np.uint64)
        check(lib().zk_sumcheck_partial_verify(claim.ctypes.data_as(u64p), rpp.ctypes.data_as(u64p), n_rounds, ncoef,
                                               transcript.h, C.byref(ok), final.ctypes.data_as(u64p),
                                               ch.ctypes.data_as(u64p)))
        return bool(ok.value), final, ch[:n_rounds]

    @staticmethod
    def verify_proof(initial_polynomial: MultiLinearPolynomial, proof: SumCheckProof) -> bool:
        """verifier.rs:64-94"""
        rp = np.ascontiguousarray(proof.round_polys, dtype=np.uint64)
        n_rounds = rp.shape[0]
        if n_rounds and rp.shape[1] != 2:
            raise ZkError(8)
        ok = C.c_int(0)
        claim = np.ascontiguousarray(proof.initial_claim_sum, dtype=np.uint64)
        rpp = rp if n_rounds else np.zeros((1, 2, 4), dtype=np.uint64)
        ctx = initial_polynomial.ctx
        check(lib().zk_sumcheck_verify_mle(ctx.h, initial_polynomial.h, claim.ctypes.data_as(u64p),
                                           rpp.ctypes.data_as(u64p), n_rounds, C.byref(ok)), ctx.h)
        return bool(ok.value)
