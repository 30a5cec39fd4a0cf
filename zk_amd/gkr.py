"""Host-side mirror of the reference's `gkr` crate, prover side (gkr/src/prover.rs:24-154,
gkr/src/gkr_protocol.rs:8-26).  generate_proof is ONE C-ABI call (zk_gkr_prove): circuit evaluation,
predicate folding and every layer's sumcheck run on the device from the gate lists.
generate_proof_dense follows the reference line by line over device tables, building the dense
add_i/mul_i tables -- small circuits only; the parity tests check both against the oracle."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from ._lib import check, lib, u64p
from .circuit import Circuit
from .polynomials import MultiLinearPolynomial, ProductPolynomial, SumPolynomial
from .sumcheck import SumCheckProof, SumcheckProver
from .transcript import Transcript


@dataclass
class GKRProof:                                            # gkr_protocol.rs:8-12
    output_poly: np.ndarray                                # (len(W_0), 4) Montgomery limbs
    w_polys_evals: list                                    # [(W_i(r_b), W_i(r_c))] for layers 1..
    sumcheck_proofs: list                                  # [SumCheckProof] per layer, output layer first
    layer_ms: list = field(default_factory=list)           # per layer device ms, then evaluate ms, W_0 absorb ms (not part of the proof)

    def elements(self) -> np.ndarray:
        """every field element in struct order: output_poly | (wb, wc) pairs | per layer claim + coefficients"""
        parts = [np.asarray(self.output_poly, dtype=np.uint64).reshape(-1, 4)]
        for wb, wc in self.w_polys_evals:
            parts += [np.asarray(wb).reshape(1, 4), np.asarray(wc).reshape(1, 4)]
        for p in self.sumcheck_proofs:
            parts += [np.asarray(p.initial_claim_sum).reshape(1, 4), np.asarray(p.round_polys).reshape(-1, 4)]
        return np.concatenate(parts).astype(np.uint64)


def _limbs_le32(limbs: np.ndarray, ctx) -> bytes:
    """LE32 of Montgomery elements through the device (to_bytes)."""
    n = limbs.shape[0]
    pad = 1 << max(n - 1, 0).bit_length()
    buf = np.zeros((pad, 4), dtype=np.uint64)
    buf[:n] = limbs
    return MultiLinearPolynomial(buf, ctx=ctx).to_bytes()[: 32 * n]


class GKRProver:
    @staticmethod
    def generate_proof(circuit: Circuit, transcript: Transcript, inputs, timing: bool = False, sharded_gather_log: int | None = None) -> GKRProof:
        """sharded_gather_log: None = this GPU alone (zk_gkr_prove); an integer = this process is a rank of the context's
        communicator (zk_amd.sharded.comm_init) and every layer's sumcheck is sharded over the ranks (zk_gkr_prove_sharded)."""
        ctx = circuit.ctx
        if not isinstance(inputs, MultiLinearPolynomial):
            inputs = MultiLinearPolynomial(inputs, ctx=ctx)
        out_len, rounds = circuit.shape()
        n_layers = len(rounds)
        output = np.empty((out_len, 4), dtype=np.uint64)
        w_evals = np.empty((max(n_layers - 1, 1), 2, 4), dtype=np.uint64)
        claims = np.empty((n_layers, 4), dtype=np.uint64)
        coeffs = np.empty((max(sum(rounds), 1), 3, 4), dtype=np.uint64)
        ms = (C.c_double * (n_layers + 2))()
        if sharded_gather_log is None:
            check(lib().zk_gkr_prove(ctx.h, circuit.h, transcript.h, inputs.h, output.ctypes.data_as(u64p),
                                     w_evals.ctypes.data_as(u64p), claims.ctypes.data_as(u64p), coeffs.ctypes.data_as(u64p),
                                     ms if timing else None), ctx.h)
        else:
            check(lib().zk_gkr_prove_sharded(ctx.h, circuit.h, transcript.h, inputs.h, int(sharded_gather_log), output.ctypes.data_as(u64p),
                                             w_evals.ctypes.data_as(u64p), claims.ctypes.data_as(u64p), coeffs.ctypes.data_as(u64p),
                                             ms if timing else None), ctx.h)
        proofs, off = [], 0
        for i, r in enumerate(rounds):
            proofs.append(SumCheckProof(claims[i].copy(), coeffs[off:off + r].copy()))
            off += r
        return GKRProof(output, [(w_evals[i, 0].copy(), w_evals[i, 1].copy()) for i in range(n_layers - 1)], proofs,
                        [ms[i] for i in range(n_layers + 2)] if timing else [])

    @staticmethod
    def generate_proof_dense(circuit: Circuit, transcript: Transcript, inputs) -> GKRProof:
        """gkr/src/prover.rs:24-154 statement by statement on device tables (dense predicates)."""
        ctx = circuit.ctx
        evals = circuit.evaluate_at_input(inputs)                                     # :30
        running = Circuit.get_w_i(0, evals)                                           # :43
        transcript.append(running.to_bytes())                                         # :46
        random_values = list(transcript.sample_n_challenges(running.number_of_variables()))  # :49-53
        w_polys_evals, proofs = [], []
        for layer_idx in range(circuit.get_layer_count()):
            mul_abc, add_abc = circuit.get_mul_i(layer_idx), circuit.get_add_i(layer_idx)   # :61-62
            nv = mul_abc.number_of_variables()
            if layer_idx == 0:                                                        # :65-79
                pts = [random_values[i] if i < len(random_values) else None for i in range(nv)]  # utils.rs:45-65
                new_mul, new_add = mul_abc.evaluate(pts), add_abc.evaluate(pts)
                claim = running.evaluate(random_values).get_evaluation_points()[0]
            else:                                                                     # :81-122
                h = len(random_values) // 2
                r_b, r_c = random_values[:h], random_values[h:]
                wb = running.evaluate(r_b).get_evaluation_points()[0]
                wc = running.evaluate(r_c).get_evaluation_points()[0]
                transcript.append(_limbs_le32(np.stack([wb, wc]), ctx))               # :102-105
                alpha, beta = transcript.sample_challenge(), transcript.sample_challenge()
                pb = [r_b[i] if i < h else None for i in range(nv)]                  # utils.rs:28-34
                pc = [r_c[i] if i < h else None for i in range(nv)]
                new_mul = mul_abc.evaluate(pb).scalar_mul(alpha) + mul_abc.evaluate(pc).scalar_mul(beta)
                new_add = add_abc.evaluate(pb).scalar_mul(alpha) + add_abc.evaluate(pc).scalar_mul(beta)
                w_polys_evals.append((wb, wc))
                # alpha*W(r_b) + beta*W(r_c) (utils.rs:5-14), computed on the device as a 2-entry table
                two = MultiLinearPolynomial(np.stack([wb, wc]), ctx=ctx)
                a = two.scalar_mul(alpha).get_evaluation_points()[0:1]
                b = two.scalar_mul(beta).get_evaluation_points()[1:2]
                s = MultiLinearPolynomial(np.concatenate([a, b]), ctx=ctx).evaluation_sum()
                claim = s
            next_w = Circuit.get_w_i(layer_idx + 1, evals)                            # :125
            f = SumPolynomial([ProductPolynomial([new_mul, MultiLinearPolynomial.w_mul(next_w, next_w)]),   # :128-137
                               ProductPolynomial([new_add, MultiLinearPolynomial.w_add(next_w, next_w)])])
            proof, pts = SumcheckProver.generate_proof_for_partial_verify(claim, f, transcript)  # :140-141
            random_values = list(pts)                                                 # :143
            running = next_w
            proofs.append(proof)
        return GKRProof(Circuit.get_w_i(0, evals).get_evaluation_points(), w_polys_evals, proofs)


class GKRVerifier:
    @staticmethod
    def verify_proof(initial_inputs, circuit: Circuit, transcript: Transcript, proof: GKRProof) -> bool:
        """gkr/src/verifier.rs:21-119 (zk_gkr_verify): predicates evaluated from the gate lists on the device."""
        ctx = circuit.ctx
        if not isinstance(initial_inputs, MultiLinearPolynomial):
            initial_inputs = MultiLinearPolynomial(initial_inputs, ctx=ctx)
        n_layers = circuit.get_layer_count()
        output = np.ascontiguousarray(proof.output_poly, dtype=np.uint64).reshape(-1, 4)
        w = np.zeros((max(n_layers - 1, 1), 2, 4), dtype=np.uint64)
        for i, (wb, wc) in enumerate(proof.w_polys_evals[: n_layers - 1]):
            w[i, 0], w[i, 1] = wb, wc
        claims = np.ascontiguousarray(np.stack([p.initial_claim_sum for p in proof.sumcheck_proofs]), dtype=np.uint64)
        _, rounds = circuit.shape()
        if len(proof.sumcheck_proofs) != n_layers or [p.round_polys.shape[0] for p in proof.sumcheck_proofs] != rounds:
            return False
        coeffs = np.ascontiguousarray(np.concatenate([np.asarray(p.round_polys).reshape(-1, 3, 4) for p in proof.sumcheck_proofs]),
                                      dtype=np.uint64)
        ok = C.c_int(0)
        check(lib().zk_gkr_verify(ctx.h, circuit.h, transcript.h, initial_inputs.h, output.ctypes.data_as(u64p), output.shape[0],
                                  w.ctypes.data_as(u64p), claims.ctypes.data_as(u64p), coeffs.ctypes.data_as(u64p), C.byref(ok)), ctx.h)
        return bool(ok.value)
