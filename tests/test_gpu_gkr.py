"""GKR layer prover on the device (zk_gkr_prove: gate lists, no dense predicate tables) against the
oracle's reference-faithful DENSE prover (oracle/gkr.c follows gkr/src/prover.rs:24-154), bit for bit
over the whole proof, and against the line-by-line dense flow over device tables
(GKRProver.generate_proof_dense).  Circuits: the reference's own test circuits
(gkr_protocol.rs:67-94, circuit.rs:159-172) and seeded random layered circuits."""
import random

import numpy as np
import pytest

import oracle_lib as O
import zk_amd
from zk_amd import Circuit, Gate, GKRProver, GKRVerifier, Operation

pytestmark = pytest.mark.gpu
SEED = 0x5EED0000

CIRCUIT_GKR = [[(0, 1, 0), (2, 3, 0), (4, 5, 0), (6, 7, 1)], [(0, 1, 1), (2, 3, 0)], [(0, 1, 0)]]  # gkr_protocol.rs:67-94
CIRCUIT_SMALL = [[(0, 1, 0), (2, 3, 1)], [(0, 1, 0)]]                                              # circuit.rs:159-172


def mk(layers, ctx=None):
    return Circuit([[Gate(l, r, Operation(o)) for (l, r, o) in gs] for gs in layers], ctx=ctx)


def random_circuit(widths, seed, full=True):
    """widths[0] = number of inputs (power of two); layer i has widths[i+1] gates reading the previous
    layer.  One gate always touches the last index so that 2^k_in == len(previous layer)."""
    rng = random.Random(seed)
    layers = []
    for i in range(1, len(widths)):
        prev = max(1 << max(widths[i - 1] - 1, 0).bit_length(), 2) if i > 1 else widths[0]
        gs = [(rng.randrange(prev), rng.randrange(prev), rng.randrange(2)) for _ in range(widths[i])]
        gs[rng.randrange(len(gs))] = (prev - 1, rng.randrange(prev), rng.randrange(2))
        layers.append(gs)
    return layers


def proof_bytes(proof: zk_amd.GKRProof, ctx) -> bytes:
    from zk_amd.gkr import _limbs_le32
    return _limbs_le32(proof.elements(), ctx)


def check_against_oracle(layers, inputs, dense_too=True):
    ctx = zk_amd.default_context()
    oc = O.OCircuit(layers)
    oproof = O.gkr_prove(oc, O.OTranscript(), inputs)
    oblob, oshape = oproof.serialize()
    circuit = mk(layers, ctx)
    tr = zk_amd.Transcript()
    proof = GKRProver.generate_proof(circuit, tr, inputs)
    assert [p.round_polys.shape[0] for p in proof.sumcheck_proofs] == oshape["rounds"]
    assert len(proof.output_poly) == oshape["output_len"] and len(proof.w_polys_evals) == oshape["n_w_evals"]
    assert proof_bytes(proof, ctx) == oblob
    # transcripts end in the same state: same bytes absorbed, same challenges drawn
    otr = O.OTranscript()
    O.gkr_prove(oc, otr, inputs)
    assert tr.peek_digest() == otr.peek_digest()
    if dense_too:
        tr2 = zk_amd.Transcript()
        dense = GKRProver.generate_proof_dense(circuit, tr2, inputs)
        assert proof_bytes(dense, ctx) == oblob
        assert tr2.peek_digest() == otr.peek_digest()
    return proof


def test_reference_gkr_circuit():                       # gkr_protocol.rs:126-142
    inputs = O.ints_to_mont([1, 2, 3, 4, 5, 6, 7, 8])
    check_against_oracle(CIRCUIT_GKR, inputs)


def test_reference_small_circuit_full_range_inputs():
    check_against_oracle(CIRCUIT_SMALL, O.gen_table(SEED, 0, 4))


def test_circuit_evaluate_and_gate_polys_match_oracle():  # circuit.rs:175-202
    ctx = zk_amd.default_context()
    c, oc = mk(CIRCUIT_SMALL, ctx), O.OCircuit(CIRCUIT_SMALL)
    ev = c.evaluate_at_input(O.ints_to_mont([1, 2, 3, 4]))
    oev = oc.evaluate(O.ints_to_mont([1, 2, 3, 4]))
    assert len(ev) == len(oev) == 3
    for a, b in zip(ev, oev):
        assert np.array_equal(a.get_evaluation_points(), b)
    assert O.mont_to_ints(ev[-1].get_evaluation_points()) == [15, 0]
    for layer_idx in (0, 1):
        for op in (0, 1):
            got = c._gate_poly(layer_idx, op).get_evaluation_points()
            assert np.array_equal(got, oc.gate_poly(layer_idx, op))
    assert O.mont_to_ints(c.get_add_i(1).get_evaluation_points()).index(1) == 1      # circuit.rs:190-196
    assert O.mont_to_ints(c.get_mul_i(1).get_evaluation_points()).index(1) == 27
    with pytest.raises(zk_amd.ZkError) as ei:
        c.get_add_i(2)                                  # "layer index out of bounds", circuit.rs:83-85
    assert ei.value.status == 4


@pytest.mark.parametrize("widths,seed", [([8, 4, 2, 1], 1), ([16, 8, 8, 3], 2), ([4, 4, 4, 4, 4], 3), ([32, 16, 5], 4), ([2, 2, 2], 5),
                                         ([64, 32, 16, 8], 6)])
def test_random_circuits_match_dense_oracle(widths, seed):
    layers = random_circuit(widths, seed)
    inputs = O.gen_table(SEED + seed, 0, widths[0])
    k_max = max(max(1 << max(w - 1, 0).bit_length(), 2).bit_length() - 1 for w in widths)
    check_against_oracle(layers, inputs, dense_too=3 * k_max <= 15)


def test_elementwise_helpers_match_oracle():
    ctx = zk_amd.default_context()
    a, b = O.gen_table(SEED + 11, 0, 8), O.gen_table(SEED + 12, 0, 8)
    A, B = zk_amd.MultiLinearPolynomial(a, ctx=ctx), zk_amd.MultiLinearPolynomial(b, ctx=ctx)
    s = O.gen_table(SEED + 13, 0, 1)[0]
    out = np.empty((64, 4), dtype=np.uint64)
    O.lib().orc_mle_w_add(O.ptr(a), 8, O.ptr(b), 8, O.ptr(out))
    assert np.array_equal(zk_amd.MultiLinearPolynomial.w_add(A, B).get_evaluation_points(), out)
    O.lib().orc_mle_w_mul(O.ptr(a), 8, O.ptr(b), 8, O.ptr(out))
    assert np.array_equal(zk_amd.MultiLinearPolynomial.w_mul(A, B).get_evaluation_points(), out)
    out8 = np.empty((8, 4), dtype=np.uint64)
    O.lib().orc_mle_scalar_mul(O.ptr(a), 8, O.ptr(s), O.ptr(out8))
    assert np.array_equal(A.scalar_mul(s).get_evaluation_points(), out8)
    O.lib().orc_mle_add(O.ptr(a), 8, O.ptr(b), 8, O.ptr(out8))
    assert np.array_equal((A + B).get_evaluation_points(), out8)
    with pytest.raises(zk_amd.ZkError) as ei:           # evaluation_form.rs:244-246
        A + zk_amd.MultiLinearPolynomial(O.gen_table(1, 0, 4), ctx=ctx)
    assert ei.value.status == 3


def test_mismatched_layer_width_is_rejected():
    # gates only reach index 1 of a 4-wide input: k_in = 1 but len(W) = 4 -> ProductPolynomial::new panics
    ctx = zk_amd.default_context()
    c = mk([[(0, 1, 0), (1, 0, 1)]], ctx)
    with pytest.raises(zk_amd.ZkError) as ei:
        GKRProver.generate_proof(c, zk_amd.Transcript(), O.gen_table(1, 0, 4))
    assert ei.value.status == 3


def test_wide_circuit_selfconsistent():
    """k = 12 per layer: far beyond the dense oracle (2^36-entry predicates); checked through the
    sumcheck verifier's round identities and the claim chain instead."""
    ctx = zk_amd.default_context()
    W, depth = 1 << 12, 3
    rng = random.Random(9)
    layers = [[((2 * g) % W, (2 * g + 1) % W, rng.randrange(2)) for g in range(W)] for _ in range(depth)]
    circuit = mk(layers, ctx)
    inputs = O.gen_table(SEED + 99, 0, W)
    tr = zk_amd.Transcript()
    proof = GKRProver.generate_proof(circuit, tr, inputs, timing=True)
    assert len(proof.sumcheck_proofs) == depth and all(p.round_polys.shape == (24, 3, 4) for p in proof.sumcheck_proofs)
    # every layer's sumcheck passes partial_verify on a transcript replaying the prover's stream
    vt = zk_amd.Transcript()
    vt.append(zk_amd.MultiLinearPolynomial(proof.output_poly, ctx=ctx).to_bytes())
    vt.sample_n_challenges(12)
    from zk_amd.gkr import _limbs_le32
    for i, p in enumerate(proof.sumcheck_proofs):
        if i > 0:
            wb, wc = proof.w_polys_evals[i - 1]
            vt.append(_limbs_le32(np.stack([wb, wc]), ctx))
            vt.sample_challenge(), vt.sample_challenge()
        ok, final, ch = zk_amd.SumcheckVerifier.partial_verify(p, vt)
        assert ok, f"layer {i}"
    assert vt.peek_digest() == tr.peek_digest()


def test_wide_circuit_proof_does_not_depend_on_the_small_table_schedule():
    """Three rounds per pass on the layer tables (k_fold_sum27, the default below 2^20 entries; in its product-plus-linear form
    P W + Q * 1, the default of the GKR prover, and in the general four-table form) against two rounds per pass (k_fold2_sum9) and
    against separate fold and sums launches: byte-identical GKR proofs and transcript states."""
    W, depth = 1 << 13, 2
    rng = random.Random(21)
    layers = [[(rng.randrange(W), rng.randrange(W), rng.randrange(2)) for g in range(W)] for _ in range(depth)]
    inputs = O.gen_table(SEED + 123, 0, W)
    outs = []
    for small_triple, fused, linear in ((20, 20, 21), (20, 20, 0), (0, 20, 21), (0, 20, 0), (0, 0, 0)):
        ctx = zk_amd.Context(0)
        ctx.set_deg2_small_triple(small_triple)
        ctx.set_deg2_fused(fused)
        ctx.set_deg2_first_triple(0, linear)   # 0: the GKR phases through the general four-table kernels
        circuit = mk(layers, ctx)
        tr = zk_amd.Transcript()
        ctx.forms_used()
        proof = GKRProver.generate_proof(circuit, tr, inputs)
        forms = ctx.forms_used()
        assert ("fold_sum27" in forms) == (small_triple > 0 or linear > 0) and ("pwq" in forms) == (linear > 0), forms
        outs.append((np.concatenate([p.round_polys.reshape(-1, 4) for p in proof.sumcheck_proofs]), np.asarray(proof.output_poly),
                     np.asarray(proof.w_polys_evals), tr.peek_digest()))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2]) and o[3] == outs[0][3]


# ---------------------------------------------------------------- verifier (gkr/src/verifier.rs:21-119)
def test_verifier_accepts_and_rejects_like_the_oracle():        # gkr_protocol.rs:126-142
    ctx = zk_amd.default_context()
    for layers, inputs in ((CIRCUIT_GKR, O.ints_to_mont([1, 2, 3, 4, 5, 6, 7, 8])),
                           (random_circuit([16, 8, 8, 3], 2), O.gen_table(SEED + 2, 0, 16)),
                           (random_circuit([2, 2, 2], 5), O.gen_table(SEED + 5, 0, 2))):
        circuit, oc = mk(layers, ctx), O.OCircuit(layers)
        tr = zk_amd.Transcript()
        proof = GKRProver.generate_proof(circuit, tr, inputs)
        vt = zk_amd.Transcript()
        assert GKRVerifier.verify_proof(inputs, circuit, vt, proof)
        ovt = O.OTranscript()
        assert O.gkr_verify(inputs, oc, ovt, O.gkr_prove(oc, O.OTranscript(), inputs))
        assert vt.peek_digest() == ovt.peek_digest()          # the verifier's transcript ends where the oracle verifier's does
        # wrong inputs
        bad_in = inputs.copy()
        bad_in[0] = O.fq1(12345)
        assert not GKRVerifier.verify_proof(bad_in, circuit, zk_amd.Transcript(), proof)
        assert not O.gkr_verify(bad_in, oc, O.OTranscript(), O.gkr_prove(oc, O.OTranscript(), inputs))
        # a coefficient of the LAST round of a layer (earlier rounds make the reference panic rather than return false)
        bad = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
        bad.sumcheck_proofs[0].round_polys[-1, 0] = O.fq1(7)
        assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), bad)
        # an earlier round
        bad = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
        bad.sumcheck_proofs[-1].round_polys[0, 1] = O.fq1(7)
        assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), bad)
        # a W evaluation sent by the prover
        if proof.w_polys_evals:
            bad = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
            bad.w_polys_evals[0] = (O.fq1(3), bad.w_polys_evals[0][1])
            assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), bad)
        # the claim of a layer
        bad = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
        bad.sumcheck_proofs[0].initial_claim_sum = O.fq1(1)
        assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), bad)


def test_verifier_on_a_wide_circuit():
    ctx = zk_amd.default_context()
    W, depth = 1 << 12, 3
    rng = random.Random(9)
    layers = [[((2 * g) % W, (2 * g + 1) % W, rng.randrange(2)) for g in range(W)] for _ in range(depth)]
    circuit = mk(layers, ctx)
    inputs = O.gen_table(SEED + 99, 0, W)
    proof = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
    assert GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof)
    proof.sumcheck_proofs[1].round_polys[-1, 2] = O.fq1(99)
    assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof)


def test_gkr_layers_of_2pow21_gates_long_eq_halves():
    """2^21 gates per layer: the eq half tables have 10 and 11 variables, one more than the host-built two-variable tables of
    k_eq_small_pairs take, so every half table of these layers comes from k_eq_small (one factor per variable).  Depth 2;
    prove -> verify accepts, a tampered coefficient is rejected (no dense oracle at this size)."""
    from zk_amd.synthetic import gkr_layers

    ctx = zk_amd.default_context()
    depth, k = 2, 21
    circuit = Circuit.from_arrays(gkr_layers(depth, k), ctx=ctx)
    inputs = zk_amd.MultiLinearPolynomial.generate(SEED + 3, 1 << k, ctx=ctx)
    proof = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
    assert len(proof.sumcheck_proofs) == depth and all(p.round_polys.shape == (2 * k, 3, 4) for p in proof.sumcheck_proofs)
    assert GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof)
    proof.sumcheck_proofs[1].round_polys[5, 1, 0] ^= 1
    assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof)


def test_gkr_config5_full_size():
    """BASELINE config 5 at its real size: depth 16, 2^20 gates per layer (bench.py's circuit, zk_amd.synthetic.gkr_layers):
    prove -> verify accepts and both transcripts end in the same state (gkr_protocol.rs:126-142, gkr/src/verifier.rs:21-119);
    a tampered coefficient, a tampered W evaluation and a wrong input are rejected.  Far beyond any dense oracle
    (2^60-entry predicates): the verifier's round identities, the predicate evaluation at the final point and
    the claim chain down to the input layer are the check."""
    from zk_amd.synthetic import gkr_layers

    ctx = zk_amd.default_context()
    depth, k = 16, 20
    circuit = Circuit.from_arrays(gkr_layers(depth, k), ctx=ctx)
    inputs = zk_amd.MultiLinearPolynomial.generate(SEED, 1 << k, ctx=ctx)
    pt = zk_amd.Transcript()
    proof = GKRProver.generate_proof(circuit, pt, inputs)
    assert len(proof.sumcheck_proofs) == depth and all(p.round_polys.shape == (2 * k, 3, 4) for p in proof.sumcheck_proofs)
    assert len(proof.w_polys_evals) == depth - 1 and len(proof.output_poly) == 1 << k
    # the output layer of the proof is the circuit's evaluation
    ev = circuit.evaluate_at_input(inputs)
    assert np.array_equal(ev[-1].get_evaluation_points(), proof.output_poly)
    vt = zk_amd.Transcript()
    assert GKRVerifier.verify_proof(inputs, circuit, vt, proof)
    # Transcript states.  (1) A host replay of the prover's stream (bytes(W_0), rho, then per layer
    # [LE32(wb) LE32(wc), alpha, beta] + the sumcheck rounds through partial_verify) ends in the PROVER's final state.
    # (2) The verifier appends the last layer's (W(r_b), W(r_c)) -- the INPUT polynomial at the two halves of the last
    # challenges -- which the prover does not (gkr/src/verifier.rs:76-104 vs prover.rs:143-147): the prover's state
    # plus those 64 bytes is the verifier's final state.
    from zk_amd.gkr import _limbs_le32
    from zk_amd.sumcheck import SumcheckVerifier

    rt = zk_amd.Transcript()
    rt.append(_limbs_le32(proof.output_poly, ctx))
    rt.sample_n_challenges(k)
    last_ch = None
    for i, sp in enumerate(proof.sumcheck_proofs):
        if i:
            rt.append(_limbs_le32(np.stack(proof.w_polys_evals[i - 1]), ctx))
            rt.sample_challenge(), rt.sample_challenge()
        ok, _, last_ch = SumcheckVerifier.partial_verify(sp, rt)
        assert ok
    assert rt.peek_digest() == pt.peek_digest()
    assert pt.peek_digest() != vt.peek_digest()
    w_in = [inputs.evaluate(list(last_ch[:k])).get_evaluation_points()[0], inputs.evaluate(list(last_ch[k:])).get_evaluation_points()[0]]
    pt2 = pt.clone()
    pt2.append(_limbs_le32(np.stack(w_in), ctx))
    assert pt2.peek_digest() == vt.peek_digest()
    # a second prove gives the same bytes (deterministic sums, fixed reduction order)
    again = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
    assert np.array_equal(again.elements(), proof.elements())
    # tampering: a coefficient in the last round of a middle layer, one in the first round of the last layer, a W evaluation
    saved = proof.sumcheck_proofs[7].round_polys[-1, 2].copy()
    proof.sumcheck_proofs[7].round_polys[-1, 2] = O.fq1(99)
    assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof)
    proof.sumcheck_proofs[7].round_polys[-1, 2] = saved
    saved = proof.sumcheck_proofs[-1].round_polys[0, 1].copy()
    proof.sumcheck_proofs[-1].round_polys[0, 1] = O.fq1(7)
    assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof)
    proof.sumcheck_proofs[-1].round_polys[0, 1] = saved
    wb, wc = proof.w_polys_evals[3]
    proof.w_polys_evals[3] = (O.fq1(3), wc)
    assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof)
    proof.w_polys_evals[3] = (wb, wc)
    assert GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof)
    # a wrong input (one entry changed)
    bad_in = inputs.get_evaluation_points().copy()
    bad_in[12345] = O.fq1(12345)
    assert not GKRVerifier.verify_proof(bad_in, circuit, zk_amd.Transcript(), proof)


def test_strict_claims_close_the_inherited_binding_gap():
    """The reference's verifier takes every layer's claim from the proof (gkr/src/verifier.rs:63-64), so a prover that absorbs
    a FALSE output_poly but sum-checks the true relation is accepted.  Built here with the line-by-line dense flow of
    prover.rs:24-154 and one changed line; zk_ctx_set_gkr_strict_claims(1) rejects it, honest proofs pass in both modes."""
    from zk_amd import MultiLinearPolynomial, ProductPolynomial, SumPolynomial
    from zk_amd.gkr import GKRProof, _limbs_le32
    from zk_amd.sumcheck import SumcheckProver

    ctx = zk_amd.Context(0)
    circuit = mk(CIRCUIT_GKR, ctx)
    inputs = O.ints_to_mont([1, 2, 3, 4, 5, 6, 7, 8])
    honest = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
    for strict in (False, True):
        ctx.set_gkr_strict_claims(strict)
        vt = zk_amd.Transcript()
        assert GKRVerifier.verify_proof(inputs, circuit, vt, honest)
    # the forgery: the output polynomial sent (and absorbed) is not the circuit's output
    tr = zk_amd.Transcript()
    evals = circuit.evaluate_at_input(inputs)
    running = Circuit.get_w_i(0, evals)
    fake_out = running.get_evaluation_points().copy()
    fake_out[0] = O.fq1(424242)
    tr.append(MultiLinearPolynomial(fake_out, ctx=ctx).to_bytes())                 # prover.rs:46 with the false table
    random_values = list(tr.sample_n_challenges(running.number_of_variables()))
    w_evals, proofs = [], []
    for layer_idx in range(circuit.get_layer_count()):
        mul_abc, add_abc = circuit.get_mul_i(layer_idx), circuit.get_add_i(layer_idx)
        nv = mul_abc.number_of_variables()
        if layer_idx == 0:
            pts = [random_values[i] if i < len(random_values) else None for i in range(nv)]
            new_mul, new_add = mul_abc.evaluate(pts), add_abc.evaluate(pts)
            claim = running.evaluate(random_values).get_evaluation_points()[0]      # the TRUE W_0 at rho: the sumcheck stays consistent
        else:
            h = len(random_values) // 2
            r_b, r_c = random_values[:h], random_values[h:]
            wb, wc = running.evaluate(r_b).get_evaluation_points()[0], running.evaluate(r_c).get_evaluation_points()[0]
            tr.append(_limbs_le32(np.stack([wb, wc]), ctx))
            alpha, beta = tr.sample_challenge(), tr.sample_challenge()
            pb = [r_b[i] if i < h else None for i in range(nv)]
            pc = [r_c[i] if i < h else None for i in range(nv)]
            new_mul = mul_abc.evaluate(pb).scalar_mul(alpha) + mul_abc.evaluate(pc).scalar_mul(beta)
            new_add = add_abc.evaluate(pb).scalar_mul(alpha) + add_abc.evaluate(pc).scalar_mul(beta)
            w_evals.append((wb, wc))
            two = MultiLinearPolynomial(np.stack([wb, wc]), ctx=ctx)
            claim = MultiLinearPolynomial(np.concatenate([two.scalar_mul(alpha).get_evaluation_points()[0:1],
                                                          two.scalar_mul(beta).get_evaluation_points()[1:2]]), ctx=ctx).evaluation_sum()
        next_w = Circuit.get_w_i(layer_idx + 1, evals)
        f = SumPolynomial([ProductPolynomial([new_mul, MultiLinearPolynomial.w_mul(next_w, next_w)]),
                           ProductPolynomial([new_add, MultiLinearPolynomial.w_add(next_w, next_w)])])
        proof, pts = SumcheckProver.generate_proof_for_partial_verify(claim, f, tr)
        random_values, running = list(pts), next_w
        proofs.append(proof)
    forged = GKRProof(fake_out, w_evals, proofs)
    ctx.set_gkr_strict_claims(False)
    assert GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), forged)          # the reference's verifier accepts the false output
    oc = O.OCircuit(CIRCUIT_GKR)
    ctx.set_gkr_strict_claims(True)
    assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), forged)      # the bound verifier does not
    # a later layer's claim replaced consistently is out of reach of a test this size; its binding is checked on the honest proof:
    bad = GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs)
    bad.w_polys_evals[0] = (O.fq1(3), bad.w_polys_evals[0][1])
    assert not GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), bad)
    del oc
