"""Every small vector the reference's own unit tests hold for the sumcheck / GKR path
(SURVEY.md section 4), replayed against BOTH CPU oracles (C: oracle/*.c, Python: oracle/pyref.py)."""
import numpy as np
import pytest

import oracle_lib as O
from oracle_lib import P, pyref

M = O.ints_to_mont
I = O.mont_to_ints

POLY4 = [0, 0, 0, 0, 0, 4, 0, 4, 0, 0, 3, 3, 5, 9, 8, 12]   # 3ac + 4bd + 5ab, evaluation_form.rs:331-350
POLY3 = [0, 0, 0, 3, 0, 0, 2, 5]                            # 2ab + 3bc, evaluation_form.rs:352-363


def _pts(vals):
    return [None if v is None else O.fq1(v) for v in vals]


def test_evaluate_4_variables():           # evaluation_form.rs:366-382
    assert I(O.mle_evaluate(M(POLY4), _pts([4, 2, 6, 1]))) == [120]
    assert pyref.evaluate(POLY4, [4, 2, 6, 1]) == [120]


def test_partially_evaluate_first_variable():   # evaluation_form.rs:385-405
    want = [0, 0, 12, 12, 20, 24, 32, 36]
    assert I(O.mle_evaluate(M(POLY4), _pts([4, None, None, None]))) == want
    assert pyref.evaluate(POLY4, [4, None, None, None]) == want
    assert I(O.mle_partially_evaluate(M(POLY4), 0, O.fq1(4))) == want


def test_partially_evaluate_last_variable():    # evaluation_form.rs:408-419
    want = [0, 9, 0, 11]
    assert I(O.mle_evaluate(M(POLY3), _pts([None, None, 3]))) == want
    assert pyref.evaluate(POLY3, [None, None, 3]) == want


def test_operation_ws():                        # evaluation_form.rs:422-473
    import ctypes as C
    wb, wc = M([0, 0, 0, 3]), M([0, 0, 0, 2])
    out = np.empty((16, 4), dtype=np.uint64)
    assert O.lib().orc_mle_w_add(O.ptr(wb), C.c_size_t(4), O.ptr(wc), C.c_size_t(4), O.ptr(out)) == 0
    want_add = [0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 2, 3, 3, 3, 5]
    assert I(out) == want_add == pyref.w_add([0, 0, 0, 3], [0, 0, 0, 2])
    assert O.lib().orc_mle_w_mul(O.ptr(wb), C.c_size_t(4), O.ptr(wc), C.c_size_t(4), O.ptr(out)) == 0
    want_mul = [0] * 15 + [6]
    assert I(out) == want_mul == pyref.w_mul([0, 0, 0, 3], [0, 0, 0, 2])


def test_mle_new_rejects_non_power_of_two():    # evaluation_form.rs:23-27
    for n in (0, 3, 5, 6, 12):
        assert O.lib().orc_mle_check(n) == 1
    for n in (1, 2, 4, 1024):
        assert O.lib().orc_mle_check(n) == 0
    with pytest.raises(ValueError):
        O.mle_evaluate(M(POLY3), _pts([1, 2]))          # :197-199 wrong number of points


def _sum_call(fn, sum_poly, *extra):
    import ctypes as C
    tables = [t for term in sum_poly for t in term]
    flat, arr, lens = O._table_ptrs([M(t) for t in tables])
    return fn(arr, C.c_size_t(len(sum_poly)), C.c_size_t(len(sum_poly[0])), lens, *extra)


def test_product_and_sum_polynomial():          # product_polynomial.rs:94-127, sum_polynomial.rs:104-149
    import ctypes as C
    prod = [[1, 2, 3, 4], [1, 2, 3, 4]]
    out = np.empty((4, 4), dtype=np.uint64)
    assert _sum_call(O.lib().orc_sum_reduce, [prod], O.ptr(out)) == 0
    assert I(out) == [1, 4, 9, 16] == pyref.sum_reduce([prod])
    r = np.empty(4, dtype=np.uint64)
    pts = M([1, 2])
    assert _sum_call(O.lib().orc_sum_evaluate, [prod], O.ptr(pts), C.c_size_t(2), O.ptr(r)) == 0
    assert I(r) == [25]
    assert pyref.sum_partial_evaluate([prod], [1, None]) == [[[3, 4], [3, 4]]]

    sp = [[[2, 3, 4, 5], [2, 3, 4, 5]], [[2, 3, 4, 5], [2, 3, 4, 5]]]
    assert _sum_call(O.lib().orc_sum_reduce, sp, O.ptr(out)) == 0
    assert I(out) == [8, 18, 32, 50] == pyref.sum_reduce(sp)
    assert _sum_call(O.lib().orc_sum_evaluate, sp, O.ptr(pts), C.c_size_t(2), O.ptr(r)) == 0
    assert I(r) == [72]
    assert pyref.sum_partial_evaluate(sp, [1, None]) == [[[4, 5], [4, 5]], [[4, 5], [4, 5]]]
    # mismatched lengths panic in the reference (product_polynomial.rs:15-19)
    assert _sum_call(O.lib().orc_sum_reduce, [[[1, 2, 3, 4], [1, 2]]], O.ptr(out)) == 3


def test_univariate():                          # dense_coefficient_form.rs:218-310
    import ctypes as C
    r = np.empty(4, dtype=np.uint64)
    O.lib().orc_uni_evaluate(O.ptr(M([20, 10, 3])), C.c_size_t(3), O.ptr(O.fq1(2)), O.ptr(r))
    assert I(r) == [52] and pyref.uni_evaluate([20, 10, 3], 2) == 52
    out = np.empty((6, 4), dtype=np.uint64)
    assert O.lib().orc_uni_mul(O.ptr(M([0, 0, 2])), C.c_size_t(3), O.ptr(M([0, 0, 0, 4])), C.c_size_t(4), O.ptr(out)) == 0
    assert I(out) == [0, 0, 0, 0, 0, 8]

    def interp(xs, ys):
        o = np.empty((len(xs), 4), dtype=np.uint64)
        assert O.lib().orc_uni_interpolate(O.ptr(M(xs)), O.ptr(M(ys)), C.c_size_t(len(xs)), O.ptr(o)) == 0
        got = I(o)
        assert got == pyref.interpolate(xs, ys)
        return got

    assert interp([0, 1, 2], [8, 10, 16]) == [8, 0, 2]
    assert interp([0, 1], [0, 2]) == [0, 2]
    assert interp([2, 4], [4, 8]) == [0, 2]
    fib = interp(list(range(9)), [0, 1, 1, 2, 3, 5, 8, 13, 21])
    assert pyref.uni_evaluate(fib, 5) == (pyref.uni_evaluate(fib, 4) + pyref.uni_evaluate(fib, 3)) % P
    # closed forms used by the device path equal the Lagrange result
    e = [123456789, P - 5, 987654321987654321]
    inv2 = pow(2, -1, P)
    c2 = (e[0] - 2 * e[1] + e[2]) * inv2 % P
    assert interp([0, 1, 2], e) == [e[0], (e[1] - e[0] - c2) % P, c2]
    assert interp([0, 1], e[:2]) == [e[0], (e[1] - e[0]) % P]


def test_full_sumcheck_pass_and_fail():         # sumcheck_protocol.rs:27-82
    table = M(POLY3)
    claim, coeffs, ch = O.sumcheck_prove(table)
    assert I(claim) == [10]
    assert O.sumcheck_verify(table, claim, coeffs)
    pp = pyref.generate_sumcheck_proof(POLY3)
    assert pyref.verify_proof(POLY3, pp)
    assert [I(c) for c in coeffs] == pp["round_polys"]
    assert I(ch) == pp["challenges"]
    # the hand-forged proof of test_full_sumcheck_fail
    bad_table = [0, 0, 0, 3, 0, 0, 2, 65]
    forged = np.stack([M([3, 7]), M([9, 10]), M([10, 97])])
    assert not O.sumcheck_verify(M(bad_table), O.fq1(10), forged)
    assert not pyref.verify_proof(bad_table, {"initial_claim_sum": 10, "round_polys": [[3, 7], [9, 10], [10, 97]]})
    # empty round_polys => false (verifier.rs:86-91): a 1-entry table has zero variables
    c1, k1, _ = O.sumcheck_prove(M([7]))
    assert k1.shape[0] == 0 and not O.sumcheck_verify(M([7]), c1, k1)
    # tampered coefficient is rejected
    bad = coeffs.copy()
    bad[1, 0] = O.fq1(I(bad[1, 0])[0] + 1)
    assert not O.sumcheck_verify(table, claim, bad)


def test_partial_sumcheck_pass():               # sumcheck_protocol.rs:85-111
    e1, e2 = [0, 0, 0, 2], [0, 0, 0, 3]
    sp = [[M(e1), M(e2)], [M(e1), M(e2)]]
    coeffs, ch = O.sumcheck_prove_sum(sp, O.fq1(12), O.OTranscript())
    ok, final, ch2 = O.sumcheck_partial_verify(O.fq1(12), coeffs, O.OTranscript())
    assert ok and I(ch) == I(ch2)
    pproof, ppts = pyref.generate_proof_for_partial_verify(12, [[e1, e2], [e1, e2]], pyref.Transcript())
    assert pyref.partial_verify(pproof, pyref.Transcript())[0]
    assert [I(c) for c in coeffs] == pproof["round_polys"] and I(ch) == ppts
    # final claim equals f evaluated at the challenges
    f_at_r = 2 * pyref.evaluate(e1, ppts)[0] * pyref.evaluate(e2, ppts)[0] % P
    assert I(final) == [f_at_r]
    # a wrong claim is caught in round 0
    assert not O.sumcheck_partial_verify(O.fq1(13), coeffs, O.OTranscript())[0]
    # a single product term cannot be expressed (F9): degree()+1 = 2 values vs 3 nodes
    with pytest.raises(ValueError):
        O.sumcheck_prove_sum([[M(e1), M(e2)]], O.fq1(6), O.OTranscript())


CIRCUIT_SMALL = [[(0, 1, 0), (2, 3, 1)], [(0, 1, 0)]]                      # circuit.rs:159-172
CIRCUIT_GKR = [[(0, 1, 0), (2, 3, 0), (4, 5, 0), (6, 7, 1)], [(0, 1, 1), (2, 3, 0)], [(0, 1, 0)]]  # gkr_protocol.rs:67-94


def test_circuit():                             # circuit.rs:175-202
    c = O.OCircuit(CIRCUIT_SMALL)
    ev = c.evaluate(M([1, 2, 3, 4]))
    assert I(ev[-1]) == [15, 0] == pyref.circuit_evaluate(CIRCUIT_SMALL, [1, 2, 3, 4])[-1]
    add1 = [0] * 32
    add1[1] = 1
    mul1 = [0] * 32
    mul1[27] = 1
    assert I(c.gate_poly(1, 0)) == add1 == pyref.gate_poly(CIRCUIT_SMALL, 1, 0)
    assert I(c.gate_poly(1, 1)) == mul1 == pyref.gate_poly(CIRCUIT_SMALL, 1, 1)
    with pytest.raises(ValueError):
        c.gate_poly(2, 0)                       # layer index out of bounds, circuit.rs:83-85


def test_gkr_sum_check():                       # gkr_protocol.rs:126-142
    inputs = list(range(1, 9))
    c = O.OCircuit(CIRCUIT_GKR)
    proof = O.gkr_prove(c, O.OTranscript(), M(inputs))
    assert O.gkr_verify(M(inputs), c, O.OTranscript(), proof)
    pproof = pyref.gkr_generate_proof(CIRCUIT_GKR, pyref.Transcript(), inputs)
    assert pyref.gkr_verify_proof(inputs, CIRCUIT_GKR, pyref.Transcript(), pproof)
    blob, shape = proof.serialize()
    assert blob == pyref.gkr_proof_bytes(pproof)
    assert shape == {"output_len": 2, "n_w_evals": 2, "n_proofs": 3, "rounds": [2, 4, 6]}
    # wrong inputs / tampered round polynomial are rejected
    assert not O.gkr_verify(M([2] + inputs[1:]), c, O.OTranscript(), proof)
    proof.tamper(1, 2)
    assert not O.gkr_verify(M(inputs), c, O.OTranscript(), proof)


def test_quotient_remainder_and_minus_c_oracle_equals_python_oracle():
    """compute_quotient_remainder / minus (evaluation_form.rs:260-301): the reference holds no vector for them (its KZG tests
    run over BLS12-381 and check only the pairing), so the two independent restatements are checked against each other and
    against the division identity f = remainder + (x_v - a) * quotient on the hypercube."""
    import random

    rng = random.Random(11)
    for log_n in (1, 2, 3, 5, 8):
        n = 1 << log_n
        vals = [rng.randrange(O.P) for _ in range(n)]
        t = O.ints_to_mont(vals)
        for var in range(log_n):
            a = rng.randrange(O.P)
            q, r = O.mle_quotient_remainder(t, O.fq1(a), var)
            pq, pr = O.pyref.compute_quotient_remainder(vals, a, var)
            assert O.mont_to_ints(q) == pq and O.mont_to_ints(r) == pr
            for k, (i1, i2) in enumerate(O.pyref.y1_y2_indexes(n, var)):
                assert (pr[k] - a * pq[k]) % O.P == vals[i1] and (pr[k] + (1 - a) * pq[k]) % O.P == vals[i2]
        v = rng.randrange(O.P)
        assert O.mont_to_ints(O.mle_minus(t, O.fq1(v))) == O.pyref.minus(vals, v)
    # the reference's own small fold vector (evaluation_form.rs:385-405) seen through the remainder
    poly4 = [0, 0, 0, 0, 0, 4, 0, 4, 0, 0, 3, 3, 5, 9, 8, 12]
    q, r = O.mle_quotient_remainder(O.ints_to_mont(poly4), O.fq1(4), 0)
    assert O.mont_to_ints(r) == [0, 0, 12, 12, 20, 24, 32, 36] and O.mont_to_ints(q) == [0, 0, 3, 3, 5, 5, 8, 8]
