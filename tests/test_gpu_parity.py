"""GPU parity tests: the HIP path (through the C ABI of libzkmi.so) against the CPU oracle
(oracle/liboracle.so) and the committed golden vectors.  Bit-exact: integer field arithmetic.
Run on an MI355X:  python -m pytest tests -m gpu"""
import ctypes as C
import random

import numpy as np
import pytest

import golden_lib as G
import oracle_lib as O
import zk_amd
from zk_amd import MultiLinearPolynomial as MLE

pytestmark = pytest.mark.gpu

SEED = 0x5EED0000
P = O.P


def rand_table(n, seed):
    """full-range values incl. the edge values 0, 1, p-1, p-2 sprinkled in"""
    rng = random.Random(seed)
    vals = [rng.randrange(P) for _ in range(n)]
    for k, v in enumerate([0, 1, P - 1, P - 2, (P - 1) // 2, 2**253]):
        if k < n:
            vals[rng.randrange(n)] = v
    return vals


@pytest.fixture(scope="module")
def ctx():
    return zk_amd.default_context()


def test_native_library_is_loaded(ctx):
    assert zk_amd.lib().zk_version().startswith(b"zkmi")
    assert ctx.stream != 0


def test_upload_download_roundtrip_and_to_bytes(ctx):
    for n in (1, 2, 8, 64, 1 << 12):
        vals = rand_table(n, n)
        t = O.ints_to_mont(vals)
        m = MLE(t)
        assert len(m) == n and m.number_of_variables() == n.bit_length() - 1
        assert np.array_equal(m.get_evaluation_points(), t)
        assert m.to_bytes() == O.mle_to_bytes(t) == O.pyref.to_bytes(vals)
        # le32 upload (reduces mod p) and back
        m2 = MLE.from_le32(m.to_bytes())
        assert np.array_equal(m2.get_evaluation_points(), t)
    # non-canonical LE32 input (>= p, up to 2^256-1) is reduced like from_le_bytes_mod_order
    raw = [P, P + 5, 2**256 - 1, 4 * P + 3]
    m3 = MLE.from_le32(b"".join(v.to_bytes(32, "little") for v in raw))
    assert O.mont_to_ints(m3.get_evaluation_points()) == [v % P for v in raw]


@pytest.mark.parametrize("log_n", [15, 16, 19, 20, 21])
def test_to_bytes_stream_chunks_vs_oracle(log_n):
    """to_bytes of tables that take one, two and several chunks of the pinned staging stream (a 2^15-entry first chunk, then 2^19-entry
    ones, alternating between two halves): the oracle's bytes, and the same transcript state when the prover absorbs the table itself."""
    t = O.gen_table(SEED + 700 + log_n, 0, 1 << log_n)
    m = MLE(t)
    want = O.mle_to_bytes(t)
    assert m.to_bytes() == want
    tr = zk_amd.Transcript()
    tr.append(want)
    claim, coeffs, ch = zk_amd.SumcheckProver.generate_round_polys(m, tr)      # the caller's transcript holds the table's bytes
    proof, ch2 = zk_amd.SumcheckProver.generate_sumcheck_proof(m, return_challenges=True)   # the library absorbs the table through the stream
    assert np.array_equal(proof.round_polys, coeffs) and np.array_equal(ch, ch2)


def test_new_rejects_non_power_of_two(ctx):
    for n in (3, 5, 6, 12, 1000):
        with pytest.raises(zk_amd.ZkError) as ei:
            MLE(np.zeros((n, 4), dtype=np.uint64))
        assert ei.value.status == 1            # evaluation_form.rs:23-27
    with pytest.raises(zk_amd.ZkError):
        MLE(np.zeros((0, 4), dtype=np.uint64))


def test_generate_matches_oracle(ctx):
    for first, n in ((0, 1), (0, 256), (12345, 4096), (2**33 + 7, 64)):
        m = MLE.generate(SEED + 3, n, first=first)
        assert np.array_equal(m.get_evaluation_points(), O.gen_table(SEED + 3, first, n))


def test_reference_fold_vectors(ctx):
    # evaluation_form.rs:366-419
    poly4 = [0, 0, 0, 0, 0, 4, 0, 4, 0, 0, 3, 3, 5, 9, 8, 12]
    poly3 = [0, 0, 0, 3, 0, 0, 2, 5]
    m4, m3 = MLE(O.ints_to_mont(poly4)), MLE(O.ints_to_mont(poly3))
    f = O.fq1
    assert O.mont_to_ints(m4.evaluate([f(4), f(2), f(6), f(1)]).get_evaluation_points()) == [120]
    assert O.mont_to_ints(m4.evaluate([f(4), None, None, None]).get_evaluation_points()) == [0, 0, 12, 12, 20, 24, 32, 36]
    assert O.mont_to_ints(m3.evaluate([None, None, f(3)]).get_evaluation_points()) == [0, 9, 0, 11]
    with pytest.raises(zk_amd.ZkError) as ei:   # :197-199
        m3.evaluate([f(1), f(2)])
    assert ei.value.status == 2
    with pytest.raises(zk_amd.ZkError):
        m3.partially_evaluate((3, f(1)))


@pytest.mark.parametrize("log_n", [1, 2, 3, 7, 12, 16])
def test_partially_evaluate_every_variable_vs_oracle(ctx, log_n):
    n = 1 << log_n
    t = O.ints_to_mont(rand_table(n, 100 + log_n)) if log_n <= 12 else O.gen_table(SEED + 9, 0, n)
    m = MLE(t)
    rng = random.Random(log_n)
    variables = range(log_n) if log_n <= 7 else (0, 1, log_n // 2, log_n - 1)
    for var in variables:
        for rv in (0, 1, P - 1, rng.randrange(P)):
            r = O.fq1(rv)
            got = m.partially_evaluate((var, r)).get_evaluation_points()
            assert np.array_equal(got, O.mle_partially_evaluate(t, var, r)), (var, rv)


def test_evaluate_mixed_points_vs_oracle(ctx):
    rng = random.Random(77)
    for log_n in (3, 6, 10):
        t = O.ints_to_mont(rand_table(1 << log_n, 200 + log_n))
        m = MLE(t)
        for trial in range(4):
            pts = [None if rng.random() < 0.4 else O.fq1(rng.randrange(P)) for _ in range(log_n)]
            got = m.evaluate(pts).get_evaluation_points()
            assert np.array_equal(got, O.mle_evaluate(t, pts))
        full = [O.fq1(rng.randrange(P)) for _ in range(log_n)]
        assert np.array_equal(m.evaluate(full).get_evaluation_points(), O.mle_evaluate(t, full))


@pytest.mark.parametrize("log_n", [1, 2, 5, 9, 14, 18])
def test_sums_vs_oracle(ctx, log_n):
    n = 1 << log_n
    t = O.gen_table(SEED + 11, 0, n)
    m = MLE(t)
    assert list(m.evaluation_sum()) == list(O.mle_sum(t))
    e = m.round_sums()
    assert list(e[0]) == list(O.mle_sum(t[: n // 2])) and list(e[1]) == list(O.mle_sum(t[n // 2:]))
    # worst case for the carry chains: every entry p-1
    tm = O.ints_to_mont([P - 1] * min(n, 4096))
    mm = MLE(tm)
    assert O.mont_to_ints(mm.evaluation_sum()) == [(P - 1) * len(tm) % P]


@pytest.mark.parametrize("case", G.vectors()["sumcheck"], ids=lambda c: c["name"])
def test_sumcheck_golden(ctx, case):
    table = G.case_table(case)
    poly = MLE(table)
    proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
    assert G.le32_of(proof.initial_claim_sum) == [case["initial_claim_sum"]]
    assert [G.le32_of(g) for g in proof.round_polys] == case["round_polys"]
    assert G.le32_of(ch) == case["challenges"]
    # the prover leaves the caller's table untouched (it borrows &MLE, prover.rs:158)
    assert np.array_equal(poly.get_evaluation_points(), table)
    # accepted by the device verifier AND by the CPU oracle's verifier
    assert zk_amd.SumcheckVerifier.verify_proof(poly, proof)
    assert O.sumcheck_verify(table, proof.initial_claim_sum, proof.round_polys)


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 11, 15, 18])
def test_sumcheck_vs_oracle_random_full_range(ctx, log_n):
    n = 1 << log_n
    table = O.ints_to_mont(rand_table(n, 300 + log_n)) if log_n <= 11 else O.gen_table(SEED + 21, 5, n)
    poly = MLE(table)
    proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
    oclaim, ocoeffs, och = O.sumcheck_prove(table)
    assert list(proof.initial_claim_sum) == list(oclaim)
    assert np.array_equal(proof.round_polys, ocoeffs)
    assert np.array_equal(ch, och)
    ok = zk_amd.SumcheckVerifier.verify_proof(poly, proof)
    assert ok == (log_n > 0)                    # zero rounds => false (verifier.rs:86-91)
    assert ok == O.sumcheck_verify(table, oclaim, ocoeffs)


@pytest.mark.parametrize("batch,db", [(1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1), (4, 0), (4, 1), (5, 0), (6, 0), (7, 0), (8, 0), (8, 1)])
def test_sumcheck_every_fold_batch_setting_gives_the_same_proof(batch, db):
    """Rounds-per-pass batching is an execution schedule, not a protocol change: proofs are identical."""
    c = zk_amd.Context(0)
    c.set_fold_batch(batch, db)
    c.set_tail_fold(0)  # the default schedule would take tables this small through eight-round tail passes instead
    for log_n in (1, 2, 3, 4, 5, 6, 7, 9, 13, 17):
        table = O.gen_table(SEED + 60 + log_n, 0, 1 << log_n)
        poly = MLE(table, ctx=c)
        proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
        oclaim, ocoeffs, och = O.sumcheck_prove(table)
        assert list(proof.initial_claim_sum) == list(oclaim), (batch, db, log_n)
        assert np.array_equal(proof.round_polys, ocoeffs), (batch, db, log_n)
        assert np.array_equal(ch, och), (batch, db, log_n)
    with pytest.raises(zk_amd.ZkError):
        c.set_fold_batch(9, 0)
    with pytest.raises(zk_amd.ZkError):
        c.set_fold_batch(4, 2)   # the losing variants left the library (tools/fold_variants.hip)


@pytest.mark.parametrize("max_log,out_below,fold_batch", [(20, 13, 4), (20, 3, 4), (20, 0, 4), (12, 13, 3), (20, 20, 2), (9, 13, 1), (20, 13, 8)])
def test_sumcheck_tail_passes_give_the_same_proof(max_log, out_below, fold_batch):
    """The tail schedule (eight rounds per pass on small tables, one wave per output when the output is tiny, a last
    batch of up to eight rounds straight from the block sums) against the oracle, for every table size it touches."""
    c = zk_amd.Context(0)
    c.set_fold_batch(fold_batch, 0)
    c.set_tail_fold(max_log, out_below)
    for log_n in range(1, 21):
        table = O.gen_table(SEED + 120 + log_n, 0, 1 << log_n)
        poly = MLE(table, ctx=c)
        proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
        oclaim, ocoeffs, och = O.sumcheck_prove(table) if log_n <= 16 else O.sumcheck_prove_fast(table)
        assert list(proof.initial_claim_sum) == list(oclaim), (max_log, out_below, fold_batch, log_n)
        assert np.array_equal(proof.round_polys, ocoeffs), (max_log, out_below, fold_batch, log_n)
        assert np.array_equal(ch, och), (max_log, out_below, fold_batch, log_n)
        pts = [r for r in ch]
        assert np.array_equal(poly.evaluate(pts).get_evaluation_points(), O.mle_evaluate(table, pts)) if log_n <= 14 else True
    with pytest.raises(zk_amd.ZkError):
        c.set_tail_fold(41, 13)


@pytest.mark.parametrize("wide,min_log_out,lds_below", [(8, 2, 19), (8, 5, 0), (7, 3, 0), (6, 0, 64), (5, 4, 0), (0, 17, 19), (8, 3, 12), (6, 1, 0)])
def test_sumcheck_wide_first_passes_give_the_same_proof(wide, min_log_out, lds_below):
    """The pass schedule (wide 2^m-stream passes while the table is large, the default batch after that) is not
    part of the protocol: the thresholds are lowered here so that small tables take one or more wide passes."""
    c = zk_amd.Context(0)
    c.set_wide_fold(wide, min_log_out, lds_below)  # lds_below = 0: the scalar-weight kernel on every wide pass
    c.set_tail_fold(0)
    for log_n in (4, 9, 10, 11, 13, 16, 18):
        table = O.gen_table(SEED + 80 + log_n, 0, 1 << log_n)
        poly = MLE(table, ctx=c)
        proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
        oclaim, ocoeffs, och = O.sumcheck_prove(table)
        assert list(proof.initial_claim_sum) == list(oclaim), (wide, min_log_out, log_n)
        assert np.array_equal(proof.round_polys, ocoeffs), (wide, min_log_out, log_n)
        assert np.array_equal(ch, och), (wide, min_log_out, log_n)
        # evaluate() at a full point, and at a prefix of it, walks the same passes (verifier.rs:15-26)
        pts = [r for r in ch]
        assert np.array_equal(poly.evaluate(pts).get_evaluation_points(), O.mle_evaluate(table, pts)), (wide, min_log_out, log_n)
        part = pts[: log_n - 2] + [None, None]
        assert np.array_equal(poly.evaluate(part).get_evaluation_points(), O.mle_evaluate(table, part)), (wide, min_log_out, log_n)
    for bad in ((4, 17, 19), (9, 17, 19), (8, -1, 19), (8, 16, 65)):
        with pytest.raises(zk_amd.ZkError):
            c.set_wide_fold(*bad)


def test_sumcheck_reference_fail_vector_and_tampering(ctx):
    # sumcheck_protocol.rs:52-82: the hand-forged proof is rejected
    bad_table = MLE(O.ints_to_mont([0, 0, 0, 3, 0, 0, 2, 65]))
    forged = zk_amd.SumCheckProof(O.fq1(10), np.stack([O.ints_to_mont([3, 7]), O.ints_to_mont([9, 10]), O.ints_to_mont([10, 97])]))
    assert not zk_amd.SumcheckVerifier.verify_proof(bad_table, forged)
    table = O.gen_table(SEED + 5, 0, 1 << 9)
    poly = MLE(table)
    proof = zk_amd.SumcheckProver.generate_sumcheck_proof(poly)
    assert zk_amd.SumcheckVerifier.verify_proof(poly, proof)
    for rnd, coef in ((0, 0), (4, 1), (8, 0)):
        rp = proof.round_polys.copy()
        rp[rnd, coef] = O.fq1(O.mont_to_ints(rp[rnd, coef])[0] + 1)
        assert not zk_amd.SumcheckVerifier.verify_proof(poly, zk_amd.SumCheckProof(proof.initial_claim_sum, rp))
    wrong_claim = O.fq1(O.mont_to_ints(proof.initial_claim_sum)[0] + 1)
    assert not zk_amd.SumcheckVerifier.verify_proof(poly, zk_amd.SumCheckProof(wrong_claim, proof.round_polys))
    # a proof for one table does not verify against another
    other = MLE.generate(SEED + 6, 1 << 9)
    assert not zk_amd.SumcheckVerifier.verify_proof(other, proof)


def test_generate_round_polys_on_caller_transcript(ctx):
    table = O.gen_table(SEED + 31, 0, 1 << 10)
    poly = MLE(table)
    tr = zk_amd.Transcript()
    tr.append(poly.to_bytes())
    claim, coeffs, ch = zk_amd.SumcheckProver.generate_round_polys(poly, tr)
    oclaim, ocoeffs, och = O.sumcheck_prove(table)
    assert list(claim) == list(oclaim) and np.array_equal(coeffs, ocoeffs) and np.array_equal(ch, och)


def _sum_poly(tabs):
    m = [MLE(t) for t in tabs]
    return zk_amd.SumPolynomial([zk_amd.ProductPolynomial([m[0], m[1]]), zk_amd.ProductPolynomial([m[2], m[3]])]), m


@pytest.mark.parametrize("case", G.vectors()["sum2x2"], ids=lambda c: c["name"])
def test_sum2x2_golden(ctx, case):
    tabs = [O.gen_table(s, 0, 1 << case["log_n"]) for s in case["seeds"]]
    f, mles = _sum_poly(tabs)
    tr = zk_amd.Transcript()
    tr.append(case["transcript_prefix"].encode())
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(G.unhex(case["claim"])), f, tr)
    assert [G.le32_of(g) for g in proof.round_polys] == case["round_polys"]
    assert G.le32_of(pts) == case["challenges"]
    for m, t in zip(mles, tabs):
        assert np.array_equal(m.get_evaluation_points(), t)   # inputs untouched
    tv = zk_amd.Transcript()
    tv.append(case["transcript_prefix"].encode())
    ok, final, ch = zk_amd.SumcheckVerifier.partial_verify(proof, tv)
    assert ok and G.le32_of(final) == [case["final_claim"]]


def test_partial_sumcheck_reference_vector(ctx):
    # sumcheck_protocol.rs:85-111: 2 x (e1 * e2), claim 12
    e1, e2 = O.ints_to_mont([0, 0, 0, 2]), O.ints_to_mont([0, 0, 0, 3])
    f, _ = _sum_poly([e1, e2, e1, e2])
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(12), f, zk_amd.Transcript())
    assert zk_amd.SumcheckVerifier.partial_verify(proof, zk_amd.Transcript())[0]
    ocoeffs, och = O.sumcheck_prove_sum([[e1, e2], [e1, e2]], O.fq1(12), O.OTranscript())
    assert np.array_equal(proof.round_polys, ocoeffs) and np.array_equal(pts, och)


@pytest.mark.parametrize("log_n", [1, 2, 4, 6, 7, 10, 13])
def test_sum2x2_vs_oracle_random_full_range(ctx, log_n):
    n = 1 << log_n
    tabs = [O.ints_to_mont(rand_table(n, 400 + 10 * log_n + k)) if log_n <= 10 else O.gen_table(SEED + 40 + k, 0, n)
            for k in range(4)]
    f, _ = _sum_poly(tabs)
    claim = O.fq1(12345)                 # the prover copies the claim unchecked (prover.rs:185)
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    tr.append(b"layer-7")
    otr.append(b"layer-7")
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(claim, f, tr)
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], claim, otr)
    assert np.array_equal(proof.round_polys, ocoeffs)
    assert np.array_equal(pts, och)
    assert tr.peek_digest() == otr.peek_digest()   # transcripts end in the same state


def test_sum2x2_shape_errors(ctx):
    a, b = MLE(O.gen_table(1, 0, 8)), MLE(O.gen_table(2, 0, 4))
    with pytest.raises(zk_amd.ZkError):
        zk_amd.ProductPolynomial([a, b])
    one_term = zk_amd.SumPolynomial([zk_amd.ProductPolynomial([a, a])])
    with pytest.raises(zk_amd.ZkError) as ei:      # F9: degree() = 1 cannot be interpolated at [0,1,2]
        zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(0), one_term, zk_amd.Transcript())
    assert ei.value.status == 5


@pytest.mark.parametrize("rounds_per_pass,lazy_from", [(1, 20), (2, 20), (2, 0), (2, 5), (2, -1)])
def test_sum2x2_both_schedules_give_the_oracle_proof(rounds_per_pass, lazy_from):
    """Two rounds per device pass (nine sums) or one (three evaluations), nine sums by per-product reduction or by
    unreduced accumulation (lazy_from = 0: every sweep): execution schedules, same proof bytes."""
    c = zk_amd.Context(0)
    c.set_deg2_rounds_per_pass(rounds_per_pass)
    c.set_deg2_lazy_sums(lazy_from)
    c.set_deg2_fused(0)        # every sweep through the nine-sum kernel under test (the fused form has its own test)
    c.set_deg2_small_triple(0)
    c.set_deg2_first_triple(0, 0)    # first pass: the nine-sum forms this test is about
    for log_n in (2, 3, 6, 7, 8, 9, 11, 12, 14, 17):
        n = 1 << log_n
        tabs = [O.gen_table(SEED + 90 + k, 0, n) for k in range(4)]
        m = [MLE(t, ctx=c) for t in tabs]
        f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
        tr, otr = zk_amd.Transcript(), O.OTranscript()
        proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
        ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
        assert np.array_equal(proof.round_polys, ocoeffs), (rounds_per_pass, log_n)
        assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()
        for t, src in zip(m, tabs):
            assert np.array_equal(t.get_evaluation_points(), src)      # the factor tables are left unmodified
    with pytest.raises(zk_amd.ZkError):
        c.set_deg2_rounds_per_pass(3)


# ---- the LDS-DMA forms of the big passes (the kernels the benchmark numbers come from), forced onto oracle-sized
# tables through zk_ctx_set_dma_gates and compared with the oracle directly -------------------------------------
@pytest.mark.parametrize("log_n", [7, 8, 14, 16, 18, 20])
def test_block_sums_dma_vs_oracle(log_n):
    c = zk_amd.Context(0)
    c.set_dma_gates(0, 22, 22)  # k_block_sums_dma on every table whose blocks are whole 64-entry rows
    n = 1 << log_n
    t = O.gen_table(SEED + 140 + log_n, 0, n) if log_n > 8 else O.ints_to_mont(rand_table(n, 140 + log_n))
    m = MLE(t, ctx=c)
    c.forms_used()
    e = m.round_sums()
    assert c.forms_used() == {"sums_dma"}
    assert list(e[0]) == list(O.mle_sum(t[: n // 2])) and list(e[1]) == list(O.mle_sum(t[n // 2:]))
    # worst case for the carry chains: every entry p - 1
    tm = O.ints_to_mont([P - 1] * 4096)
    e = MLE(tm, ctx=c).round_sums()
    assert O.mont_to_ints(e[0]) == [(P - 1) * 2048 % P] and O.mont_to_ints(e[1]) == [(P - 1) * 2048 % P]


@pytest.mark.parametrize("log_n,wide,min_log_out,fold_batch", [(14, 6, 8, 2), (16, 8, 8, 2), (17, 8, 9, 3), (18, 8, 10, 4), (20, 8, 12, 4),
                                                               (20, 7, 13, 4), (19, 5, 14, 2)])
@pytest.mark.parametrize("stream_stores", [False, True])
def test_sumcheck_dma_sums_and_dma_wide_fold_vs_oracle(log_n, wide, min_log_out, fold_batch, stream_stores):
    """k_block_sums_dma and k_multifold_wide_dma<BIG> (the two sweeps of the 2^28 benchmark) on 2^14 .. 2^20 tables:
    claim, coefficients and challenges against the oracle, and the form that ran is the one forced."""
    c = zk_amd.Context(0)
    c.set_dma_gates(0, 0 if stream_stores else 64, 22)
    c.set_wide_mfma(64)                    # the vector-ALU form of the large pass
    c.set_fold_batch(fold_batch, 0)
    c.set_wide_fold(wide, min_log_out, 0)  # weights by scalar loads
    c.set_tail_fold(0)
    table = O.gen_table(SEED + 160 + log_n, 3, 1 << log_n)
    poly = MLE(table, ctx=c)
    c.forms_used()
    proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
    forms = c.forms_used()
    assert "sums_dma" in forms and ("wide_dma_stream" if stream_stores else "wide_dma") in forms, forms
    assert not forms & {"wide_lds", "coop", "wide_dma" if stream_stores else "wide_dma_stream"}, forms
    oclaim, ocoeffs, och = O.sumcheck_prove(table) if log_n <= 16 else O.sumcheck_prove_fast(table)
    assert list(proof.initial_claim_sum) == list(oclaim)
    assert np.array_equal(proof.round_polys, ocoeffs)
    assert np.array_equal(ch, och)
    assert np.array_equal(poly.get_evaluation_points(), table)  # the input table is left unmodified
    if log_n <= 16:
        pts = [r for r in ch]
        assert np.array_equal(poly.evaluate(pts).get_evaluation_points(), O.mle_evaluate(table, pts))


def test_dma_forms_on_worst_case_limbs():
    """every entry p - 1 (all radix-2^28 columns at their bound) through the DMA sums and the DMA wide fold"""
    c = zk_amd.Context(0)
    c.set_dma_gates(0, 0, 22)
    c.set_wide_mfma(64)
    c.set_fold_batch(2, 0)
    c.set_wide_fold(8, 8, 0)
    c.set_tail_fold(0)
    table = O.ints_to_mont([P - 1] * (1 << 16))
    poly = MLE(table, ctx=c)
    c.forms_used()
    proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
    assert {"sums_dma", "wide_dma_stream"} <= c.forms_used()
    oclaim, ocoeffs, och = O.sumcheck_prove(table)
    assert list(proof.initial_claim_sum) == list(oclaim) and np.array_equal(proof.round_polys, ocoeffs) and np.array_equal(ch, och)


@pytest.mark.parametrize("log_n", [8, 10, 13, 14, 16])
@pytest.mark.parametrize("stream_stores", [False, True])
@pytest.mark.parametrize("lazy_from", [0, 20])
def test_sum2x2_dma_two_variable_fold_vs_oracle(log_n, stream_stores, lazy_from):
    """the m = 2 LDS-DMA fold of the degree-2 branch (4 x 2^26 benchmark shape) on small tables against the oracle"""
    c = zk_amd.Context(0)
    c.set_dma_gates(27, 0 if stream_stores else 64, 0)
    c.set_deg2_lazy_sums(lazy_from)
    c.set_deg2_fused(0)        # separate fold and sums launches: the forms under test
    c.set_deg2_small_triple(0)
    c.set_deg2_first_triple(0, 0)    # first pass: the nine-sum forms this test is about
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 180 + k, 0, n) for k in range(4)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
    forms = c.forms_used()
    assert "fold2_dma" in forms, forms
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs), log_n
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()
    for t, src in zip(m, tabs):
        assert np.array_equal(t.get_evaluation_points(), src)


def test_dma_gate_arguments():
    c = zk_amd.Context(0)
    for bad in ((-1, 22, 22), (27, 65, 22), (27, 22, -3)):
        with pytest.raises(zk_amd.ZkError):
            c.set_dma_gates(*bad)
    c.set_dma_gates(64, 64, 64)  # never
    m = MLE(O.gen_table(SEED, 0, 1 << 14), ctx=c)
    c.forms_used()
    m.round_sums()
    assert c.forms_used() == {"sums_reg"}


# ---- compute_quotient_remainder / minus (evaluation_form.rs:260-301; the division step of kzg/src/multilinear/prover.rs:59-96)
@pytest.mark.parametrize("log_n", [1, 2, 3, 6, 11, 16])
def test_quotient_remainder_every_variable_vs_oracle(ctx, log_n):
    n = 1 << log_n
    t = O.ints_to_mont(rand_table(n, 500 + log_n)) if log_n <= 11 else O.gen_table(SEED + 71, 0, n)
    m = MLE(t)
    rng = random.Random(log_n)
    variables = range(log_n) if log_n <= 6 else (0, 1, log_n // 2, log_n - 1)
    for var in variables:
        for dv in (0, 1, P - 1, rng.randrange(P)):
            d = O.fq1(dv)
            q, r = m.compute_quotient_remainder(d, var)
            oq, orr = O.mle_quotient_remainder(t, d, var)
            assert np.array_equal(q, oq), (var, dv)
            assert np.array_equal(r.get_evaluation_points(), orr), (var, dv)
            assert np.array_equal(r.get_evaluation_points(), m.partially_evaluate((var, d)).get_evaluation_points())
    assert np.array_equal(m.get_evaluation_points(), t)   # &self: the dividend is left unmodified
    with pytest.raises(zk_amd.ZkError) as ei:
        m.compute_quotient_remainder(O.fq1(1), log_n)      # variable index out of range
    assert ei.value.status == 2
    v = O.fq1(rng.randrange(P))
    assert np.array_equal(m.minus(v).get_evaluation_points(), O.mle_minus(t, v))


def test_kzg_opening_division_chain_ends_in_zero(ctx):
    """The loop of kzg/src/multilinear/prover.rs:59-96 on the device: f - f(a) divided by (x_i - a_i) one variable after
    the other leaves remainder 0, and every step satisfies y1 = rem - a q, y2 = rem + (1 - a) q."""
    rng = random.Random(5)
    log_n = 10
    t = O.gen_table(SEED + 72, 0, 1 << log_n)
    f = MLE(t)
    openings = [O.fq1(rng.randrange(P)) for _ in range(log_n)]
    v = f.evaluate(openings).get_evaluation_points()[0]
    dividend = f.minus(v)
    odiv = O.mle_minus(t, v)
    for a in openings:
        q, rem = dividend.compute_quotient_remainder(a, 0)
        oq, orem = O.mle_quotient_remainder(odiv, a, 0)
        assert np.array_equal(q, oq) and np.array_equal(rem.get_evaluation_points(), orem)
        cur = O.mont_to_ints(dividend.get_evaluation_points())
        qi, ri, ai = O.mont_to_ints(q), O.mont_to_ints(orem), O.mont_to_ints(a)[0]
        half = len(cur) // 2
        assert all((ri[j] - ai * qi[j]) % P == cur[j] and (ri[j] + (1 - ai) * qi[j]) % P == cur[half + j] for j in range(half))
        dividend, odiv = rem, orem
    assert O.mont_to_ints(dividend.get_evaluation_points()) == [0]


def test_upload_rejects_limbs_that_are_not_fully_reduced():
    """Limbs >= p are outside the contract (the radix-2^28 accumulations size their carries for canonical operands): an
    upload refuses them instead of proving garbage; wrap_device leaves the check to the caller."""
    c = zk_amd.Context(0)
    good = O.gen_table(SEED + 3, 0, 1 << 10)
    p_limbs = np.array([(P >> (64 * k)) & (2**64 - 1) for k in range(4)], dtype=np.uint64)
    for bad_entry in (p_limbs, np.full(4, 2**64 - 1, dtype=np.uint64)):
        t = good.copy()
        t[777] = bad_entry
        with pytest.raises(zk_amd.ZkError) as ei:
            MLE(t, ctx=c)
        assert ei.value.status == 8
    t = good.copy()
    t[5] = np.array([(P - 1 >> (64 * k)) & (2**64 - 1) for k in range(4)], dtype=np.uint64)   # p - 1 as raw limbs: the largest valid entry
    m = MLE(t, ctx=c)
    n = C.c_uint64(99)
    assert zk_amd.lib().zk_mle_count_noncanonical(m.h, C.byref(n)) == 0 and n.value == 0
    c.set_validate(False)
    t[9], t[10] = p_limbs, np.full(4, 2**64 - 1, dtype=np.uint64)
    m2 = MLE(t, ctx=c)                       # accepted unchecked ...
    assert zk_amd.lib().zk_mle_count_noncanonical(m2.h, C.byref(n)) == 0 and n.value == 2   # ... and the explicit check finds both


# ---- the wide pass on the matrix cores (k_multifold_wide_mfma): same field elements as every other form ------------------
@pytest.mark.parametrize("log_n,wide,min_log_out,fold_batch", [(14, 6, 8, 2), (16, 8, 8, 2), (17, 8, 9, 3), (18, 8, 10, 4), (20, 8, 12, 4),
                                                               (20, 7, 13, 4), (19, 5, 14, 2), (13, 5, 8, 2), (22, 8, 14, 4)])
@pytest.mark.parametrize("stream_stores,wg_per_cu,depth", [(False, 2, 2), (True, 1, 3), (True, 1, 4)])
def test_sumcheck_mfma_wide_fold_vs_oracle(log_n, wide, min_log_out, fold_batch, stream_stores, wg_per_cu, depth):
    c = zk_amd.Context(0)
    c.set_dma_gates(27, 0 if stream_stores else 64, 22)
    c.set_wide_mfma(0, wg_per_cu, depth)
    c.set_fold_batch(fold_batch, 0)
    c.set_wide_fold(wide, min_log_out, 0)
    c.set_tail_fold(0)
    table = O.gen_table(SEED + 200 + log_n, 1, 1 << log_n)
    poly = MLE(table, ctx=c)
    c.forms_used()
    proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
    forms = c.forms_used()
    assert "wide_mfma" in forms and not forms & {"wide_dma", "wide_dma_stream", "wide_lds", "coop"}, forms
    oclaim, ocoeffs, och = O.sumcheck_prove(table) if log_n <= 16 else O.sumcheck_prove_fast(table)
    assert list(proof.initial_claim_sum) == list(oclaim)
    assert np.array_equal(proof.round_polys, ocoeffs)
    assert np.array_equal(ch, och)
    if log_n <= 16:
        pts = [r for r in ch]
        assert np.array_equal(poly.evaluate(pts).get_evaluation_points(), O.mle_evaluate(table, pts))


@pytest.mark.parametrize("fill", ["p-1", "zero", "one", "byte80", "byteff"])
def test_mfma_wide_fold_on_extreme_bytes(fill):
    """every table byte 0x80 / 0xff-ish / 0 (the signed-byte offset at its limits), every entry p - 1 (largest columns)"""
    c = zk_amd.Context(0)
    c.set_wide_mfma(0, 2)
    c.set_fold_batch(2, 0)
    c.set_wide_fold(8, 8, 0)
    c.set_tail_fold(0)
    n = 1 << 16
    if fill == "p-1":
        table = O.ints_to_mont([P - 1] * n)
    elif fill == "zero":
        table = np.zeros((n, 4), dtype=np.uint64)
    elif fill == "one":
        table = O.ints_to_mont([1] * n)
    else:  # raw limbs (canonical: top byte below p's) with every other byte at the extreme
        b = 0x80 if fill == "byte80" else 0xff
        raw = int.from_bytes(bytes([b] * 31 + [0x2f]), "little")
        assert raw < P
        table = np.tile(np.array([(raw >> (64 * k)) & (2**64 - 1) for k in range(4)], dtype=np.uint64), (n, 1))
    poly = MLE(table, ctx=c)
    c.forms_used()
    proof, ch = zk_amd.SumcheckProver.generate_sumcheck_proof(poly, return_challenges=True)
    assert "wide_mfma" in c.forms_used()
    oclaim, ocoeffs, och = O.sumcheck_prove(table)
    assert list(proof.initial_claim_sum) == list(oclaim) and np.array_equal(proof.round_polys, ocoeffs) and np.array_equal(ch, och)


# ---- the nine sums on the matrix cores (k_sum16_mfma) -----------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [8, 9, 10, 12, 14, 16, 17])
def test_sum2x2_mfma_nine_sums_vs_oracle(log_n):
    c = zk_amd.Context(0)
    c.set_deg2_mfma_sums(0)   # every sweep whose quarter-tables are whole 64-entry rows
    c.set_deg2_fused(0)
    c.set_deg2_small_triple(0)
    c.set_deg2_first_triple(0, 0)    # first pass: the nine-sum forms this test is about
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 220 + k, 0, n) for k in range(4)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
    assert "sum9_mfma" in c.forms_used()
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs), log_n
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()


# ---- fold + nine sums in one kernel (k_fold2_sum9: small, cache-resident tables) ---------------------------------------------------
@pytest.mark.parametrize("fused_below", [0, 64])
@pytest.mark.parametrize("log_n", [7, 8, 9, 10, 12, 13, 15, 16, 18])
def test_sum2x2_fused_fold_and_sums_vs_oracle(log_n, fused_below):
    """fused_below = 64: every two-round pass that still needs sums runs k_fold2_sum9; 0: never (separate fold and sums launches).
    Same proof bytes as the oracle either way."""
    c = zk_amd.Context(0)
    c.set_deg2_fused(fused_below)
    c.set_deg2_triple(64, 3)
    c.set_deg2_small_triple(0)
    c.set_deg2_first_triple(0, 0)    # first pass: the nine-sum forms this test is about
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 300 + k, 0, n) for k in range(4)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
    forms = c.forms_used()
    assert ("fold2_sum9" in forms) == (fused_below == 64 and log_n >= 9), forms
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs), log_n
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()


# ---- three rounds per pass on small tables (k_fold_sum27: fold + the 27 sums of the next three rounds) ---------------------------
@pytest.mark.parametrize("cross_from", [64, 12])
@pytest.mark.parametrize("log_n", [7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18])
def test_sum2x2_small_tables_three_rounds_per_pass_vs_oracle(log_n, cross_from):
    """Every pass whose folded tables keep more than 64 entries runs k_fold_sum27 (M = 2 after a nine-sum pass, M = 3 after a 27-sum
    pass or -- cross_from = 12 -- after a matrix-core cross-sum pass): the host runs three rounds from 27 sums.  Few workgroups leave
    their shares to the host, more go through k_reduce27.  Same proof bytes as the oracle."""
    c = zk_amd.Context(0)
    c.set_deg2_small_triple(62)
    c.set_deg2_first_triple(0, 0)    # first pass: the nine-sum forms this test is about
    c.set_deg2_triple(cross_from, 4)
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 380 + k, 0, n) for k in range(4)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
    forms = c.forms_used()
    assert ("fold_sum27" in forms) == (log_n >= 9), forms
    if cross_from == 12 and log_n >= 12:
        assert "cross64" in forms, forms
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs), log_n
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()
    for t, src in zip(m, tabs):
        assert np.array_equal(t.get_evaluation_points(), src)      # the factor tables are left unmodified


@pytest.mark.parametrize("fill", ["p-1", "zero", "mixed"])
def test_sum2x2_small_tables_three_rounds_extreme_values(fill):
    """k_fold_sum27 on tables of p - 1 (every limb at its bound), of zeros, and a mix: same proof bytes as the oracle."""
    c = zk_amd.Context(0)
    c.set_deg2_small_triple(62)
    c.set_deg2_first_triple(0, 0)    # first pass: the nine-sum forms this test is about
    c.set_deg2_triple(64, 4)
    n = 1 << 11
    # raw Montgomery limbs of value p - 1 (the largest canonical limbs: every radix-2^28 column at its bound)
    pm1 = np.array([((P - 1) >> (64 * k)) & (2**64 - 1) for k in range(4)], dtype=np.uint64)
    tabs = []
    for k in range(4):
        if fill == "p-1":
            t = np.tile(pm1, (n, 1))
        elif fill == "zero":
            t = np.zeros((n, 4), dtype=np.uint64)
        else:
            t = O.gen_table(SEED + 390 + k, 0, n)
            t[::3] = pm1
            t[1::5] = 0
        tabs.append(np.ascontiguousarray(t))
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
    assert "fold_sum27" in c.forms_used()
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs)
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()


# ---- the product-plus-linear form F G + H * 1 of the 27-sum passes (the GKR phases) and the first pass as 27 sums ---------------------
def _ones_table(n):
    return np.ascontiguousarray(np.tile(O.fq1(1), (n, 1)))   # Montgomery form of 1


@pytest.mark.parametrize("mode", ["default", "all27", "off", "lazy9"])
@pytest.mark.parametrize("log_n", [0, 1, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 18])
def test_prod_plus_linear_vs_oracle(log_n, mode):
    """zk_sumcheck_prove_prod_plus_linear on {f, g, h} against the oracle's prover on SumPolynomial[Prod[f, g], Prod[h, ones]]:
    same round polynomials, challenges and transcript state.  default: the library's thresholds; all27: every pass above the host
    tail through k_fold_sum27<M, PWQ> (first pass M = 0, then M = 3); off: the general four-table kernels (the form turned off);
    lazy9: the first pass through k_sum9_lazy<PWQ> (the GKR phases' first sweep: the linear term as additions), then the 27-sum passes."""
    c = zk_amd.Context(0)
    if mode == "lazy9":
        c.set_deg2_first_triple(0, 1)     # the product + linear form on, but no table is small enough for a 27-sum FIRST pass
        c.set_deg2_triple(64, 4)
        c.set_deg2_lazy_sums(0)           # the unreduced-accumulation nine-sum sweep from 64 quarter-entries
    if mode == "all27":
        c.set_deg2_small_triple(62)
        c.set_deg2_first_triple(0, 62)
        c.set_deg2_triple(64, 4)
    elif mode == "off":
        c.set_deg2_first_triple(0, 0)
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 520 + k, 0, n) for k in range(3)]
    m = [MLE(t, ctx=c) for t in tabs]
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_product_plus_linear(O.fq1(5), m[0], m[1], m[2], tr)
    forms = c.forms_used()
    if mode == "lazy9":
        assert log_n < 8 or ("pwq" in forms and "sum9_lazy" in forms and "sum27_first" not in forms), forms
    else:
        assert ("pwq" in forms) == (mode != "off" and log_n >= 7), forms
    if mode == "all27" and log_n >= 10:
        assert "sum27_first" in forms and "fold_sum27" in forms and "sum9_lazy" not in forms and "sum9" not in forms, forms
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], _ones_table(n)]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs), log_n
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()
    for t, src in zip(m, tabs):
        assert np.array_equal(t.get_evaluation_points(), src)      # the factor tables are left unmodified


@pytest.mark.parametrize("first", ["27", "lazy9"])
@pytest.mark.parametrize("fill", ["p-1", "zero", "mixed"])
def test_prod_plus_linear_extreme_values(fill, first):
    c = zk_amd.Context(0)
    c.set_deg2_small_triple(62)
    if first == "27":
        c.set_deg2_first_triple(0, 62)
    else:                                  # first pass: k_sum9_lazy<PWQ> (plain 256-bit sums of up to four entries of p - 1 each)
        c.set_deg2_first_triple(0, 1)
        c.set_deg2_triple(64, 4)
        c.set_deg2_lazy_sums(0)
    n = 1 << 11
    pm1 = np.array([((P - 1) >> (64 * k)) & (2**64 - 1) for k in range(4)], dtype=np.uint64)
    tabs = []
    for k in range(3):
        if fill == "p-1":
            t = np.tile(pm1, (n, 1))
        elif fill == "zero":
            t = np.zeros((n, 4), dtype=np.uint64)
        else:
            t = O.gen_table(SEED + 530 + k, 0, n)
            t[::3] = pm1
            t[1::5] = 0
        tabs.append(np.ascontiguousarray(t))
    m = [MLE(t, ctx=c) for t in tabs]
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_product_plus_linear(O.fq1(5), m[0], m[1], m[2], tr)
    assert "pwq" in c.forms_used()
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], _ones_table(n)]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs)
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()


@pytest.mark.parametrize("log_n", [7, 8, 9, 10, 12, 13, 15, 17])
def test_sum2x2_first_pass_as_27_sums_vs_oracle(log_n):
    """The general four-table prover with its FIRST pass as 27 sums (k_fold_sum27<0>: three rounds before the first fold)."""
    c = zk_amd.Context(0)
    c.set_deg2_small_triple(62)
    c.set_deg2_first_triple(62, 21)
    c.set_deg2_triple(64, 4)
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 540 + k, 0, n) for k in range(4)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
    forms = c.forms_used()
    assert "sum27_first" in forms and "pwq" not in forms and "sum9_lazy" not in forms and "sum9" not in forms, forms
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs), log_n
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()


# ---- three rounds per pass: 8 x 8 cross sums per term (k_cross64_mfma) + the m = 3 fold -------------------------------------------
@pytest.mark.parametrize("ring", [3, 4])
@pytest.mark.parametrize("log_n", [9, 10, 11, 12, 13, 15, 17])
def test_sum2x2_three_rounds_per_pass_vs_oracle(log_n, ring):
    """Every pass whose tables are whole 64-entry eighths serves three rounds (cross sums on the matrix cores, host rounds from
    them, one three-variable fold); the rest of the prove runs the two-round and host forms.  Same proof bytes as the oracle."""
    c = zk_amd.Context(0)
    c.set_deg2_triple(0, ring)
    c.set_deg2_fold3_fused(0)   # separate fold and cross-sum sweeps (the fused sweep has its own test)
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 260 + k, 0, n) for k in range(4)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
    forms = c.forms_used()
    # folds with >= 2^12 outputs per table stream through the LDS-DMA kernel, smaller ones are one launch for all four tables
    assert "cross64" in forms and "fold3_reg" in forms and ("fold3_dma" in forms) == (log_n >= 15), forms
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs), log_n
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()
    for t, src in zip(m, tabs):
        assert np.array_equal(t.get_evaluation_points(), src)      # the factor tables are left unmodified


@pytest.mark.parametrize("raw", [True, False])
@pytest.mark.parametrize("slots", [3, 4, 5, 8])
@pytest.mark.parametrize("log_n", [12, 13, 15, 16, 18])
def test_sum2x2_fold_fused_with_next_cross_sums_vs_oracle(log_n, slots, raw):
    """k_fold3_cross64 (slots 3..5) / k_fold3_cross64_mfma (slots = 8: the fold on the matrix cores too): the three-variable fold and
    the next level's cross sums in one sweep (tables of >= 2^12 entries have a successor pass of whole 64-entry eighths).  Same proof
    bytes as the oracle.  raw: the sweeps leave integer digits for the second stage (default) / field elements per workgroup."""
    c = zk_amd.Context(0)
    c.set_deg2_triple(0, 4)
    c.set_deg2_fold3_fused(slots)
    c.set_deg2_raw_cross(raw)
    n = 1 << log_n
    tabs = [O.gen_table(SEED + 340 + k, 0, n) for k in range(4)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(5), f, tr)
    forms = c.forms_used()
    assert ("fold3_cross64_mfma" if slots == 8 else "fold3_cross64") in forms and "cross64" in forms, forms
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(5), otr)
    assert np.array_equal(proof.round_polys, ocoeffs), log_n
    assert np.array_equal(pts, och) and tr.peek_digest() == otr.peek_digest()
    for t, src in zip(m, tabs):
        assert np.array_equal(t.get_evaluation_points(), src)


@pytest.mark.parametrize("raw", [True, False])
@pytest.mark.parametrize("fill", ["p-1", "zero", "byteff", "mixed"])
def test_sum2x2_three_rounds_per_pass_extreme_bytes(fill, raw):
    c = zk_amd.Context(0)
    c.set_deg2_triple(0, 3)
    c.set_deg2_raw_cross(raw)
    n = 1 << 12

    def const_table(raw):
        assert raw < P
        return np.tile(np.array([(raw >> (64 * k)) & (2**64 - 1) for k in range(4)], dtype=np.uint64), (n, 1))

    if fill == "p-1":
        tabs = [O.ints_to_mont([P - 1] * n)] * 4
    elif fill == "zero":
        tabs = [np.zeros((n, 4), dtype=np.uint64)] * 4
    elif fill == "byteff":
        tabs = [const_table(int.from_bytes(bytes([0xff] * 31 + [0x2f]), "little"))] * 4
    else:
        tabs = [const_table(int.from_bytes(bytes([0xff] * 31 + [0x2f]), "little")), np.zeros((n, 4), dtype=np.uint64),
                const_table(int.from_bytes(bytes([0x7f, 0x80] * 15 + [0x00, 0x30]), "little")), O.gen_table(SEED, 0, n)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(1), f, tr)
    assert "cross64" in c.forms_used()
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(1), otr)
    assert np.array_equal(proof.round_polys, ocoeffs) and np.array_equal(pts, och)


@pytest.mark.parametrize("fill", ["p-1", "zero", "byte80", "byteff", "mixed"])
def test_sum2x2_mfma_nine_sums_extreme_bytes(fill):
    c = zk_amd.Context(0)
    c.set_deg2_mfma_sums(0)
    c.set_deg2_fused(0)
    c.set_deg2_small_triple(0)
    c.set_deg2_first_triple(0, 0)    # first pass: the nine-sum forms this test is about
    n = 1 << 12

    def const_table(raw):
        assert raw < P
        return np.tile(np.array([(raw >> (64 * k)) & (2**64 - 1) for k in range(4)], dtype=np.uint64), (n, 1))

    if fill == "p-1":
        tabs = [O.ints_to_mont([P - 1] * n)] * 4
    elif fill == "zero":
        tabs = [np.zeros((n, 4), dtype=np.uint64)] * 4
    elif fill == "byte80":
        tabs = [const_table(int.from_bytes(bytes([0x80] * 31 + [0x2f]), "little"))] * 4
    elif fill == "byteff":
        tabs = [const_table(int.from_bytes(bytes([0xff] * 31 + [0x2f]), "little"))] * 4
    else:
        tabs = [const_table(int.from_bytes(bytes([0xff] * 31 + [0x2f]), "little")), np.zeros((n, 4), dtype=np.uint64),
                const_table(int.from_bytes(bytes([0x7f, 0x80] * 15 + [0x00, 0x30]), "little")), O.gen_table(SEED, 0, n)]
    m = [MLE(t, ctx=c) for t in tabs]
    f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(m[:2]), zk_amd.ProductPolynomial(m[2:])])
    tr, otr = zk_amd.Transcript(), O.OTranscript()
    c.forms_used()
    proof, pts = zk_amd.SumcheckProver.generate_proof_for_partial_verify(O.fq1(1), f, tr)
    assert "sum9_mfma" in c.forms_used()
    ocoeffs, och = O.sumcheck_prove_sum([[tabs[0], tabs[1]], [tabs[2], tabs[3]]], O.fq1(1), otr)
    assert np.array_equal(proof.round_polys, ocoeffs) and np.array_equal(pts, och)
