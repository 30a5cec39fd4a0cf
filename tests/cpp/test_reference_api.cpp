// tests/cpp/test_reference_api.cpp -- the reference's own unit tests for the sumcheck / GKR path, replayed
// through the C++ host layer (include/zkmi.hpp) on the GPU.  Each test names the Rust test it restates; the
// vectors are the small-integer vectors those tests hold.  Needs an MI355X.
// Build: g++ -O2 -std=c++17 -Iinclude tests/cpp/test_reference_api.cpp -Lzk_amd -lzkmi -Wl,-rpath,'$ORIGIN/../../zk_amd' -o tests/cpp/test_reference_api
#include <cstdio>
#include <functional>
#include <string>

#include "zkmi.hpp"

using namespace zk;
using MLE = MultiLinearPolynomial;

static int g_fail = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("    FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)

static std::vector<Fq> F(std::initializer_list<uint64_t> v) { return std::vector<Fq>(v.begin(), v.end()); }
static bool eq(const std::vector<Fq>& a, std::initializer_list<uint64_t> v) { return a == F(v); }
template <class Fn>
static bool panics_with(int status, Fn&& fn) {
    try { fn(); } catch (const Error& e) { return e.status == status; }
    return false;
}
static std::string hex(const std::array<uint8_t, 32>& d) {
    std::string s;
    char b[3];
    for (uint8_t x : d) { std::snprintf(b, 3, "%02x", x); s += b; }
    return s;
}

// 3ac + 4bd + 5ab and 2ab + 3bc (evaluation_form.rs:331-363)
static const std::initializer_list<uint64_t> POLY4 = {0, 0, 0, 0, 0, 4, 0, 4, 0, 0, 3, 3, 5, 9, 8, 12};
static const std::initializer_list<uint64_t> POLY3 = {0, 0, 0, 3, 0, 0, 2, 5};

static void test_evaluate_4_variables() { // evaluation_form.rs:366-382
    MLE p(F(POLY4));
    CHECK(eq(p.evaluate({Fq(4), Fq(2), Fq(6), Fq(1)}).get_evaluation_points(), {120}));
}
static void test_partially_evaluate_incomplete_points() { // evaluation_form.rs:385-419
    MLE p(F(POLY4));
    CHECK(eq(p.evaluate({Fq(4), std::nullopt, std::nullopt, std::nullopt}).get_evaluation_points(), {0, 0, 12, 12, 20, 24, 32, 36}));
    CHECK(eq(p.partially_evaluate({0, Fq(4)}).get_evaluation_points(), {0, 0, 12, 12, 20, 24, 32, 36}));
    MLE q(F(POLY3));
    CHECK(eq(q.evaluate({std::nullopt, std::nullopt, Fq(3)}).get_evaluation_points(), {0, 9, 0, 11}));
    CHECK(panics_with(ZK_E_POINTS_LEN, [&] { q.evaluate({Fq(1), Fq(2)}); })); // :197-199
    CHECK(panics_with(ZK_E_NOT_POW2, [&] { MLE bad(F({1, 2, 3})); }));        // :23-27
}
static void test_quotient_remainder() { // evaluation_form.rs:280-301 on the fold vector of :385-405, and the KZG division chain
    MLE p(F(POLY4));
    auto [q, r] = p.compute_quotient_remainder(Fq(4), 0);
    CHECK(eq(r.get_evaluation_points(), {0, 0, 12, 12, 20, 24, 32, 36}));
    CHECK(eq(q, {0, 0, 3, 3, 5, 5, 8, 8}));
    // kzg/src/multilinear/prover.rs:59-96: (f - f(a)) divided by every (x_i - a_i) leaves 0
    const std::vector<Fq> a = {Fq(4), Fq(2), Fq(6), Fq(1)};
    MLE dividend = p.minus(Fq(120));
    for (const Fq& ai : a) dividend = dividend.compute_quotient_remainder(ai, 0).second;
    CHECK(eq(dividend.get_evaluation_points(), {0}));
    CHECK(panics_with(ZK_E_POINTS_LEN, [&] { p.compute_quotient_remainder(Fq(1), 4); }));
}
static void test_operation_ws() { // evaluation_form.rs:422-473
    MLE wb(F({0, 0, 0, 3})), wc(F({0, 0, 0, 2}));
    CHECK(eq(MLE::w_add(wb, wc).get_evaluation_points(), {0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 2, 3, 3, 3, 5}));
    CHECK(eq(wb.w_mul(wc).get_evaluation_points(), {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6}));
    CHECK(eq(wb.scalar_mul(Fq(5)).get_evaluation_points(), {0, 0, 0, 15}));
    CHECK(eq((wb + wc).get_evaluation_points(), {0, 0, 0, 5}));
    CHECK(wb.evaluation_sum() == Fq(3));
    CHECK(wb.to_bytes().size() == 128 && wb.to_bytes()[96] == 3);
}
static void test_transcript() { // transcript.rs:122-157 + the public Keccak-256 known answers
    Transcript a, b;
    CHECK(hex(a.digest()) == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470");
    const uint8_t abc[3] = {'a', 'b', 'c'};
    a.append(abc, 3);
    b.append(abc, 3);
    CHECK(hex(a.digest()) == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45");
    CHECK(a.sample_challenge() == b.sample_challenge());
    auto x = a.sample_n_challenges(3), y = b.sample_n_challenges(3);
    CHECK(x == y && x[0] != x[1]);
}
static void test_full_sumcheck_pass() { // sumcheck_protocol.rs:27-49
    MLE poly(F(POLY3));
    SumCheckProof proof = SumcheckProver::generate_sumcheck_proof(poly);
    CHECK(proof.initial_claim_sum == Fq(10) && proof.round_polys.size() == 3);
    CHECK(SumcheckVerifier::verify_proof(poly, proof));
    proof.round_polys[1].coefficients[0] = proof.round_polys[1].coefficients[0] + Fq(1);
    CHECK(!SumcheckVerifier::verify_proof(poly, proof));
}
static void test_full_sumcheck_fail() { // sumcheck_protocol.rs:52-82
    MLE poly(F({0, 0, 0, 3, 0, 0, 2, 65}));
    SumCheckProof forged{Fq(10), {UnivariatePolynomial{F({3, 7})}, UnivariatePolynomial{F({9, 10})}, UnivariatePolynomial{F({10, 97})}}};
    CHECK(!SumcheckVerifier::verify_proof(poly, forged));
}
static void test_partial_sumcheck_pass() { // sumcheck_protocol.rs:85-111
    auto e1 = F({0, 0, 0, 2}), e2 = F({0, 0, 0, 3});
    SumPolynomial f({ProductPolynomial({MLE(e1), MLE(e2)}), ProductPolynomial({MLE(e1), MLE(e2)})});
    Transcript pt, vt;
    auto [proof, pts] = SumcheckProver::generate_proof_for_partial_verify(Fq(12), f, pt);
    auto [ok, final_claim, ch] = SumcheckVerifier::partial_verify(proof, vt);
    CHECK(ok && ch.size() == 2 && *ch[0] == pts[0] && *ch[1] == pts[1]);
    // the final claim is f at the challenges: 2 * e1(r) * e2(r)
    Fq a = MLE(e1).evaluate({pts[0], pts[1]}).get_evaluation_points()[0], b = MLE(e2).evaluate({pts[0], pts[1]}).get_evaluation_points()[0];
    CHECK(final_claim == Fq(2) * a * b);
    CHECK(proof.round_polys[1].evaluate(pts[1]) == final_claim);
    // a single product term cannot go through the degree-2 branch
    SumPolynomial one({ProductPolynomial({MLE(e1), MLE(e2)})});
    Transcript t3;
    CHECK(panics_with(ZK_E_DEGREE, [&] { SumcheckProver::generate_proof_for_partial_verify(Fq(6), one, t3); }));
    CHECK(panics_with(ZK_E_LEN_MISMATCH, [&] { ProductPolynomial({MLE(e1), MLE(F({1, 2}))}); })); // product_polynomial.rs:15-19
}
static void test_product_plus_linear_equals_the_four_table_form() { // the GKR phase relation f g + h (gkr/src/prover.rs:55-147: Product[h, ones])
    std::vector<Fq> fv, gv, hv, ones;
    for (unsigned i = 0; i < 256; ++i) { // 256 entries: above the host tail, so the device kernels of the product-plus-linear form run
        fv.push_back(Fq(3 * i + 1)); gv.push_back(Fq(i * i + 7)); hv.push_back(Fq(1000003u * i + 11)); ones.push_back(Fq(1));
    }
    SumPolynomial four({ProductPolynomial({MLE(fv), MLE(gv)}), ProductPolynomial({MLE(hv), MLE(ones)})});
    Transcript t4, t3;
    auto [p4, pts4] = SumcheckProver::generate_proof_for_partial_verify(Fq(5), four, t4);
    auto [p3, pts3] = SumcheckProver::generate_proof_product_plus_linear(Fq(5), MLE(fv), MLE(gv), MLE(hv), t3);
    CHECK(p4.round_polys.size() == 8 && p3.round_polys.size() == 8 && pts3.size() == 8);
    for (size_t k = 0; k < 8; ++k) {
        CHECK(pts3[k] == pts4[k]);
        for (size_t j = 0; j < 3; ++j) CHECK(p3.round_polys[k].coefficients[j] == p4.round_polys[k].coefficients[j]);
    }
    CHECK(t3.sample_challenge() == t4.sample_challenge()); // the transcripts end in the same state
}
static void test_circuit() { // circuit.rs:159-202
    Circuit c({{Gate(0, 1, Operation::Add), Gate(2, 3, Operation::Mul)}, {Gate(0, 1, Operation::Add)}});
    auto ev = c.evaluate_at_input(F({1, 2, 3, 4}));
    CHECK(ev.size() == 3 && eq(ev[2].get_evaluation_points(), {15, 0}) && eq(ev[1].get_evaluation_points(), {3, 12}));
    auto add1 = c.get_add_i(1).get_evaluation_points(), mul1 = c.get_mul_i(1).get_evaluation_points();
    CHECK(add1.size() == 32 && mul1.size() == 32);
    for (size_t i = 0; i < 32; ++i) { CHECK(add1[i] == Fq(i == 1 ? 1 : 0)); CHECK(mul1[i] == Fq(i == 27 ? 1 : 0)); }
    CHECK(panics_with(ZK_E_LAYER_OOB, [&] { c.get_add_i(2); }));
}
static void test_gkr_sum_check() { // gkr_protocol.rs:67-142
    Circuit circuit({{Gate(0, 1, Operation::Add), Gate(2, 3, Operation::Add), Gate(4, 5, Operation::Add), Gate(6, 7, Operation::Mul)},
                     {Gate(0, 1, Operation::Mul), Gate(2, 3, Operation::Add)},
                     {Gate(0, 1, Operation::Add)}});
    auto inputs = F({1, 2, 3, 4, 5, 6, 7, 8});
    Transcript pt, vt, vt2;
    GKRProof proof = GKRProver::generate_proof(circuit, pt, inputs);
    CHECK(proof.sumcheck_proofs.size() == 3 && proof.w_polys_evals.size() == 2 && proof.output_poly.size() == 2);
    CHECK(GKRVerifier::verify_proof(inputs, circuit, vt, proof));
    auto wrong = inputs;
    wrong[0] = Fq(2);
    CHECK(!GKRVerifier::verify_proof(wrong, circuit, vt2, proof));
}

int main() {
    const std::pair<const char*, std::function<void()>> tests[] = {
        {"test_evaluate_4_variables", test_evaluate_4_variables},
        {"test_partially_evaluate_incomplete_points", test_partially_evaluate_incomplete_points},
        {"test_quotient_remainder", test_quotient_remainder},
        {"test_operation_ws", test_operation_ws},
        {"test_transcript", test_transcript},
        {"test_full_sumcheck_pass", test_full_sumcheck_pass},
        {"test_full_sumcheck_fail", test_full_sumcheck_fail},
        {"test_partial_sumcheck_pass", test_partial_sumcheck_pass},
        {"test_product_plus_linear_equals_the_four_table_form", test_product_plus_linear_equals_the_four_table_form},
        {"test_circuit", test_circuit},
        {"test_gkr_sum_check", test_gkr_sum_check},
    };
    for (auto& t : tests) {
        const int before = g_fail;
        try { t.second(); } catch (const std::exception& e) { std::printf("    EXCEPTION %s\n", e.what()); ++g_fail; }
        std::printf("%s %s\n", g_fail == before ? "ok    " : "FAILED", t.first);
    }
    std::printf("%d failure(s)\n", g_fail);
    return g_fail ? 1 : 0;
}
