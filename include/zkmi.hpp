// zkmi.hpp -- C++ host layer over the C ABI of libzkmi.so (include/zkmi.h), mirroring the public API of
// the reference's `polynomials`, `fiat_shamir`, `sumcheck`, `arithmetic_circuit` and `gkr` crates for the
// sumcheck path: same type and method names, same argument meaning, and where the reference panics a
// zk::Error carrying the matching zk_status is thrown.  Header-only; link with -lzkmi.
//
// The reference is Rust and no Rust toolchain exists in the build image, so this is the
// compiled-language host side; tests/cpp/test_reference_api.cpp replays the reference's own unit tests
// against it.  Reference files are cited per item (paths under /root/reference).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "zkmi.h"

namespace zk {

struct Error : std::runtime_error { // a reference panic
    int status;
    explicit Error(int s, const std::string& detail = "") : std::runtime_error(std::string(zk_strerror(s)) + (detail.empty() ? "" : " [" + detail + "]")), status(s) {}
};

// ark_bn254::Fq in ark-ff's in-memory form: 4 x u64 Montgomery limbs
struct Fq {
    uint64_t l[4] = {0, 0, 0, 0};
    Fq() = default;
    Fq(uint64_t v) { zk_fq_from_u64(v, l); } // T::from(v)
    static Fq from_limbs(const uint64_t* p) { Fq f; std::memcpy(f.l, p, 32); return f; }
    bool operator==(const Fq& o) const { return std::memcmp(l, o.l, 32) == 0; }
    bool operator!=(const Fq& o) const { return !(*this == o); }
    Fq operator+(const Fq& o) const { Fq r; zk_fq_add(l, o.l, r.l); return r; }
    Fq operator-(const Fq& o) const { Fq r; zk_fq_sub(l, o.l, r.l); return r; }
    Fq operator*(const Fq& o) const { Fq r; zk_fq_mul(l, o.l, r.l); return r; }
    std::array<uint8_t, 32> to_bytes_le() const { std::array<uint8_t, 32> b; zk_fq_to_le32(l, b.data()); return b; } // into_bigint().to_bytes_le()
};
static_assert(sizeof(Fq) == 32, "Fq must be 4 plain limbs");

class Context { // one GPU (zk_ctx)
  public:
    explicit Context(int device = 0) { int rc = zk_ctx_create(device, &h_); if (rc) throw Error(rc); }
    ~Context() { zk_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    zk_ctx* get() const { return h_; }
    void check(int rc) const { if (rc) throw Error(rc, zk_last_error(h_)); }
    static const std::shared_ptr<Context>& default_context() { static std::shared_ptr<Context> c = std::make_shared<Context>(0); return c; }
  private:
    zk_ctx* h_ = nullptr;
};
using Ctx = std::shared_ptr<Context>;

// fiat_shamir::Transcript (fiat_shamir/src/transcript.rs:6-42)
class Transcript {
  public:
    Transcript() : h_(zk_transcript_new()) { if (!h_) throw Error(ZK_E_ALLOC); } // Transcript::new :12-17
    Transcript(const Transcript& o) : h_(zk_transcript_clone(o.h_)) { if (!h_) throw Error(ZK_E_ALLOC); }
    Transcript& operator=(const Transcript&) = delete;
    ~Transcript() { zk_transcript_free(h_); }
    void append(const uint8_t* data, size_t len) { zk_transcript_append(h_, data, len); } // :20-22
    void append(const std::vector<uint8_t>& d) { append(d.data(), d.size()); }
    void append_n(const std::vector<std::vector<uint8_t>>& parts) { for (auto& p : parts) append(p); } // :24-26
    Fq sample_challenge() { Fq r; zk_transcript_sample_challenge(h_, r.l); return r; } // :28-36
    std::vector<Fq> sample_n_challenges(size_t n) { std::vector<Fq> v(n); for (auto& x : v) x = sample_challenge(); return v; } // :39-41
    std::array<uint8_t, 32> digest() const { std::array<uint8_t, 32> d; zk_transcript_peek_digest(h_, d.data()); return d; }
    zk_transcript* get() const { return h_; }
  private:
    zk_transcript* h_;
};

// polynomials::multilinear_polynomial::evaluation_form::MultiLinearPolynomial (evaluation_form.rs:17-313);
// the evaluation table lives in HBM.  Copying is a deep copy (#[derive(Clone)]).
class MultiLinearPolynomial {
  public:
    explicit MultiLinearPolynomial(const std::vector<Fq>& evaluation_points, Ctx ctx = Context::default_context()) : ctx_(std::move(ctx)) { // new :22-32
        if (evaluation_points.empty()) throw Error(ZK_E_NOT_POW2);
        ctx_->check(zk_mle_upload_mont(ctx_->get(), evaluation_points[0].l, evaluation_points.size(), &h_));
    }
    MultiLinearPolynomial(const MultiLinearPolynomial& o) : ctx_(o.ctx_) { ctx_->check(zk_mle_clone(o.h_, &h_)); }
    MultiLinearPolynomial(MultiLinearPolynomial&& o) noexcept : ctx_(std::move(o.ctx_)), h_(o.h_) { o.h_ = nullptr; }
    MultiLinearPolynomial& operator=(MultiLinearPolynomial o) { std::swap(ctx_, o.ctx_); std::swap(h_, o.h_); return *this; }
    ~MultiLinearPolynomial() { zk_mle_free(h_); }
    static MultiLinearPolynomial adopt(zk_mle* h, Ctx ctx) { return MultiLinearPolynomial(h, std::move(ctx)); }

    size_t len() const { return zk_mle_len(h_); }
    uint32_t number_of_variables() const { uint32_t k = 0; for (size_t n = len(); n > 1; n >>= 1) ++k; return k; } // :159-161
    std::vector<Fq> get_evaluation_points() const { std::vector<Fq> v(len()); ctx_->check(zk_mle_download_mont(h_, v[0].l)); return v; } // :155-157
    MultiLinearPolynomial partially_evaluate(std::pair<size_t, Fq> variable) const { // :163-192
        zk_mle* o = nullptr;
        ctx_->check(zk_mle_partially_evaluate(h_, variable.first, variable.second.l, &o));
        return adopt(o, ctx_);
    }
    MultiLinearPolynomial evaluate(const std::vector<std::optional<Fq>>& points) const { // :194-223
        std::vector<uint8_t> has(points.size() + 1, 0);
        std::vector<Fq> vals(points.size() + 1);
        for (size_t i = 0; i < points.size(); ++i)
            if (points[i]) { has[i] = 1; vals[i] = *points[i]; }
        zk_mle* o = nullptr;
        ctx_->check(zk_mle_evaluate(h_, has.data(), vals[0].l, points.size(), &o));
        return adopt(o, ctx_);
    }
    std::vector<uint8_t> to_bytes() const { std::vector<uint8_t> b(32 * len()); ctx_->check(zk_mle_to_le32(h_, b.data())); return b; } // :225-234
    Fq evaluation_sum() const { Fq s; ctx_->check(zk_mle_evaluation_sum(h_, s.l)); return s; } // :236-238
    MultiLinearPolynomial scalar_mul(const Fq& s) const { zk_mle* o = nullptr; ctx_->check(zk_mle_scalar_mul(h_, s.l, &o)); return adopt(o, ctx_); } // :151-153
    MultiLinearPolynomial minus(const Fq& other) const { zk_mle* o = nullptr; ctx_->check(zk_mle_minus(h_, other.l, &o)); return adopt(o, ctx_); } // :260-272
    // compute_quotient_remainder(&divisor, variable_index) :280-301 -> (quotient: Vec<T>, remainder)
    std::pair<std::vector<Fq>, MultiLinearPolynomial> compute_quotient_remainder(const Fq& divisor, size_t variable_index) const {
        zk_mle *q = nullptr, *r = nullptr;
        ctx_->check(zk_mle_quotient_remainder(h_, variable_index, divisor.l, &q, &r));
        MultiLinearPolynomial quotient = adopt(q, ctx_);
        return {quotient.get_evaluation_points(), adopt(r, ctx_)};
    }
    MultiLinearPolynomial operator+(const MultiLinearPolynomial& b) const { zk_mle* o = nullptr; ctx_->check(zk_mle_add(h_, b.h_, &o)); return adopt(o, ctx_); } // :241-258
    static MultiLinearPolynomial w_add(const MultiLinearPolynomial& w_b, const MultiLinearPolynomial& w_c) { // :303-308
        zk_mle* o = nullptr; w_b.ctx_->check(zk_mle_w_add(w_b.h_, w_c.h_, &o)); return adopt(o, w_b.ctx_);
    }
    MultiLinearPolynomial w_mul(const MultiLinearPolynomial& other) const { zk_mle* o = nullptr; ctx_->check(zk_mle_w_mul(h_, other.h_, &o)); return adopt(o, ctx_); } // :310-312
    zk_mle* get() const { return h_; }
    const Ctx& ctx() const { return ctx_; }
  private:
    MultiLinearPolynomial(zk_mle* h, Ctx ctx) : ctx_(std::move(ctx)), h_(h) {}
    Ctx ctx_;
    zk_mle* h_ = nullptr;
};

// polynomials::product_polynomial::ProductPolynomial (product_polynomial.rs:7-78)
struct ProductPolynomial {
    std::vector<MultiLinearPolynomial> polys;
    explicit ProductPolynomial(std::vector<MultiLinearPolynomial> p) : polys(std::move(p)) { // new :12-22
        if (polys.empty()) throw Error(ZK_E_LEN_MISMATCH);
        for (auto& q : polys) if (q.len() != polys[0].len()) throw Error(ZK_E_LEN_MISMATCH);
    }
    size_t length() const { return polys[0].len(); }
};
// polynomials::sum_polynomial::SumPolynomial (sum_polynomial.rs:7-82)
struct SumPolynomial {
    std::vector<ProductPolynomial> prod_polys;
    explicit SumPolynomial(std::vector<ProductPolynomial> p) : prod_polys(std::move(p)) { // new :12-22
        if (prod_polys.empty()) throw Error(ZK_E_LEN_MISMATCH);
        for (auto& q : prod_polys) if (q.length() != prod_polys[0].length()) throw Error(ZK_E_LEN_MISMATCH);
    }
    size_t degree() const { return prod_polys.size(); } // :79-81 -- the number of product terms
    size_t number_of_variables() const { size_t k = 0; for (size_t n = prod_polys[0].length(); n > 1; n >>= 1) ++k; return k; }
};

// polynomials::univariate_polynomial::dense_coefficient_form::UnivariatePolynomial (coefficients ascending)
struct UnivariatePolynomial {
    std::vector<Fq> coefficients;
    Fq evaluate(const Fq& x) const { Fq r; zk_unipoly_evaluate(coefficients.empty() ? nullptr : coefficients[0].l, coefficients.size(), x.l, r.l); return r; } // :20-34
};

// sumcheck::sumcheck_protocol::SumCheckProof (sumcheck_protocol.rs:5-9)
struct SumCheckProof {
    Fq initial_claim_sum;
    std::vector<UnivariatePolynomial> round_polys;
};

namespace detail {
inline std::vector<Fq> flatten(const std::vector<UnivariatePolynomial>& rp, size_t ncoef) {
    std::vector<Fq> f;
    for (auto& g : rp) {
        if (g.coefficients.size() != ncoef) throw Error(ZK_E_ARG);
        f.insert(f.end(), g.coefficients.begin(), g.coefficients.end());
    }
    if (f.empty()) f.resize(1);
    return f;
}
inline std::vector<UnivariatePolynomial> unflatten(const std::vector<Fq>& c, size_t rounds, size_t ncoef) {
    std::vector<UnivariatePolynomial> rp(rounds);
    for (size_t k = 0; k < rounds; ++k) rp[k].coefficients.assign(c.begin() + k * ncoef, c.begin() + (k + 1) * ncoef);
    return rp;
}
} // namespace detail

// sumcheck::prover::SumcheckProver (sumcheck/src/prover.rs:17-191)
struct SumcheckProver {
    static SumCheckProof generate_sumcheck_proof(const MultiLinearPolynomial& init_polynomial) { // :158-173
        const size_t n = init_polynomial.number_of_variables();
        SumCheckProof p;
        std::vector<Fq> coeffs(2 * n + 1);
        auto& c = init_polynomial.ctx();
        c->check(zk_sumcheck_prove_mle(c->get(), init_polynomial.get(), 1, nullptr, p.initial_claim_sum.l, coeffs[0].l, nullptr));
        p.round_polys = detail::unflatten(coeffs, n, 2);
        return p;
    }
    static std::pair<SumCheckProof, std::vector<Fq>> generate_proof_for_partial_verify(const Fq& initial_claim_sum, const SumPolynomial& init_poly,
                                                                                      Transcript& transcript) { // :175-190
        if (init_poly.degree() != 2) throw Error(ZK_E_DEGREE);
        for (auto& pp : init_poly.prod_polys) if (pp.polys.size() != 2) throw Error(ZK_E_DEGREE);
        zk_mle* t[4] = {init_poly.prod_polys[0].polys[0].get(), init_poly.prod_polys[0].polys[1].get(), init_poly.prod_polys[1].polys[0].get(),
                        init_poly.prod_polys[1].polys[1].get()};
        const size_t n = init_poly.number_of_variables();
        std::vector<Fq> coeffs(3 * n + 1), pts(n + 1);
        auto& c = init_poly.prod_polys[0].polys[0].ctx();
        c->check(zk_sumcheck_prove_sum2x2(c->get(), t, initial_claim_sum.l, transcript.get(), coeffs[0].l, pts[0].l));
        pts.resize(n);
        return {SumCheckProof{initial_claim_sum, detail::unflatten(coeffs, n, 3)}, pts};
    }
    // The same for SumPolynomial[Product[f, g], Product[h, ones]] -- the relation f g + h of a GKR phase (gkr/src/prover.rs:55-147) -- without
    // the table of ones: same proof bytes, the constant factor is never read (zk_sumcheck_prove_prod_plus_linear).
    static std::pair<SumCheckProof, std::vector<Fq>> generate_proof_product_plus_linear(const Fq& initial_claim_sum, const MultiLinearPolynomial& f,
                                                                                       const MultiLinearPolynomial& g, const MultiLinearPolynomial& h,
                                                                                       Transcript& transcript) {
        zk_mle* t[3] = {f.get(), g.get(), h.get()};
        const size_t n = f.number_of_variables();
        std::vector<Fq> coeffs(3 * n + 1), pts(n + 1);
        auto& c = f.ctx();
        c->check(zk_sumcheck_prove_prod_plus_linear(c->get(), t, initial_claim_sum.l, transcript.get(), coeffs[0].l, pts[0].l));
        pts.resize(n);
        return {SumCheckProof{initial_claim_sum, detail::unflatten(coeffs, n, 3)}, pts};
    }
};

// sumcheck::verifier::SumcheckVerifier (sumcheck/src/verifier.rs:9-95)
struct SumcheckVerifier {
    static std::tuple<bool, Fq, std::vector<std::optional<Fq>>> partial_verify(const SumCheckProof& proof, Transcript& transcript) { // :29-62
        const size_t rounds = proof.round_polys.size(), ncoef = rounds ? proof.round_polys[0].coefficients.size() : 2;
        auto flat = detail::flatten(proof.round_polys, ncoef);
        int ok = 0;
        Fq final_claim;
        std::vector<Fq> ch(rounds + 1);
        int rc = zk_sumcheck_partial_verify(proof.initial_claim_sum.l, flat[0].l, rounds, ncoef, transcript.get(), &ok, final_claim.l, ch[0].l);
        if (rc) throw Error(rc);
        std::vector<std::optional<Fq>> pts;
        for (size_t k = 0; k < rounds; ++k) pts.emplace_back(ch[k]);
        return {ok != 0, final_claim, pts};
    }
    static bool verify_proof(const MultiLinearPolynomial& initial_polynomial, const SumCheckProof& proof) { // :64-94
        const size_t rounds = proof.round_polys.size();
        auto flat = detail::flatten(proof.round_polys, 2);
        int ok = 0;
        auto& c = initial_polynomial.ctx();
        c->check(zk_sumcheck_verify_mle(c->get(), initial_polynomial.get(), proof.initial_claim_sum.l, flat[0].l, rounds, &ok));
        return ok != 0;
    }
};

// arithmetic_circuit::gate (gate.rs) and ::circuit::Circuit (circuit.rs:9-152)
enum class Operation : uint64_t { Add = 0, Mul = 1 };
struct Gate {
    size_t left, right;
    Operation operation;
    Gate(size_t l, size_t r, Operation op) : left(l), right(r), operation(op) {} // Gate::new
};
class Circuit {
  public:
    explicit Circuit(const std::vector<std::vector<Gate>>& layers, Ctx ctx = Context::default_context()) : ctx_(std::move(ctx)) { // new :15-20
        std::vector<uint64_t> flat;
        std::vector<size_t> lens;
        for (auto& gs : layers) {
            lens.push_back(gs.size());
            for (auto& g : gs) { flat.push_back(g.left); flat.push_back(g.right); flat.push_back((uint64_t)g.operation); }
        }
        if (flat.empty()) flat.push_back(0);
        if (lens.empty()) lens.push_back(0);
        ctx_->check(zk_circuit_create(ctx_->get(), flat.data(), lens.data(), layers.size(), &h_));
    }
    Circuit(const Circuit&) = delete;
    Circuit& operator=(const Circuit&) = delete;
    ~Circuit() { zk_circuit_free(h_); }
    size_t get_layer_count() const { return zk_circuit_layer_count(h_); } // :149-151
    std::vector<MultiLinearPolynomial> evaluate_at_input(const std::vector<Fq>& inputs) { // :37-59
        MultiLinearPolynomial in(inputs, ctx_);
        std::vector<zk_mle*> out(get_layer_count() + 1, nullptr);
        ctx_->check(zk_circuit_evaluate(h_, in.get(), out.data()));
        std::vector<MultiLinearPolynomial> r;
        for (auto* m : out) r.push_back(MultiLinearPolynomial::adopt(m, ctx_));
        return r;
    }
    MultiLinearPolynomial get_w_i(size_t layer_idx, const std::vector<MultiLinearPolynomial>& layer_evaluations) const { // :126-138
        if (layer_idx >= layer_evaluations.size()) throw Error(ZK_E_LAYER_OOB);
        return layer_evaluations[layer_evaluations.size() - layer_idx - 1];
    }
    MultiLinearPolynomial get_add_i(size_t layer_idx) const { return gate_poly(layer_idx, 0); } // :140-142
    MultiLinearPolynomial get_mul_i(size_t layer_idx) const { return gate_poly(layer_idx, 1); } // :144-146
    zk_circuit* get() const { return h_; }
    const Ctx& ctx() const { return ctx_; }
  private:
    MultiLinearPolynomial gate_poly(size_t layer_idx, int op) const {
        zk_mle* o = nullptr;
        ctx_->check(zk_circuit_gate_poly(h_, layer_idx, op, &o));
        return MultiLinearPolynomial::adopt(o, ctx_);
    }
    Ctx ctx_;
    zk_circuit* h_ = nullptr;
};

// gkr::gkr_protocol::GKRProof (gkr_protocol.rs:8-26)
struct GKRProof {
    std::vector<Fq> output_poly;                  // the evaluation points of W_0
    std::vector<std::pair<Fq, Fq>> w_polys_evals; // (W_i(r_b), W_i(r_c)), layers 1..
    std::vector<SumCheckProof> sumcheck_proofs;
};

namespace detail {
inline void shape(const Circuit& c, size_t& out_len, std::vector<size_t>& rounds) {
    rounds.assign(c.get_layer_count() ? c.get_layer_count() : 1, 0);
    int rc = zk_circuit_shape(c.get(), &out_len, rounds.data());
    if (rc) throw Error(rc);
    rounds.resize(c.get_layer_count());
}
} // namespace detail

// gkr::prover::GKRProver (gkr/src/prover.rs:18-154)
struct GKRProver {
    static GKRProof generate_proof(Circuit& circuit, Transcript& transcript, const std::vector<Fq>& inputs) { // :24-154
        size_t out_len = 0, total = 0;
        std::vector<size_t> rounds;
        detail::shape(circuit, out_len, rounds);
        for (size_t r : rounds) total += r;
        const size_t L = rounds.size();
        std::vector<Fq> output(out_len), w(2 * L + 2), claims(L), coeffs(3 * total + 1);
        MultiLinearPolynomial in(inputs, circuit.ctx());
        circuit.ctx()->check(zk_gkr_prove(circuit.ctx()->get(), circuit.get(), transcript.get(), in.get(), output[0].l, w[0].l, claims[0].l,
                                          coeffs[0].l, nullptr));
        GKRProof p;
        p.output_poly = output;
        for (size_t i = 0; i + 1 < L; ++i) p.w_polys_evals.emplace_back(w[2 * i], w[2 * i + 1]);
        size_t off = 0;
        for (size_t i = 0; i < L; ++i) {
            std::vector<Fq> c(coeffs.begin() + 3 * off, coeffs.begin() + 3 * (off + rounds[i]));
            p.sumcheck_proofs.push_back(SumCheckProof{claims[i], detail::unflatten(c, rounds[i], 3)});
            off += rounds[i];
        }
        return p;
    }
};

// gkr::verifier::GKRVerifier (gkr/src/verifier.rs:15-119)
struct GKRVerifier {
    static bool verify_proof(const std::vector<Fq>& initial_inputs, Circuit& circuit, Transcript& transcript, const GKRProof& proof) { // :21-119
        size_t out_len = 0;
        std::vector<size_t> rounds;
        detail::shape(circuit, out_len, rounds);
        const size_t L = rounds.size();
        if (proof.sumcheck_proofs.size() != L) return false;
        std::vector<Fq> w(2 * L + 2), claims, coeffs;
        for (size_t i = 0; i < proof.w_polys_evals.size() && i + 1 < L; ++i) { w[2 * i] = proof.w_polys_evals[i].first; w[2 * i + 1] = proof.w_polys_evals[i].second; }
        for (size_t i = 0; i < L; ++i) {
            if (proof.sumcheck_proofs[i].round_polys.size() != rounds[i]) return false;
            claims.push_back(proof.sumcheck_proofs[i].initial_claim_sum);
            auto f = detail::flatten(proof.sumcheck_proofs[i].round_polys, 3);
            coeffs.insert(coeffs.end(), f.begin(), f.begin() + 3 * rounds[i]);
        }
        if (coeffs.empty()) coeffs.resize(1);
        MultiLinearPolynomial in(initial_inputs, circuit.ctx());
        int ok = 0;
        circuit.ctx()->check(zk_gkr_verify(circuit.ctx()->get(), circuit.get(), transcript.get(), in.get(), proof.output_poly[0].l,
                                           proof.output_poly.size(), w[0].l, claims[0].l, coeffs[0].l, &ok));
        return ok != 0;
    }
};

} // namespace zk
