// gkr.cpp -- arithmetic_circuit::Circuit and GKRProver::generate_proof on the device
// (arithmetic_circuit/src/circuit.rs:14-152, gkr/src/prover.rs:24-154, gkr/src/utils.rs:5-65).
//
// The circuit keeps its wiring as gate lists plus two CSR indices per layer (gates grouped by left
// and by right input), built once on the host.  A layer's sumcheck over the 2 k_in variables (b, c)
// of  f(b,c) = mul'(b,c) W(b) W(c) + add'(b,c) (W(b) + W(c))  runs as two degree-2 sumchecks over
// 2^k_in-entry tables on the SAME transcript (gkr_kernels.hip explains why the round polynomials are
// the field elements the reference's dense computation produces):
//   rounds over b:  Sum[ Prod[A, W], Prod[H2, 1] ]              A, H2 from k_gkr_phase1
//   rounds over c:  Sum[ Prod[W(u) M, W], Prod[Ad, W(u) + W] ]  M, Ad from k_gkr_phase2 (u = the b-challenges)
// Both use the degree-2 prover of zkmi.cpp unchanged.  W(u) and W(v) fall out of the last folds and
// are the next layer's W(r_b), W(r_c) (prover.rs:87-99).
#include <chrono>
#include <future>

#include "gkr_kernels.h"
#include "zkmi_internal.hpp"

namespace {

struct LayerDev {
    size_t n_gates = 0, n_out = 0; // n_out = max(next_pow2(n_gates), 2), circuit.rs:43, :89-94
    int k_out = 0, k_in = 0;       // k_in = log2(next_pow2(max index + 1)), circuit.rs:96-109
    size_t max_index = 0;
    uint32_t *left = nullptr, *right = nullptr, *lptr = nullptr, *lidx = nullptr, *rptr = nullptr, *ridx = nullptr;
    uint8_t* op = nullptr;
    std::vector<uint32_t> h_left, h_right; // host copies (dense gate polys, checks)
    std::vector<uint8_t> h_op;
};

size_t next_pow2(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

template <class T>
int to_device(zk_ctx* c, T** dst, const std::vector<T>& src) {
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(c, hipMalloc((void**)dst, bytes));
    if (!src.empty()) HIPCHK(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return ZK_OK;
}

// counting sort of gate ids by key (< n_keys) -> CSR
void build_csr(const std::vector<uint32_t>& key, size_t n_keys, std::vector<uint32_t>& ptr, std::vector<uint32_t>& idx) {
    ptr.assign(n_keys + 1, 0);
    for (uint32_t k : key) ++ptr[k + 1];
    for (size_t i = 0; i < n_keys; ++i) ptr[i + 1] += ptr[i];
    idx.resize(key.size());
    std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
    for (uint32_t g = 0; g < key.size(); ++g) idx[fill[key[g]]++] = g;
}

} // namespace

struct zk_circuit {
    zk_ctx* ctx;
    int device;
    std::vector<LayerDev> layers; // layers[0] is next to the inputs (circuit.rs:11)
    // the prover's layer evaluations (circuit_evaluations, gkr/src/prover.rs:30), kept between proofs (grow-only): 17 hipMalloc + 17
    // hipFree of 32 MiB per prove at BASELINE config 5 were 3 ms of host time
    void* ev_buf = nullptr;
    size_t ev_cap = 0; // bytes
};

namespace {

// eq(x, .) * scale over k variables into `out` (2^k entries); hi/lo: scratch of >= 2^ceil(k/2) entries each
void build_eq(zk_ctx* c, void* out, int k, const uint64_t* x, const Fq& scale, void* hi, void* lo) {
    const int k_lo = k / 2, k_hi = k - k_lo;
    const Fq one = zkh::one();
    void* const o[2] = {hi, lo};
    const int kk[2] = {k_hi, k_lo};
    const uint64_t* const xx[2] = {x, x + 4 * k_hi};
    const uint64_t* const ss[2] = {scale.l, one.l};
    zkk::launch_eq_small_multi(2, o, kk, xx, ss, c->stream);
    zkk::launch_eq_outer(out, (size_t)1 << k, k_lo, hi, lo, nullptr, nullptr, c->stream);
}
// alpha * eq(xa, .) + beta * eq(xb, .)
void build_eq2(zk_ctx* c, void* out, int k, const uint64_t* xa, const Fq& alpha, const uint64_t* xb, const Fq& beta, void* scratch4[4]) {
    const int k_lo = k / 2, k_hi = k - k_lo;
    const Fq one = zkh::one();
    const int kk[4] = {k_hi, k_lo, k_hi, k_lo};
    const uint64_t* const xx[4] = {xa, xa + 4 * k_hi, xb, xb + 4 * k_hi};
    const uint64_t* const ss[4] = {alpha.l, one.l, beta.l, one.l};
    zkk::launch_eq_small_multi(4, scratch4, kk, xx, ss, c->stream);
    zkk::launch_eq_outer(out, (size_t)1 << k, k_lo, scratch4[0], scratch4[1], scratch4[2], scratch4[3], c->stream);
}

struct DevBuf { // hipFree on scope exit
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

} // namespace

extern "C" {

int zk_circuit_create(zk_ctx* c, const uint64_t* gates, const size_t* layer_len, size_t n_layers, zk_circuit** out) {
    if (!c || !out || (n_layers && (!gates || !layer_len))) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    zk_circuit* ci = new (std::nothrow) zk_circuit{c, c->device, {}, nullptr, 0};
    if (!ci) return ZK_E_ALLOC;
    ci->layers.resize(n_layers);
    size_t off = 0;
    for (size_t i = 0; i < n_layers; ++i) {
        LayerDev& L = ci->layers[i];
        L.n_gates = layer_len[i];
        if (L.n_gates == 0 || L.n_gates > ((size_t)1 << 30)) { zk_circuit_free(ci); return ZK_E_ARG; } // `.max().unwrap()` on no gates panics, circuit.rs:104-107
        L.n_out = std::max<size_t>(next_pow2(L.n_gates), 2);
        L.k_out = (int)ilog2(L.n_out);
        L.h_left.resize(L.n_gates);
        L.h_right.resize(L.n_gates);
        L.h_op.resize(L.n_gates);
        size_t mx = 0;
        for (size_t g = 0; g < L.n_gates; ++g, ++off) {
            const uint64_t l = gates[3 * off], r = gates[3 * off + 1], o = gates[3 * off + 2];
            if (l >= ((uint64_t)1 << 31) || r >= ((uint64_t)1 << 31) || o > 1) { zk_circuit_free(ci); return ZK_E_ARG; }
            L.h_left[g] = (uint32_t)l;
            L.h_right[g] = (uint32_t)r;
            L.h_op[g] = (uint8_t)o;
            mx = std::max<size_t>(mx, std::max<size_t>(l, r));
        }
        L.max_index = mx;
        L.k_in = (int)ilog2(next_pow2(mx + 1));
        if (L.k_in > 2 * zkk::kMaxEqVars || L.k_out > 2 * zkk::kMaxEqVars) { zk_circuit_free(ci); return ZK_E_ARG; }
        std::vector<uint32_t> lptr, lidx, rptr, ridx;
        build_csr(L.h_left, (size_t)1 << L.k_in, lptr, lidx);
        build_csr(L.h_right, (size_t)1 << L.k_in, rptr, ridx);
        if ((rc = to_device(c, &L.left, L.h_left)) || (rc = to_device(c, &L.right, L.h_right)) || (rc = to_device(c, &L.op, L.h_op)) ||
            (rc = to_device(c, &L.lptr, lptr)) || (rc = to_device(c, &L.lidx, lidx)) || (rc = to_device(c, &L.rptr, rptr)) ||
            (rc = to_device(c, &L.ridx, ridx))) {
            zk_circuit_free(ci);
            return rc;
        }
    }
    *out = ci;
    return ZK_OK;
}

void zk_circuit_free(zk_circuit* ci) {
    if (!ci) return;
    (void)hipSetDevice(ci->device);
    for (auto& L : ci->layers)
        for (void* p : {(void*)L.left, (void*)L.right, (void*)L.op, (void*)L.lptr, (void*)L.lidx, (void*)L.rptr, (void*)L.ridx})
            if (p) (void)hipFree(p);
    if (ci->ev_buf) (void)hipFree(ci->ev_buf);
    delete ci;
}

size_t zk_circuit_layer_count(const zk_circuit* ci) { return ci ? ci->layers.size() : 0; } // get_layer_count, circuit.rs:149-151

int zk_circuit_shape(const zk_circuit* ci, size_t* output_len, size_t* rounds_per_layer) {
    if (!ci || ci->layers.empty()) return ZK_E_ARG;
    const size_t L = ci->layers.size();
    if (output_len) *output_len = ci->layers[L - 1].n_out;
    if (rounds_per_layer)
        for (size_t i = 0; i < L; ++i) rounds_per_layer[i] = 2 * (size_t)ci->layers[L - 1 - i].k_in;
    return ZK_OK;
}

// evaluate_at_input (circuit.rs:37-59): layers_out[0] = the inputs, layers_out[i+1] = outputs of layers[i]
int zk_circuit_evaluate(zk_circuit* ci, const zk_mle* inputs, zk_mle** layers_out) {
    if (!ci || !inputs || !layers_out || inputs->ctx != ci->ctx) return ZK_E_ARG;
    zk_ctx* c = ci->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    const size_t L = ci->layers.size();
    for (size_t i = 0; i <= L; ++i) layers_out[i] = nullptr;
    auto fail = [&](int code) {
        for (size_t i = 0; i <= L; ++i) { zk_mle_free(layers_out[i]); layers_out[i] = nullptr; }
        return code;
    };
    if ((rc = zk_mle_clone(inputs, &layers_out[0]))) return rc;
    for (size_t i = 0; i < L; ++i) {
        const LayerDev& Ly = ci->layers[i];
        if (Ly.max_index >= layers_out[i]->len) return fail(ZK_E_ARG); // running_inputs[gate.left] out of bounds panics, circuit.rs:47-48
        if ((rc = new_mle(c, Ly.n_out, &layers_out[i + 1]))) return fail(rc);
        zkk::launch_circuit_layer(layers_out[i]->d, layers_out[i + 1]->d, Ly.left, Ly.right, Ly.op, Ly.n_gates, Ly.n_out, c->stream);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("circuit evaluate: ") + hipGetErrorString(e);
        return fail(ZK_E_HIP);
    }
    return ZK_OK;
}

// get_add_i / get_mul_i (circuit.rs:82-146), DENSE: 2^(k_out + 2 k_in) entries -- small circuits only
int zk_circuit_gate_poly(zk_circuit* ci, size_t layer_idx, int op, zk_mle** out) {
    if (!ci || !out || (op != 0 && op != 1)) return ZK_E_ARG;
    if (layer_idx >= ci->layers.size()) return ZK_E_LAYER_OOB; // circuit.rs:83-85
    zk_ctx* c = ci->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    const LayerDev& Ly = ci->layers[ci->layers.size() - layer_idx - 1];
    const int bits = Ly.k_out + 2 * Ly.k_in;
    if (bits > 34) return ZK_E_ALLOC; // 2^34 x 32 B = 512 GiB: the dense table does not exist at this size
    zk_mle* m = nullptr;
    if ((rc = new_mle(c, (size_t)1 << bits, &m))) return rc;
    zkk::launch_gate_poly(m->d, m->len, Ly.left, Ly.right, Ly.op, Ly.n_gates, Ly.k_in, op, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("gate poly: ") + hipGetErrorString(e);
        zk_mle_free(m);
        return ZK_E_HIP;
    }
    *out = m;
    return ZK_OK;
}

// ---- elementwise MultiLinearPolynomial helpers the GKR prover uses (evaluation_form.rs) ----------
static int finish_new(zk_ctx* c, zk_mle* m, zk_mle** out, const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string(what) + ": " + hipGetErrorString(e);
        zk_mle_free(m);
        return ZK_E_HIP;
    }
    *out = m;
    return ZK_OK;
}
static int w_op(const zk_mle* wb, const zk_mle* wc, int mul, zk_mle** out) {
    if (!wb || !wc || !out || wb->ctx != wc->ctx) return ZK_E_ARG;
    // operate_w indexes w_b by i / w_b.len and w_c by i % w_c.len over w_b.len * w_c.len entries: only
    // meaningful (and only used) with equal lengths; unequal lengths index out of bounds in the reference
    if (wb->len != wc->len) return ZK_E_LEN_MISMATCH;
    zk_ctx* c = wb->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* m = nullptr;
    if ((rc = new_mle(c, wb->len * wc->len, &m))) return rc;
    zkk::launch_w_op(m->d, wb->d, wc->d, wb->len, mul, c->stream);
    return finish_new(c, m, out, "w_add/w_mul");
}
int zk_mle_w_add(const zk_mle* wb, const zk_mle* wc, zk_mle** out) { return w_op(wb, wc, 0, out); } // w_add :303-308
// w_mul(&self, other) = operate_w(other, self, Mul) (:310-312): `other` supplies the HIGH index bits
int zk_mle_w_mul(const zk_mle* self, const zk_mle* other, zk_mle** out) { return w_op(other, self, 1, out); }
int zk_mle_scalar_mul(const zk_mle* m, const uint64_t s[4], zk_mle** out) { // :151-153
    if (!m || !s || !out) return ZK_E_ARG;
    zk_ctx* c = m->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* o = nullptr;
    if ((rc = new_mle(c, m->len, &o))) return rc;
    zkk::launch_scalar_mul(o->d, m->d, s, m->len, c->stream);
    return finish_new(c, o, out, "scalar_mul");
}
int zk_mle_add(const zk_mle* a, const zk_mle* b, zk_mle** out) { // _add :241-258
    if (!a || !b || !out || a->ctx != b->ctx) return ZK_E_ARG;
    if (a->len != b->len) return ZK_E_LEN_MISMATCH; // :244-246
    zk_ctx* c = a->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* o = nullptr;
    if ((rc = new_mle(c, a->len, &o))) return rc;
    zkk::launch_add(o->d, a->d, b->d, a->len, c->stream);
    return finish_new(c, o, out, "add");
}

} // extern "C"

// ---- GKRProver::generate_proof (gkr/src/prover.rs:24-154) ----------------------------------------
// world > 1: this process is rank `rank` of a sharded prove (last row).  Circuit, inputs and the layer evaluations
// are replicated (32 MiB per layer at 2^20 gates); what is sharded is each layer's sumcheck: rank q builds the phase tables for
// the indices b (resp. c) that are q mod world -- the gates partitioned by their left (resp. right) input, tables sharded on their
// LOW index bits -- and the ranks run the sharded degree-2 prover of sharded.cpp on them (nine sums per exchange, then the table
// gather).  Every rank absorbs the same bytes and ends with the same proof.  Layers with fewer than two entries per rank run
// replicated.
static int gkr_prove_impl(zk_ctx* c, zk_circuit* ci, zk_transcript* tr, const zk_mle* inputs, int world, int rank, int gather_log,
                          uint64_t* output_poly_out, uint64_t* w_evals_out, uint64_t* claims_out, uint64_t* coeffs_out, double* layer_ms_out) {
    if (!c || !ci || !tr || !inputs || !output_poly_out || !claims_out || !coeffs_out || ci->ctx != c || inputs->ctx != c) return ZK_E_ARG;
    const size_t L = ci->layers.size();
    if (L == 0) return ZK_E_LAYER_OOB; // get_w_i(0) of a lone input layer: prover.rs:38-43 would proceed, but no layer to prove
    if (L > 1 && !w_evals_out) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;

    // circuit_evaluations (:30): ev[0] = the inputs (read in place), ev[i + 1] = outputs of layers[i], in the circuit's own buffer
    struct Tab { const void* d; size_t len; };
    std::vector<Tab> ev(L + 1);
    const auto t_start = std::chrono::steady_clock::now();
    {
        size_t total = 0;
        for (size_t i = 0; i < L; ++i) total += ci->layers[i].n_out;
        if (ci->ev_cap < total * kElem) {
            if (ci->ev_buf) { HIPCHK(c, hipFree(ci->ev_buf)); ci->ev_buf = nullptr; ci->ev_cap = 0; }
            HIPCHK(c, hipMalloc(&ci->ev_buf, total * kElem));
            ci->ev_cap = total * kElem;
        }
        ev[0] = Tab{inputs->d, inputs->len};
        uint8_t* at = (uint8_t*)ci->ev_buf;
        for (size_t i = 0; i < L; ++i) { // evaluate_at_input, circuit.rs:37-59
            const LayerDev& Ly = ci->layers[i];
            if (Ly.max_index >= ev[i].len) return ZK_E_ARG; // running_inputs[gate.left] out of bounds panics, circuit.rs:47-48
            zkk::launch_circuit_layer(ev[i].d, at, Ly.left, Ly.right, Ly.op, Ly.n_gates, Ly.n_out, c->stream);
            ev[i + 1] = Tab{at, Ly.n_out};
            at += Ly.n_out * kElem;
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const auto t_eval = std::chrono::steady_clock::now();

    // every layer's relation needs 2^k_in == len(W_{i+1}) (ProductPolynomial::new panics otherwise, product_polynomial.rs:15-19)
    size_t max_in = 2, max_out = 2;
    for (size_t i = 0; i < L; ++i) {
        const LayerDev& Ly = ci->layers[i];
        if (((size_t)1 << Ly.k_in) != ev[i].len || Ly.k_in < 1) return ZK_E_LEN_MISMATCH;
        max_in = std::max(max_in, ev[i].len);
        max_out = std::max(max_out, Ly.n_out);
    }
    const size_t half_tab = (size_t)1 << zkk::kMaxEqVars;
    const size_t eq_half = std::min(half_tab, std::max(max_in, max_out));
    // scratch: wt[max_out] | Eu[max_in] | A H2 ones T0 T2 Wl [max_in each] | 4 eq half tables
    const size_t n_elems = max_out + 7 * max_in + 4 * eq_half;
    if (c->gkr_scratch_cap < n_elems * kElem) { // kept in the context between proofs (a few hundred MiB at 2^20 gates per layer)
        if (c->gkr_scratch) { HIPCHK(c, hipFree(c->gkr_scratch)); c->gkr_scratch = nullptr; c->gkr_scratch_cap = 0; }
        HIPCHK(c, hipMalloc(&c->gkr_scratch, n_elems * kElem));
        c->gkr_scratch_cap = n_elems * kElem;
    }
    uint8_t* base = (uint8_t*)c->gkr_scratch;
    auto take = [&](size_t n) { void* p = base; base += n * kElem; return p; };
    void* wt = take(max_out);
    void* Eu = take(max_in);
    void *A = take(max_in), *H2 = take(max_in), *ones = take(max_in), *T0 = take(max_in), *T2 = take(max_in);
    void* Wl = take(max_in); // a rank's shard of the next layer's table (sharded proves)
    void* eqs[4] = {take(eq_half), take(eq_half), take(eq_half), take(eq_half)};
    zkk::launch_fill_one(ones, max_in, c->stream);

    const zk_mle w0_view{c, const_cast<void*>(ev[L].d), ev[L].len, false, c->device}; // get_w_i(0), :43 (not owned)
    const zk_mle* W0 = &w0_view;
    const unsigned k0 = ilog2(W0->len);
    // proof.output_poly: the copy into the caller's (pageable) buffer runs on a helper thread and its own stream while
    // this thread feeds the output layer through the Keccak sponge (:46) -- both read W0, neither writes it
    HIPCHK(c, hipStreamSynchronize(c->stream)); // W0 is complete before the helper stream reads it
    const int dev = c->device;
    const void* w0_dev = W0->d;
    const size_t w0_bytes = W0->len * kElem;
    std::future<hipError_t> out_copy = std::async(std::launch::async, [dev, w0_dev, w0_bytes, output_poly_out]() -> hipError_t {
        hipError_t e = hipSetDevice(dev);
        hipStream_t s2 = nullptr;
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMemcpyAsync(output_poly_out, w0_dev, w0_bytes, hipMemcpyDeviceToHost, s2);
        if (e == hipSuccess) e = hipStreamSynchronize(s2);
        if (s2) (void)hipStreamDestroy(s2);
        return e;
    });
    rc = stream_le32(c, W0->d, W0->len, [&](const uint8_t* p, size_t nbytes, size_t) { tr->append(p, nbytes); }); // :46
    {
        const hipError_t e = out_copy.get(); // joined before any return: the lambda holds no reference to this frame
        if (rc) return rc;
        if (e != hipSuccess) {
            c->err = std::string("output layer download: ") + hipGetErrorString(e);
            return ZK_E_HIP;
        }
    }
    if (layer_ms_out) {
        const auto t_abs = std::chrono::steady_clock::now();
        layer_ms_out[L] = std::chrono::duration<double, std::milli>(t_eval - t_start).count();
        layer_ms_out[L + 1] = std::chrono::duration<double, std::milli>(t_abs - t_eval).count();
    }
    std::vector<uint64_t> rnd(4 * std::max<size_t>(k0, 1)); // random_values, :49-53
    for (unsigned i = 0; i < k0; ++i) zkh::store(&rnd[4 * i], tr->sample());

    Fq wb = zkh::zero(), wc = zkh::zero();
    size_t coeff_off = 0;
    struct Events { // destroyed on every exit path
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evs;
    if (layer_ms_out) { HIPCHK(c, hipEventCreate(&evs.a)); HIPCHK(c, hipEventCreate(&evs.b)); }
    hipEvent_t& ev_a = evs.a;
    hipEvent_t& ev_b = evs.b;
    for (size_t layer_idx = 0; layer_idx < L; ++layer_idx) {
        const LayerDev& Ly = ci->layers[L - 1 - layer_idx];
        const Tab* Wn = &ev[L - 1 - layer_idx]; // next_w_i, :125
        const int k_in = Ly.k_in, k_out = Ly.k_out;
        const size_t n_in = (size_t)1 << k_in;
        if (layer_ms_out) HIPCHK(c, hipEventRecord(ev_a, c->stream));
        Fq claim;
        if (layer_idx == 0) {
            // claim = W_0(rho) (:70-75); mul'/add' = predicates folded at rho over the k_out "a" variables (utils.rs:45-65)
            std::vector<uint8_t> has(std::max<unsigned>(k0, 1), 1);
            zk_mle* e = nullptr;
            if ((rc = zk_mle_evaluate(W0, has.data(), rnd.data(), k0, &e))) return rc;
            uint64_t v[4];
            rc = zk_mle_download_mont(e, v);
            zk_mle_free(e);
            if (rc) return rc;
            claim = zkh::load(v);
            build_eq(c, wt, k_out, rnd.data(), zkh::one(), eqs[0], eqs[1]);
        } else {
            // r_b | r_c = halves of the previous challenges (:82-85); W_i(r_b), W_i(r_c) are the last folds of the
            // previous layer's two phases (:87-99)
            tr->append_fq(wb); // :102-105
            tr->append_fq(wc);
            const Fq alpha = tr->sample(), beta = tr->sample(); // :107-108
            zkh::store(w_evals_out + 8 * (layer_idx - 1), wb);  // :115
            zkh::store(w_evals_out + 8 * (layer_idx - 1) + 4, wc);
            claim = zkh::add(zkh::mul(alpha, wb), zkh::mul(beta, wc)); // utils.rs:5-14
            build_eq2(c, wt, k_out, rnd.data(), alpha, rnd.data() + 4 * k_out, beta, eqs); // utils.rs:16-43
        }
        zkh::store(claims_out + 4 * layer_idx, claim); // SumCheckProof.initial_claim_sum = the claim passed in, sumcheck/src/prover.rs:185

        std::vector<uint64_t> ch(4 * 2 * (size_t)k_in);
        uint64_t fin[16];
        const bool shard_layer = world > 1 && n_in >= 2 * (size_t)world;
        const size_t n_loc = shard_layer ? n_in / (size_t)world : n_in, step = shard_layer ? (size_t)world : 1, first = shard_layer ? (size_t)rank : 0;
        const void* Wt = Wn->d; // the W factor table of both phases: the whole table, or this rank's shard of it
        if (shard_layer) {
            zkk::launch_stride_gather(Wl, Wn->d, n_loc, step, first, c->stream);
            Wt = Wl;
        }
        // both phases have the shape t[0] t[1] + t[2] * 1 (t[3] = the table of ones): the single-GPU prover is told so
        auto prove = [&](const void* const t[4], uint64_t* co, uint64_t* chal) -> int {
            return shard_layer ? prove_sum2x2_sharded_dev(c, t, n_loc, tr, gather_log, co, chal, fin)
                               : prove_sum2x2_dev(c, t, n_in, tr, co, chal, fin, /*ones3=*/true);
        };
        // rounds over b
        zkk::launch_gkr_phase1(A, H2, Ly.lptr, Ly.lidx, Ly.right, Ly.op, wt, Wn->d, n_loc, step, first, c->stream);
        HIPCHK(c, hipGetLastError());
        {
            const void* t[4] = {A, Wt, H2, ones};
            if ((rc = prove(t, coeffs_out + 12 * coeff_off, ch.data()))) return rc;
        }
        const Fq wu = zkh::load(fin + 4); // W_{i+1}(u)
        // rounds over c
        build_eq(c, Eu, k_in, ch.data(), zkh::one(), eqs[0], eqs[1]);
        zkk::launch_gkr_phase2(T0, T2, Ly.rptr, Ly.ridx, Ly.left, Ly.op, wt, Eu, wu.l, n_loc, step, first, c->stream);
        HIPCHK(c, hipGetLastError());
        {
            const void* t[4] = {T0, Wt, T2, ones}; // P W + Q * 1 with P = wu m + ad, Q = wu ad (gkr_kernels.hip)
            if ((rc = prove(t, coeffs_out + 12 * (coeff_off + k_in), ch.data() + 4 * k_in))) return rc;
        }
        wb = wu;
        wc = zkh::load(fin + 4); // W_{i+1}(v)
        coeff_off += 2 * (size_t)k_in;
        rnd = ch; // random_values = random_points, :143
        if (layer_ms_out) {
            HIPCHK(c, hipEventRecord(ev_b, c->stream));
            HIPCHK(c, hipEventSynchronize(ev_b));
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, ev_a, ev_b));
            layer_ms_out[layer_idx] = ms;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

extern "C" {

int zk_gkr_prove(zk_ctx* c, zk_circuit* ci, zk_transcript* tr, const zk_mle* inputs, uint64_t* output_poly_out, uint64_t* w_evals_out,
                 uint64_t* claims_out, uint64_t* coeffs_out, double* layer_ms_out) {
    return gkr_prove_impl(c, ci, tr, inputs, 1, 0, 0, output_poly_out, w_evals_out, claims_out, coeffs_out, layer_ms_out);
}

int zk_gkr_prove_sharded(zk_ctx* c, zk_circuit* ci, zk_transcript* tr, const zk_mle* inputs, int gather_log, uint64_t* output_poly_out,
                         uint64_t* w_evals_out, uint64_t* claims_out, uint64_t* coeffs_out, double* layer_ms_out) {
    if (!c) return ZK_E_ARG;
    if (!c->comm) {
        c->err = "no communicator on this context (zk_comm_init_rccl / zk_comm_init_host)";
        return ZK_E_COMM;
    }
    if (gather_log < 0) return ZK_E_ARG; // W(u), W(v) come out of the replicated finish of each phase
    return gkr_prove_impl(c, ci, tr, inputs, zk_comm_world(c), zk_comm_rank(c), gather_log, output_poly_out, w_evals_out, claims_out, coeffs_out,
                          layer_ms_out);
}

// ---- GKRVerifier::verify_proof (gkr/src/verifier.rs:21-119) --------------------------------------
// Same transcript stream as the prover.  The reference rebuilds the dense predicates and evaluates them at
// the sumcheck's challenges (:39-73); here add'(u,v), mul'(u,v) come from the gate list
// (k_gkr_pred_eval).  Like the reference, the layer-0 claim is taken from the proof, not recomputed from
// output_poly (:63-64).  A sumcheck round that fails makes the reference panic in evaluate() on the short
// challenge vector (evaluation_form.rs:197-199) unless it is the last round; both are reported as ok = 0.
int zk_gkr_verify(zk_ctx* c, zk_circuit* ci, zk_transcript* tr, const zk_mle* inputs, const uint64_t* output_poly, size_t output_len,
                  const uint64_t* w_evals, const uint64_t* claims, const uint64_t* coeffs, int* ok_out) {
    if (!c || !ci || !tr || !inputs || !output_poly || !claims || !coeffs || !ok_out || ci->ctx != c || inputs->ctx != c) return ZK_E_ARG;
    *ok_out = 0;
    const size_t L = ci->layers.size();
    if (L == 0) return ZK_E_LAYER_OOB;
    if (L > 1 && !w_evals) return ZK_E_ARG;
    if (!is_pow2(output_len)) return ZK_E_NOT_POW2;
    int rc = set_device(c);
    if (rc) return rc;

    size_t max_in = 2, max_out = std::max<size_t>(2, output_len);
    for (const LayerDev& Ly : ci->layers) {
        max_in = std::max(max_in, (size_t)1 << Ly.k_in);
        max_out = std::max(max_out, Ly.n_out);
    }
    const size_t eq_half = std::min((size_t)1 << zkk::kMaxEqVars, std::max(max_in, max_out));
    DevBuf buf;
    HIPCHK(c, hipMalloc(&buf.p, (max_out + 2 * max_in + 4 * eq_half) * kElem));
    uint8_t* base = (uint8_t*)buf.p;
    auto take = [&](size_t n) { void* p = base; base += n * kElem; return p; };
    void *wt = take(max_out), *Eu = take(max_in), *Ev = take(max_in);
    void* eqs[4] = {take(eq_half), take(eq_half), take(eq_half), take(eq_half)};

    for (size_t i = 0; i < output_len; ++i) tr->append_fq(zkh::load(output_poly + 4 * i)); // :30
    const unsigned k0 = ilog2(output_len);
    std::vector<uint64_t> rnd(4 * std::max<size_t>(k0, 1));
    for (unsigned i = 0; i < k0; ++i) zkh::store(&rnd[4 * i], tr->sample()); // :32-36
    size_t n_rnd = k0, coeff_off = 0;
    Fq prev_wb = zkh::zero(), prev_wc = zkh::zero();

    for (size_t layer_idx = 0; layer_idx < L; ++layer_idx) {
        const LayerDev& Ly = ci->layers[L - 1 - layer_idx];
        const int k_in = Ly.k_in, k_out = Ly.k_out;
        const size_t n_in = (size_t)1 << k_in, rounds = 2 * (size_t)k_in;
        if (layer_idx == 0) {
            if (n_rnd != (size_t)k_out) return ZK_OK; // a proof whose output_poly does not fit the circuit: rejected
            build_eq(c, wt, k_out, rnd.data(), zkh::one(), eqs[0], eqs[1]); // :43
            if (c->gkr_strict_claims) { // not in the reference: the output layer's claim must be output_poly(rho)
                zk_mle* w0 = nullptr;
                if ((rc = zk_mle_upload_mont(c, output_poly, output_len, &w0))) return rc;
                std::vector<uint8_t> has(std::max<size_t>(k0, 1), 1);
                zk_mle* e = nullptr;
                rc = zk_mle_evaluate(w0, has.data(), rnd.data(), k0, &e);
                zk_mle_free(w0);
                if (rc) return rc;
                uint64_t v[4];
                rc = zk_mle_download_mont(e, v);
                zk_mle_free(e);
                if (rc) return rc;
                if (zkh::load(v) != zkh::load(claims)) return ZK_OK;
            }
        } else {
            if (n_rnd != 2 * (size_t)k_out) return ZK_E_LEN_MISMATCH; // the circuit's layers do not chain (the prover refuses it too)
            const Fq alpha = tr->sample(), beta = tr->sample(); // :45-46
            build_eq2(c, wt, k_out, rnd.data(), alpha, rnd.data() + 4 * k_out, beta, eqs); // :49-56
            // not in the reference: this layer's claim must be alpha W(r_b) + beta W(r_c) of the values the last layer ended on
            if (c->gkr_strict_claims &&
                zkh::add(zkh::mul(alpha, prev_wb), zkh::mul(beta, prev_wc)) != zkh::load(claims + 4 * layer_idx))
                return ZK_OK;
        }
        int ok = 0;
        uint64_t final_claim[4];
        std::vector<uint64_t> ch(4 * std::max<size_t>(rounds, 1));
        if ((rc = zk_sumcheck_partial_verify(claims + 4 * layer_idx, coeffs + 12 * coeff_off, rounds, 3, tr, &ok, final_claim, ch.data()))) return rc; // :63-64
        if (!ok) return ZK_OK;
        // add'(u,v), mul'(u,v) :67-73
        build_eq(c, Eu, k_in, ch.data(), zkh::one(), eqs[0], eqs[1]);
        build_eq(c, Ev, k_in, ch.data() + 4 * k_in, zkh::one(), eqs[2], eqs[3]);
        const int g = zkk::launch_gkr_pred_eval(Ly.left, Ly.right, Ly.op, wt, Eu, Ev, Ly.n_gates, c->partials, c->stream);
        HIPCHK(c, hipGetLastError());
        Fq pe[2];
        if ((rc = fetch_sums(c, g, 2, pe))) return rc;
        Fq wb, wc;
        if (layer_idx + 1 == L) { // the input layer: W from the inputs themselves, :76-94
            if (inputs->len != n_in) return ZK_E_POINTS_LEN; // evaluate() panics on the length mismatch
            std::vector<uint8_t> has(k_in, 1);
            for (int side = 0; side < 2; ++side) {
                zk_mle* e = nullptr;
                if ((rc = zk_mle_evaluate(inputs, has.data(), ch.data() + 4 * (size_t)side * k_in, (size_t)k_in, &e))) return rc;
                uint64_t v[4];
                rc = zk_mle_download_mont(e, v);
                zk_mle_free(e);
                if (rc) return rc;
                (side ? wc : wb) = zkh::load(v);
            }
        } else { // :96-98
            wb = zkh::load(w_evals + 8 * layer_idx);
            wc = zkh::load(w_evals + 8 * layer_idx + 4);
        }
        tr->append_fq(wb); // :101-104
        tr->append_fq(wc);
        const Fq fbc = zkh::add(zkh::mul(pe[0], zkh::add(wb, wc)), zkh::mul(pe[1], zkh::mul(wb, wc))); // :106-107
        if (fbc != zkh::load(final_claim)) return ZK_OK; // :111-113
        prev_wb = wb;
        prev_wc = wc;
        rnd = ch;
        n_rnd = rounds;
        coeff_off += rounds;
    }
    *ok_out = 1;
    return ZK_OK;
}

} // extern "C"
