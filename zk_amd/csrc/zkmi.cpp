// zkmi.cpp -- the C ABI of include/zkmi.h: contexts, device-resident tables, the host
// transcript and the sumcheck prover round loops that drive the gfx950 kernels.
//
// Round loop: reduce_k -> host: coefficients, Keccak absorb, challenge r_k ->
// fold_k fused with reduce_{k+1}.  The table never leaves HBM; per round (d+1) x 32 B come
// back to the host and 32 B (the challenge) go out as a kernel argument.
#include "zkmi_internal.hpp"
#include "gkr_kernels.h" // launch_fill_one

extern "C" {

const char* zk_strerror(int s) {
    switch (s) {
        case ZK_OK: return "ok";
        case ZK_E_NOT_POW2: return "Invalid Multilinear Polynomial: evaluation points length is not a power of two";
        case ZK_E_POINTS_LEN: return "points length does not match number of variables";
        case ZK_E_LEN_MISMATCH: return "All polynomials must have the same length";
        case ZK_E_LAYER_OOB: return "layer index out of bounds";
        case ZK_E_DEGREE: return "sum polynomial must be 2 product terms of 2 factors";
        case ZK_E_ALLOC: return "allocation failed";
        case ZK_E_HIP: return "HIP runtime error";
        case ZK_E_ARG: return "bad argument";
        case ZK_E_NO_DEVICE: return "no gfx950 device available (this library has no CPU path)";
        case ZK_E_COMM: return "collective communication error";
        case ZK_E_TOO_LARGE: return "input larger than the maximum the static scratch buffer is sized for";
        default: return "unknown status";
    }
}
const char* zk_last_error(const zk_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }
const char* zk_version(void) { return "zkmi 0.1 (gfx950)"; }

// ---- context --------------------------------------------------------------------------------
static int ctx_create(int device_id, bool own_stream, hipStream_t stream, zk_ctx** out) {
    if (!out) return ZK_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return ZK_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return ZK_E_NO_DEVICE;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return ZK_E_NO_DEVICE; // kernels are gfx950-only
    zk_ctx* c = new (std::nothrow) zk_ctx();
    if (!c) return ZK_E_ALLOC;
    c->device = device_id;
    zkh::store(c->two24, zkh::from_u64((uint64_t)1 << 24));
    c->own_stream = own_stream;
    c->stream = stream;
    int rc = ZK_OK;
    auto fail = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return false;
        fprintf(stderr, "zkmi: %s: %s\n", what, hipGetErrorString(e));
        rc = ZK_E_HIP;
        return true;
    };
    void* static_scratch = nullptr; // set-up of the no-allocation entry point's scratch for this device, with it current
    if (fail(hipSetDevice(device_id), "hipSetDevice") || fail(zkk::static_scratch_ptr(&static_scratch), "hipGetSymbolAddress(scratch)") ||
        (own_stream && fail(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) ||
        fail(hipMalloc(&c->partials, std::max((size_t)zkk::kMaxBlocks * 3 + 16, zkk::kSum9PartialElems) * kElem), "hipMalloc partials") ||
        fail(hipMalloc(&c->sums_dev, kSums * kElem), "hipMalloc sums") ||
        fail(hipMalloc(&c->wide_w, zkk::kWideWeightWords * sizeof(uint32_t)), "hipMalloc wide weights") ||
        fail(hipHostMalloc((void**)&c->sums_host, kHostStage * kElem, hipHostMallocMapped), "hipHostMalloc sums") ||
        fail(hipHostGetDevicePointer(&c->sums_host_dev, c->sums_host, 0), "hipHostGetDevicePointer")) {
        zk_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return ZK_OK;
}
int zk_ctx_create(int device_id, zk_ctx** out) { return ctx_create(device_id, true, nullptr, out); }
int zk_ctx_create_on_stream(int device_id, void* hip_stream, zk_ctx** out) {
    return ctx_create(device_id, false, (hipStream_t)hip_stream, out);
}

void zk_ctx_destroy(zk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->prof_pending) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto& p : c->prof_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (c->partials) (void)hipFree(c->partials);
    if (c->cross_raw) (void)hipFree(c->cross_raw);
    if (c->sums_dev) (void)hipFree(c->sums_dev);
    if (c->wide_w) (void)hipFree(c->wide_w);
    if (c->sums_host) (void)hipHostFree(c->sums_host);
    if (c->le_host) (void)hipHostFree(c->le_host);
    for (hipEvent_t e : c->le_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->work[0]) (void)hipFree(c->work[0]);
    if (c->work[1]) (void)hipFree(c->work[1]);
    if (c->stage) (void)hipFree(c->stage);
    if (c->gkr_scratch) (void)hipFree(c->gkr_scratch);
    zk_comm_destroy(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
void* zk_ctx_stream(zk_ctx* c) { return c ? (void*)c->stream : nullptr; }
int zk_ctx_synchronize(zk_ctx* c) {
    if (!c) return ZK_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

// ---- host field helpers (what ark-ff gives the Rust caller; used by the C++/Python host layers) ----
void zk_fq_from_u64(uint64_t v, uint64_t out[4]) { zkh::store(out, zkh::from_u64(v)); }          // T::from(u64)
void zk_fq_to_le32(const uint64_t a[4], uint8_t out[32]) { zkh::to_le32(zkh::load(a), out); }     // into_bigint().to_bytes_le()
void zk_fq_from_le32_mod_order(const uint8_t in[32], uint64_t out[4]) { zkh::store(out, zkh::from_le32_mod_order(in)); }
void zk_fq_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { zkh::store(out, zkh::add(zkh::load(a), zkh::load(b))); }
void zk_fq_sub(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { zkh::store(out, zkh::sub(zkh::load(a), zkh::load(b))); }
void zk_fq_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { zkh::store(out, zkh::mul(zkh::load(a), zkh::load(b))); }
// UnivariatePolynomial::evaluate (dense_coefficient_form.rs:20-34): sum_i coeffs[i] x^i
void zk_unipoly_evaluate(const uint64_t* coeffs, size_t n, const uint64_t x[4], uint64_t out[4]) {
    std::vector<Fq> c(n);
    for (size_t i = 0; i < n; ++i) c[i] = zkh::load(coeffs + 4 * i);
    zkh::store(out, zkh::poly_eval(c.data(), n, zkh::load(x)));
}

// ---- transcript -----------------------------------------------------------------------------
zk_transcript* zk_transcript_new(void) { return new (std::nothrow) zk_transcript(); }
zk_transcript* zk_transcript_clone(const zk_transcript* t) { return t ? new (std::nothrow) zk_transcript(*t) : nullptr; }
void zk_transcript_free(zk_transcript* t) { delete t; }
int zk_transcript_append(zk_transcript* t, const uint8_t* data, size_t len) {
    if (!t || (!data && len)) return ZK_E_ARG;
    if (len) t->append(data, len);
    return ZK_OK;
}
int zk_transcript_sample_challenge(zk_transcript* t, uint64_t out[4]) {
    if (!t || !out) return ZK_E_ARG;
    zkh::store(out, t->sample());
    return ZK_OK;
}
int zk_transcript_sample_n_challenges(zk_transcript* t, size_t n, uint64_t* out) {
    if (!t || (!out && n)) return ZK_E_ARG;
    for (size_t i = 0; i < n; ++i) zkh::store(out + 4 * i, t->sample());
    return ZK_OK;
}
int zk_transcript_selftest(size_t nblocks) { return zkh::Keccak256::selftest(nblocks); }
int zk_transcript_peek_digest(const zk_transcript* t, uint8_t out[32]) {
    if (!t || !out) return ZK_E_ARG;
    t->hasher.peek(out);
    return ZK_OK;
}

// ---- MLE --------------------------------------------------------------------------------------
// number of entries that are not fully reduced (>= p)
int zk_mle_count_noncanonical(const zk_mle* m, uint64_t* count_out) {
    if (!m || !count_out) return ZK_E_ARG;
    zk_ctx* c = m->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->sums_dev, 0, sizeof(uint64_t), c->stream)); // the counter lives in device memory (atomics)
    zkk::launch_count_noncanonical(m->d, m->len, c->sums_dev, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->sums_host, c->sums_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *count_out = c->sums_host[0];
    return ZK_OK;
}

int zk_mle_upload_mont(zk_ctx* c, const uint64_t* limbs, size_t len, zk_mle** out) {
    if (!c || !limbs || !out) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* m = nullptr;
    if ((rc = new_mle(c, len, &m))) return rc;
    hipError_t e = hipMemcpyAsync(m->d, limbs, len * kElem, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("upload: ") + hipGetErrorString(e);
        zk_mle_free(m);
        return ZK_E_HIP;
    }
    if (c->validate_uploads) { // limbs >= p are outside the contract (zkmi.h): refuse them here instead of proving garbage
        uint64_t bad = 0;
        rc = zk_mle_count_noncanonical(m, &bad);
        if (!rc && bad) {
            c->err = "upload: " + std::to_string(bad) + " entries are not fully reduced Montgomery limbs (>= p)";
            rc = ZK_E_ARG;
        }
        if (rc) {
            zk_mle_free(m);
            return rc;
        }
    }
    *out = m;
    return ZK_OK;
}

int zk_mle_upload_le32(zk_ctx* c, const uint8_t* bytes, size_t len, zk_mle** out) {
    if (!c || !bytes || !out) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* m = nullptr;
    if ((rc = new_mle(c, len, &m))) return rc;
    if ((rc = ensure_stage(c))) { zk_mle_free(m); return rc; }
    for (size_t off = 0; off < len; off += kChunkElems) {
        size_t n = std::min(kChunkElems, len - off);
        hipError_t e = hipMemcpyAsync(c->stage, bytes + off * kElem, n * kElem, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            zkk::launch_from_le32(c->stage, (uint8_t*)m->d + off * kElem, n, c->stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            c->err = std::string("upload_le32: ") + hipGetErrorString(e);
            zk_mle_free(m);
            return ZK_E_HIP;
        }
    }
    *out = m;
    return ZK_OK;
}

int zk_mle_wrap_device(zk_ctx* c, void* dptr, size_t len, zk_mle** out) {
    if (!c || !dptr || !out) return ZK_E_ARG;
    if (!is_pow2(len)) return ZK_E_NOT_POW2;
    if ((uintptr_t)dptr % 16) return ZK_E_ARG; // 16-byte vector loads
    zk_mle* m = new (std::nothrow) zk_mle{c, dptr, len, false, c->device};
    if (!m) return ZK_E_ALLOC;
    *out = m;
    return ZK_OK;
}

int zk_mle_generate_strided(zk_ctx* c, uint64_t seed, uint64_t first, uint64_t step, size_t len, zk_mle** out) {
    if (!c || !out) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* m = nullptr;
    if ((rc = new_mle(c, len, &m))) return rc;
    zkk::launch_generate(m->d, seed, first, step, len, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("generate: ") + hipGetErrorString(e);
        zk_mle_free(m);
        return ZK_E_HIP;
    }
    *out = m;
    return ZK_OK;
}
int zk_mle_generate(zk_ctx* c, uint64_t seed, uint64_t first, size_t len, zk_mle** out) {
    return zk_mle_generate_strided(c, seed, first, 1, len, out);
}

// shard `rank` of `world` (a power of two) of a host table: local entry j = global entry j*world + rank
int zk_mle_upload_mont_shard(zk_ctx* c, const uint64_t* limbs, size_t global_len, int rank, int world, zk_mle** out) {
    if (!c || !limbs || !out || world <= 0 || rank < 0 || rank >= world || !is_pow2((size_t)world)) return ZK_E_ARG;
    if (!is_pow2(global_len)) return ZK_E_NOT_POW2;
    if (global_len < (size_t)world) return ZK_E_ARG;
    const size_t len = global_len / (size_t)world;
    std::vector<uint64_t> local;
    try { local.resize(4 * len); } catch (...) { return ZK_E_ALLOC; }
    for (size_t j = 0; j < len; ++j) std::memcpy(&local[4 * j], limbs + 4 * (j * (size_t)world + (size_t)rank), kElem);
    return zk_mle_upload_mont(c, local.data(), len, out);
}

int zk_mle_clone(const zk_mle* src, zk_mle** out) {
    if (!src || !out) return ZK_E_ARG;
    zk_ctx* c = src->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* m = nullptr;
    if ((rc = new_mle(c, src->len, &m))) return rc;
    hipError_t e = hipMemcpyAsync(m->d, src->d, src->len * kElem, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("clone: ") + hipGetErrorString(e);
        zk_mle_free(m);
        return ZK_E_HIP;
    }
    *out = m;
    return ZK_OK;
}

void zk_mle_free(zk_mle* m) {
    if (!m) return;
    if (m->owned && m->d) {
        (void)hipSetDevice(m->device);
        (void)hipFree(m->d);
    }
    delete m;
}
size_t zk_mle_len(const zk_mle* m) { return m ? m->len : 0; }
void* zk_mle_device_ptr(const zk_mle* m) { return m ? m->d : nullptr; }

int zk_mle_download_mont(const zk_mle* m, uint64_t* out) {
    if (!m || !out) return ZK_E_ARG;
    zk_ctx* c = m->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, m->d, m->len * kElem, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

int zk_mle_to_le32(const zk_mle* m, uint8_t* out) {
    if (!m || !out) return ZK_E_ARG;
    zk_ctx* c = m->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    return stream_le32(c, m->d, m->len, [&](const uint8_t* p, size_t nbytes, size_t off) { std::memcpy(out + off * kElem, p, nbytes); });
}

int zk_mle_partially_evaluate(const zk_mle* m, size_t var, const uint64_t r[4], zk_mle** out) {
    if (!m || !r || !out) return ZK_E_ARG;
    zk_ctx* c = m->ctx;
    unsigned nv = ilog2(m->len);
    if (var >= nv) return ZK_E_POINTS_LEN; // Rust: `number_of_variables() - 1 - idx` underflows and panics
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* o = nullptr;
    if ((rc = new_mle(c, m->len / 2, &o))) return rc;
    {
        ProfScope ps(c, ZK_K_FOLD, (uint64_t)m->len * 48);
        zkk::launch_fold_var(m->d, o->d, m->len / 2, (int)(nv - 1 - var), r, c->stream);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("partially_evaluate: ") + hipGetErrorString(e);
        zk_mle_free(o);
        return ZK_E_HIP;
    }
    prof_collect(c);
    *out = o;
    return ZK_OK;
}

// compute_quotient_remainder (evaluation_form.rs:280-301): one pass, two output streams
int zk_mle_quotient_remainder(const zk_mle* m, size_t var, const uint64_t divisor[4], zk_mle** quotient_out, zk_mle** remainder_out) {
    if (!m || !divisor || !quotient_out || !remainder_out) return ZK_E_ARG;
    zk_ctx* c = m->ctx;
    const unsigned nv = ilog2(m->len);
    if (var >= nv) return ZK_E_POINTS_LEN; // get_y1_y2_indexes: `number_of_variables() - 1 - idx` underflows and panics
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle *q = nullptr, *r = nullptr;
    if ((rc = new_mle(c, m->len / 2, &q))) return rc;
    if ((rc = new_mle(c, m->len / 2, &r))) { zk_mle_free(q); return rc; }
    {
        ProfScope ps(c, ZK_K_FOLD, (uint64_t)m->len * 64); // 32 B per entry read, 2 x 16 B per entry written
        zkk::launch_quot_rem_var(m->d, q->d, r->d, m->len / 2, (int)(nv - 1 - var), divisor, c->stream);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("quotient_remainder: ") + hipGetErrorString(e);
        zk_mle_free(q);
        zk_mle_free(r);
        return ZK_E_HIP;
    }
    prof_collect(c);
    *quotient_out = q;
    *remainder_out = r;
    return ZK_OK;
}

// minus (evaluation_form.rs:260-272)
int zk_mle_minus(const zk_mle* m, const uint64_t v[4], zk_mle** out) {
    if (!m || !v || !out) return ZK_E_ARG;
    zk_ctx* c = m->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    zk_mle* o = nullptr;
    if ((rc = new_mle(c, m->len, &o))) return rc;
    zkk::launch_minus(m->d, o->d, m->len, v, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("minus: ") + hipGetErrorString(e);
        zk_mle_free(o);
        return ZK_E_HIP;
    }
    *out = o;
    return ZK_OK;
}

int zk_mle_evaluate(const zk_mle* m, const uint8_t* has, const uint64_t* vals, size_t npts, zk_mle** out) {
    if (!m || !out || (npts && (!has || !vals))) return ZK_E_ARG;
    if (npts != ilog2(m->len)) return ZK_E_POINTS_LEN; // evaluation_form.rs:197-199
    zk_ctx* c = m->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    // A leading run of Some(..) binds the TOP variables one after the other (:206-217 with idx - done = 0
    // each time): the batched fold pass of the prover does up to kMaxFoldBatch of them per sweep.  This is
    // the verifier's oracle check (verifier.rs:15-26) and GKR's W(r) evaluations (gkr/src/prover.rs:70-75).
    size_t pre = 0;
    while (pre < npts && has[pre]) ++pre;
    zk_mle* acc = nullptr;
    size_t first = 0;
    if (pre >= 2) {
        zk_shard st{c, m->d, (unsigned)npts, 0};
        while (first < pre) {
            const int mm = (int)std::min<size_t>(next_batch(c, st.ell), pre - first);
            if ((rc = zk_shard_fold(&st, mm, vals + 4 * first, 0, nullptr))) return rc;
            first += (size_t)mm;
        }
        if ((rc = new_mle(c, (size_t)1 << st.ell, &acc))) return rc;
        hipError_t e = hipMemcpyAsync(acc->d, st.cur, acc->len * kElem, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            c->err = std::string("evaluate: ") + hipGetErrorString(e);
            zk_mle_free(acc);
            return ZK_E_HIP;
        }
        prof_collect(c);
    } else {
        if ((rc = zk_mle_clone(m, &acc))) return rc; // MultiLinearPolynomial::new(&self.evaluation_points), :204
    }
    size_t done = first;
    for (size_t idx = first; idx < npts; ++idx) {
        if (!has[idx]) continue;
        zk_mle* next = nullptr;
        rc = zk_mle_partially_evaluate(acc, idx - done, vals + 4 * idx, &next); // :208
        zk_mle_free(acc);
        if (rc) return rc;
        acc = next;
        ++done;
    }
    *out = acc;
    return ZK_OK;
}

int zk_mle_evaluation_sum(const zk_mle* m, uint64_t out[4]) {
    if (!m || !out) return ZK_E_ARG;
    zk_ctx* c = m->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    int g = zkk::launch_sum_all(m->d, m->len, c->partials, c->stream);
    HIPCHK(c, hipGetLastError());
    Fq s;
    if ((rc = fetch_sums(c, g, 1, &s))) return rc;
    zkh::store(out, s);
    return ZK_OK;
}

int zk_mle_round_sums(const zk_mle* m, uint64_t out[8]) {
    if (!m || !out) return ZK_E_ARG;
    if (m->len < 2) return ZK_E_POINTS_LEN;
    zk_ctx* c = m->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    {
        ProfScope ps(c, ZK_K_BLOCK_SUMS, (uint64_t)m->len * 32);
        zkk::launch_block_sums(m->d, m->len, 1, c->partials, c->sums_host_dev, c->stream, c->gates);
    }
    Fq e[2];
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s2 = 0; s2 < 2; ++s2) e[s2] = zkh::load(c->sums_host + 4 * s2);
    prof_collect(c);
    zkh::store(out, e[0]);
    zkh::store(out + 4, e[1]);
    return ZK_OK;
}

// ---- sumcheck prover: single MLE ---------------------------------------------------------------
// Batched rounds (kernels.hip): S = 2^m block sums of the current table; the host runs m rounds
// on S (half sums of S == half sums of the table, prover.rs:108-116); one device pass then binds
// those m variables and produces the next batch's block sums.
//
// Sharded tables: rank q of `world` holds the entries whose index is q mod world,
// i.e. a table in the top n-g variables for one assignment q of the g variables bound last.  Sums
// over index halves add across ranks, so S = sum_q S_q; folds are rank-local.  `sums` is the
// gathered [world][2^m] array.  folded_out[q] (optional) = rank q's block sums folded at the m
// challenges: when the local tables are down to 2^m entries these are the entries of the remaining
// g-variable table, index q.
} // extern "C"
namespace zki {
void host_rounds(zk_transcript* tr, const uint64_t* sums, int world, unsigned m, uint64_t* coeffs_out, uint64_t* ch_out,
                 uint64_t* folded_out, uint64_t* total_out) {
    const unsigned nb = 1u << m;
    std::vector<Fq> S(nb, zkh::zero()), per(folded_out ? (size_t)world * nb : 0);
    for (int q = 0; q < world; ++q)
        for (unsigned b = 0; b < nb; ++b) {
            const Fq v = zkh::load(sums + 4 * ((size_t)q * nb + b));
            S[b] = zkh::add(S[b], v);
            if (folded_out) per[(size_t)q * nb + b] = v;
        }
    if (total_out) {
        Fq total = zkh::zero();
        for (unsigned b = 0; b < nb; ++b) total = zkh::add(total, S[b]);
        zkh::store(total_out, total); // evaluation_sum(), prover.rs:170
    }
    unsigned slen = nb;
    for (unsigned t = 0; t < m; ++t) {
        const unsigned half = slen / 2;
        Fq e0 = zkh::zero(), e1 = zkh::zero();
        for (unsigned b = 0; b < half; ++b) { e0 = zkh::add(e0, S[b]); e1 = zkh::add(e1, S[half + b]); }
        Fq claimed = zkh::add(e0, e1);      // prover.rs:118
        Fq c1 = zkh::sub(e1, e0);           // interpolate([0,1],[e0,e1]) = [e0, e1-e0], :120-123
        tr->append_fq(claimed);             // :126-129
        tr->append_fq(e0);
        tr->append_fq(c1);
        Fq r = tr->sample();                // :141-144
        zkh::store(coeffs_out + 8 * t, e0);
        zkh::store(coeffs_out + 8 * t + 4, c1);
        if (ch_out) zkh::store(ch_out + 4 * t, r);
        for (unsigned b = 0; b < half; ++b) S[b] = zkh::add(S[b], zkh::mul(zkh::sub(S[half + b], S[b]), r)); // evaluation_form.rs:184
        if (folded_out)
            for (int q = 0; q < world; ++q) {
                Fq* P = &per[(size_t)q * nb];
                for (unsigned b = 0; b < half; ++b) P[b] = zkh::add(P[b], zkh::mul(zkh::sub(P[half + b], P[b]), r));
            }
        slen = half;
    }
    if (folded_out)
        for (int q = 0; q < world; ++q) zkh::store(folded_out + 4 * q, per[(size_t)q * nb]);
}
// one fold pass: dst (lout entries) = src (lout << m entries) with its top m variables bound at r, sums (2^m_next elements)
// = the block sums of dst.  `k` supplies the schedule knobs only (kernel form thresholds); every buffer is an argument, so
// the entry point that allocates nothing shares this dispatch.  Returns the workgroups launched, < 0 if no kernel takes the shape.
int fold_pass(const zk_ctx& k, const uint64_t two24[4], const void* src, void* dst, size_t lout, int m, const uint64_t* r_mont, int m_next,
              void* partials, void* wide_w, void* sums, hipStream_t stream) {
    if (m > zkk::kMaxLazyBatch)
        return zkk::launch_multifold_wide(m, src, dst, lout, r_mont, two24, wide_w, k.wide_lds_below_log, k.tail_max_log > 0 ? k.tail_out_below_log : 0,
                                          m_next, partials, sums, stream, k.gates);
    if (k.lazy && m >= 2) {
        uint32_t w28[(1 << zkk::kMaxLazyBatch) * 10];
        eq_weights28(r_mont, (unsigned)m, w28);
        return zkk::launch_multifold_lazy(m, src, dst, lout, w28, m_next, partials, sums, stream);
    }
    return zkk::launch_multifold(m, src, dst, lout, r_mont, m_next, partials, sums, stream);
}
} // namespace zki
extern "C" {


int zk_sumcheck_rounds_from_block_sums(zk_transcript* t, const uint64_t* sums, int world, int m, uint64_t* coeffs_out,
                                       uint64_t* challenges_out, uint64_t* folded_out, uint64_t total_out[4]) {
    if (!t || !sums || !coeffs_out || world < 1 || m < 1 || m > 30) return ZK_E_ARG;
    host_rounds(t, sums, world, (unsigned)m, coeffs_out, challenges_out, folded_out, total_out);
    return ZK_OK;
}

int zk_shard_begin(zk_ctx* c, const zk_mle* local, zk_shard** out) {
    if (!c || !local || !out || local->ctx != c) return ZK_E_ARG;
    zk_shard* s = new (std::nothrow) zk_shard{c, local->d, ilog2(local->len), 0};
    if (!s) return ZK_E_ALLOC;
    *out = s;
    return ZK_OK;
}
void zk_shard_end(zk_shard* s) { delete s; }
int zk_shard_vars_left(const zk_shard* s) { return s ? (int)s->ell : -1; }

int zk_shard_next_batch(const zk_shard* s, int after_binding) {
    if (!s || after_binding < 0 || (unsigned)after_binding > s->ell) return -1;
    return (int)next_batch(s->ctx, s->ell - (unsigned)after_binding);
}

// dst_dev (2^vars_left elements) = the current local table; asynchronous on the context stream
int zk_shard_copy_table(zk_shard* s, void* dst_dev) {
    if (!s || !dst_dev) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(dst_dev, s->cur, ((size_t)1 << s->ell) * kElem, hipMemcpyDeviceToDevice, c->stream));
    return ZK_OK;
}

// sums_dev_out[b] (b < 2^m) = sum of the b-th contiguous block of the current local table; asynchronous
int zk_shard_block_sums(zk_shard* s, int m, void* sums_dev_out) {
    if (!s || !sums_dev_out || m < 1 || m > zkk::kMaxFoldBatch || (unsigned)m > s->ell) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    const size_t len = (size_t)1 << s->ell;
    {
        ProfScope ps(c, ZK_K_BLOCK_SUMS, (uint64_t)len * 32);
        zkk::launch_block_sums(s->cur, len, m, c->partials, sums_dev_out, c->stream, c->gates);
    }
    HIPCHK(c, hipGetLastError());
    return ZK_OK;
}

// bind the top m variables of the current local table at r[0..m) (m successive
// partially_evaluate((0, r_l)), evaluation_form.rs:163-192) into the context's scratch, and leave the
// 2^m_next block sums of the result in sums_dev_out; asynchronous.  m_next = 0: no sums (last pass).
int zk_shard_fold(zk_shard* s, int m, const uint64_t* r_mont, int m_next, void* sums_dev_out) {
    if (!s || !r_mont || m < 1 || m > zkk::kMaxFoldBatch || (unsigned)m > s->ell) return ZK_E_ARG;
    if (m_next < 0 || m_next > zkk::kMaxFoldBatch || (unsigned)(m + m_next) > s->ell || (m_next && !sums_dev_out)) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    const unsigned ell_out = s->ell - (unsigned)m;
    const size_t lout = (size_t)1 << ell_out;
    if ((rc = ensure_work(c, s->pass & 1, lout * kElem))) return rc;
    void* dst = c->work[s->pass & 1];
    void* sums = m_next ? sums_dev_out : c->sums_dev;
    int launched = 0; // workgroups of the pass, < 0: the launcher rejected the shape and launched nothing
    {
        const bool tail = m >= 6 && c->tail_max_log > 0 && lout < ((size_t)1 << c->tail_out_below_log); // as launch_multifold_wide decides
        ProfScope ps(c, m <= zkk::kMaxLazyBatch ? ZK_K_MULTIFOLD : tail ? ZK_K_MULTIFOLD_TAIL : ZK_K_MULTIFOLD_WIDE,
                     (uint64_t)(lout << m) * 32 + (uint64_t)lout * 32);
        launched = fold_pass(*c, c->two24, s->cur, dst, lout, m, r_mont, m_next, c->partials, c->wide_w, sums, c->stream);
    }
    HIPCHK(c, hipGetLastError());
    if (launched < 0) {
        c->err = "fold pass: shape not supported by the kernel";
        return ZK_E_ARG;
    }
    s->cur = dst;
    s->ell = ell_out;
    ++s->pass;
    return ZK_OK;
}

// The fold pass of zk_shard_fold for callers that cannot allocate: every buffer it touches is an argument or the
// library's static __device__ scratch array (kernels.hip).  No hipMalloc / hipFree / allocating call on this path;
// asynchronous on `hip_stream`, device = the calling thread's current device.
int zk_fold_block_sums_noalloc(void* hip_stream, const void* src_dev, size_t len, int m, const uint64_t* r_mont, void* dst_dev, int m_next,
                               void* sums_dev_out) {
    if (!src_dev || !dst_dev || !r_mont || ((uintptr_t)src_dev | (uintptr_t)dst_dev | (uintptr_t)sums_dev_out) % 16) return ZK_E_ARG;
    if (!is_pow2(len)) return ZK_E_NOT_POW2;
    if (len > ((size_t)1 << ZK_NOALLOC_MAX_LOG_LEN)) return ZK_E_TOO_LARGE;
    const unsigned ell = ilog2(len);
    if (m < 1 || m > zkk::kMaxFoldBatch || (unsigned)m > ell) return ZK_E_ARG;
    if (m_next < 0 || m_next > zkk::kMaxFoldBatch || (unsigned)(m + m_next) > ell || (m_next && !sums_dev_out)) return ZK_E_ARG;
    { // the kernels stream src while they store dst (and the sums): ranges that intersect would silently yield garbage
        const uintptr_t s0 = (uintptr_t)src_dev, s1 = s0 + len * kElem, d0 = (uintptr_t)dst_dev, d1 = d0 + (len >> m) * kElem;
        if (s0 < d1 && d0 < s1) return ZK_E_ARG;
        if (m_next) {
            const uintptr_t q0 = (uintptr_t)sums_dev_out, q1 = q0 + ((size_t)kElem << m_next);
            if ((s0 < q1 && q0 < s1) || (d0 < q1 && q0 < d1)) return ZK_E_ARG;
        }
    }
    void* base = nullptr; // this device's copy of the array (first call on a device: hipGetSymbolAddress; cached after that)
    if (zkk::static_scratch_ptr(&base) != hipSuccess) return ZK_E_HIP;
    void* partials = base;
    void* wide_w = (uint8_t*)base + zkk::kStaticPartialsBytes;
    void* own_sum = (uint8_t*)base + zkk::kStaticPartialsBytes + zkk::kStaticWeightsBytes; // m_next = 0: the pass still writes one total
    // Every scratch word a kernel of the pass reads was written earlier in the same pass (each workgroup stores its partial
    // unconditionally, k_eq_weights28 writes all 2^m weights), so nothing has to start at zero.
    static const zk_ctx defaults{}; // the library's default schedule knobs (no device state)
    uint64_t two24[4];
    zkh::store(two24, zkh::from_u64((uint64_t)1 << 24));
    const size_t lout = len >> m;
    const int launched = fold_pass(defaults, two24, src_dev, dst_dev, lout, m, r_mont, m_next, partials, wide_w, m_next ? sums_dev_out : own_sum,
                                   (hipStream_t)hip_stream);
    if (launched < 0) return ZK_E_ARG;
    return hipGetLastError() == hipSuccess ? ZK_OK : ZK_E_HIP;
}

int zk_sumcheck_prove_mle(zk_ctx* c, const zk_mle* poly, int absorb_table, zk_transcript* transcript,
                          uint64_t claim_out[4], uint64_t* coeffs_out, uint64_t* challenges_out) {
    if (!c || !poly || !claim_out) return ZK_E_ARG;
    if (poly->ctx != c) return ZK_E_ARG;
    const size_t len = poly->len;
    const unsigned n = ilog2(len);
    if (n && !coeffs_out) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;

    zk_transcript local;
    zk_transcript* tr = transcript ? transcript : &local; // Transcript::new(), prover.rs:159
    if (!transcript || absorb_table) {                      // transcript.append(&init_polynomial.to_bytes()), :162
        rc = stream_le32(c, poly->d, len, [&](const uint8_t* p, size_t nbytes, size_t) { tr->append(p, nbytes); });
        if (rc) return rc;
    }
    return prove_mle_dev(c, poly->d, n, tr, claim_out, coeffs_out, challenges_out);
}
} // extern "C"

// generate_round_polys (prover.rs:45-155, MLE branch) on a device table of 2^n entries and the caller's transcript
int zki::prove_mle_dev(zk_ctx* c, const void* table, unsigned n, zk_transcript* tr, uint64_t claim_out[4], uint64_t* coeffs_out,
                       uint64_t* challenges_out) {
    if (n == 0) { // zero variables: no rounds; initial_claim_sum = the single entry
        HIPCHK(c, hipMemcpyAsync(c->sums_host, table, kElem, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::memcpy(claim_out, c->sums_host, kElem);
        return ZK_OK;
    }
    int rc;
    zk_shard st{c, table, n, 0};
    unsigned m = next_batch(c, st.ell), k = 0;
    if ((rc = zk_shard_block_sums(&st, (int)m, c->sums_host_dev))) return rc; // sums land in pinned host memory
    for (;;) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        uint64_t rs[4 * zkk::kMaxFoldBatch];
        host_rounds(tr, c->sums_host, 1, m, coeffs_out + 8 * k, rs, nullptr, k == 0 ? claim_out : nullptr);
        if (challenges_out) std::memcpy(challenges_out + 4 * k, rs, (size_t)m * kElem);
        k += m;
        if (st.ell == m) break; // the fully folded table (one entry) is never used by the prover
        const unsigned m_next = next_batch(c, st.ell - m);
        if ((rc = zk_shard_fold(&st, (int)m, rs, (int)m_next, c->sums_host_dev))) return rc;
        m = m_next;
    }
    prof_collect(c);
    return ZK_OK;
}

// ---- sumcheck prover: Sum[Prod[t0,t1], Prod[t2,t3]] ---------------------------------------------

namespace {
// one round of the SumPolynomial branch from the three evaluations (prover.rs:102-107, :126-135);
// interpolate([0,1,2], ev) in closed form -- dense_coefficient_form.rs:52-88 gives the same canonical
// coefficients: c0 = ev0, c2 = (ev0 - 2 ev1 + ev2)/2, c1 = ev1 - ev0 - c2
Fq host_round_deg2(zk_transcript* tr, const Fq ev[3], uint64_t* coeffs_out) {
    const Fq inv2 = zkh::inv2();
    Fq claimed = zkh::add(ev[0], ev[1]);
    Fq c2 = zkh::mul(zkh::add(zkh::sub(ev[0], zkh::add(ev[1], ev[1])), ev[2]), inv2);
    Fq c1 = zkh::sub(zkh::sub(ev[1], ev[0]), c2);
    tr->append_fq(claimed);
    tr->append_fq(ev[0]);
    tr->append_fq(c1);
    tr->append_fq(c2);
    zkh::store(coeffs_out, ev[0]);
    zkh::store(coeffs_out + 4, c1);
    zkh::store(coeffs_out + 8, c2);
    return tr->sample();
}

// all log2(len) rounds of the SumPolynomial branch on four host tables T[q*len + j]; final_out (optional):
// the four tables folded at every challenge
void host_sum2x2_rounds(zk_transcript* tr, std::vector<Fq>& T, size_t len, uint64_t* coeffs_out, uint64_t* challenges_out,
                        uint64_t* final_out) {
    const unsigned n = ilog2(len);
    size_t cur = len;
    for (unsigned k = 0; k < n; ++k) {
        const size_t half = cur / 2;
        Fq ev[3] = {zkh::zero(), zkh::zero(), zkh::zero()};
        for (size_t j = 0; j < half; ++j) {
            Fq lo[4], hi[4], x2[4];
            for (int q = 0; q < 4; ++q) {
                lo[q] = T[q * len + j];
                hi[q] = T[q * len + half + j];
                x2[q] = zkh::sub(zkh::add(hi[q], hi[q]), lo[q]); // lo + 2 (hi - lo)
            }
            ev[0] = zkh::add(ev[0], zkh::add(zkh::mul(lo[0], lo[1]), zkh::mul(lo[2], lo[3])));
            ev[1] = zkh::add(ev[1], zkh::add(zkh::mul(hi[0], hi[1]), zkh::mul(hi[2], hi[3])));
            ev[2] = zkh::add(ev[2], zkh::add(zkh::mul(x2[0], x2[1]), zkh::mul(x2[2], x2[3])));
        }
        const Fq r = host_round_deg2(tr, ev, coeffs_out + 12 * k);
        if (challenges_out) zkh::store(challenges_out + 4 * k, r);
        for (int q = 0; q < 4; ++q)
            for (size_t j = 0; j < half; ++j)
                T[q * len + j] = zkh::add(T[q * len + j], zkh::mul(zkh::sub(T[q * len + half + j], T[q * len + j]), r));
        cur = half;
    }
    if (final_out)
        for (int q = 0; q < 4; ++q) zkh::store(final_out + 4 * q, T[q * len]);
}
} // namespace

// the round loop of the SumPolynomial branch on four device tables of `len` entries each;
// final_out (optional, 4 elements): every table folded at ALL n challenges (the reference's
// sum_poly after the last partial_evaluate, prover.rs:139) -- GKR needs W(r_b), W(r_c) from it
// one round per device pass (k_eval3 / k_fold4_eval3)
// The raw-digit buffer of the cross-sum sweeps for tables of 8 q entries: grow-only, made on first use (a prove at a size the context
// has seen allocates nothing).  Null (the sweeps then form their field elements per workgroup) when switched off or when the
// allocation fails.
static void* cross_raw_for(zk_ctx* c, size_t q) {
    if (!c->cross_raw_on) return nullptr;
    const size_t need = zkk::cross64_raw_bytes(q);
    if (need > c->cross_raw_cap) {
        if (c->cross_raw) {
            (void)hipStreamSynchronize(c->stream); // no queued kernel still writes the old buffer
            (void)hipFree(c->cross_raw);
            c->cross_raw = nullptr;
            c->cross_raw_cap = 0;
        }
        if (hipMalloc(&c->cross_raw, need) != hipSuccess) {
            (void)hipGetLastError();
            c->cross_raw = nullptr;
            return nullptr;
        }
        c->cross_raw_cap = need;
    }
    return c->cross_raw;
}

static int prove_sum2x2_single(zk_ctx* c, const void* const tables[4], size_t len, zk_transcript* tr, uint64_t* coeffs_out,
                          uint64_t* challenges_out, uint64_t* final_out) {
    const unsigned n = ilog2(len);
    if (n == 0) { // zero variables: no rounds, nothing appended
        if (final_out)
            for (int t = 0; t < 4; ++t) {
                HIPCHK(c, hipMemcpyAsync(c->sums_host + 4 * t, tables[t], kElem, hipMemcpyDeviceToHost, c->stream));
            }
        if (final_out) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::memcpy(final_out, c->sums_host, 4 * kElem);
        }
        return ZK_OK;
    }
    if (!coeffs_out) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    // ping-pong scratch: 4 tables of len/2 in work[0], 4 of len/4 in work[1]
    if (n >= 2 && (rc = ensure_work(c, 0, 4 * (len / 2) * kElem))) return rc;
    if (n >= 3 && (rc = ensure_work(c, 1, 4 * (len / 4) * kElem))) return rc;

    const void* cur[4];
    for (int k = 0; k < 4; ++k) cur[k] = tables[k];
    size_t curlen = len;
    Fq ev[3];
    int g;
    for (unsigned k = 0; k < n; ++k) {
        if (curlen <= kHostTail && curlen >= 2) {
            // small tables: one D2H of the four tables, the remaining rounds (and the last fold) on the host --
            // a round on the device costs a launch, a second stage and a sync whatever the size
            zkk::launch_gather4(cur, curlen, c->sums_host_dev, c->stream);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::vector<Fq> T(4 * curlen);
            for (size_t i = 0; i < 4 * curlen; ++i) T[i] = zkh::load(c->sums_host + 4 * i);
            host_sum2x2_rounds(tr, T, curlen, coeffs_out + 12 * k, challenges_out ? challenges_out + 4 * k : nullptr, final_out);
            break;
        }
        if (k == 0) { // first round's evaluations; later rounds get theirs from the fused fold
            {
                ProfScope ps(c, ZK_K_EVAL3, (uint64_t)len * 32 * 4);
                g = zkk::launch_eval3(cur, len / 2, c->partials, c->stream);
            }
            HIPCHK(c, hipGetLastError());
            if ((rc = fetch_sums(c, g, 3, ev))) return rc;
        }
        // interpolate([0,1,2], ev) in closed form (dense_coefficient_form.rs:52-88 gives the same canonical
        // coefficients): host_round_deg2
        Fq r = host_round_deg2(tr, ev, coeffs_out + 12 * k); // prover.rs:102-107, :126-135
        if (challenges_out) zkh::store(challenges_out + 4 * k, r);
        if (k + 1 == n) {
            if (final_out) { // the last fold, on the host: every table is down to two entries
                for (int t = 0; t < 4; ++t)
                    HIPCHK(c, hipMemcpyAsync(c->sums_host + 8 * t, cur[t], 2 * kElem, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                for (int t = 0; t < 4; ++t) {
                    const Fq y1 = zkh::load(c->sums_host + 8 * t), y2 = zkh::load(c->sums_host + 8 * t + 4);
                    zkh::store(final_out + 4 * t, zkh::add(y1, zkh::mul(zkh::sub(y2, y1), r)));
                }
            }
            break;
        }
        void* dst[4];
        const size_t outlen = curlen / 2;
        for (int t = 0; t < 4; ++t) dst[t] = (uint8_t*)c->work[k & 1] + (size_t)t * outlen * kElem;
        {
            ProfScope ps(c, ZK_K_FOLD4_EVAL3, (uint64_t)curlen * 48 * 4);
            g = zkk::launch_fold4_eval3(cur, dst, curlen / 4, r.l, c->partials, c->stream);
        }
        HIPCHK(c, hipGetLastError());
        if ((rc = fetch_sums(c, g, 3, ev))) return rc;
        for (int t = 0; t < 4; ++t) cur[t] = dst[t];
        curlen = outlen;
    }
    prof_collect(c);
    return ZK_OK;
}

// the host part of two rounds from the nine sums S[3Y + t] (deg2_kernels.hip, "degree-2 branch, TWO rounds per pass")
static void host_round_pair_from_sum9(zk_transcript* tr, const Fq S[9], uint64_t* coeffs_out /* 2 x 3 */, uint64_t rs[8]) {
    // round k binds b1: ev(X) = sum_{Y in {0,1}} [S0 + X (S1 - S0 - S2) + X^2 S2](Y)
    Fq mid[3]; // S1 - S0 - S2 per Y
    for (int Y = 0; Y < 3; ++Y) mid[Y] = zkh::sub(zkh::sub(S[3 * Y + 1], S[3 * Y]), S[3 * Y + 2]);
    Fq ev[3];
    ev[0] = zkh::add(S[0], S[3]);
    ev[1] = zkh::add(S[1], S[4]);
    {
        Fq e2 = zkh::zero();
        for (int Y = 0; Y < 2; ++Y) {
            const Fq two_mid = zkh::add(mid[Y], mid[Y]), two_s2 = zkh::add(S[3 * Y + 2], S[3 * Y + 2]);
            e2 = zkh::add(e2, zkh::add(zkh::add(S[3 * Y], two_mid), zkh::add(two_s2, two_s2)));
        }
        ev[2] = e2;
    }
    const Fq r = host_round_deg2(tr, ev, coeffs_out);
    zkh::store(rs, r);
    // round k+1 binds b2 with b1 = r: ev'(Y) = S0 + r (S1 - S0 - S2) + r^2 S2
    const Fq r2 = zkh::mul(r, r);
    for (int Y = 0; Y < 3; ++Y) ev[Y] = zkh::add(zkh::add(S[3 * Y], zkh::mul(r, mid[Y])), zkh::mul(r2, S[3 * Y + 2]));
    const Fq s2 = host_round_deg2(tr, ev, coeffs_out + 12);
    zkh::store(rs + 4, s2);
}

// the host part of THREE rounds from the 27 sums T[9a + 3b + c], a, b, c in {0, 1, oo} (deg2_kernels.hip, k_fold_sum27): along each
// variable the summed product is P(X) = P[0] + X (P[1] - P[0] - P[oo]) + X^2 P[oo].  Round i sums the later variables over {0, 1} and
// reads ev(0) = A[0], ev(1) = A[1], ev(2) = 2 A[1] - A[0] + 2 A[oo]; binding X = r leaves the same structure one variable shorter.
// Exact arithmetic: the evaluations are the field elements prover.rs:93-100 sums.
static void host_round_triple_from_sum27(zk_transcript* tr, const Fq T[27], uint64_t* coeffs_out /* 3 x 3 */, uint64_t rs[12]) {
    Fq cur[27];
    for (int i = 0; i < 27; ++i) cur[i] = T[i];
    int rest = 9; // entries per value of the leading index
    for (int i = 0; i < 3; ++i) {
        Fq A[3];
        for (int a = 0; a < 3; ++a) {
            Fq acc = zkh::zero();
            for (int idx = 0; idx < rest; ++idx)
                if (idx % 3 != 2 && (idx / 3) % 3 != 2) acc = zkh::add(acc, cur[a * rest + idx]); // later variables over {0, 1}
            A[a] = acc;
        }
        const Fq ev[3] = {A[0], A[1], zkh::add(zkh::sub(zkh::add(A[1], A[1]), A[0]), zkh::add(A[2], A[2]))};
        const Fq r = host_round_deg2(tr, ev, coeffs_out + 12 * i);
        zkh::store(rs + 4 * i, r);
        const Fq r2 = zkh::mul(r, r);
        for (int idx = 0; idx < rest; ++idx) {
            const Fq p0 = cur[idx], p1 = cur[rest + idx], pi = cur[2 * rest + idx];
            cur[idx] = zkh::add(zkh::add(p0, zkh::mul(r, zkh::sub(zkh::sub(p1, p0), pi))), zkh::mul(r2, pi));
        }
        rest /= 3;
    }
}

// the host part of K rounds from the cross sums P[term][e][e'] = sum_j F_e[j] G_e'[j] (e, e' < 2^K index the K top variables, the
// first to bind on the most significant bit; deg2_kernels.hip, k_cross64_mfma).  With x the point of the K variables,
// sum_j F(x, j) G(x, j) = sum_{e,e'} eq(x, e) eq(x, e') P[e][e'], so round i (r_0..r_{i-1} bound, X free, the rest y boolean) reads
// only the entries whose remaining bits agree:  ev(X) = sum_y sum_{a,a'} L_a(X) L_a'(X) Q[(a,y)][(a',y)],  L_0 = 1 - X, L_1 = X,
// and binding r contracts Q's top bit on both sides.  Exact arithmetic: the evaluations are the field elements prover.rs:93-100 sums.
static void host_rounds_from_cross(zk_transcript* tr, int K, const Fq* P /* [2][4^K] */, uint64_t* coeffs_out /* K x 3 */, uint64_t* rs /* K x 4 */) {
    size_t n = (size_t)1 << K;
    std::vector<Fq> Q(P, P + 2 * n * n), Qn;
    for (int i = 0; i < K; ++i) {
        const size_t h = n / 2;
        Fq ev[3] = {zkh::zero(), zkh::zero(), zkh::zero()};
        for (int t = 0; t < 2; ++t) {
            const Fq* q = Q.data() + (size_t)t * n * n;
            for (size_t y = 0; y < h; ++y) {
                const Fq q00 = q[y * n + y], q01 = q[y * n + h + y], q10 = q[(h + y) * n + y], q11 = q[(h + y) * n + h + y];
                ev[0] = zkh::add(ev[0], q00);
                ev[1] = zkh::add(ev[1], q11);
                // X = 2: L_0 = -1, L_1 = 2
                const Fq cross = zkh::add(q01, q10), q11x2 = zkh::add(q11, q11);
                ev[2] = zkh::add(ev[2], zkh::sub(zkh::add(q00, zkh::add(q11x2, q11x2)), zkh::add(cross, cross)));
            }
        }
        const Fq r = host_round_deg2(tr, ev, coeffs_out + 12 * i);
        zkh::store(rs + 4 * i, r);
        const Fq omr = zkh::sub(zkh::one(), r);
        const Fq c00 = zkh::mul(omr, omr), c01 = zkh::mul(omr, r), c11 = zkh::mul(r, r);
        Qn.assign(2 * h * h, zkh::zero());
        for (int t = 0; t < 2; ++t) {
            const Fq* q = Q.data() + (size_t)t * n * n;
            Fq* o = Qn.data() + (size_t)t * h * h;
            for (size_t y = 0; y < h; ++y)
                for (size_t y2 = 0; y2 < h; ++y2)
                    o[y * h + y2] = zkh::add(zkh::add(zkh::mul(c00, q[y * n + y2]), zkh::mul(c01, zkh::add(q[y * n + h + y2], q[(h + y) * n + y2]))),
                                             zkh::mul(c11, q[(h + y) * n + h + y2]));
        }
        Q.swap(Qn);
        n = h;
    }
}

void zki::host_rounds_from_cross64(zk_transcript* tr, const uint64_t* cross, int world, uint64_t* coeffs_out, uint64_t* rs) {
    Fq P[128];
    for (int i = 0; i < 128; ++i) {
        Fq a = zkh::zero();
        for (int q = 0; q < world; ++q) a = zkh::add(a, zkh::load(cross + 4 * ((size_t)q * 128 + i)));
        P[i] = a;
    }
    host_rounds_from_cross(tr, 3, P, coeffs_out, rs);
}

// two (nine sums) or three (8 x 8 cross sums) rounds per device pass
// ones3: tables[3] is a table of ones (the relation is tables[0] tables[1] + tables[2] * 1: both phases of a GKR layer); the 27-sum
// passes then run in their product-plus-linear form, which neither reads nor folds it (cur[3] stays the caller's table: a prefix of
// a table of ones is its own fold), every other kernel treats it as the ordinary table it also is
static int prove_sum2x2_pairs(zk_ctx* c, const void* const tables[4], size_t len, zk_transcript* tr, uint64_t* coeffs_out,
                              uint64_t* challenges_out, uint64_t* final_out, bool ones3) {
    int rc = set_device(c);
    if (rc) return rc;
    const void* cur[4];
    for (int k = 0; k < 4; ++k) cur[k] = tables[k];
    size_t curlen = len;
    unsigned k = 0; // round index
    int pass = 0;
    // the sums sweep over four tables of n entries; returns the rounds it serves (3: cross sums, 2: nine sums)
    // what the pinned buffer holds after a two-round sums launch: the nine sums (0 groups), G workgroups' shares of them (G > 0), or the
    // nine U sums of k_sum9_lazy (kU) -- the last two are finished here on the host instead of by one more small launch
    constexpr int kU = -1;
    int sums_kind = 0;
    bool gathered = false; // the last fold has left the four small tables in the pinned buffer already
    bool have27 = false; // the pinned buffer holds k_fold_sum27's 27 sums (or sums_kind workgroups' shares of them): three rounds
    const bool pwq = ones3 && c->gates.deg2_pwq_first_below_log > 0; // product + linear relation in its own form (k_fold_sum27<M, true>)
    auto launch_sums = [&](const void* const t[4], size_t n) -> int {
        ProfScope ps(c, ZK_K_SUM9, (uint64_t)n * 32 * 4);
        have27 = false;
        if (deg2_triple_ok(c, n, pwq)) {
            zkk::launch_cross64(t, n / 8, c->gates.deg2_triple_ring, c->partials, cross_raw_for(c, n / 8), c->sums_host_dev, c->stream); // 128 cross sums straight into pinned host memory
            return 3;
        }
        sums_kind = zkk::launch_sum9(t, n / 4, c->two24, c->deg2_lazy_from_log, c->partials, c->sums_host_dev, c->stream, c->gates, true, pwq) ? kU : 0;
        return 2;
    };
    auto nine_sums = [&](Fq S[9]) {
        if (sums_kind > 0) { // shares of sums_kind workgroups, fixed order
            for (int i = 0; i < 9; ++i) {
                Fq a = zkh::zero();
                for (int g = 0; g < sums_kind; ++g) a = zkh::add(a, zkh::load(c->sums_host + 4 * ((size_t)g * 9 + i)));
                S[i] = a;
            }
        } else if (sums_kind == kU) { // k_sum9_from_u's maps: V[a][Y] = U[a][Y] (Y < 2), V[a][2] = 3 U[a][0] + 6 U[a][1] - 2 U[a][+]
            Fq U[9];
            for (int i = 0; i < 9; ++i) U[i] = zkh::load(c->sums_host + 4 * i);
            for (int Y = 0; Y < 3; ++Y) {
                Fq V[3];
                for (int a = 0; a < 3; ++a) {
                    if (Y < 2) { V[a] = U[3 * Y + a]; continue; }
                    const Fq u0 = U[a], u1 = U[3 + a], up = U[6 + a], u1x2 = zkh::add(u1, u1);
                    V[a] = zkh::sub(zkh::add(zkh::add(zkh::add(u0, u0), u0), zkh::add(zkh::add(u1x2, u1x2), u1x2)), zkh::add(up, up));
                }
                S[3 * Y] = V[0];
                S[3 * Y + 1] = V[1];
                const Fq v01 = zkh::add(V[0], V[1]);
                S[3 * Y + 2] = zkh::sub(zkh::add(v01, v01), V[2]);
            }
        } else {
            for (int i = 0; i < 9; ++i) S[i] = zkh::load(c->sums_host + 4 * i);
        }
    };
    int K = 0;
    if (curlen > kHostTail) {
        const int fb0 = pwq ? c->gates.deg2_pwq_first_below_log : c->gates.deg2_sum27_first_below_log;
        if (fb0 > 0 && curlen < ((size_t)1 << std::min(fb0, 62)) && curlen % 16 == 0 && !deg2_triple_ok(c, curlen, pwq)) {
            // the first pass as 27 sums: three rounds before the first fold
            ProfScope ps(c, ZK_K_SUM9, (uint64_t)curlen * 32 * (pwq ? 3 : 4));
            sums_kind = zkk::launch_fold_sum27(0, cur, nullptr, curlen, nullptr, c->partials, c->sums_host_dev, c->stream, kHostSumGroups, pwq);
            if (sums_kind < 0) return ZK_E_ARG;
            zkk::g_forms |= zkk::kFormSum27First;
            have27 = true;
            K = 3;
        } else {
            K = launch_sums(cur, curlen);
        }
    }
    while (curlen > kHostTail) {
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        uint64_t rs[12];
        if (K == 3 && have27) {
            Fq T[27];
            for (int i = 0; i < 27; ++i) {
                Fq a = zkh::load(c->sums_host + 4 * i);
                for (int g = 1; g < sums_kind; ++g) a = zkh::add(a, zkh::load(c->sums_host + 4 * ((size_t)g * 27 + i))); // fixed order
                T[i] = a;
            }
            host_round_triple_from_sum27(tr, T, coeffs_out + 12 * k, rs);
        } else if (K == 3) {
            Fq P[128];
            for (int i = 0; i < 128; ++i) P[i] = zkh::load(c->sums_host + 4 * i);
            host_rounds_from_cross(tr, 3, P, coeffs_out + 12 * k, rs);
        } else {
            Fq S[9];
            nine_sums(S);
            host_round_pair_from_sum9(tr, S, coeffs_out + 12 * k, rs);
        }
        if (challenges_out) std::memcpy(challenges_out + 4 * k, rs, (size_t)K * kElem);
        k += (unsigned)K;
        // bind the K variables in one sweep, then the sums of the result unless the host takes over
        const size_t outlen = curlen >> K;
        if ((rc = ensure_work(c, pass & 1, 4 * outlen * kElem))) return rc;
        void* dst[4];
        for (int t = 0; t < 4; ++t) dst[t] = (uint8_t*)c->work[pass & 1] + (size_t)t * outlen * kElem;
        const int fb = c->gates.deg2_fused_below_log, tb = c->gates.deg2_sum27_below_log;
        const bool cont3 = K == 3 && !have27 && c->gates.deg2_fold3_fused && deg2_triple_cont_ok(c, outlen, pwq); // this fold leaves the next cross sums
        if (tb > 0 && outlen > kHostTail && outlen < ((size_t)1 << std::min(tb, 62)) && !deg2_triple_ok(c, outlen, pwq) && !cont3) {
            // small tables: the fold of this pass's K variables and the 27 sums that serve the next THREE rounds in one launch
            ProfScope ps(c, ZK_K_FOLD2X4_SUM9, ((uint64_t)curlen * 32 + (uint64_t)outlen * 32) * (pwq ? 3 : 4));
            uint32_t w28[8 * 10];
            eq_weights28(rs, (unsigned)K, w28);
            sums_kind = zkk::launch_fold_sum27(K, cur, dst, outlen, w28, c->partials, c->sums_host_dev, c->stream, kHostSumGroups, pwq);
            if (sums_kind < 0) return ZK_E_ARG;
            if (pwq) dst[3] = const_cast<void*>(cur[3]); // not folded: still the table of ones
            have27 = true;
            K = 3;
        } else if (cont3) {
            // the next pass serves three rounds too: its cross sums come out of this fold's sweep
            ProfScope ps(c, ZK_K_FOLD2X4_SUM9, (uint64_t)curlen * 32 * 4 + (uint64_t)outlen * 32 * 4);
            zkk::launch_fold3_cross64(cur, dst, outlen, rs, c->two24, c->wide_w, c->gates.deg2_fold3_fused, c->partials, cross_raw_for(c, outlen / 8), c->sums_host_dev, c->stream,
                                      c->gates);
        } else if (K == 2 && outlen > kHostTail && fb > 0 && outlen < ((size_t)1 << std::min(fb, 62))) {
            // small tables: the fold and the nine sums of its result in one launch
            ProfScope ps(c, ZK_K_FOLD2X4_SUM9, (uint64_t)curlen * 32 * 4 + (uint64_t)outlen * 32 * 4);
            uint32_t w28[4 * 10];
            eq_weights28(rs, 2, w28);
            sums_kind = zkk::launch_fold2_sum9(cur, dst, outlen, w28, c->partials, c->sums_host_dev, c->stream, kHostSumGroups);
        } else {
            {
                ProfScope ps(c, ZK_K_FOLD2X4_SUM9, (uint64_t)curlen * 32 * 4 + (uint64_t)outlen * 32 * 4);
                if (K == 3 && (outlen < kFold3SmallBelow || outlen % 64)) {
                    uint32_t w28[8 * 10]; // small tables: one launch, weights as kernel arguments
                    eq_weights28(rs, 3, w28);
                    // the host takes over after this fold: the folded tables go to the pinned buffer too (no k_gather4 launch)
                    gathered = outlen <= kHostTail;
                    zkk::launch_fold3_tables_small(cur, dst, outlen, w28, gathered ? c->sums_host_dev : nullptr, c->stream);
                } else if (K == 3) {
                    zkk::launch_fold3_tables(cur, dst, outlen, rs, c->two24, c->wide_w, c->partials, c->stream, c->gates);
                } else {
                    uint32_t w28[4 * 10];
                    eq_weights28(rs, 2, w28);
                    zkk::launch_fold2_tables(cur, dst, outlen, w28, c->wide_w, c->partials, c->stream, c->gates);
                }
            }
            if (outlen > kHostTail) K = launch_sums(dst, outlen);
        }
        for (int t = 0; t < 4; ++t) cur[t] = dst[t];
        curlen = outlen;
        ++pass;
    }
    HIPCHK(c, hipGetLastError());
    // the rest (<= kHostTail entries per table, at least 2 variables' worth unless len was tiny) on the host
    if (!gathered) zkk::launch_gather4(cur, curlen, c->sums_host_dev, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<Fq> T(4 * curlen);
    for (size_t i = 0; i < 4 * curlen; ++i) T[i] = zkh::load(c->sums_host + 4 * i);
    host_sum2x2_rounds(tr, T, curlen, coeffs_out + 12 * k, challenges_out ? challenges_out + 4 * k : nullptr, final_out);
    prof_collect(c);
    return ZK_OK;
}

int zki::prove_sum2x2_dev(zk_ctx* c, const void* const tables[4], size_t len, zk_transcript* tr, uint64_t* coeffs_out,
                          uint64_t* challenges_out, uint64_t* final_out, bool ones3) {
    if (len >= 2 && !coeffs_out) return ZK_E_ARG;
    if (c->deg2_pairs && len >= 4) return prove_sum2x2_pairs(c, tables, len, tr, coeffs_out, challenges_out, final_out, ones3);
    return prove_sum2x2_single(c, tables, len, tr, coeffs_out, challenges_out, final_out);
}

extern "C" {

int zk_sumcheck_prove_sum2x2(zk_ctx* c, zk_mle* const tables[4], const uint64_t claim[4], zk_transcript* tr,
                             uint64_t* coeffs_out, uint64_t* challenges_out) {
    if (!c || !tables || !claim || !tr) return ZK_E_ARG;
    for (int k = 0; k < 4; ++k)
        if (!tables[k] || tables[k]->ctx != c) return ZK_E_ARG;
    const size_t len = tables[0]->len;
    for (int k = 1; k < 4; ++k)
        if (tables[k]->len != len) return ZK_E_LEN_MISMATCH; // product_polynomial.rs:15-19, sum_polynomial.rs:15-19
    const void* t[4] = {tables[0]->d, tables[1]->d, tables[2]->d, tables[3]->d};
    return prove_sum2x2_dev(c, t, len, tr, coeffs_out, challenges_out, nullptr);
}

int zk_sumcheck_prove_prod_plus_linear(zk_ctx* c, zk_mle* const tables[3], const uint64_t claim[4], zk_transcript* tr,
                                       uint64_t* coeffs_out, uint64_t* challenges_out) {
    if (!c || !tables || !claim || !tr) return ZK_E_ARG;
    for (int k = 0; k < 3; ++k)
        if (!tables[k] || tables[k]->ctx != c) return ZK_E_ARG;
    const size_t len = tables[0]->len;
    for (int k = 1; k < 3; ++k)
        if (tables[k]->len != len) return ZK_E_LEN_MISMATCH; // product_polynomial.rs:15-19, sum_polynomial.rs:15-19
    int rc = set_device(c);
    if (rc) return rc;
    void* ones = nullptr;
    HIPCHK(c, hipMalloc(&ones, len * kElem));
    zkk::launch_fill_one(ones, len, c->stream);
    const void* t[4] = {tables[0]->d, tables[1]->d, tables[2]->d, ones};
    rc = prove_sum2x2_dev(c, t, len, tr, coeffs_out, challenges_out, nullptr, /*ones3=*/true);
    (void)hipStreamSynchronize(c->stream); // no kernel reads `ones` after this
    (void)hipFree(ones);
    return rc;
}

// ---- degree-2 prover in steps (sharded tables) ---------------------------------------------------

// evals: [world][3] partial evaluations (every rank's share of ev[0..3)); sums them, runs the round
int zk_sumcheck_round_from_evals(zk_transcript* t, const uint64_t* evals, int world, uint64_t coeffs_out[12], uint64_t challenge_out[4]) {
    if (!t || !evals || !coeffs_out || !challenge_out || world < 1) return ZK_E_ARG;
    Fq ev[3] = {zkh::zero(), zkh::zero(), zkh::zero()};
    for (int q = 0; q < world; ++q)
        for (int x = 0; x < 3; ++x) ev[x] = zkh::add(ev[x], zkh::load(evals + 4 * (3 * (size_t)q + x)));
    zkh::store(challenge_out, host_round_deg2(t, ev, coeffs_out));
    return ZK_OK;
}

// the whole SumPolynomial-branch round loop on four small HOST tables (the tail of a sharded prove:
// one entry per rank and table).  tables: 4 x len elements; coeffs_out log2(len) x 3.
int zk_sumcheck_prove_sum2x2_host(zk_transcript* t, const uint64_t* tables, size_t len, uint64_t* coeffs_out, uint64_t* challenges_out) {
    if (!t || !tables) return ZK_E_ARG;
    if (!is_pow2(len)) return ZK_E_NOT_POW2;
    const unsigned n = ilog2(len);
    if (n && !coeffs_out) return ZK_E_ARG;
    std::vector<Fq> T(4 * len);
    for (size_t i = 0; i < 4 * len; ++i) T[i] = zkh::load(tables + 4 * i);
    host_sum2x2_rounds(t, T, len, coeffs_out, challenges_out, nullptr);
    return ZK_OK;
}

int zk_shard4_begin(zk_ctx* c, zk_mle* const tables[4], zk_shard4** out) {
    if (!c || !tables || !out) return ZK_E_ARG;
    for (int k = 0; k < 4; ++k)
        if (!tables[k] || tables[k]->ctx != c) return ZK_E_ARG;
    for (int k = 1; k < 4; ++k)
        if (tables[k]->len != tables[0]->len) return ZK_E_LEN_MISMATCH;
    zk_shard4* s = new (std::nothrow) zk_shard4{c, {tables[0]->d, tables[1]->d, tables[2]->d, tables[3]->d}, tables[0]->len, 0};
    if (!s) return ZK_E_ALLOC;
    *out = s;
    return ZK_OK;
}
void zk_shard4_end(zk_shard4* s) { delete s; }
size_t zk_shard4_len(const zk_shard4* s) { return s ? s->len : 0; }

// ev_dev_out (3 elements) = this rank's share of the three evaluations of the current tables; asynchronous
int zk_shard4_eval(zk_shard4* s, void* ev_dev_out) {
    if (!s || !ev_dev_out || s->len < 2) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    int g;
    {
        ProfScope ps(c, ZK_K_EVAL3, (uint64_t)s->len * 32 * 4);
        g = zkk::launch_eval3(s->cur, s->len / 2, c->partials, c->stream);
    }
    zkk::launch_reduce_partials(c->partials, g, 3, ev_dev_out, c->stream);
    HIPCHK(c, hipGetLastError());
    return ZK_OK;
}

// fold the four tables at r (prover.rs:131-139).  While more than one entry per table remains, out_dev
// (3 elements) receives the next round's evaluation shares; when the fold leaves ONE entry per table,
// out_dev (4 elements) receives those entries instead (the rank's row of the tail tables).
int zk_shard4_fold(zk_shard4* s, const uint64_t r[4], void* out_dev) {
    if (!s || !r || !out_dev || s->len < 2) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    const size_t outlen = s->len / 2;
    if ((rc = ensure_work(c, s->pass & 1, 4 * outlen * kElem))) return rc;
    void* dst[4];
    for (int t = 0; t < 4; ++t) dst[t] = (uint8_t*)c->work[s->pass & 1] + (size_t)t * outlen * kElem;
    if (outlen >= 2) {
        int g;
        {
            ProfScope ps(c, ZK_K_FOLD4_EVAL3, (uint64_t)s->len * 48 * 4);
            g = zkk::launch_fold4_eval3(s->cur, dst, s->len / 4, r, c->partials, c->stream);
        }
        zkk::launch_reduce_partials(c->partials, g, 3, out_dev, c->stream);
    } else {
        zkk::launch_fold4(s->cur, dst, outlen, r, c->stream);
        HIPCHK(c, hipMemcpyAsync(out_dev, c->work[s->pass & 1], 4 * kElem, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipGetLastError());
    for (int t = 0; t < 4; ++t) s->cur[t] = dst[t];
    s->len = outlen;
    ++s->pass;
    return ZK_OK;
}

// The same branch two rounds per exchange: every rank's share of the nine sums (linear in the entries, so the
// shares add), the host part of both rounds, one 4-to-1 fold.
int zk_sumcheck_round_pair_from_sum9(zk_transcript* t, const uint64_t* sums, int world, uint64_t coeffs_out[24], uint64_t challenges_out[8]) {
    if (!t || !sums || !coeffs_out || !challenges_out || world < 1) return ZK_E_ARG;
    Fq S[9];
    for (int i = 0; i < 9; ++i) S[i] = zkh::zero();
    for (int q = 0; q < world; ++q)
        for (int i = 0; i < 9; ++i) S[i] = zkh::add(S[i], zkh::load(sums + 4 * (9 * (size_t)q + i)));
    host_round_pair_from_sum9(t, S, coeffs_out, challenges_out);
    return ZK_OK;
}

int zk_shard4_sum9(zk_shard4* s, void* sums_dev_out) {
    if (!s || !sums_dev_out || s->len < 4) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    {
        ProfScope ps(c, ZK_K_SUM9, (uint64_t)s->len * 32 * 4);
        zkk::launch_sum9(s->cur, s->len / 4, c->two24, c->deg2_lazy_from_log, c->partials, sums_dev_out, c->stream, c->gates);
    }
    HIPCHK(c, hipGetLastError());
    return ZK_OK;
}

// bind the two top variables of the four tables at r2[0], r2[1]; sums_dev_out (9 elements, may be NULL) receives
// the nine sums of the result when it keeps at least 4 entries per table
int zk_shard4_fold2(zk_shard4* s, const uint64_t r2[8], void* sums_dev_out) {
    if (!s || !r2 || s->len < 4) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    const size_t outlen = s->len / 4;
    if ((rc = ensure_work(c, s->pass & 1, 4 * outlen * kElem))) return rc;
    void* dst[4];
    for (int t = 0; t < 4; ++t) dst[t] = (uint8_t*)c->work[s->pass & 1] + (size_t)t * outlen * kElem;
    uint32_t w28[4 * 10];
    eq_weights28(r2, 2, w28);
    {
        ProfScope ps(c, ZK_K_FOLD2X4_SUM9, (uint64_t)s->len * 32 * 4 + (uint64_t)outlen * 32 * 4);
        zkk::launch_fold2_tables(s->cur, dst, outlen, w28, c->wide_w, c->partials, c->stream, c->gates);
    }
    if (sums_dev_out && outlen >= 4) {
        ProfScope ps(c, ZK_K_SUM9, (uint64_t)outlen * 32 * 4);
        zkk::launch_sum9(dst, outlen / 4, c->two24, c->deg2_lazy_from_log, c->partials, sums_dev_out, c->stream, c->gates);
    }
    HIPCHK(c, hipGetLastError());
    for (int t = 0; t < 4; ++t) s->cur[t] = dst[t];
    s->len = outlen;
    ++s->pass;
    return ZK_OK;
}

// ---- three rounds per step (the 8 x 8 cross sums of both terms) -----------------------------------
int zk_shard4_can_triple(const zk_shard4* s) { return s && deg2_triple_ok(s->ctx, s->len) ? 1 : 0; }

int zk_shard4_cross64(zk_shard4* s, void* cross_dev_out) {
    if (!s || !cross_dev_out || s->len < 512 || !zkk::cross64_fits(s->len / 8)) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    {
        ProfScope ps(c, ZK_K_SUM9, (uint64_t)s->len * 32 * 4);
        zkk::launch_cross64(s->cur, s->len / 8, c->gates.deg2_triple_ring, c->partials, cross_raw_for(c, s->len / 8), cross_dev_out, c->stream);
    }
    HIPCHK(c, hipGetLastError());
    return ZK_OK;
}

int zk_sumcheck_rounds_from_cross64(zk_transcript* t, const uint64_t* cross, int world, uint64_t coeffs_out[36], uint64_t challenges_out[12]) {
    if (!t || !cross || !coeffs_out || !challenges_out || world < 1) return ZK_E_ARG;
    host_rounds_from_cross64(t, cross, world, coeffs_out, challenges_out);
    return ZK_OK;
}

// bind the three top variables of the four tables at r3[0..3); next_dev_out (may be NULL) receives the sums the next step needs
// and *next_rounds_out says which: 3 = the 2 x 64 cross sums (128 elements), 2 = the nine sums, 0 = none (tables too short / NULL)
int zk_shard4_fold3(zk_shard4* s, const uint64_t r3[12], void* next_dev_out, int* next_rounds_out) {
    if (!s || !r3 || s->len < 512 || (s->len / 8) % 64 != 0) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    const size_t outlen = s->len / 8;
    if ((rc = ensure_work(c, s->pass & 1, 4 * outlen * kElem))) return rc;
    void* dst[4];
    for (int t = 0; t < 4; ++t) dst[t] = (uint8_t*)c->work[s->pass & 1] + (size_t)t * outlen * kElem;
    int next = 0;
    if (next_dev_out && c->gates.deg2_fold3_fused && deg2_triple_ok(c, outlen)) {
        ProfScope ps(c, ZK_K_FOLD2X4_SUM9, (uint64_t)s->len * 32 * 4 + (uint64_t)outlen * 32 * 4);
        zkk::launch_fold3_cross64(s->cur, dst, outlen, r3, c->two24, c->wide_w, c->gates.deg2_fold3_fused, c->partials, cross_raw_for(c, outlen / 8), next_dev_out, c->stream, c->gates);
        next = 3;
    } else {
        {
            ProfScope ps(c, ZK_K_FOLD2X4_SUM9, (uint64_t)s->len * 32 * 4 + (uint64_t)outlen * 32 * 4);
            zkk::launch_fold3_tables(s->cur, dst, outlen, r3, c->two24, c->wide_w, c->partials, c->stream, c->gates);
        }
        if (next_dev_out && deg2_triple_ok(c, outlen)) {
            ProfScope ps(c, ZK_K_SUM9, (uint64_t)outlen * 32 * 4);
            zkk::launch_cross64(dst, outlen / 8, c->gates.deg2_triple_ring, c->partials, cross_raw_for(c, outlen / 8), next_dev_out, c->stream);
            next = 3;
        } else if (next_dev_out && outlen >= 4) {
            ProfScope ps(c, ZK_K_SUM9, (uint64_t)outlen * 32 * 4);
            zkk::launch_sum9(dst, outlen / 4, c->two24, c->deg2_lazy_from_log, c->partials, next_dev_out, c->stream, c->gates);
            next = 2;
        }
    }
    HIPCHK(c, hipGetLastError());
    for (int t = 0; t < 4; ++t) s->cur[t] = dst[t];
    s->len = outlen;
    ++s->pass;
    if (next_rounds_out) *next_rounds_out = next;
    return ZK_OK;
}

// dst_dev (4 x len elements, table after table) = the four current local tables; asynchronous
int zk_shard4_copy_tables(zk_shard4* s, void* dst_dev) {
    if (!s || !dst_dev) return ZK_E_ARG;
    zk_ctx* c = s->ctx;
    int rc = set_device(c);
    if (rc) return rc;
    for (int t = 0; t < 4; ++t)
        HIPCHK(c, hipMemcpyAsync((uint8_t*)dst_dev + (size_t)t * s->len * kElem, s->cur[t], s->len * kElem, hipMemcpyDeviceToDevice, c->stream));
    return ZK_OK;
}

// ---- verifier ------------------------------------------------------------------------------------
int zk_sumcheck_partial_verify(const uint64_t claim[4], const uint64_t* coeffs, size_t n_rounds, size_t ncoef,
                               zk_transcript* tr, int* ok_out, uint64_t final_claim_out[4], uint64_t* challenges_out) {
    if (!claim || !tr || !ok_out || (n_rounds && !coeffs) || ncoef == 0 || ncoef > 16) return ZK_E_ARG;
    Fq cur = zkh::load(claim); // verifier.rs:34
    int ok = 1;
    const Fq one = zkh::one();
    for (size_t k = 0; k < n_rounds; ++k) {
        Fq g[16];
        for (size_t i = 0; i < ncoef; ++i) g[i] = zkh::load(coeffs + 4 * (k * ncoef + i));
        Fq s = zkh::add(zkh::poly_eval(g, ncoef, zkh::zero()), zkh::poly_eval(g, ncoef, one)); // g(0)+g(1), :41
        if (s != cur) { ok = 0; break; }
        tr->append_fq(cur); // :47-50
        for (size_t i = 0; i < ncoef; ++i) tr->append_fq(g[i]);
        Fq r = tr->sample(); // :52
        if (challenges_out) zkh::store(challenges_out + 4 * k, r);
        cur = zkh::poly_eval(g, ncoef, r); // :58
    }
    *ok_out = ok;
    if (final_claim_out) zkh::store(final_claim_out, cur);
    return ZK_OK;
}

int zk_sumcheck_verify_mle(zk_ctx* c, const zk_mle* poly, const uint64_t claim[4], const uint64_t* coeffs, size_t n_rounds,
                           int* ok_out) {
    if (!c || !poly || !claim || !ok_out || poly->ctx != c) return ZK_E_ARG;
    *ok_out = 0;
    int rc = set_device(c);
    if (rc) return rc;
    zk_transcript tr;
    rc = stream_le32(c, poly->d, poly->len, [&](const uint8_t* p, size_t nbytes, size_t) { tr.append(p, nbytes); }); // verifier.rs:71
    if (rc) return rc;
    uint64_t s[4];
    if ((rc = zk_mle_evaluation_sum(poly, s))) return rc;
    if (zkh::load(s) != zkh::load(claim)) return ZK_OK; // :74-76
    std::vector<uint64_t> ch(4 * (n_rounds ? n_rounds : 1));
    uint64_t final_claim[4];
    int ok = 0;
    if ((rc = zk_sumcheck_partial_verify(claim, coeffs, n_rounds, 2, &tr, &ok, final_claim, ch.data()))) return rc;
    if (!ok || n_rounds == 0) return ZK_OK; // :81-91
    // perform_oracle_check :15-26
    if (n_rounds != ilog2(poly->len)) return ZK_E_POINTS_LEN; // evaluate() panics on a length mismatch
    std::vector<uint8_t> has(n_rounds, 1);
    zk_mle* e = nullptr;
    if ((rc = zk_mle_evaluate(poly, has.data(), ch.data(), n_rounds, &e))) return rc;
    uint64_t v[4];
    rc = zk_mle_download_mont(e, v);
    zk_mle_free(e);
    if (rc) return rc;
    *ok_out = zkh::load(v) == zkh::load(final_claim);
    return ZK_OK;
}

// ---- measurement ----------------------------------------------------------------------------------
int zk_ctx_set_fold_batch(zk_ctx* c, int rounds_per_pass, int variant) {
    if (!c || rounds_per_pass < 1 || rounds_per_pass > zkk::kMaxFoldBatch || variant < 0 || variant > 1) return ZK_E_ARG;
    c->fold_batch = rounds_per_pass;
    c->lazy = variant == 0;
    return ZK_OK;
}
int zk_ctx_set_wide_fold(zk_ctx* c, int rounds, int min_log_out, int lds_weights_below_log) {
    if (!c || (rounds != 0 && (rounds <= zkk::kMaxLazyBatch || rounds > zkk::kMaxFoldBatch)) || min_log_out < 0 || min_log_out > 40 ||
        lds_weights_below_log < 0 || lds_weights_below_log > 64)
        return ZK_E_ARG;
    c->wide_batch = rounds;
    c->wide_min_log_out = min_log_out;
    c->wide_lds_below_log = lds_weights_below_log;
    return ZK_OK;
}
int zk_ctx_set_tail_fold(zk_ctx* c, int max_log_table, int out_below_log) {
    if (!c || max_log_table < 0 || max_log_table > 40 || out_below_log < 0 || out_below_log > 40) return ZK_E_ARG;
    c->tail_max_log = max_log_table;
    c->tail_out_below_log = out_below_log;
    return ZK_OK;
}
int zk_ctx_set_deg2_lazy_sums(zk_ctx* c, int from_log) {
    if (!c || from_log > 62) return ZK_E_ARG;
    c->deg2_lazy_from_log = from_log;
    return ZK_OK;
}
int zk_ctx_set_deg2_mfma_sums(zk_ctx* c, int from_log) {
    if (!c || from_log < 0 || from_log > 64) return ZK_E_ARG;
    c->gates.sum9_mfma_from_log = from_log;
    return ZK_OK;
}
int zk_ctx_set_deg2_small_triple(zk_ctx* c, int below_log) {
    if (!c || below_log < 0 || below_log > 62) return ZK_E_ARG;
    c->gates.deg2_sum27_below_log = below_log;
    return ZK_OK;
}
int zk_ctx_set_deg2_first_triple(zk_ctx* c, int below_log, int linear_below_log) {
    if (!c || below_log < 0 || below_log > 62 || linear_below_log < 0 || linear_below_log > 62) return ZK_E_ARG;
    c->gates.deg2_sum27_first_below_log = below_log;
    c->gates.deg2_pwq_first_below_log = linear_below_log;
    return ZK_OK;
}
int zk_ctx_set_deg2_fused(zk_ctx* c, int below_log) {
    if (!c || below_log < 0 || below_log > 64) return ZK_E_ARG;
    c->gates.deg2_fused_below_log = below_log;
    return ZK_OK;
}
int zk_ctx_set_deg2_triple(zk_ctx* c, int from_log, int ring) {
    if (!c || from_log < 0 || from_log > 64 || (ring < 3 || ring > 5)) return ZK_E_ARG;
    c->gates.deg2_triple_from_log = from_log;
    c->gates.deg2_triple_ring = ring;
    return ZK_OK;
}
int zk_ctx_set_deg2_fold3_fused(zk_ctx* c, int dma_slots) {
    if (!c || (dma_slots != 0 && dma_slots != 8 && (dma_slots < 3 || dma_slots > 5) && (dma_slots < 12 || dma_slots > 14))) return ZK_E_ARG;
    c->gates.deg2_fold3_fused = dma_slots;
    return ZK_OK;
}
int zk_ctx_set_deg2_rounds_per_pass(zk_ctx* c, int rounds) {
    if (!c || (rounds != 1 && rounds != 2)) return ZK_E_ARG;
    c->deg2_pairs = rounds == 2;
    return ZK_OK;
}
int zk_ctx_set_validate(zk_ctx* c, int on) {
    if (!c) return ZK_E_ARG;
    c->validate_uploads = on != 0;
    return ZK_OK;
}
int zk_ctx_set_gkr_strict_claims(zk_ctx* c, int on) {
    if (!c) return ZK_E_ARG;
    c->gkr_strict_claims = on != 0;
    return ZK_OK;
}
int zk_ctx_set_dma_gates(zk_ctx* c, int sums_from_log, int wide_stream_from_log, int fold2_from_log) {
    if (!c || sums_from_log < 0 || sums_from_log > 64 || wide_stream_from_log < 0 || wide_stream_from_log > 64 || fold2_from_log < 0 ||
        fold2_from_log > 64)
        return ZK_E_ARG;
    c->gates.sums_from_log = sums_from_log;
    c->gates.wide_stream_from_log = wide_stream_from_log;
    c->gates.fold2_from_log = fold2_from_log;
    return ZK_OK;
}
int zk_ctx_set_deg2_raw_cross(zk_ctx* c, int on) {
    if (!c) return ZK_E_ARG;
    c->cross_raw_on = on != 0;
    return ZK_OK;
}
int zk_ctx_set_wide_mfma(zk_ctx* c, int from_log, int workgroups_per_cu, int depth) {
    if (!c || from_log < 0 || from_log > 64 || workgroups_per_cu < 1 || workgroups_per_cu > 8 || depth < 2 || depth > 4) return ZK_E_ARG;
    c->gates.wide_mfma_from_log = from_log;
    c->gates.wide_mfma_wg_per_cu = workgroups_per_cu;
    c->gates.wide_mfma_depth = depth;
    return ZK_OK;
}
unsigned zk_ctx_forms_used(zk_ctx* c, int reset) {
    (void)c; // the record is per host thread (a context is used from one thread at a time)
    const unsigned f = zkk::g_forms;
    if (reset) zkk::g_forms = 0;
    return f;
}
int zk_prof_enable(zk_ctx* c, int on) {
    if (!c) return ZK_E_ARG;
    c->prof_on = on != 0;
    return ZK_OK;
}
int zk_prof_collect(zk_ctx* c) { // accounts the launches recorded since the last collect; synchronizes the stream
    if (!c) return ZK_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZK_OK;
}
int zk_prof_reset(zk_ctx* c) {
    if (!c) return ZK_E_ARG;
    for (int k = 0; k < ZK_K_COUNT; ++k) { c->prof_launches[k] = 0; c->prof_ms[k] = 0; c->prof_bytes[k] = 0; }
    return ZK_OK;
}
int zk_prof_read(zk_ctx* c, int kernel, uint64_t* launches, double* ms, uint64_t* bytes) {
    if (!c || kernel < 0 || kernel >= ZK_K_COUNT) return ZK_E_ARG;
    if (launches) *launches = c->prof_launches[kernel];
    if (ms) *ms = c->prof_ms[kernel];
    if (bytes) *bytes = c->prof_bytes[kernel];
    return ZK_OK;
}
int zk_bench_mad(zk_ctx* c, int blocks, int iters, double* ms_out) {
    if (!c || blocks <= 0 || iters <= 0 || !ms_out) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    void* buf = nullptr;
    HIPCHK(c, hipMalloc(&buf, (size_t)blocks * 256 * sizeof(uint32_t)));
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    zkk::launch_mad_probe(buf, blocks, 2, c->stream); // warm-up
    HIPCHK(c, hipEventRecord(a, c->stream));
    zkk::launch_mad_probe(buf, blocks, iters, c->stream);
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, a, b));
    *ms_out = ms;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIPCHK(c, hipFree(buf));
    return ZK_OK;
}
int zk_bench_modmul(zk_ctx* c, int blocks, int iters, double* ms_out) {
    if (!c || blocks <= 0 || iters <= 0 || !ms_out) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    void* buf = nullptr;
    HIPCHK(c, hipMalloc(&buf, (size_t)blocks * 256 * kElem));
    zkk::launch_generate(buf, 42, 0, 1, (size_t)blocks * 256, c->stream);
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    zkk::launch_modmul_probe(buf, blocks, 4, c->stream); // warm-up
    HIPCHK(c, hipEventRecord(a, c->stream));
    zkk::launch_modmul_probe(buf, blocks, iters, c->stream);
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, a, b));
    *ms_out = ms;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIPCHK(c, hipFree(buf));
    return ZK_OK;
}

} // extern "C"
