/*
 * oracle/api.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * Flat, ctypes-friendly entry points over the restatement in fq.c / keccak.c / poly.c /
 * sumcheck.c / gkr.c.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library.  Field elements cross this boundary as 4 x u64 Montgomery
 * limbs (ark-ff's in-memory form) unless a name says le32.
 */
#include "protocol.h"
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- field */
API void orc_fq_from_canonical(const uint64_t c[4], uint64_t out[4]) { fq_t r; fq_from_canonical(&r, c); memcpy(out, r.l, 32); }
API void orc_fq_to_canonical(const uint64_t a[4], uint64_t out[4]) { fq_t x; memcpy(x.l, a, 32); fq_to_canonical(out, &x); }
API void orc_fq_to_le32(const uint64_t a[4], uint8_t out[32]) { fq_t x; memcpy(x.l, a, 32); fq_to_le32(out, &x); }
API void orc_fq_from_u64(uint64_t v, uint64_t out[4]) { fq_t r; fq_from_u64(&r, v); memcpy(out, r.l, 32); }
API void orc_fq_from_le_bytes_mod_order(const uint8_t *b, size_t len, uint64_t out[4]) { fq_t r; fq_from_le_bytes_mod_order(&r, b, len); memcpy(out, r.l, 32); }
#define BINOP(name, fn) \
    API void name(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { \
        fq_t x, y, r; memcpy(x.l, a, 32); memcpy(y.l, b, 32); fn(&r, &x, &y); memcpy(out, r.l, 32); }
BINOP(orc_fq_add, fq_add)
BINOP(orc_fq_sub, fq_sub)
BINOP(orc_fq_mul, fq_mul)
BINOP(orc_fq_div, fq_div)
API void orc_fq_inv(const uint64_t a[4], uint64_t out[4]) { fq_t x, r; memcpy(x.l, a, 32); fq_inv(&r, &x); memcpy(out, r.l, 32); }
API void orc_fq_neg(const uint64_t a[4], uint64_t out[4]) { fq_t x, r; memcpy(x.l, a, 32); fq_neg(&r, &x); memcpy(out, r.l, 32); }

/* ---------------------------------------------------------------- synthetic tables
 * entry j = four consecutive outputs of the splitmix64 stream whose state
 * starts at `seed` (output i, 1-based, is mix(seed + i*0x9E3779B97F4A7C15)), draws
 * 4j+1 .. 4j+4, assembled little-endian into 256 bits, reduced mod p, then converted to
 * Montgomery form.  Counter-based: entry j depends on (seed, j) only. */
static uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + i * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
API void orc_gen_table(uint64_t seed, uint64_t first, size_t count, uint64_t *out /* 4*count */) {
    for (size_t k = 0; k < count; ++k) {
        uint64_t j = first + k;
        uint8_t b[32];
        for (int w = 0; w < 4; ++w) {
            uint64_t v = splitmix64_at(seed, 4 * j + w + 1);
            for (int q = 0; q < 8; ++q) b[8 * w + q] = (uint8_t)(v >> (8 * q));
        }
        fq_t r;
        fq_from_le_bytes_mod_order(&r, b, 32);
        memcpy(out + 4 * k, r.l, 32);
    }
}

/* ---------------------------------------------------------------- keccak / transcript */
API void orc_keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
    keccak256_t k; keccak256_init(&k); keccak256_update(&k, data, len); keccak256_peek(&k, out);
}
API void *orc_transcript_new(void) { transcript_t *t = (transcript_t *)malloc(sizeof *t); if (t) transcript_new(t); return t; }
API void orc_transcript_free(void *t) { free(t); }
API void orc_transcript_append(void *t, const uint8_t *d, size_t len) { transcript_append((transcript_t *)t, d, len); }
API void orc_transcript_sample(void *t, uint64_t out[4]) { fq_t r; transcript_sample_challenge((transcript_t *)t, &r); memcpy(out, r.l, 32); }
API void orc_transcript_peek_digest(void *t, uint8_t out[32]) { keccak256_peek(&((transcript_t *)t)->hasher, out); }

/* ---------------------------------------------------------------- MLE */
static opt_fq_t *mk_pts(const uint8_t *has, const uint64_t *vals, size_t n) {
    opt_fq_t *p = (opt_fq_t *)calloc(n ? n : 1, sizeof(opt_fq_t));
    for (size_t i = 0; p && i < n; ++i) { p[i].has = has ? has[i] : 1; memcpy(p[i].v.l, vals + 4 * i, 32); }
    return p;
}
API int orc_mle_check(size_t len) { mle_t m; fq_t z; fq_zero(&z); if (len == 0) return ORC_E_NOT_POW2; fq_t *t = (fq_t *)calloc(len, sizeof(fq_t)); int rc = mle_new(&m, t, len); free(t); if (!rc) mle_free(&m); return rc; }
API int orc_mle_partially_evaluate(const uint64_t *table, size_t len, size_t var, const uint64_t r[4], uint64_t *out /* 4*len/2 */) {
    mle_t m, o; int rc = mle_new(&m, (const fq_t *)table, len); if (rc) return rc;
    fq_t rr; memcpy(rr.l, r, 32);
    rc = mle_partially_evaluate(&o, &m, var, &rr);
    if (!rc) { memcpy(out, o.ev, o.len * 32); mle_free(&o); }
    mle_free(&m); return rc;
}
/* out must hold 4*len limbs; *out_len receives the resulting length */
API int orc_mle_evaluate(const uint64_t *table, size_t len, const uint8_t *has, const uint64_t *vals, size_t npts, uint64_t *out, size_t *out_len) {
    mle_t m, o; int rc = mle_new(&m, (const fq_t *)table, len); if (rc) return rc;
    opt_fq_t *p = mk_pts(has, vals, npts);
    rc = mle_evaluate(&o, &m, p, npts);
    if (!rc) { memcpy(out, o.ev, o.len * 32); *out_len = o.len; mle_free(&o); }
    free(p); mle_free(&m); return rc;
}
API int orc_mle_to_bytes(const uint64_t *table, size_t len, uint8_t *out) { mle_t m = {(fq_t *)table, len}; mle_to_bytes(out, &m); return 0; }
API int orc_mle_sum(const uint64_t *table, size_t len, uint64_t out[4]) { mle_t m = {(fq_t *)table, len}; fq_t s; mle_evaluation_sum(&s, &m); memcpy(out, s.l, 32); return 0; }
API int orc_mle_scalar_mul(const uint64_t *table, size_t len, const uint64_t s[4], uint64_t *out) {
    mle_t m = {(fq_t *)table, len}, o; fq_t ss; memcpy(ss.l, s, 32);
    int rc = mle_scalar_mul(&o, &m, &ss); if (!rc) { memcpy(out, o.ev, o.len * 32); mle_free(&o); } return rc;
}
API int orc_mle_minus(const uint64_t *table, size_t len, const uint64_t v[4], uint64_t *out) {
    mle_t m = {(fq_t *)table, len}, o; fq_t vv; memcpy(vv.l, v, 32);
    int rc = mle_minus(&o, &m, &vv); if (!rc) { memcpy(out, o.ev, o.len * 32); mle_free(&o); } return rc;
}
/* quotient_out, remainder_out: 4 * len/2 limbs each */
API int orc_mle_quotient_remainder(const uint64_t *table, size_t len, const uint64_t divisor[4], size_t var, uint64_t *quotient_out,
                                   uint64_t *remainder_out) {
    mle_t m, r; int rc = mle_new(&m, (const fq_t *)table, len); if (rc) return rc;
    fq_t d; memcpy(d.l, divisor, 32);
    rc = mle_compute_quotient_remainder((fq_t *)quotient_out, &r, &m, &d, var);
    if (!rc) { memcpy(remainder_out, r.ev, r.len * 32); mle_free(&r); }
    mle_free(&m); return rc;
}
API int orc_mle_add(const uint64_t *a, size_t la, const uint64_t *b, size_t lb, uint64_t *out) {
    mle_t x = {(fq_t *)a, la}, y = {(fq_t *)b, lb}, o;
    int rc = mle_add(&o, &x, &y); if (!rc) { memcpy(out, o.ev, o.len * 32); mle_free(&o); } return rc;
}
API int orc_mle_w_add(const uint64_t *wb, size_t lb, const uint64_t *wc, size_t lc, uint64_t *out) {
    mle_t x = {(fq_t *)wb, lb}, y = {(fq_t *)wc, lc}, o;
    int rc = mle_w_add(&o, &x, &y); if (!rc) { memcpy(out, o.ev, o.len * 32); mle_free(&o); } return rc;
}
API int orc_mle_w_mul(const uint64_t *self, size_t ls, const uint64_t *other, size_t lo, uint64_t *out) {
    mle_t x = {(fq_t *)self, ls}, y = {(fq_t *)other, lo}, o;
    int rc = mle_w_mul(&o, &x, &y); if (!rc) { memcpy(out, o.ev, o.len * 32); mle_free(&o); } return rc;
}

/* ---------------------------------------------------------------- univariate */
API int orc_uni_interpolate(const uint64_t *xs, const uint64_t *ys, size_t n, uint64_t *out) {
    unipoly_t u; int rc = uni_interpolate(&u, (const fq_t *)xs, (const fq_t *)ys, n);
    if (!rc) { memcpy(out, u.c, u.n * 32); uni_free(&u); } return rc;
}
API void orc_uni_evaluate(const uint64_t *c, size_t n, const uint64_t x[4], uint64_t out[4]) {
    unipoly_t u = {(fq_t *)c, n}; fq_t xx, r; memcpy(xx.l, x, 32); uni_evaluate(&r, &u, &xx); memcpy(out, r.l, 32);
}
API int orc_uni_mul(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out) {
    unipoly_t x = {(fq_t *)a, na}, y = {(fq_t *)b, nb}, o; int rc = uni_mul(&o, &x, &y);
    if (!rc) { memcpy(out, o.c, o.n * 32); uni_free(&o); } return rc;
}

/* ---------------------------------------------------------------- product / sum polynomial helpers */
/* tables: n_terms * n_factors tables of `len` entries each, term-major */
static int build_sum(sumpoly_t *f, const uint64_t *const *tables, size_t n_terms, size_t n_factors, const size_t *lens) {
    prodpoly_t *prods = (prodpoly_t *)calloc(n_terms ? n_terms : 1, sizeof(prodpoly_t));
    int rc = ORC_OK; size_t made = 0;
    for (size_t p = 0; p < n_terms && !rc; ++p) {
        mle_t *ms = (mle_t *)calloc(n_factors ? n_factors : 1, sizeof(mle_t)); size_t mm = 0;
        for (size_t k = 0; k < n_factors && !rc; ++k) { rc = mle_new(&ms[k], (const fq_t *)tables[p * n_factors + k], lens[p * n_factors + k]); if (!rc) ++mm; }
        if (!rc) rc = prod_new(&prods[p], ms, n_factors);
        if (rc) for (size_t k = 0; k < mm; ++k) mle_free(&ms[k]); else ++made;
        free(ms);
    }
    if (!rc) rc = sum_new(f, prods, n_terms);
    if (rc) for (size_t p = 0; p < made; ++p) prod_free(&prods[p]);
    free(prods);
    return rc;
}
API int orc_sum_reduce(const uint64_t *const *tables, size_t n_terms, size_t n_factors, const size_t *lens, uint64_t *out) {
    sumpoly_t f; int rc = build_sum(&f, tables, n_terms, n_factors, lens); if (rc) return rc;
    fq_t *res; size_t l; rc = sum_reduce(&res, &l, &f);
    if (!rc) { memcpy(out, res, l * 32); free(res); }
    sum_free(&f); return rc;
}
API int orc_sum_evaluate(const uint64_t *const *tables, size_t n_terms, size_t n_factors, const size_t *lens, const uint64_t *pts, size_t npts, uint64_t out[4]) {
    sumpoly_t f; int rc = build_sum(&f, tables, n_terms, n_factors, lens); if (rc) return rc;
    opt_fq_t *p = mk_pts(NULL, pts, npts); fq_t r;
    rc = sum_evaluate(&r, &f, p, npts); if (!rc) memcpy(out, r.l, 32);
    free(p); sum_free(&f); return rc;
}

/* ---------------------------------------------------------------- sumcheck */
/* generate_sumcheck_proof (prover.rs:158-173).  coeffs_out: n_vars * 2 elements; challenges_out
 * (optional, may be NULL): n_vars elements -- the reference discards them, they are returned
 * here only so tests can compare them. */
API int orc_sumcheck_prove(const uint64_t *table, size_t len, uint64_t claim_out[4], uint64_t *coeffs_out, uint64_t *challenges_out) {
    mle_t m; int rc = mle_new(&m, (const fq_t *)table, len); if (rc) return rc;
    /* same sequence as sumcheck_generate_proof, but keeping the challenges */
    transcript_t tr; transcript_new(&tr);
    uint8_t *bytes = (uint8_t *)malloc(32 * len);
    mle_to_bytes(bytes, &m); transcript_append(&tr, bytes, 32 * len); free(bytes);
    mle_t copy; rc = mle_clone(&copy, &m);
    if (rc) { mle_free(&m); return rc; }
    unipoly_t *rp; fq_t *ch; size_t nr;
    rc = sumcheck_generate_round_polys(NULL, &copy, &tr, &rp, &ch, &nr);
    mle_free(&copy);
    if (!rc) {
        fq_t s; mle_evaluation_sum(&s, &m); memcpy(claim_out, s.l, 32);
        for (size_t k = 0; k < nr; ++k) { memcpy(coeffs_out + 8 * k, rp[k].c, 64); uni_free(&rp[k]); }
        if (challenges_out) memcpy(challenges_out, ch, nr * 32);
        free(rp); free(ch);
    }
    mle_free(&m); return rc;
}
/* for timing the CPU baseline: absorb != 0 is the unmodified reference entry point
 * generate_sumcheck_proof (prover.rs:158-173); absorb == 0 runs only generate_round_polys
 * (prover.rs:45-155) on a fresh transcript -- the round loop the GPU figure is quoted on. */
API int orc_sumcheck_prove_timed(const uint64_t *table, size_t len, int absorb, uint64_t claim_out[4]) {
    mle_t m = {(fq_t *)table, len};
    if (absorb) {
        sumcheck_proof_t p;
        int rc = sumcheck_generate_proof(&p, &m); if (rc) return rc;
        memcpy(claim_out, p.initial_claim_sum.l, 32); sumcheck_proof_free(&p); return 0;
    }
    transcript_t tr; transcript_new(&tr);
    unipoly_t *rp; fq_t *ch; size_t nr;
    int rc = sumcheck_generate_round_polys(NULL, &m, &tr, &rp, &ch, &nr); if (rc) return rc;
    if (nr) { fq_t s; fq_add(&s, &rp[0].c[0], &rp[0].c[0]); fq_add(&s, &s, &rp[0].c[1]); memcpy(claim_out, s.l, 32); }
    for (size_t k = 0; k < nr; ++k) uni_free(&rp[k]);
    free(rp); free(ch); return 0;
}
static int mk_proof(sumcheck_proof_t *p, const uint64_t claim[4], const uint64_t *coeffs, size_t n_rounds, size_t ncoef) {
    memcpy(p->initial_claim_sum.l, claim, 32);
    p->n_rounds = n_rounds;
    p->round_polys = (unipoly_t *)calloc(n_rounds ? n_rounds : 1, sizeof(unipoly_t));
    for (size_t k = 0; k < n_rounds; ++k) uni_new(&p->round_polys[k], (const fq_t *)(coeffs + 4 * ncoef * k), ncoef);
    return 0;
}
API int orc_sumcheck_verify(const uint64_t *table, size_t len, const uint64_t claim[4], const uint64_t *coeffs, size_t n_rounds, size_t ncoef) {
    mle_t m; if (mle_new(&m, (const fq_t *)table, len)) return -1;
    sumcheck_proof_t p; mk_proof(&p, claim, coeffs, n_rounds, ncoef);
    int ok = sumcheck_verify_proof(&m, &p);
    sumcheck_proof_free(&p); mle_free(&m); return ok;
}
/* generate_proof_for_partial_verify (prover.rs:175-190): coeffs_out n_vars*3, challenges_out n_vars */
API int orc_sumcheck_prove_sum(const uint64_t *const *tables, size_t n_terms, size_t n_factors, const size_t *lens,
                               const uint64_t claim[4], void *transcript, uint64_t *coeffs_out, uint64_t *challenges_out) {
    sumpoly_t f; int rc = build_sum(&f, tables, n_terms, n_factors, lens); if (rc) return rc;
    sumcheck_proof_t p; fq_t *pts; fq_t c; memcpy(c.l, claim, 32);
    rc = sumcheck_generate_proof_for_partial_verify(&p, &pts, &c, &f, (transcript_t *)transcript);
    if (!rc) {
        for (size_t k = 0; k < p.n_rounds; ++k) memcpy(coeffs_out + 12 * k, p.round_polys[k].c, 96);
        if (challenges_out) memcpy(challenges_out, pts, p.n_rounds * 32);
        free(pts); sumcheck_proof_free(&p);
    }
    sum_free(&f); return rc;
}
API int orc_sumcheck_partial_verify(const uint64_t claim[4], const uint64_t *coeffs, size_t n_rounds, size_t ncoef, void *transcript,
                                    uint64_t final_claim_out[4], uint64_t *challenges_out, size_t *n_challenges) {
    sumcheck_proof_t p; mk_proof(&p, claim, coeffs, n_rounds, ncoef);
    fq_t fc, *ch; size_t nch;
    int ok = sumcheck_partial_verify(&p, (transcript_t *)transcript, &fc, &ch, &nch);
    memcpy(final_claim_out, fc.l, 32);
    if (challenges_out) memcpy(challenges_out, ch, nch * 32);
    if (n_challenges) *n_challenges = nch;
    free(ch); sumcheck_proof_free(&p); return ok;
}

/* ---------------------------------------------------------------- circuit / GKR */
/* gates: per layer, triples (left, right, op) as u64; layers[0] is next to the inputs */
API void *orc_circuit_new(const uint64_t *gates, const size_t *layer_len, size_t n_layers) {
    circuit_t *c = (circuit_t *)calloc(1, sizeof *c);
    gate_t **ls = (gate_t **)calloc(n_layers ? n_layers : 1, sizeof(gate_t *));
    size_t off = 0;
    for (size_t i = 0; i < n_layers; ++i) {
        ls[i] = (gate_t *)calloc(layer_len[i] ? layer_len[i] : 1, sizeof(gate_t));
        for (size_t g = 0; g < layer_len[i]; ++g, ++off) {
            ls[i][g].left = (size_t)gates[3 * off]; ls[i][g].right = (size_t)gates[3 * off + 1]; ls[i][g].op = (int)gates[3 * off + 2];
        }
    }
    int rc = circuit_new(c, (const gate_t *const *)ls, layer_len, n_layers);
    for (size_t i = 0; i < n_layers; ++i) free(ls[i]);
    free(ls);
    if (rc) { circuit_free(c); free(c); return NULL; }
    return c;
}
API void orc_circuit_free(void *c) { if (c) { circuit_free((circuit_t *)c); free(c); } }
/* layer k of the result (k = 0 inputs .. n_layers) is written at out + 4*offsets; lens_out[k] = its length */
API int orc_circuit_evaluate(void *c, const uint64_t *inputs, size_t n_inputs, uint64_t *out, size_t out_cap_elems, size_t *lens_out) {
    circuit_t *cc = (circuit_t *)c; mle_t *ev;
    int rc = circuit_evaluate_at_input(cc, (const fq_t *)inputs, n_inputs, &ev); if (rc) return rc;
    size_t off = 0;
    for (size_t k = 0; k <= cc->n_layers; ++k) {
        if (off + ev[k].len <= out_cap_elems) memcpy(out + 4 * off, ev[k].ev, ev[k].len * 32);
        lens_out[k] = ev[k].len; off += ev[k].len; mle_free(&ev[k]);
    }
    free(ev); return off <= out_cap_elems ? 0 : ORC_E_ALLOC;
}
/* returns the table length (0 on error, *rc_out set); out may be NULL to query the length */
API size_t orc_circuit_gate_poly(void *c, size_t layer_idx, int op, uint64_t *out, size_t out_cap_elems, int *rc_out) {
    mle_t m; int rc = circuit_get_gate_poly(&m, (circuit_t *)c, layer_idx, op);
    if (rc_out) *rc_out = rc;
    if (rc) return 0;
    size_t n = m.len;
    if (out && n <= out_cap_elems) memcpy(out, m.ev, n * 32);
    mle_free(&m); return n;
}

API void *orc_gkr_prove(void *c, void *transcript, const uint64_t *inputs, size_t n_inputs, int *rc_out) {
    gkr_proof_t *p = (gkr_proof_t *)calloc(1, sizeof *p);
    int rc = gkr_generate_proof(p, (circuit_t *)c, (transcript_t *)transcript, (const fq_t *)inputs, n_inputs);
    if (rc_out) *rc_out = rc;
    if (rc) { free(p); return NULL; }
    return p;
}
API void orc_gkr_proof_free(void *p) { if (p) { gkr_proof_free((gkr_proof_t *)p); free(p); } }
API int orc_gkr_verify(const uint64_t *inputs, size_t n_inputs, void *c, void *transcript, void *proof) {
    return gkr_verify_proof((const fq_t *)inputs, n_inputs, (circuit_t *)c, (transcript_t *)transcript, (gkr_proof_t *)proof);
}
/* Proof in struct order (gkr_protocol.rs:8-12) as LE32 per field element:
 * output_poly entries | (wb,wc) pairs | per layer: initial_claim_sum, then per round the 3 coefficients.
 * Returns the byte count; writes only if cap is large enough.  shape_out (optional):
 * [output_len, n_w_evals, n_proofs, rounds_0, rounds_1, ...]. */
API size_t orc_gkr_proof_serialize(void *proof, uint8_t *out, size_t cap, size_t *shape_out, size_t shape_cap) {
    gkr_proof_t *p = (gkr_proof_t *)proof;
    size_t n = p->output_poly.len + 2 * p->n_w_evals;
    for (size_t i = 0; i < p->n_proofs; ++i) { n += 1; for (size_t k = 0; k < p->sumcheck_proofs[i].n_rounds; ++k) n += p->sumcheck_proofs[i].round_polys[k].n; }
    if (shape_out && shape_cap >= 3 + p->n_proofs) {
        shape_out[0] = p->output_poly.len; shape_out[1] = p->n_w_evals; shape_out[2] = p->n_proofs;
        for (size_t i = 0; i < p->n_proofs; ++i) shape_out[3 + i] = p->sumcheck_proofs[i].n_rounds;
    }
    if (out && cap >= 32 * n) {
        uint8_t *w = out;
        for (size_t i = 0; i < p->output_poly.len; ++i, w += 32) fq_to_le32(w, &p->output_poly.ev[i]);
        for (size_t i = 0; i < 2 * p->n_w_evals; ++i, w += 32) fq_to_le32(w, &p->w_evals[i]);
        for (size_t i = 0; i < p->n_proofs; ++i) {
            fq_to_le32(w, &p->sumcheck_proofs[i].initial_claim_sum); w += 32;
            for (size_t k = 0; k < p->sumcheck_proofs[i].n_rounds; ++k)
                for (size_t q = 0; q < p->sumcheck_proofs[i].round_polys[k].n; ++q, w += 32)
                    fq_to_le32(w, &p->sumcheck_proofs[i].round_polys[k].c[q]);
        }
    }
    return 32 * n;
}
/* tamper helper for reject tests: add 1 to the first coefficient of round `round` of layer `layer` */
API void orc_gkr_proof_tamper(void *proof, size_t layer, size_t round) {
    gkr_proof_t *p = (gkr_proof_t *)proof; fq_t one; fq_one(&one);
    fq_add(&p->sumcheck_proofs[layer].round_polys[round].c[0], &p->sumcheck_proofs[layer].round_polys[round].c[0], &one);
}
