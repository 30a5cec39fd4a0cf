/* oracle/poly.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). See poly.h for provenance. */
#include "poly.h"
#include <stdlib.h>
#include <string.h>

static int is_pow2(size_t n) { return n != 0 && (n & (n - 1)) == 0; }
static unsigned ilog2(size_t n) {
    unsigned k = 0;
    while (n > 1) { n >>= 1; ++k; }
    return k;
}

/* ------------------------------------------------------------------ MLE */

int mle_new(mle_t *out, const fq_t *pts, size_t len) {
    out->ev = NULL;
    out->len = 0;
    if (!is_pow2(len)) return ORC_E_NOT_POW2; /* evaluation_form.rs:23-27 */
    out->ev = (fq_t *)malloc(len * sizeof(fq_t));
    if (!out->ev) return ORC_E_ALLOC;
    memcpy(out->ev, pts, len * sizeof(fq_t)); /* to_vec(), :30 */
    out->len = len;
    return ORC_OK;
}

void mle_free(mle_t *m) {
    free(m->ev);
    m->ev = NULL;
    m->len = 0;
}

int mle_clone(mle_t *out, const mle_t *m) { return mle_new(out, m->ev, m->len); }

unsigned mle_num_vars(const mle_t *m) { return ilog2(m->len); }

typedef struct { size_t y1, y2; } idx_pair_t;

/* evaluation_form.rs:45-80 -- the pair list is materialised exactly like the reference
 * does (it is part of the reference's cost, and of the pairing order being pinned). */
static idx_pair_t *get_y1_y2_indexes(const mle_t *m, size_t var) {
    size_t half = m->len / 2;
    unsigned power = mle_num_vars(m) - 1 - (unsigned)var;
    size_t target = (size_t)1 << power;
    idx_pair_t *pairs = (idx_pair_t *)malloc((half ? half : 1) * sizeof(idx_pair_t));
    if (!pairs) return NULL;
    size_t y1 = 0;
    for (size_t k = 0; k < half; ++k) {
        size_t y2 = y1 | ((size_t)1 << power); /* :35-43 */
        pairs[k].y1 = y1;
        pairs[k].y2 = y2;
        y1 = ((y1 + 1) % target == 0) ? y2 + 1 : y1 + 1; /* :72-76 */
    }
    return pairs;
}

int mle_partially_evaluate(mle_t *out, const mle_t *m, size_t var, const fq_t *r) {
    if (var >= mle_num_vars(m)) return ORC_E_POINTS_LEN; /* Rust: subtract-with-overflow panic */
    size_t half = m->len / 2;
    idx_pair_t *pairs = get_y1_y2_indexes(m, var);
    fq_t *nv = (fq_t *)malloc((half ? half : 1) * sizeof(fq_t));
    if (!pairs || !nv) { free(pairs); free(nv); return ORC_E_ALLOC; }
    for (size_t k = 0; k < half; ++k) {
        const fq_t *y1 = &m->ev[pairs[k].y1], *y2 = &m->ev[pairs[k].y2];
        fq_t d, t;
        fq_sub(&d, y2, y1);
        fq_mul(&t, &d, r);
        fq_add(&nv[k], y1, &t); /* y1 + ((y2 - y1) * r), :184 */
    }
    int rc = mle_new(out, nv, half); /* Self::new(&vec) copies again, :191 */
    free(nv);
    free(pairs);
    return rc;
}

/* minus (evaluation_form.rs:260-272): F(x) - V, entry by entry */
int mle_minus(mle_t *out, const mle_t *m, const fq_t *other) {
    fq_t *nv = (fq_t *)malloc(m->len * sizeof(fq_t));
    if (!nv) return ORC_E_ALLOC;
    for (size_t i = 0; i < m->len; ++i) fq_sub(&nv[i], &m->ev[i], other); /* *val - *other, :267 */
    int rc = mle_new(out, nv, m->len);
    free(nv);
    return rc;
}

/* compute_quotient_remainder (evaluation_form.rs:280-301): remainder = partially_evaluate((variable_index, divisor)) (:285),
 * quotient[k] = y2 - y1 over the pairs of get_y1_y2_indexes(variable_index) in their order (:286-296).  quotient: len/2 elements. */
int mle_compute_quotient_remainder(fq_t *quotient, mle_t *remainder, const mle_t *m, const fq_t *divisor, size_t var) {
    if (var >= mle_num_vars(m)) return ORC_E_POINTS_LEN; /* Rust: subtract-with-overflow panic in get_y1_y2_indexes */
    idx_pair_t *pairs = get_y1_y2_indexes(m, var); /* :283 */
    if (!pairs) return ORC_E_ALLOC;
    int rc = mle_partially_evaluate(remainder, m, var, divisor); /* :285 */
    if (!rc)
        for (size_t k = 0; k < m->len / 2; ++k) fq_sub(&quotient[k], &m->ev[pairs[k].y2], &m->ev[pairs[k].y1]); /* y2 - y1, :294 */
    free(pairs);
    return rc;
}

int mle_evaluate(mle_t *out, const mle_t *m, const opt_fq_t *pts, size_t npts) {
    if (npts != mle_num_vars(m)) return ORC_E_POINTS_LEN; /* :197-199 */
    mle_t acc;
    int rc = mle_new(&acc, m->ev, m->len); /* :204 */
    if (rc) return rc;
    size_t done = 0;
    for (size_t idx = 0; idx < npts; ++idx) {
        if (pts[idx].has) {
            mle_t next;
            rc = mle_partially_evaluate(&next, &acc, idx - done, &pts[idx].v); /* :208 */
            mle_free(&acc);
            if (rc) return rc;
            acc = next;
            ++done;
        }
    }
    *out = acc;
    return ORC_OK;
}

void mle_to_bytes(uint8_t *out, const mle_t *m) {
    for (size_t i = 0; i < m->len; ++i) fq_to_le32(out + 32 * i, &m->ev[i]);
}

void mle_evaluation_sum(fq_t *out, const mle_t *m) {
    fq_t s;
    fq_zero(&s);
    for (size_t i = 0; i < m->len; ++i) fq_add(&s, &s, &m->ev[i]);
    *out = s;
}

int mle_scalar_mul(mle_t *out, const mle_t *m, const fq_t *s) {
    fq_t *nv = (fq_t *)malloc(m->len * sizeof(fq_t));
    if (!nv) return ORC_E_ALLOC;
    for (size_t i = 0; i < m->len; ++i) fq_mul(&nv[i], &m->ev[i], s);
    int rc = mle_new(out, nv, m->len);
    free(nv);
    return rc;
}

int mle_add(mle_t *out, const mle_t *a, const mle_t *b) {
    if (mle_num_vars(a) != mle_num_vars(b)) return ORC_E_LEN_MISMATCH; /* :244-246 */
    fq_t *nv = (fq_t *)calloc(b->len, sizeof(fq_t));
    if (!nv) return ORC_E_ALLOC;
    for (size_t i = 0; i < a->len; ++i) {
        fq_t t;
        fq_add(&t, &a->ev[i], &b->ev[i]);
        fq_add(&nv[i], &nv[i], &t); /* new_evals[idx] += a + b, :251-252 */
    }
    int rc = mle_new(out, nv, b->len);
    free(nv);
    return rc;
}

/* operate_w, evaluation_form.rs:82-112.  Note the reference's index quirk
 * idx_b = i / w_b_len (not / w_c_len): only right when both operands have equal length. */
static int operate_w(mle_t *out, const mle_t *wb, const mle_t *wc, int is_mul) {
    size_t lb = wb->len, lc = wc->len, n = lb * lc;
    fq_t *res = (fq_t *)calloc(n ? n : 1, sizeof(fq_t));
    if (!res) return ORC_E_ALLOC;
    for (size_t i = 0; i < n; ++i) {
        size_t ib = i / lb, ic = i % lc; /* :101 */
        if (ib >= lb) { free(res); return ORC_E_LEN_MISMATCH; } /* Rust: index out of bounds panic */
        if (is_mul) fq_mul(&res[i], &wb->ev[ib], &wc->ev[ic]);
        else fq_add(&res[i], &wb->ev[ib], &wc->ev[ic]);
    }
    int rc = mle_new(out, res, n);
    free(res);
    return rc;
}

int mle_w_add(mle_t *out, const mle_t *wb, const mle_t *wc) { return operate_w(out, wb, wc, 0); }
/* w_mul(&self, other) = operate_w(other, self, Mul), :310-312 */
int mle_w_mul(mle_t *out, const mle_t *self, const mle_t *other) { return operate_w(out, other, self, 1); }

/* ------------------------------------------------------------------ ProductPolynomial */

int prod_new(prodpoly_t *out, mle_t *polys, size_t n) {
    out->polys = NULL;
    out->n = 0;
    if (n == 0) return ORC_E_LEN_MISMATCH; /* first().unwrap() panics */
    for (size_t i = 0; i < n; ++i)
        if (polys[i].len != polys[0].len) return ORC_E_LEN_MISMATCH; /* :15-19 */
    out->polys = (mle_t *)malloc(n * sizeof(mle_t));
    if (!out->polys) return ORC_E_ALLOC;
    memcpy(out->polys, polys, n * sizeof(mle_t));
    out->n = n;
    return ORC_OK;
}

void prod_free(prodpoly_t *p) {
    for (size_t i = 0; i < p->n; ++i) mle_free(&p->polys[i]);
    free(p->polys);
    p->polys = NULL;
    p->n = 0;
}

int prod_clone(prodpoly_t *out, const prodpoly_t *p) {
    out->polys = (mle_t *)calloc(p->n, sizeof(mle_t));
    if (!out->polys) return ORC_E_ALLOC;
    out->n = p->n;
    for (size_t i = 0; i < p->n; ++i) {
        int rc = mle_clone(&out->polys[i], &p->polys[i]);
        if (rc) { prod_free(out); return rc; }
    }
    return ORC_OK;
}

size_t prod_length(const prodpoly_t *p) { return p->polys[0].len; }

int prod_partial_evaluate(prodpoly_t *out, const prodpoly_t *p, const opt_fq_t *t, size_t nt) {
    if (nt != ilog2(prod_length(p))) return ORC_E_POINTS_LEN; /* :32-34 */
    out->polys = (mle_t *)calloc(p->n, sizeof(mle_t));
    if (!out->polys) return ORC_E_ALLOC;
    out->n = p->n;
    for (size_t i = 0; i < p->n; ++i) {
        int rc = mle_evaluate(&out->polys[i], &p->polys[i], t, nt); /* :36 */
        if (rc) { prod_free(out); return rc; }
    }
    return ORC_OK;
}

int prod_evaluate(fq_t *out, const prodpoly_t *p, const opt_fq_t *t, size_t nt) {
    if (nt != ilog2(prod_length(p))) return ORC_E_POINTS_LEN; /* :44-46 */
    fq_t acc;
    fq_one(&acc);
    for (size_t i = 0; i < p->n; ++i) {
        mle_t e;
        int rc = mle_evaluate(&e, &p->polys[i], t, nt);
        if (rc) return rc;
        fq_mul(&acc, &acc, &e.ev[0]);
        mle_free(&e);
    }
    *out = acc;
    return ORC_OK;
}

int prod_reduce(fq_t **out, size_t *len, const prodpoly_t *p) {
    size_t n = prod_length(p);
    fq_t *res = (fq_t *)malloc(n * sizeof(fq_t));
    if (!res) return ORC_E_ALLOC;
    for (size_t idx = 0; idx < n; ++idx) {
        fq_t run;
        fq_one(&run); /* T::from(1), :61 */
        for (size_t k = 0; k < p->n; ++k) fq_mul(&run, &run, &p->polys[k].ev[idx]);
        res[idx] = run;
    }
    *out = res;
    *len = n;
    return ORC_OK;
}

/* ------------------------------------------------------------------ SumPolynomial */

int sum_new(sumpoly_t *out, prodpoly_t *prods, size_t n) {
    out->prods = NULL;
    out->n = 0;
    if (n == 0) return ORC_E_LEN_MISMATCH;
    for (size_t i = 0; i < n; ++i)
        if (prod_length(&prods[i]) != prod_length(&prods[0])) return ORC_E_LEN_MISMATCH; /* :15-19 */
    out->prods = (prodpoly_t *)malloc(n * sizeof(prodpoly_t));
    if (!out->prods) return ORC_E_ALLOC;
    memcpy(out->prods, prods, n * sizeof(prodpoly_t));
    out->n = n;
    return ORC_OK;
}

void sum_free(sumpoly_t *s) {
    for (size_t i = 0; i < s->n; ++i) prod_free(&s->prods[i]);
    free(s->prods);
    s->prods = NULL;
    s->n = 0;
}

int sum_clone(sumpoly_t *out, const sumpoly_t *s) {
    out->prods = (prodpoly_t *)calloc(s->n, sizeof(prodpoly_t));
    if (!out->prods) return ORC_E_ALLOC;
    out->n = s->n;
    for (size_t i = 0; i < s->n; ++i) {
        int rc = prod_clone(&out->prods[i], &s->prods[i]);
        if (rc) { sum_free(out); return rc; }
    }
    return ORC_OK;
}

size_t sum_length(const sumpoly_t *s) { return prod_length(&s->prods[0]); }
unsigned sum_num_vars(const sumpoly_t *s) { return ilog2(sum_length(s)); }
size_t sum_degree(const sumpoly_t *s) { return s->n; }

int sum_partial_evaluate(sumpoly_t *out, const sumpoly_t *s, const opt_fq_t *t, size_t nt) {
    out->prods = (prodpoly_t *)calloc(s->n, sizeof(prodpoly_t));
    if (!out->prods) return ORC_E_ALLOC;
    out->n = s->n;
    for (size_t i = 0; i < s->n; ++i) {
        int rc = prod_partial_evaluate(&out->prods[i], &s->prods[i], t, nt);
        if (rc) { sum_free(out); return rc; }
    }
    return ORC_OK;
}

int sum_evaluate(fq_t *out, const sumpoly_t *s, const opt_fq_t *t, size_t nt) {
    fq_t acc;
    fq_zero(&acc);
    for (size_t i = 0; i < s->n; ++i) {
        fq_t e;
        int rc = prod_evaluate(&e, &s->prods[i], t, nt);
        if (rc) return rc;
        fq_add(&acc, &acc, &e);
    }
    *out = acc;
    return ORC_OK;
}

int sum_reduce(fq_t **out, size_t *len, const sumpoly_t *s) {
    size_t n = sum_length(s);
    fq_t **red = (fq_t **)calloc(s->n, sizeof(fq_t *));
    fq_t *res = (fq_t *)malloc(n * sizeof(fq_t));
    if (!red || !res) { free(red); free(res); return ORC_E_ALLOC; }
    int rc = ORC_OK;
    for (size_t i = 0; i < s->n && !rc; ++i) {
        size_t l;
        rc = prod_reduce(&red[i], &l, &s->prods[i]); /* :47-48 */
    }
    if (!rc) {
        for (size_t idx = 0; idx < n; ++idx) {
            fq_t run;
            fq_zero(&run);
            for (size_t i = 0; i < s->n; ++i) fq_add(&run, &run, &red[i][idx]);
            res[idx] = run;
        }
    }
    for (size_t i = 0; i < s->n; ++i) free(red[i]);
    free(red);
    if (rc) { free(res); return rc; }
    *out = res;
    *len = n;
    return ORC_OK;
}

/* ------------------------------------------------------------------ UnivariatePolynomial */

int uni_new(unipoly_t *out, const fq_t *c, size_t n) {
    out->c = (fq_t *)malloc((n ? n : 1) * sizeof(fq_t));
    if (!out->c) return ORC_E_ALLOC;
    if (n) memcpy(out->c, c, n * sizeof(fq_t));
    out->n = n;
    return ORC_OK;
}

void uni_free(unipoly_t *u) {
    free(u->c);
    u->c = NULL;
    u->n = 0;
}

void uni_evaluate(fq_t *out, const unipoly_t *u, const fq_t *x) {
    fq_t result, running_x, t;
    fq_zero(&result);
    fq_one(&running_x);
    for (size_t i = 0; i < u->n; ++i) {
        fq_mul(&t, &u->c[i], &running_x);
        fq_add(&result, &result, &t);
        fq_mul(&running_x, &running_x, x);
    }
    *out = result;
}

void uni_eval_sum_01(fq_t *out, const unipoly_t *u) {
    fq_t z, o, a, b;
    fq_zero(&z);
    fq_one(&o);
    uni_evaluate(&a, u, &z);
    uni_evaluate(&b, u, &o);
    fq_add(out, &a, &b);
}

int uni_scalar_mul(unipoly_t *out, const unipoly_t *u, const fq_t *k) {
    int rc = uni_new(out, u->c, u->n);
    if (rc) return rc;
    for (size_t i = 0; i < u->n; ++i) fq_mul(&out->c[i], &u->c[i], k);
    return ORC_OK;
}

int uni_mul(unipoly_t *out, const unipoly_t *a, const unipoly_t *b) {
    size_t l1 = a->n, l2 = b->n;
    size_t max_len = l1 > l2 ? l1 : l2;
    const unipoly_t *greater = a, *lesser = b;
    if (l2 > l1) { greater = b; lesser = a; } /* :119-124 */
    size_t n = l1 + l2 - 1;
    out->c = (fq_t *)calloc(n, sizeof(fq_t));
    if (!out->c) return ORC_E_ALLOC;
    out->n = n;
    for (size_t i = 0; i < max_len; ++i) {
        size_t idx = i;
        for (size_t j = 0; j < lesser->n; ++j) {
            fq_t t;
            fq_mul(&t, &greater->c[i], &lesser->c[j]);
            fq_add(&out->c[idx], &out->c[idx], &t);
            ++idx;
        }
    }
    return ORC_OK;
}

int uni_add(unipoly_t *out, const unipoly_t *a, const unipoly_t *b) {
    size_t n = a->n < b->n ? a->n : b->n; /* the reference's min-length quirk, :157 */
    out->c = (fq_t *)calloc(n ? n : 1, sizeof(fq_t));
    if (!out->c) return ORC_E_ALLOC;
    out->n = n;
    for (size_t i = 0; i < n; ++i) {
        fq_add(&out->c[i], &out->c[i], &a->c[i]);
        fq_add(&out->c[i], &out->c[i], &b->c[i]);
    }
    return ORC_OK;
}

int uni_interpolate(unipoly_t *out, const fq_t *xs, const fq_t *ys, size_t n) {
    unipoly_t res;
    res.c = (fq_t *)calloc(n ? n : 1, sizeof(fq_t)); /* vec![0; n], :57-59 */
    if (!res.c) return ORC_E_ALLOC;
    res.n = n;
    for (size_t i = 0; i < n; ++i) {
        fq_t denom;
        fq_one(&denom);
        unipoly_t numer;
        fq_t one;
        fq_one(&one);
        int rc = uni_new(&numer, &one, 1);
        if (rc) { uni_free(&res); return rc; }
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;
            fq_t lin[2], d;
            fq_neg(&lin[0], &xs[j]);
            fq_one(&lin[1]);
            unipoly_t ip = {lin, 2}, prod;
            fq_sub(&d, &xs[i], &xs[j]);
            fq_mul(&denom, &denom, &d); /* :78 */
            rc = uni_mul(&prod, &numer, &ip); /* :79 */
            uni_free(&numer);
            if (rc) { uni_free(&res); return rc; }
            numer = prod;
        }
        fq_t k;
        fq_div(&k, &ys[i], &denom); /* y_i / denominator, :82 */
        unipoly_t scaled, sum;
        rc = uni_scalar_mul(&scaled, &numer, &k);
        uni_free(&numer);
        if (rc) { uni_free(&res); return rc; }
        rc = uni_add(&sum, &res, &scaled);
        uni_free(&scaled);
        uni_free(&res);
        if (rc) return rc;
        res = sum;
    }
    *out = res;
    return ORC_OK;
}

void uni_to_bytes(uint8_t *out, const unipoly_t *u) {
    for (size_t i = 0; i < u->n; ++i) fq_to_le32(out + 32 * i, &u->c[i]);
}
