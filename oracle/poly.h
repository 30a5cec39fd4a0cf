/*
 * oracle/poly.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * CPU restatement of the reference's `polynomials` crate, the parts on the sumcheck /
 * GKR path.  Same passes, same copies, single thread (the reference has no threading,
 * SURVEY.md F1) so that it can double as the timed "port" CPU baseline.
 *   MultiLinearPolynomial  polynomials/src/multilinear_polynomial/evaluation_form.rs:17-313
 *   ProductPolynomial      polynomials/src/product_polynomial.rs:7-78
 *   SumPolynomial          polynomials/src/sum_polynomial.rs:7-82
 *   UnivariatePolynomial   polynomials/src/univariate_polynomial/dense_coefficient_form.rs:7-208
 * Where the reference panics these functions return a non-zero ORC_E_* code.
 */
#ifndef ORACLE_POLY_H
#define ORACLE_POLY_H
#include "fq.h"

enum {
    ORC_OK = 0,
    ORC_E_NOT_POW2 = 1,     /* evaluation_form.rs:23-27 */
    ORC_E_POINTS_LEN = 2,   /* evaluation_form.rs:197-199, product_polynomial.rs:32-34 */
    ORC_E_LEN_MISMATCH = 3, /* product_polynomial.rs:15-19, sum_polynomial.rs:15-19, evaluation_form.rs:244-246 */
    ORC_E_LAYER_OOB = 4,    /* circuit.rs:83-85, 131-133 */
    ORC_E_DEGREE = 5,       /* interpolation nodes [0,1,2] vs degree()+1 values, prover.rs:90-107 (F9) */
    ORC_E_ALLOC = 6
};

/* Option<T>: has == 0 is None */
typedef struct { int has; fq_t v; } opt_fq_t;

typedef struct { fq_t *ev; size_t len; } mle_t;
typedef struct { mle_t *polys; size_t n; } prodpoly_t;
typedef struct { prodpoly_t *prods; size_t n; } sumpoly_t;
typedef struct { fq_t *c; size_t n; } unipoly_t;

/* ---- MultiLinearPolynomial ---- */
int mle_new(mle_t *out, const fq_t *pts, size_t len);             /* :22-32 (copies) */
void mle_free(mle_t *m);
int mle_clone(mle_t *out, const mle_t *m);
unsigned mle_num_vars(const mle_t *m);                             /* :159-161 */
int mle_partially_evaluate(mle_t *out, const mle_t *m, size_t var, const fq_t *r); /* :163-192 */
int mle_evaluate(mle_t *out, const mle_t *m, const opt_fq_t *pts, size_t npts);    /* :194-223 */
void mle_to_bytes(uint8_t *out, const mle_t *m);                   /* :225-234, 32*len bytes */
void mle_evaluation_sum(fq_t *out, const mle_t *m);                /* :236-238 */
int mle_scalar_mul(mle_t *out, const mle_t *m, const fq_t *s);     /* :151-153 */
int mle_minus(mle_t *out, const mle_t *m, const fq_t *other);      /* :260-272 */
int mle_compute_quotient_remainder(fq_t *quotient, mle_t *remainder, const mle_t *m, const fq_t *divisor, size_t var); /* :280-301 */
int mle_add(mle_t *out, const mle_t *a, const mle_t *b);           /* :241-258 */
int mle_w_add(mle_t *out, const mle_t *wb, const mle_t *wc);       /* :303-308 */
int mle_w_mul(mle_t *out, const mle_t *self, const mle_t *other);  /* :310-312 */

/* ---- ProductPolynomial / SumPolynomial ---- */
int prod_new(prodpoly_t *out, mle_t *polys, size_t n);             /* takes ownership of polys[] contents */
void prod_free(prodpoly_t *p);
int prod_clone(prodpoly_t *out, const prodpoly_t *p);
size_t prod_length(const prodpoly_t *p);
int prod_partial_evaluate(prodpoly_t *out, const prodpoly_t *p, const opt_fq_t *t, size_t nt);
int prod_evaluate(fq_t *out, const prodpoly_t *p, const opt_fq_t *t, size_t nt);
int prod_reduce(fq_t **out, size_t *len, const prodpoly_t *p);     /* product_polynomial.rs:54-73 */

int sum_new(sumpoly_t *out, prodpoly_t *prods, size_t n);          /* takes ownership */
void sum_free(sumpoly_t *s);
int sum_clone(sumpoly_t *out, const sumpoly_t *s);
size_t sum_length(const sumpoly_t *s);
unsigned sum_num_vars(const sumpoly_t *s);
size_t sum_degree(const sumpoly_t *s);                              /* sum_polynomial.rs:79-81 */
int sum_partial_evaluate(sumpoly_t *out, const sumpoly_t *s, const opt_fq_t *t, size_t nt);
int sum_evaluate(fq_t *out, const sumpoly_t *s, const opt_fq_t *t, size_t nt);
int sum_reduce(fq_t **out, size_t *len, const sumpoly_t *s);       /* sum_polynomial.rs:44-65 */

/* ---- UnivariatePolynomial (dense coefficient form, ascending degree) ---- */
int uni_new(unipoly_t *out, const fq_t *c, size_t n);
void uni_free(unipoly_t *u);
void uni_evaluate(fq_t *out, const unipoly_t *u, const fq_t *x);   /* :20-34 */
void uni_eval_sum_01(fq_t *out, const unipoly_t *u);               /* :37-45 */
int uni_interpolate(unipoly_t *out, const fq_t *xs, const fq_t *ys, size_t n); /* :52-88 */
int uni_scalar_mul(unipoly_t *out, const unipoly_t *u, const fq_t *k);         /* :94-107 */
int uni_mul(unipoly_t *out, const unipoly_t *a, const unipoly_t *b);           /* :111-147 */
int uni_add(unipoly_t *out, const unipoly_t *a, const unipoly_t *b);           /* :151-180 */
void uni_to_bytes(uint8_t *out, const unipoly_t *u);               /* :182-191, 32*n bytes */

#endif
