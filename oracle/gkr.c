/* oracle/gkr.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). See protocol.h for provenance.
 * Dense restatement of the reference's Circuit and GKR prover/verifier: the wiring
 * predicates are the dense 2^(k_out+2k_in) tables of circuit.rs:82-123, so it is only
 * usable on small circuits -- exactly like the reference. */
#include "protocol.h"
#include <stdlib.h>
#include <string.h>

static size_t next_pow2(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}
static unsigned ilog2z(size_t n) {
    unsigned k = 0;
    while (n > 1) { n >>= 1; ++k; }
    return k;
}

/* ------------------------------------------------------------------ Circuit */

int circuit_new(circuit_t *c, const gate_t *const *layers, const size_t *layer_len, size_t n_layers) {
    c->layers = (gate_t **)calloc(n_layers ? n_layers : 1, sizeof(gate_t *));
    c->layer_len = (size_t *)calloc(n_layers ? n_layers : 1, sizeof(size_t));
    c->n_layers = n_layers;
    if (!c->layers || !c->layer_len) return ORC_E_ALLOC;
    for (size_t i = 0; i < n_layers; ++i) {
        c->layer_len[i] = layer_len[i];
        c->layers[i] = (gate_t *)malloc((layer_len[i] ? layer_len[i] : 1) * sizeof(gate_t));
        if (!c->layers[i]) return ORC_E_ALLOC;
        memcpy(c->layers[i], layers[i], layer_len[i] * sizeof(gate_t));
    }
    return ORC_OK;
}

void circuit_free(circuit_t *c) {
    for (size_t i = 0; i < c->n_layers; ++i) free(c->layers[i]);
    free(c->layers);
    free(c->layer_len);
    memset(c, 0, sizeof *c);
}

int circuit_evaluate_at_input(const circuit_t *c, const fq_t *inputs, size_t n_inputs, mle_t **evals_out) {
    mle_t *evals = (mle_t *)calloc(c->n_layers + 1, sizeof(mle_t));
    if (!evals) return ORC_E_ALLOC;
    int rc = mle_new(&evals[0], inputs, n_inputs); /* circuit.rs:38 */
    size_t made = rc ? 0 : 1;
    fq_t *running = NULL;
    size_t running_len = n_inputs;
    if (!rc) {
        running = (fq_t *)malloc((n_inputs ? n_inputs : 1) * sizeof(fq_t));
        if (!running) rc = ORC_E_ALLOC; else memcpy(running, inputs, n_inputs * sizeof(fq_t));
    }
    for (size_t li = 0; li < c->n_layers && !rc; ++li) {
        size_t ng = c->layer_len[li];
        size_t width = next_pow2(ng);
        if (width < 2) width = 2; /* max(next_power_of_two, 2), :43 */
        fq_t *next = (fq_t *)calloc(width, sizeof(fq_t));
        if (!next) { rc = ORC_E_ALLOC; break; }
        for (size_t g = 0; g < ng; ++g) {
            const gate_t *gt = &c->layers[li][g];
            if (gt->left >= running_len || gt->right >= running_len) { rc = ORC_E_LAYER_OOB; break; }
            if (gt->op == GATE_ADD) fq_add(&next[g], &running[gt->left], &running[gt->right]);
            else fq_mul(&next[g], &running[gt->left], &running[gt->right]);
        }
        if (!rc) rc = mle_new(&evals[li + 1], next, width);
        if (!rc) ++made;
        free(running);
        running = next;
        running_len = width;
    }
    free(running);
    if (rc) {
        for (size_t i = 0; i < made; ++i) mle_free(&evals[i]);
        free(evals);
        return rc;
    }
    *evals_out = evals;
    return ORC_OK;
}

int circuit_get_gate_poly(mle_t *out, const circuit_t *c, size_t layer_idx, int op) {
    if (layer_idx >= c->n_layers) return ORC_E_LAYER_OOB; /* :83-85 */
    size_t li = c->n_layers - layer_idx - 1;              /* :87 */
    const gate_t *gates = c->layers[li];
    size_t ng = c->layer_len[li];
    if (ng == 0) return ORC_E_LAYER_OOB; /* Rust: max().unwrap() on an empty list panics */
    size_t output_length = next_pow2(ng);
    if (output_length == 1) output_length = 2; /* :91-94 */
    size_t mx = 0;
    for (size_t g = 0; g < ng; ++g) {
        if (gates[g].left + 1 > mx) mx = gates[g].left + 1;
        if (gates[g].right + 1 > mx) mx = gates[g].right + 1;
    }
    unsigned ibl = ilog2z(next_pow2(mx)); /* :104-109 */
    size_t n = output_length << (2 * ibl);
    fq_t *ev = (fq_t *)calloc(n, sizeof(fq_t));
    if (!ev) return ORC_E_ALLOC;
    for (size_t g = 0; g < ng; ++g)
        if (gates[g].op == op) {
            size_t idx = (((g << ibl) | gates[g].left) << ibl) | gates[g].right; /* get_bit_idx, :66-74 */
            fq_one(&ev[idx]);
        }
    int rc = mle_new(out, ev, n);
    free(ev);
    return rc;
}

int circuit_get_w_i(mle_t *out, size_t layer_idx, const mle_t *evals, size_t n_evals) {
    if (layer_idx >= n_evals) return ORC_E_LAYER_OOB; /* :131-133 */
    const mle_t *src = &evals[n_evals - layer_idx - 1];
    return mle_new(out, src->ev, src->len);
}

/* ------------------------------------------------------------------ gkr/src/utils.rs */

/* utils.rs:45-65 */
static int get_evaluated_muli_addi_at_a(mle_t *mul_out, mle_t *add_out, const mle_t *mul, const mle_t *add,
                                        const fq_t *rv, size_t nrv) {
    size_t nv = mle_num_vars(mul);
    opt_fq_t *pts = (opt_fq_t *)calloc(nv ? nv : 1, sizeof(opt_fq_t));
    if (!pts) return ORC_E_ALLOC;
    for (size_t i = 0; i < nv && i < nrv; ++i) { pts[i].has = 1; pts[i].v = rv[i]; }
    int rc = mle_evaluate(mul_out, mul, pts, nv);
    if (!rc) {
        rc = mle_evaluate(add_out, add, pts, nv);
        if (rc) mle_free(mul_out);
    }
    free(pts);
    return rc;
}

/* alpha * p(rb,.,.) + beta * p(rc,.,.), utils.rs:36-40 */
static int fold_one(mle_t *out, const mle_t *p, const opt_fq_t *prb, const opt_fq_t *prc, size_t nv,
                    const fq_t *alpha, const fq_t *beta) {
    mle_t eb = {0}, ec = {0}, sb = {0}, sc = {0};
    int rc = mle_evaluate(&eb, p, prb, nv);
    if (!rc) rc = mle_scalar_mul(&sb, &eb, alpha);
    if (!rc) rc = mle_evaluate(&ec, p, prc, nv);
    if (!rc) rc = mle_scalar_mul(&sc, &ec, beta);
    if (!rc) rc = mle_add(out, &sb, &sc);
    mle_free(&eb); mle_free(&ec); mle_free(&sb); mle_free(&sc);
    return rc;
}

/* utils.rs:16-43 */
static int get_folded_polys(mle_t *mul_out, mle_t *add_out, const fq_t *alpha, const fq_t *beta,
                            const mle_t *mul, const mle_t *add, const fq_t *rb, size_t nrb, const fq_t *rc_,
                            size_t nrc) {
    size_t nv = mle_num_vars(mul);
    opt_fq_t *prb = (opt_fq_t *)calloc(nv ? nv : 1, sizeof(opt_fq_t));
    opt_fq_t *prc = (opt_fq_t *)calloc(nv ? nv : 1, sizeof(opt_fq_t));
    if (!prb || !prc) { free(prb); free(prc); return ORC_E_ALLOC; }
    int rc = ORC_OK;
    for (size_t i = 0; i < nrb; ++i) {
        if (i >= nv || i >= nrc) { rc = ORC_E_POINTS_LEN; break; } /* Rust: index out of bounds */
        prb[i].has = 1; prb[i].v = rb[i];
        prc[i].has = 1; prc[i].v = rc_[i];
    }
    if (!rc) rc = fold_one(mul_out, mul, prb, prc, nv, alpha, beta);
    if (!rc) {
        rc = fold_one(add_out, add, prb, prc, nv, alpha, beta);
        if (rc) mle_free(mul_out);
    }
    free(prb);
    free(prc);
    return rc;
}

static int eval_all(fq_t *out, const mle_t *m, const fq_t *pts, size_t n) {
    opt_fq_t *o = (opt_fq_t *)calloc(n ? n : 1, sizeof(opt_fq_t));
    if (!o) return ORC_E_ALLOC;
    for (size_t i = 0; i < n; ++i) { o[i].has = 1; o[i].v = pts[i]; }
    mle_t e;
    int rc = mle_evaluate(&e, m, o, n);
    free(o);
    if (rc) return rc;
    *out = e.ev[0];
    mle_free(&e);
    return ORC_OK;
}

static void append_fq(transcript_t *t, const fq_t *v) {
    uint8_t b[32];
    fq_to_le32(b, v);
    transcript_append(t, b, 32);
}

static void append_mle(transcript_t *t, const mle_t *m) {
    uint8_t *b = (uint8_t *)malloc(32 * m->len);
    mle_to_bytes(b, m);
    transcript_append(t, b, 32 * m->len);
    free(b);
}

/* ------------------------------------------------------------------ GKR prover */

void gkr_proof_free(gkr_proof_t *p) {
    mle_free(&p->output_poly);
    free(p->w_evals);
    for (size_t i = 0; i < p->n_proofs; ++i) sumcheck_proof_free(&p->sumcheck_proofs[i]);
    free(p->sumcheck_proofs);
    memset(p, 0, sizeof *p);
}

int gkr_generate_proof(gkr_proof_t *proof, const circuit_t *c, transcript_t *t, const fq_t *inputs,
                       size_t n_inputs) {
    memset(proof, 0, sizeof *proof);
    mle_t *evals = NULL;
    int rc = circuit_evaluate_at_input(c, inputs, n_inputs, &evals); /* prover.rs:30 */
    if (rc) return rc;
    size_t n_evals = c->n_layers + 1;

    mle_t running = {0};
    fq_t *random_values = NULL;
    size_t n_rv = 0;
    proof->w_evals = (fq_t *)calloc(2 * (c->n_layers ? c->n_layers : 1), sizeof(fq_t));
    proof->sumcheck_proofs = (sumcheck_proof_t *)calloc(c->n_layers ? c->n_layers : 1, sizeof(sumcheck_proof_t));
    if (!proof->w_evals || !proof->sumcheck_proofs) { rc = ORC_E_ALLOC; goto done; }

    rc = circuit_get_w_i(&running, 0, evals, n_evals); /* :43 */
    if (rc) goto done;
    append_mle(t, &running);                            /* :46 */
    n_rv = mle_num_vars(&running);                      /* :38-40, :49-53 */
    random_values = (fq_t *)calloc(n_rv ? n_rv : 1, sizeof(fq_t));
    if (!random_values) { rc = ORC_E_ALLOC; goto done; }
    for (size_t i = 0; i < n_rv; ++i) transcript_sample_challenge(t, &random_values[i]);

    for (size_t layer_idx = 0; layer_idx < c->n_layers; ++layer_idx) { /* :55 */
        mle_t mul = {0}, add = {0}, new_mul = {0}, new_add = {0}, next_w = {0}, ww = {0}, wpw = {0};
        fq_t claim_sum;
        rc = circuit_get_gate_poly(&mul, c, layer_idx, GATE_MUL);
        if (!rc) rc = circuit_get_gate_poly(&add, c, layer_idx, GATE_ADD);
        if (!rc) {
            if (layer_idx == 0) {
                rc = get_evaluated_muli_addi_at_a(&new_mul, &new_add, &mul, &add, random_values, n_rv); /* :66-67 */
                if (!rc) rc = eval_all(&claim_sum, &running, random_values, n_rv); /* :70-75 */
            } else {
                size_t nb = n_rv / 2, nc = n_rv - nb; /* :82-85 */
                const fq_t *rb = random_values, *rcv = random_values + nb;
                fq_t wb, wc, alpha, beta;
                rc = eval_all(&wb, &running, rb, nb);             /* :87 */
                if (!rc) rc = eval_all(&wc, &running, rcv, nc);   /* :88 */
                if (!rc) {
                    append_fq(t, &wb);                             /* :102-105 */
                    append_fq(t, &wc);
                    transcript_sample_challenge(t, &alpha);        /* :107-108 */
                    transcript_sample_challenge(t, &beta);
                    rc = get_folded_polys(&new_mul, &new_add, &alpha, &beta, &mul, &add, rb, nb, rcv, nc);
                }
                if (!rc) {
                    proof->w_evals[2 * proof->n_w_evals] = wb;     /* :115 */
                    proof->w_evals[2 * proof->n_w_evals + 1] = wc;
                    ++proof->n_w_evals;
                    fq_t a, b;                                     /* utils.rs:13 (args swapped at :118; same value) */
                    fq_mul(&a, &alpha, &wb);
                    fq_mul(&b, &beta, &wc);
                    fq_add(&claim_sum, &a, &b);
                }
            }
        }
        if (!rc) rc = circuit_get_w_i(&next_w, layer_idx + 1, evals, n_evals); /* :125 */
        if (!rc) rc = mle_w_mul(&ww, &next_w, &next_w);                        /* :131 */
        if (!rc) rc = mle_w_add(&wpw, &next_w, &next_w);                       /* :135 */
        if (!rc) {
            /* f(b,c) = Sum[ Prod[mul', W*W], Prod[add', W+W] ], :128-137 */
            mle_t p0[2] = {new_mul, ww}, p1[2] = {new_add, wpw};
            prodpoly_t prods[2] = {{0}, {0}};
            sumpoly_t f = {0};
            rc = prod_new(&prods[0], p0, 2);
            if (!rc) { memset(&new_mul, 0, sizeof new_mul); memset(&ww, 0, sizeof ww); }
            if (!rc) {
                rc = prod_new(&prods[1], p1, 2);
                if (!rc) { memset(&new_add, 0, sizeof new_add); memset(&wpw, 0, sizeof wpw); }
            }
            if (!rc) {
                rc = sum_new(&f, prods, 2);
                if (!rc) memset(prods, 0, sizeof prods);
            }
            if (!rc) {
                fq_t *pts = NULL;
                rc = sumcheck_generate_proof_for_partial_verify(&proof->sumcheck_proofs[proof->n_proofs], &pts,
                                                                &claim_sum, &f, t); /* :140-141 */
                if (!rc) {
                    free(random_values);
                    random_values = pts; /* :143 */
                    n_rv = proof->sumcheck_proofs[proof->n_proofs].n_rounds;
                    ++proof->n_proofs;
                }
            }
            sum_free(&f);
            prod_free(&prods[0]);
            prod_free(&prods[1]);
        }
        mle_free(&mul); mle_free(&add); mle_free(&new_mul); mle_free(&new_add); mle_free(&ww); mle_free(&wpw);
        if (rc) { mle_free(&next_w); goto done; }
        mle_free(&running);
        running = next_w; /* :144 */
    }
    rc = circuit_get_w_i(&proof->output_poly, 0, evals, n_evals); /* :149-153 */

done:
    mle_free(&running);
    free(random_values);
    for (size_t i = 0; i < n_evals; ++i) mle_free(&evals[i]);
    free(evals);
    if (rc) gkr_proof_free(proof);
    return rc;
}

/* ------------------------------------------------------------------ GKR verifier */

int gkr_verify_proof(const fq_t *inputs, size_t n_inputs, const circuit_t *c, transcript_t *t,
                     const gkr_proof_t *proof) {
    size_t n_rv = mle_num_vars(&proof->output_poly); /* verifier.rs:28 */
    append_mle(t, &proof->output_poly);              /* :30 */
    fq_t *rv = (fq_t *)calloc(n_rv ? n_rv : 1, sizeof(fq_t));
    for (size_t i = 0; i < n_rv; ++i) transcript_sample_challenge(t, &rv[i]);
    int ok = 1;

    for (size_t layer_idx = 0; layer_idx < c->n_layers && ok; ++layer_idx) {
        mle_t mul = {0}, add = {0}, new_mul = {0}, new_add = {0};
        int rc = circuit_get_gate_poly(&mul, c, layer_idx, GATE_MUL);
        if (!rc) rc = circuit_get_gate_poly(&add, c, layer_idx, GATE_ADD);
        if (!rc) {
            if (layer_idx == 0) {
                rc = get_evaluated_muli_addi_at_a(&new_mul, &new_add, &mul, &add, rv, n_rv); /* :43 */
            } else {
                fq_t alpha, beta;
                transcript_sample_challenge(t, &alpha); /* :45-46 */
                transcript_sample_challenge(t, &beta);
                size_t nb = n_rv / 2;
                rc = get_folded_polys(&new_mul, &new_add, &alpha, &beta, &mul, &add, rv, nb, rv + nb, n_rv - nb);
            }
        }
        if (rc || layer_idx >= proof->n_proofs) { ok = 0; mle_free(&mul); mle_free(&add); mle_free(&new_mul); mle_free(&new_add); break; }

        fq_t final_claim, *next_vals;
        size_t n_next;
        int verified = sumcheck_partial_verify(&proof->sumcheck_proofs[layer_idx], t, &final_claim, &next_vals, &n_next); /* :63-64 */

        fq_t add_eval, mul_eval, wb, wc;
        rc = eval_all(&add_eval, &new_add, next_vals, n_next); /* :67 (Rust panics on length mismatch) */
        if (!rc) rc = eval_all(&mul_eval, &new_mul, next_vals, n_next);
        if (!rc) {
            if (layer_idx + 1 == c->n_layers) { /* :76-94 */
                mle_t w_in;
                rc = mle_new(&w_in, inputs, n_inputs);
                size_t nb = n_next / 2;
                if (!rc) rc = eval_all(&wb, &w_in, next_vals, nb);
                if (!rc) rc = eval_all(&wc, &w_in, next_vals + nb, n_next - nb);
                mle_free(&w_in);
            } else if (layer_idx < proof->n_w_evals) {
                wb = proof->w_evals[2 * layer_idx]; /* :97 */
                wc = proof->w_evals[2 * layer_idx + 1];
            } else {
                rc = ORC_E_LAYER_OOB;
            }
        }
        if (rc) {
            ok = 0;
        } else {
            append_fq(t, &wb); /* :101-104 */
            append_fq(t, &wc);
            fq_t s, p, a, b, fbc;
            fq_add(&s, &wb, &wc);
            fq_mul(&p, &wb, &wc);
            fq_mul(&a, &add_eval, &s);
            fq_mul(&b, &mul_eval, &p);
            fq_add(&fbc, &a, &b); /* :106-107 */
            if (!verified || !fq_eq(&fbc, &final_claim)) ok = 0; /* :111-113 */
        }
        free(rv);
        rv = next_vals; /* :115 */
        n_rv = n_next;
        mle_free(&mul); mle_free(&add); mle_free(&new_mul); mle_free(&new_add);
    }
    free(rv);
    return ok;
}
