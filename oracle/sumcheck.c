/* oracle/sumcheck.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). See protocol.h for provenance. */
#include "protocol.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Transcript */

void transcript_new(transcript_t *t) { keccak256_init(&t->hasher); }

void transcript_append(transcript_t *t, const uint8_t *data, size_t len) {
    keccak256_update(&t->hasher, data, len);
}

void transcript_sample_challenge(transcript_t *t, fq_t *out) {
    uint8_t h[32];
    keccak256_peek(&t->hasher, h);            /* hasher.clone().finalize(), transcript.rs:30 */
    transcript_append(t, h, 32);              /* :33 */
    fq_from_le_bytes_mod_order(out, h, 32);   /* :35 */
}

/* ------------------------------------------------------------------ prover */

void sumcheck_proof_free(sumcheck_proof_t *p) {
    for (size_t i = 0; i < p->n_rounds; ++i) uni_free(&p->round_polys[i]);
    free(p->round_polys);
    p->round_polys = NULL;
    p->n_rounds = 0;
}

static void append_fq(transcript_t *t, const fq_t *v) {
    uint8_t b[32];
    fq_to_le32(b, v);
    transcript_append(t, b, 32);
}

static void append_uni(transcript_t *t, const unipoly_t *u) {
    uint8_t *b = (uint8_t *)malloc(32 * (u->n ? u->n : 1));
    uni_to_bytes(b, u);
    transcript_append(t, b, 32 * u->n);
    free(b);
}

/* prover.rs:23-43: r = sample_challenge(); points = [Some(r), None, ...] */
static opt_fq_t *generate_evaluation_points(transcript_t *t, size_t nvars, fq_t *challenge) {
    transcript_sample_challenge(t, challenge);
    opt_fq_t *pts = (opt_fq_t *)calloc(nvars ? nvars : 1, sizeof(opt_fq_t));
    if (pts && nvars) { pts[0].has = 1; pts[0].v = *challenge; }
    return pts;
}

int sumcheck_generate_round_polys(const sumpoly_t *sum_poly, const mle_t *multi_poly,
                                  transcript_t *transcript, unipoly_t **round_polys_out,
                                  fq_t **challenges_out, size_t *n_rounds_out) {
    int rc = ORC_OK;
    sumpoly_t cur_sum = {NULL, 0};
    mle_t cur_mle = {NULL, 0};
    size_t nvars;
    if (sum_poly) { /* polynomial.clone(), prover.rs:58 */
        rc = sum_clone(&cur_sum, sum_poly);
        nvars = sum_num_vars(sum_poly);
    } else {        /* prover.rs:64 */
        rc = mle_clone(&cur_mle, multi_poly);
        nvars = mle_num_vars(multi_poly);
    }
    if (rc) return rc;

    unipoly_t *round_polys = (unipoly_t *)calloc(nvars ? nvars : 1, sizeof(unipoly_t));
    fq_t *challenges = (fq_t *)calloc(nvars ? nvars : 1, sizeof(fq_t));
    size_t done = 0;
    if (!round_polys || !challenges) { rc = ORC_E_ALLOC; goto fail; }

    for (size_t round = 0; round < nvars; ++round) { /* prover.rs:83 */
        fq_t claimed_sum;
        unipoly_t round_poly = {NULL, 0};
        fq_zero(&claimed_sum);

        if (sum_poly) {
            size_t degree = sum_degree(&cur_sum);            /* :90 */
            size_t cur_vars = sum_num_vars(&cur_sum);
            fq_t *evaluation_points = (fq_t *)calloc(degree + 1, sizeof(fq_t));
            if (!evaluation_points) { rc = ORC_E_ALLOC; goto fail; }
            for (size_t i = 0; i < degree + 1; ++i) {          /* :93 */
                opt_fq_t *pts = (opt_fq_t *)calloc(cur_vars ? cur_vars : 1, sizeof(opt_fq_t));
                if (!pts) { free(evaluation_points); rc = ORC_E_ALLOC; goto fail; }
                pts[0].has = 1;
                fq_from_u64(&pts[0].v, (uint64_t)(uint8_t)i); /* T::from(i as u8), :95 */
                sumpoly_t pe;
                rc = sum_partial_evaluate(&pe, &cur_sum, pts, cur_vars); /* :97 */
                free(pts);
                if (rc) { free(evaluation_points); goto fail; }
                fq_t *res;
                size_t rl;
                rc = sum_reduce(&res, &rl, &pe);
                sum_free(&pe);
                if (rc) { free(evaluation_points); goto fail; }
                fq_t s;
                fq_zero(&s);
                for (size_t k = 0; k < rl; ++k) fq_add(&s, &s, &res[k]); /* :99 */
                free(res);
                evaluation_points[i] = s;
            }
            if (degree + 1 < 3) { /* Rust: y_points[2] index out of bounds (F9) */
                free(evaluation_points);
                rc = ORC_E_DEGREE;
                goto fail;
            }
            fq_add(&claimed_sum, &evaluation_points[0], &evaluation_points[1]); /* :102 */
            fq_t xs[3];
            fq_from_u64(&xs[0], 0);
            fq_from_u64(&xs[1], 1);
            fq_from_u64(&xs[2], 2);
            rc = uni_interpolate(&round_poly, xs, evaluation_points, 3); /* :104-107 */
            free(evaluation_points);
            if (rc) goto fail;
        } else {
            size_t half = cur_mle.len / 2; /* split_at(len/2), :110-111 */
            fq_t e0, e1;
            fq_zero(&e0);
            fq_zero(&e1);
            for (size_t k = 0; k < half; ++k) fq_add(&e0, &e0, &cur_mle.ev[k]);
            for (size_t k = half; k < cur_mle.len; ++k) fq_add(&e1, &e1, &cur_mle.ev[k]);
            fq_add(&claimed_sum, &e0, &e1); /* :118 */
            fq_t xs[2], ys[2] = {e0, e1};
            fq_from_u64(&xs[0], 0);
            fq_from_u64(&xs[1], 1);
            rc = uni_interpolate(&round_poly, xs, ys, 2); /* :120-123 */
            if (rc) goto fail;
        }

        append_fq(transcript, &claimed_sum);   /* :126-129 */
        append_uni(transcript, &round_poly);

        fq_t challenge;
        if (sum_poly) {
            size_t cur_vars = sum_num_vars(&cur_sum);
            opt_fq_t *pts = generate_evaluation_points(transcript, cur_vars, &challenge);
            if (!pts) { uni_free(&round_poly); rc = ORC_E_ALLOC; goto fail; }
            sumpoly_t next;
            rc = sum_partial_evaluate(&next, &cur_sum, pts, cur_vars); /* :139 */
            free(pts);
            if (rc) { uni_free(&round_poly); goto fail; }
            sum_free(&cur_sum);
            cur_sum = next;
        } else {
            size_t cur_vars = mle_num_vars(&cur_mle);
            opt_fq_t *pts = generate_evaluation_points(transcript, cur_vars, &challenge);
            if (!pts) { uni_free(&round_poly); rc = ORC_E_ALLOC; goto fail; }
            mle_t next;
            rc = mle_evaluate(&next, &cur_mle, pts, cur_vars); /* :148 */
            free(pts);
            if (rc) { uni_free(&round_poly); goto fail; }
            mle_free(&cur_mle);
            cur_mle = next;
        }
        challenges[done] = challenge;
        round_polys[done] = round_poly; /* :151 */
        ++done;
    }

    if (sum_poly) sum_free(&cur_sum); else mle_free(&cur_mle);
    *round_polys_out = round_polys;
    *challenges_out = challenges;
    *n_rounds_out = nvars;
    return ORC_OK;

fail:
    if (sum_poly) sum_free(&cur_sum); else mle_free(&cur_mle);
    if (round_polys) for (size_t i = 0; i < done; ++i) uni_free(&round_polys[i]);
    free(round_polys);
    free(challenges);
    return rc;
}

int sumcheck_generate_proof(sumcheck_proof_t *proof, const mle_t *init_poly) {
    transcript_t tr;
    transcript_new(&tr); /* prover.rs:159 */
    uint8_t *bytes = (uint8_t *)malloc(32 * init_poly->len);
    if (!bytes) return ORC_E_ALLOC;
    mle_to_bytes(bytes, init_poly);
    transcript_append(&tr, bytes, 32 * init_poly->len); /* :162 */
    free(bytes);

    mle_t copy; /* init_polynomial.clone(), :165 */
    int rc = mle_clone(&copy, init_poly);
    if (rc) return rc;
    fq_t *challenges;
    rc = sumcheck_generate_round_polys(NULL, &copy, &tr, &proof->round_polys, &challenges, &proof->n_rounds);
    mle_free(&copy);
    if (rc) return rc;
    free(challenges);
    mle_evaluation_sum(&proof->initial_claim_sum, init_poly); /* :170 */
    return ORC_OK;
}

int sumcheck_generate_proof_for_partial_verify(sumcheck_proof_t *proof, fq_t **random_points,
                                               const fq_t *initial_claim_sum, const sumpoly_t *init_poly,
                                               transcript_t *transcript) {
    int rc = sumcheck_generate_round_polys(init_poly, NULL, transcript, &proof->round_polys, random_points,
                                           &proof->n_rounds);
    if (rc) return rc;
    proof->initial_claim_sum = *initial_claim_sum; /* :185 -- never checked by the prover */
    return ORC_OK;
}

/* ------------------------------------------------------------------ verifier */

int sumcheck_partial_verify(const sumcheck_proof_t *proof, transcript_t *transcript, fq_t *final_claim,
                            fq_t **challenges_out, size_t *n_challenges) {
    fq_t *challenges = (fq_t *)calloc(proof->n_rounds ? proof->n_rounds : 1, sizeof(fq_t));
    size_t nch = 0;
    fq_t cur = proof->initial_claim_sum; /* verifier.rs:34 */
    int ok = 1;
    for (size_t k = 0; k < proof->n_rounds; ++k) {
        const unipoly_t *g = &proof->round_polys[k];
        fq_t s;
        uni_eval_sum_01(&s, g);
        if (!fq_eq(&s, &cur)) { ok = 0; break; } /* :41-45 */
        append_fq(transcript, &cur);              /* :47-50 */
        append_uni(transcript, g);
        fq_t ch;
        transcript_sample_challenge(transcript, &ch); /* :52 */
        challenges[nch++] = ch;
        uni_evaluate(&cur, g, &ch);               /* :58 */
    }
    *final_claim = cur;
    *challenges_out = challenges;
    *n_challenges = nch;
    return ok;
}

int sumcheck_verify_proof(const mle_t *initial_polynomial, const sumcheck_proof_t *proof) {
    transcript_t tr;
    transcript_new(&tr);
    uint8_t *bytes = (uint8_t *)malloc(32 * initial_polynomial->len);
    mle_to_bytes(bytes, initial_polynomial);
    transcript_append(&tr, bytes, 32 * initial_polynomial->len); /* verifier.rs:71 */
    free(bytes);

    fq_t s;
    mle_evaluation_sum(&s, initial_polynomial);
    if (!fq_eq(&s, &proof->initial_claim_sum)) return 0; /* :74-76 */

    fq_t final_claim, *challenges;
    size_t nch;
    int ok = sumcheck_partial_verify(proof, &tr, &final_claim, &challenges, &nch);
    if (!ok) { free(challenges); return 0; }
    if (proof->n_rounds == 0) { free(challenges); return 0; } /* :86-91 */

    /* perform_oracle_check, :15-26 */
    opt_fq_t *pts = (opt_fq_t *)calloc(nch ? nch : 1, sizeof(opt_fq_t));
    for (size_t i = 0; i < nch; ++i) { pts[i].has = 1; pts[i].v = challenges[i]; }
    mle_t e;
    int rc = mle_evaluate(&e, initial_polynomial, pts, nch);
    free(pts);
    free(challenges);
    if (rc) return 0; /* Rust would panic on a length mismatch */
    int good = fq_eq(&e.ev[0], &final_claim);
    mle_free(&e);
    return good;
}
