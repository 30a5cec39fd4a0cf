/*
 * oracle/fast.c -- TEST INFRASTRUCTURE ONLY (CPU baseline, second line).
 *
 * An honestly optimised CPU variant of the single-MLE round loop (sumcheck/src/prover.rs:45-155, MLE branch):
 * same field, same transcript bytes, same proof -- but one working copy folded IN PLACE, the fold of round k
 * fused with the half sums of round k+1, and OpenMP over all cores.  The reference itself is single-threaded
 * and copies the table several times per round; this variant exists so the GPU/CPU ratio in
 * bench.py is not inflated by that ("CPU-baseline plan").  Checked against the faithful
 * restatement in tests/test_oracle_fast.py.
 */
#include "protocol.h"
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* sums of the two halves of t[0..len) */
static void half_sums(const fq_t *t, size_t len, fq_t *e0, fq_t *e1) {
    const size_t h = len / 2;
    int nt = omp_get_max_threads();
    fq_t *p0 = (fq_t *)calloc((size_t)nt, sizeof(fq_t)), *p1 = (fq_t *)calloc((size_t)nt, sizeof(fq_t));
#pragma omp parallel
    {
        const int id = omp_get_thread_num();
        fq_t a, b;
        fq_zero(&a);
        fq_zero(&b);
#pragma omp for schedule(static)
        for (long long j = 0; j < (long long)h; ++j) { fq_add(&a, &a, &t[j]); fq_add(&b, &b, &t[h + j]); }
        p0[id] = a;
        p1[id] = b;
    }
    fq_zero(e0);
    fq_zero(e1);
    for (int i = 0; i < nt; ++i) { fq_add(e0, e0, &p0[i]); fq_add(e1, e1, &p1[i]); }
    free(p0);
    free(p1);
}

/* coeffs_out: n x 2 elements, challenges_out (optional): n elements.  absorb != 0: the table's LE32 bytes go into the
 * transcript first (generate_sumcheck_proof, prover.rs:158-173); serial, as Keccak is. */
API int orc_sumcheck_prove_fast(const uint64_t *table, size_t len, int absorb, int threads, uint64_t claim_out[4],
                                uint64_t *coeffs_out, uint64_t *challenges_out) {
    if (len == 0 || (len & (len - 1))) return ORC_E_NOT_POW2;
    if (threads > 0) omp_set_num_threads(threads);
    transcript_t tr;
    transcript_new(&tr);
    if (absorb) {
        enum { CH = 1 << 16 };
        uint8_t *buf = (uint8_t *)malloc((size_t)CH * 32);
        for (size_t off = 0; off < len; off += CH) {
            const size_t n = len - off < CH ? len - off : CH;
#pragma omp parallel for schedule(static)
            for (long long i = 0; i < (long long)n; ++i) fq_to_le32(buf + 32 * i, (const fq_t *)(table + 4 * (off + i)));
            transcript_append(&tr, buf, 32 * n);
        }
        free(buf);
    }
    fq_t *t = (fq_t *)malloc(len * sizeof(fq_t));
    if (!t) return ORC_E_ALLOC;
    memcpy(t, table, len * sizeof(fq_t));
    size_t cur = len, k = 0;
    fq_t e0, e1;
    if (cur >= 2) half_sums(t, cur, &e0, &e1);
    else { e0 = t[0]; fq_zero(&e1); }
    { fq_t s; fq_add(&s, &e0, &e1); memcpy(claim_out, s.l, 32); }
    while (cur >= 2) {
        fq_t claimed, c1, r;
        uint8_t b[96];
        fq_add(&claimed, &e0, &e1);
        fq_sub(&c1, &e1, &e0);
        fq_to_le32(b, &claimed);
        fq_to_le32(b + 32, &e0);
        fq_to_le32(b + 64, &c1);
        transcript_append(&tr, b, 96);
        transcript_sample_challenge(&tr, &r);
        memcpy(coeffs_out + 8 * k, e0.l, 32);
        memcpy(coeffs_out + 8 * k + 4, c1.l, 32);
        if (challenges_out) memcpy(challenges_out + 4 * k, r.l, 32);
        ++k;
        /* fold in place, fused with the next round's half sums */
        const size_t h = cur / 2, q = h / 2;
        int nt = omp_get_max_threads();
        fq_t *p0 = (fq_t *)calloc((size_t)nt, sizeof(fq_t)), *p1 = (fq_t *)calloc((size_t)nt, sizeof(fq_t));
#pragma omp parallel
        {
            const int id = omp_get_thread_num();
            fq_t a, bsum;
            fq_zero(&a);
            fq_zero(&bsum);
#pragma omp for schedule(static)
            for (long long j = 0; j < (long long)h; ++j) {
                fq_t d;
                fq_sub(&d, &t[h + j], &t[j]);
                fq_mul(&d, &d, &r);
                fq_add(&t[j], &t[j], &d); /* evaluation_form.rs:184 */
                if ((size_t)j < q || h == 1) fq_add(&a, &a, &t[j]);
                else fq_add(&bsum, &bsum, &t[j]);
            }
            p0[id] = a;
            p1[id] = bsum;
        }
        fq_zero(&e0);
        fq_zero(&e1);
        for (int i = 0; i < nt; ++i) { fq_add(&e0, &e0, &p0[i]); fq_add(&e1, &e1, &p1[i]); }
        free(p0);
        free(p1);
        cur = h;
    }
    free(t);
    return 0;
}
API int orc_max_threads(void) { return omp_get_max_threads(); }
