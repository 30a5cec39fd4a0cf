/*
 * oracle/fq.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). See fq.h for provenance.
 * 4 x 64-bit CIOS Montgomery arithmetic over BN254 Fq, written from the textbook
 * definition (Koc/Acar/Kaliski CIOS); no code from ark-ff (its source is not in the
 * container).
 */
#include "fq.h"
#include <string.h>

typedef unsigned __int128 u128;

const uint64_t FQ_P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL,
                          0xb85045b68181585dULL, 0x30644e72e131a029ULL};
/* R mod p and R^2 mod p, R = 2^256 */
static const uint64_t FQ_R[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL,
                                 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
static const uint64_t FQ_R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL,
                                  0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
static const uint64_t FQ_NINV = 0x87d20782e4866389ULL; /* -p^-1 mod 2^64 */

static int geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > FQ_P[i]) return 1;
        if (a[i] < FQ_P[i]) return 0;
    }
    return 1;
}

static void sub_p(uint64_t a[4]) {
    u128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - FQ_P[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}

void fq_zero(fq_t *r) { memset(r, 0, sizeof *r); }
void fq_one(fq_t *r) { memcpy(r->l, FQ_R, sizeof FQ_R); }

int fq_eq(const fq_t *a, const fq_t *b) { return memcmp(a, b, sizeof *a) == 0; }
int fq_is_zero(const fq_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }

void fq_add(fq_t *r, const fq_t *a, const fq_t *b) {
    uint64_t t[4];
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    /* p < 2^254 so a+b < 2^255: no carry out of limb 3 */
    if (geq_p(t)) sub_p(t);
    memcpy(r->l, t, sizeof t);
}

void fq_sub(fq_t *r, const fq_t *a, const fq_t *b) {
    uint64_t t[4];
    u128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        t[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)t[i] + FQ_P[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, sizeof t);
}

void fq_neg(fq_t *r, const fq_t *a) {
    fq_t z;
    fq_zero(&z);
    fq_sub(r, &z, a);
}

void fq_mul(fq_t *r, const fq_t *a, const fq_t *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 cur;
        uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {
            cur = (u128)a->l[j] * b->l[i] + t[j] + carry;
            t[j] = (uint64_t)cur;
            carry = (uint64_t)(cur >> 64);
        }
        cur = (u128)t[4] + carry;
        t[4] = (uint64_t)cur;
        t[5] = (uint64_t)(cur >> 64);

        uint64_t m = t[0] * FQ_NINV;
        cur = (u128)m * FQ_P[0] + t[0];
        carry = (uint64_t)(cur >> 64);
        for (int j = 1; j < 4; ++j) {
            cur = (u128)m * FQ_P[j] + t[j] + carry;
            t[j - 1] = (uint64_t)cur;
            carry = (uint64_t)(cur >> 64);
        }
        cur = (u128)t[4] + carry;
        t[3] = (uint64_t)cur;
        t[4] = t[5] + (uint64_t)(cur >> 64);
    }
    /* t < 2p here and t[4] == 0 because p < 2^254 */
    if (geq_p(t)) sub_p(t);
    memcpy(r->l, t, 4 * sizeof(uint64_t));
}

void fq_from_canonical(fq_t *r, const uint64_t c[4]) {
    fq_t a, r2;
    memcpy(a.l, c, sizeof a.l);
    memcpy(r2.l, FQ_R2, sizeof r2.l);
    fq_mul(r, &a, &r2);
}

void fq_from_u64(fq_t *r, uint64_t v) {
    uint64_t c[4] = {v, 0, 0, 0};
    fq_from_canonical(r, c);
}

void fq_to_canonical(uint64_t c[4], const fq_t *a) {
    fq_t one_raw = {{1, 0, 0, 0}}, t;
    fq_mul(&t, a, &one_raw);
    memcpy(c, t.l, sizeof t.l);
}

void fq_to_le32(uint8_t out[32], const fq_t *a) {
    uint64_t c[4];
    fq_to_canonical(c, a);
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 8; ++k) out[8 * i + k] = (uint8_t)(c[i] >> (8 * k));
}

/* Integer value of the little-endian byte string, reduced mod p.  Restated from the
 * documented semantics of PrimeField::from_le_bytes_mod_order (transcript.rs:35); done
 * by Horner's rule from the most significant byte, which is exact for any length. */
void fq_from_le_bytes_mod_order(fq_t *r, const uint8_t *bytes, size_t len) {
    fq_t acc, c256, b;
    fq_zero(&acc);
    fq_from_u64(&c256, 256);
    for (size_t i = len; i-- > 0;) {
        fq_mul(&acc, &acc, &c256);
        fq_from_u64(&b, bytes[i]);
        fq_add(&acc, &acc, &b);
    }
    *r = acc;
}

void fq_inv(fq_t *r, const fq_t *a) {
    /* Fermat: a^(p-2) */
    uint64_t e[4];
    memcpy(e, FQ_P, sizeof e);
    e[0] -= 2; /* p is odd and its low limb is > 2: no borrow */
    fq_t acc, base = *a;
    fq_one(&acc);
    for (int i = 0; i < 256; ++i) {
        if ((e[i / 64] >> (i % 64)) & 1) fq_mul(&acc, &acc, &base);
        fq_mul(&base, &base, &base);
    }
    *r = acc;
}

void fq_div(fq_t *r, const fq_t *a, const fq_t *b) {
    fq_t bi;
    fq_inv(&bi, b);
    fq_mul(r, a, &bi);
}
