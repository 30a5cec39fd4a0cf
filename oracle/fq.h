/*
 * oracle/fq.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * BN254 base field Fq, restated from the public mathematical definition of
 * ark-ff 0.5.0 `Fp256<MontBackend<FqConfig,4>>` / ark-bn254 0.5.0 (the reference's
 * un-vendored dependency, /root/reference/Cargo.lock:56-59,88-91).  In-memory form is
 * the one ark-ff keeps: 4 x u64 little-endian limbs of a*R mod p, R = 2^256, fully
 * reduced (< p).  Call sites being restated: evaluation_form.rs:184 (mul/sub/add),
 * :230 (into_bigint().to_bytes_le()), :237 (Sum); transcript.rs:35
 * (from_le_bytes_mod_order); dense_coefficient_form.rs:75-82 (neg, sub, mul, div).
 *
 * Parity status: the reference holds no byte-level golden vectors for this arithmetic
 *  -- "parity unpinned" at the ark-ff boundary; pinned here by the
 * public BN254 constants and cross-checked against Python big-int arithmetic
 * (oracle/pyref.py, tests/test_oracle_field.py).
 */
#ifndef ORACLE_FQ_H
#define ORACLE_FQ_H
#include <stddef.h>
#include <stdint.h>

typedef struct { uint64_t l[4]; } fq_t; /* Montgomery form, little-endian limbs */

extern const uint64_t FQ_P[4];

void fq_zero(fq_t *r);
void fq_one(fq_t *r);
void fq_from_u64(fq_t *r, uint64_t v);            /* T::from(u64) */
void fq_from_canonical(fq_t *r, const uint64_t c[4]); /* c < p, canonical -> Montgomery */
void fq_to_canonical(uint64_t c[4], const fq_t *a);   /* into_bigint() */
void fq_to_le32(uint8_t out[32], const fq_t *a);      /* into_bigint().to_bytes_le() */
void fq_from_le_bytes_mod_order(fq_t *r, const uint8_t *bytes, size_t len);
void fq_add(fq_t *r, const fq_t *a, const fq_t *b);
void fq_sub(fq_t *r, const fq_t *a, const fq_t *b);
void fq_neg(fq_t *r, const fq_t *a);
void fq_mul(fq_t *r, const fq_t *a, const fq_t *b);
void fq_inv(fq_t *r, const fq_t *a);                  /* a != 0 */
void fq_div(fq_t *r, const fq_t *a, const fq_t *b);
int fq_eq(const fq_t *a, const fq_t *b);
int fq_is_zero(const fq_t *a);

#endif
