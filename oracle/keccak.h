/*
 * oracle/keccak.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * Keccak-256 (original Keccak padding 0x01 ... 0x80, rate 136 B, capacity 512), the
 * function `sha3::Keccak256` of crate sha3 0.10.8 computes (Cargo.lock:545-548; used at
 * fiat_shamir/src/transcript.rs:3,15,21,30).  Restated from the Keccak reference
 * specification; pinned by the public known answers Keccak-256("") and
 * Keccak-256("abc") (tests/test_oracle_keccak.py).
 */
#ifndef ORACLE_KECCAK_H
#define ORACLE_KECCAK_H
#include <stddef.h>
#include <stdint.h>

typedef struct {
    uint64_t s[25];
    uint8_t buf[136];
    size_t fill; /* bytes in buf, < 136 */
} keccak256_t;

void keccak256_init(keccak256_t *k);
void keccak256_update(keccak256_t *k, const uint8_t *data, size_t len);
/* digest of everything absorbed so far, WITHOUT consuming the state
 * (the reference's `hasher.clone().finalize()`, transcript.rs:30) */
void keccak256_peek(const keccak256_t *k, uint8_t out[32]);

#endif
