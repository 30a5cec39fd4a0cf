/* oracle/keccak.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). See keccak.h. */
#include "keccak.h"
#include <string.h>

#define RATE 136

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

/* rotation offsets r[x][y] and the pi permutation, lane index = x + 5*y */
static const int ROT[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                            25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};

static inline uint64_t rol(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

static void keccak_f1600(uint64_t a[25]) {
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
        /* rho + pi: B[y, 2x+3y] = rot(A[x,y], r[x,y]) */
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) {
                int src = x + 5 * y;
                int dst = y + 5 * ((2 * x + 3 * y) % 5);
                b[dst] = rol(a[src], ROT[src]);
            }
        /* chi */
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x)
                a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}

static void absorb_block(uint64_t s[25], const uint8_t *blk) {
    for (int i = 0; i < RATE / 8; ++i) {
        uint64_t w = 0;
        for (int k = 0; k < 8; ++k) w |= (uint64_t)blk[8 * i + k] << (8 * k);
        s[i] ^= w;
    }
    keccak_f1600(s);
}

void keccak256_init(keccak256_t *k) { memset(k, 0, sizeof *k); }

void keccak256_update(keccak256_t *k, const uint8_t *data, size_t len) {
    if (k->fill) {
        size_t take = RATE - k->fill;
        if (take > len) take = len;
        memcpy(k->buf + k->fill, data, take);
        k->fill += take;
        data += take;
        len -= take;
        if (k->fill == RATE) {
            absorb_block(k->s, k->buf);
            k->fill = 0;
        }
    }
    while (len >= RATE) {
        absorb_block(k->s, data);
        data += RATE;
        len -= RATE;
    }
    if (len) {
        memcpy(k->buf, data, len);
        k->fill = len;
    }
}

void keccak256_peek(const keccak256_t *k, uint8_t out[32]) {
    uint64_t s[25];
    uint8_t blk[RATE];
    memcpy(s, k->s, sizeof s);
    memset(blk, 0, sizeof blk);
    memcpy(blk, k->buf, k->fill);
    blk[k->fill] ^= 0x01; /* original Keccak domain/pad bit (NOT the 0x06 of NIST SHA3-256) */
    blk[RATE - 1] ^= 0x80;
    absorb_block(s, blk);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(s[i] >> (8 * b));
}
