// fq_host.hpp -- host-side BN254 Fq for the prover's per-round scalar work (a handful of
// field operations per round: sums -> coefficients -> LE32 bytes, digest -> challenge).
// 4 x u64 limbs, Montgomery form (R = 2^256), the layout ark-ff keeps in memory.
// Multiplication is a full 512-bit schoolbook product followed by a separate word-by-word
// Montgomery reduction (SOS form); product code only -- never links the test oracle.
#pragma once
#include <cstdint>
#include <cstring>

namespace zkh {

typedef unsigned __int128 u128;

struct Fq {
    uint64_t l[4];
    bool operator==(const Fq& o) const { return std::memcmp(l, o.l, sizeof l) == 0; }
    bool operator!=(const Fq& o) const { return !(*this == o); }
};

static constexpr uint64_t kP[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                                   0x30644e72e131a029ULL};
static constexpr uint64_t kR2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL,
                                    0x06d89f71cab8351fULL};
static constexpr uint64_t kNinv = 0x87d20782e4866389ULL; // -p^-1 mod 2^64
// (p+1)/2 = 2^-1 mod p, canonical
static constexpr uint64_t kInv2[4] = {0x9e10460b6c3e7ea4ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL,
                                      0x183227397098d014ULL};

inline bool ge_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] != kP[i]) return a[i] > kP[i];
    }
    return true;
}
inline void sub_p_inplace(uint64_t a[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t t = a[i] - kP[i];
        uint64_t b1 = a[i] < kP[i];
        uint64_t t2 = t - borrow;
        uint64_t b2 = t < borrow;
        a[i] = t2;
        borrow = b1 | b2;
    }
}

inline Fq zero() { return Fq{{0, 0, 0, 0}}; }

inline Fq add(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 s = (u128)a.l[i] + b.l[i] + carry;
        r.l[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    if (ge_p(r.l)) sub_p_inplace(r.l);
    return r;
}

inline Fq sub(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 127);
    }
    if (borrow) {
        uint64_t carry = 0;
        for (int i = 0; i < 4; ++i) {
            u128 s = (u128)r.l[i] + kP[i] + carry;
            r.l[i] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
    }
    return r;
}

// Montgomery reduction of an 8-limb value T < p * 2^256: returns T * 2^-256 mod p
inline Fq redc(uint64_t T[9]) {
    for (int i = 0; i < 4; ++i) {
        uint64_t m = T[i] * kNinv;
        uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {
            u128 s = (u128)m * kP[j] + T[i + j] + carry;
            T[i + j] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        for (int k = i + 4; carry && k < 9; ++k) {
            u128 s = (u128)T[k] + carry;
            T[k] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
    }
    Fq r = {{T[4], T[5], T[6], T[7]}};
    if (T[8] || ge_p(r.l)) sub_p_inplace(r.l);
    return r;
}

inline Fq mul(const Fq& a, const Fq& b) {
    uint64_t T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {
            u128 s = (u128)a.l[i] * b.l[j] + T[i + j] + carry;
            T[i + j] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        T[i + 4] = carry;
    }
    return redc(T);
}

inline Fq from_canonical(const uint64_t c[4]) {
    Fq a = {{c[0], c[1], c[2], c[3]}}, r2 = {{kR2[0], kR2[1], kR2[2], kR2[3]}};
    return mul(a, r2);
}
inline Fq from_u64(uint64_t v) {
    uint64_t c[4] = {v, 0, 0, 0};
    return from_canonical(c);
}
inline Fq one() { return from_u64(1); }
inline Fq inv2() { return from_canonical(kInv2); }

inline void to_canonical(const Fq& a, uint64_t out[4]) {
    uint64_t T[9] = {a.l[0], a.l[1], a.l[2], a.l[3], 0, 0, 0, 0, 0};
    Fq r = redc(T);
    std::memcpy(out, r.l, sizeof r.l);
}

// into_bigint().to_bytes_le()
inline void to_le32(const Fq& a, uint8_t out[32]) {
    uint64_t c[4];
    to_canonical(a, c);
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 8; ++k) out[8 * i + k] = (uint8_t)(c[i] >> (8 * k));
}

// PrimeField::from_le_bytes_mod_order for a 32-byte string: 2^256 < 6p, so at most five
// subtractions bring the integer below p
inline Fq from_le32_mod_order(const uint8_t in[32]) {
    uint64_t c[4];
    for (int i = 0; i < 4; ++i) {
        uint64_t w = 0;
        for (int k = 0; k < 8; ++k) w |= (uint64_t)in[8 * i + k] << (8 * k);
        c[i] = w;
    }
    while (ge_p(c)) sub_p_inplace(c);
    return from_canonical(c);
}

inline Fq load(const uint64_t* p) { return Fq{{p[0], p[1], p[2], p[3]}}; }
inline void store(uint64_t* p, const Fq& a) { std::memcpy(p, a.l, sizeof a.l); }

// Horner evaluation of sum c_i x^i -- value equals UnivariatePolynomial::evaluate
// (dense_coefficient_form.rs:20-34)
inline Fq poly_eval(const Fq* c, size_t n, const Fq& x) {
    Fq acc = zero();
    for (size_t i = n; i-- > 0;) acc = add(mul(acc, x), c[i]);
    return acc;
}

} // namespace zkh
