// keccak_host.hpp -- host Keccak-256 sponge for the Fiat-Shamir transcript
// (fiat_shamir/src/transcript.rs:6-42 uses sha3::Keccak256: rate 136 B, capacity 512,
// ORIGINAL Keccak multi-rate padding 0x01 .. 0x80 -- not NIST SHA3-256's 0x06).
// Product code: independent of the test oracle.  The state is kept as 25 named lanes and the
// round is written plane by plane so the compiler keeps everything in registers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace zkh {

class Keccak256 {
  public:
    static constexpr size_t kRate = 136;

    Keccak256() { reset(); }
    void reset() {
        std::memset(st_, 0, sizeof st_);
        fill_ = 0;
    }

    void update(const uint8_t* data, size_t len) {
        if (fill_) {
            size_t take = kRate - fill_;
            if (take > len) take = len;
            std::memcpy(buf_ + fill_, data, take);
            fill_ += take;
            data += take;
            len -= take;
            if (fill_ < kRate) return;
            absorb(st_, buf_);
            fill_ = 0;
        }
        if (len >= kRate) { // the bulk of a table absorb: whole blocks in one call of the form this CPU runs fastest
            const size_t nblocks = len / kRate;
            absorb_blocks(st_, data, nblocks);
            data += nblocks * kRate;
            len -= nblocks * kRate;
        }
        if (len) {
            std::memcpy(buf_, data, len);
            fill_ = len;
        }
    }

    // digest of everything absorbed so far; the sponge itself is left untouched
    // (the reference's `self.hasher.clone().finalize()`, transcript.rs:30)
    void peek(uint8_t out[32]) const {
        uint64_t s[25];
        std::memcpy(s, st_, sizeof s);
        uint8_t last[kRate];
        std::memset(last, 0, sizeof last);
        std::memcpy(last, buf_, fill_);
        last[fill_] ^= 0x01;
        last[kRate - 1] ^= 0x80;
        absorb(s, last);
        std::memcpy(out, s, 32); // little-endian host
    }

    // Both builds of the round function (baseline ISA; BMI1/BMI2 when this CPU has them) on the same `nblocks` pseudo-random blocks:
    // 0 if the states agree (or there is only one build), 1 otherwise.  The library runs only one of the two on a given machine.
    static int selftest(size_t nblocks) {
#if defined(__x86_64__)
        if (!(__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2"))) return 0;
        uint8_t block[kRate];
        uint64_t a[25] = {0}, b[25] = {0}, x = 0x9e3779b97f4a7c15ULL;
        for (size_t n = 0; n < nblocks; ++n) {
            for (size_t i = 0; i < kRate; ++i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                block[i] = (uint8_t)x;
            }
            absorb_blocks_base(a, block, 1);
            absorb_blocks_bmi(b, block, 1);
        }
        return std::memcmp(a, b, sizeof a) ? 1 : 0;
#else
        (void)nblocks;
        return 0;
#endif
    }

  private:
    static inline uint64_t rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

    // Whole blocks.  The round function is the same source compiled twice: for the baseline x86-64 ISA and with BMI1/BMI2 (ANDN for
    // chi's ~a & b, RORX for non-destructive rotations: fewer instructions and moves per round), chosen once per process.
    static void absorb_blocks(uint64_t* s, const uint8_t* data, size_t nblocks) {
#if defined(__x86_64__)
        static const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
        if (bmi) { absorb_blocks_bmi(s, data, nblocks); return; }
#endif
        absorb_blocks_base(s, data, nblocks);
    }
    static void absorb_blocks_base(uint64_t* s, const uint8_t* data, size_t nblocks) {
        for (size_t b = 0; b < nblocks; ++b) absorb_one(s, data + b * kRate);
    }
#if defined(__x86_64__)
    __attribute__((target("bmi,bmi2"))) static void absorb_blocks_bmi(uint64_t* s, const uint8_t* data, size_t nblocks) {
        for (size_t b = 0; b < nblocks; ++b) absorb_one(s, data + b * kRate);
    }
#endif
    static void absorb(uint64_t* s, const uint8_t* block) { absorb_blocks(s, block, 1); }

    __attribute__((always_inline)) static inline void absorb_one(uint64_t* s, const uint8_t* block) {
        for (size_t i = 0; i < kRate / 8; ++i) {
            uint64_t w;
            std::memcpy(&w, block + 8 * i, 8);
            s[i] ^= w;
        }
        permute(s);
    }

    __attribute__((always_inline)) static inline void permute(uint64_t* s) {
        static const uint64_t kRC[24] = {
            0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
            0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
            0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
            0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
            0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
            0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        uint64_t a00 = s[0], a10 = s[1], a20 = s[2], a30 = s[3], a40 = s[4];
        uint64_t a01 = s[5], a11 = s[6], a21 = s[7], a31 = s[8], a41 = s[9];
        uint64_t a02 = s[10], a12 = s[11], a22 = s[12], a32 = s[13], a42 = s[14];
        uint64_t a03 = s[15], a13 = s[16], a23 = s[17], a33 = s[18], a43 = s[19];
        uint64_t a04 = s[20], a14 = s[21], a24 = s[22], a34 = s[23], a44 = s[24];
        for (int r = 0; r < 24; ++r) {
            // theta
            uint64_t c0 = a00 ^ a01 ^ a02 ^ a03 ^ a04, c1 = a10 ^ a11 ^ a12 ^ a13 ^ a14;
            uint64_t c2 = a20 ^ a21 ^ a22 ^ a23 ^ a24, c3 = a30 ^ a31 ^ a32 ^ a33 ^ a34;
            uint64_t c4 = a40 ^ a41 ^ a42 ^ a43 ^ a44;
            uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1);
            uint64_t d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1);
            // rho + pi: b[y][2x+3y] = rot(a[x][y] ^ d[x], r[x][y]); names are bXY = b[x][y]
            uint64_t b00 = a00 ^ d0;
            uint64_t b13 = rotl(a01 ^ d0, 36), b21 = rotl(a02 ^ d0, 3), b34 = rotl(a03 ^ d0, 41), b42 = rotl(a04 ^ d0, 18);
            uint64_t b02 = rotl(a10 ^ d1, 1), b10 = rotl(a11 ^ d1, 44), b23 = rotl(a12 ^ d1, 10), b31 = rotl(a13 ^ d1, 45), b44 = rotl(a14 ^ d1, 2);
            uint64_t b04 = rotl(a20 ^ d2, 62), b12 = rotl(a21 ^ d2, 6), b20 = rotl(a22 ^ d2, 43), b33 = rotl(a23 ^ d2, 15), b41 = rotl(a24 ^ d2, 61);
            uint64_t b01 = rotl(a30 ^ d3, 28), b14 = rotl(a31 ^ d3, 55), b22 = rotl(a32 ^ d3, 25), b30 = rotl(a33 ^ d3, 21), b43 = rotl(a34 ^ d3, 56);
            uint64_t b03 = rotl(a40 ^ d4, 27), b11 = rotl(a41 ^ d4, 20), b24 = rotl(a42 ^ d4, 39), b32 = rotl(a43 ^ d4, 8), b40 = rotl(a44 ^ d4, 14);
            // chi (+ iota on lane 0,0)
            a00 = b00 ^ (~b10 & b20) ^ kRC[r]; a10 = b10 ^ (~b20 & b30); a20 = b20 ^ (~b30 & b40); a30 = b30 ^ (~b40 & b00); a40 = b40 ^ (~b00 & b10);
            a01 = b01 ^ (~b11 & b21); a11 = b11 ^ (~b21 & b31); a21 = b21 ^ (~b31 & b41); a31 = b31 ^ (~b41 & b01); a41 = b41 ^ (~b01 & b11);
            a02 = b02 ^ (~b12 & b22); a12 = b12 ^ (~b22 & b32); a22 = b22 ^ (~b32 & b42); a32 = b32 ^ (~b42 & b02); a42 = b42 ^ (~b02 & b12);
            a03 = b03 ^ (~b13 & b23); a13 = b13 ^ (~b23 & b33); a23 = b23 ^ (~b33 & b43); a33 = b33 ^ (~b43 & b03); a43 = b43 ^ (~b03 & b13);
            a04 = b04 ^ (~b14 & b24); a14 = b14 ^ (~b24 & b34); a24 = b24 ^ (~b34 & b44); a34 = b34 ^ (~b44 & b04); a44 = b44 ^ (~b04 & b14);
        }
        s[0] = a00; s[1] = a10; s[2] = a20; s[3] = a30; s[4] = a40;
        s[5] = a01; s[6] = a11; s[7] = a21; s[8] = a31; s[9] = a41;
        s[10] = a02; s[11] = a12; s[12] = a22; s[13] = a32; s[14] = a42;
        s[15] = a03; s[16] = a13; s[17] = a23; s[18] = a33; s[19] = a43;
        s[20] = a04; s[21] = a14; s[22] = a24; s[23] = a34; s[24] = a44;
    }

    uint64_t st_[25];
    uint8_t buf_[kRate];
    size_t fill_;
};

} // namespace zkh
