// keccak_host.hpp -- host Keccak-256 sponge for the Fiat-Shamir transcript
// (fiat_shamir/src/transcript.rs:6-42 uses sha3::Keccak256: rate 136 B, capacity 512,
// ORIGINAL Keccak multi-rate padding 0x01 .. 0x80 -- not NIST SHA3-256's 0x06).
// Product code: independent of the test oracle.  The state is kept as 25 named lanes and the
// round is written plane by plane so the compiler keeps everything in registers.
// Three builds of the round function, one chosen per process from the CPU's feature bits (absorb_blocks):
//   baseline x86-64 / any other ISA;  BMI1/BMI2 (ANDN, RORX);  and a HYBRID for CPUs with AVX-512VL + BMI: planes y = 0, 1 in general
//   registers, planes y = 2, 3, 4 one lane per xmm register (vprolq, vpternlogq for xor3 and chi) -- the two halves run on different
//   execution ports of the core; per round 10 column parities cross for theta (each side forms its own D) and 12 lanes for pi.
//   On the GPU box's EPYC: 675 / 780 / 985 MB/s (a 32 MiB table absorb 49.8 / 43.0 / 34.1 ms).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace zkh {

class Keccak256 {
  public:
    static constexpr size_t kRate = 136;
    static constexpr uint64_t kRCh[24] = {
            0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
            0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
            0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
            0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
            0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
            0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

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
        const int form = best_form();
        if (form == 0) return 0;
        uint8_t block[kRate];
        uint64_t a[25] = {0}, b[25] = {0}, h[25] = {0}, x = 0x9e3779b97f4a7c15ULL;
        for (size_t n = 0; n < nblocks; ++n) {
            for (size_t i = 0; i < kRate; ++i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                block[i] = (uint8_t)x;
            }
            absorb_blocks_base(a, block, 1);
            absorb_blocks_bmi(b, block, 1);
            if (form == 2) absorb_blocks_hybrid(h, block, 1);
        }
        if (std::memcmp(a, b, sizeof a)) return 1;
        return form == 2 && std::memcmp(a, h, sizeof a) ? 1 : 0;
#else
        (void)nblocks;
        return 0;
#endif
    }

  private:
    static inline uint64_t rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

    // Whole blocks, in the build this CPU runs fastest (chosen once per process): the scalar round function compiled for the baseline
    // x86-64 ISA and again with BMI1/BMI2 (ANDN for chi's ~a & b, RORX for non-destructive rotations: fewer instructions and moves
    // per round), or the hybrid scalar + vector form below.
    static void absorb_blocks(uint64_t* s, const uint8_t* data, size_t nblocks) {
#if defined(__x86_64__)
        static const int form = best_form();
        if (form == 2) { absorb_blocks_hybrid(s, data, nblocks); return; }
        if (form == 1) { absorb_blocks_bmi(s, data, nblocks); return; }
#endif
        absorb_blocks_base(s, data, nblocks);
    }
#if defined(__x86_64__)
    static int best_form() {
        const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
        if (bmi && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")) return 2;
        return bmi ? 1 : 0;
    }
#endif
    static void absorb_blocks_base(uint64_t* s, const uint8_t* data, size_t nblocks) {
        for (size_t b = 0; b < nblocks; ++b) absorb_one(s, data + b * kRate);
    }
#if defined(__x86_64__)
    __attribute__((target("bmi,bmi2"))) static void absorb_blocks_bmi(uint64_t* s, const uint8_t* data, size_t nblocks) {
        for (size_t b = 0; b < nblocks; ++b) absorb_one(s, data + b * kRate);
    }
#endif
#if defined(__x86_64__)
#define ZKH_X3(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0x96)
#define ZKH_CHI(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0xD2)
#define ZKH_VX(a, b) _mm_xor_si128(a, b)
#define ZKH_VROL(a, n) _mm_rol_epi64(a, n)
#define ZKH_LD(p) _mm_loadl_epi64((const __m128i*)(p))
#define ZKH_ST(p, v) _mm_storel_epi64((__m128i*)(p), v)
#define ZKH_TOV(s) _mm_cvtsi64_si128((long long)(s))
#define ZKH_TOS(v) ((uint64_t)_mm_cvtsi128_si64(v))
    static inline uint64_t ld64(const uint8_t* p) { uint64_t w; std::memcpy(&w, p, 8); return w; }
    // the hybrid build: see the header comment.  bXY = b[x][y] after rho and pi; s* live in general registers, w* in xmm registers,
    // t* / u* are the lanes that change sides
    __attribute__((target("avx512f,avx512vl,bmi,bmi2"))) static void absorb_blocks_hybrid(uint64_t* s, const uint8_t* data, size_t nblocks) {
        uint64_t a00 = s[0], a10 = s[1], a20 = s[2], a30 = s[3], a40 = s[4];
        uint64_t a01 = s[5], a11 = s[6], a21 = s[7], a31 = s[8], a41 = s[9];
        __m128i a02 = ZKH_LD(s + 10), a12 = ZKH_LD(s + 11), a22 = ZKH_LD(s + 12), a32 = ZKH_LD(s + 13), a42 = ZKH_LD(s + 14);
        __m128i a03 = ZKH_LD(s + 15), a13 = ZKH_LD(s + 16), a23 = ZKH_LD(s + 17), a33 = ZKH_LD(s + 18), a43 = ZKH_LD(s + 19);
        __m128i a04 = ZKH_LD(s + 20), a14 = ZKH_LD(s + 21), a24 = ZKH_LD(s + 22), a34 = ZKH_LD(s + 23), a44 = ZKH_LD(s + 24);
        for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
            a00 ^= ld64(data + 0); a10 ^= ld64(data + 8); a20 ^= ld64(data + 16); a30 ^= ld64(data + 24); a40 ^= ld64(data + 32);
            a01 ^= ld64(data + 40); a11 ^= ld64(data + 48); a21 ^= ld64(data + 56); a31 ^= ld64(data + 64); a41 ^= ld64(data + 72);
            a02 = ZKH_VX(a02, ZKH_LD(data + 80)); a12 = ZKH_VX(a12, ZKH_LD(data + 88)); a22 = ZKH_VX(a22, ZKH_LD(data + 96)); a32 = ZKH_VX(a32, ZKH_LD(data + 104)); a42 = ZKH_VX(a42, ZKH_LD(data + 112));
            a03 = ZKH_VX(a03, ZKH_LD(data + 120)); a13 = ZKH_VX(a13, ZKH_LD(data + 128));
            for (int r = 0; r < 24; ++r) {
                // theta: column parities, the vector planes' share crosses to the scalar side, D goes back
                const __m128i cv0 = ZKH_X3(a02, a03, a04), cv1 = ZKH_X3(a12, a13, a14), cv2 = ZKH_X3(a22, a23, a24), cv3 = ZKH_X3(a32, a33, a34), cv4 = ZKH_X3(a42, a43, a44);
                const uint64_t c0 = a00 ^ a01 ^ ZKH_TOS(cv0), c1 = a10 ^ a11 ^ ZKH_TOS(cv1), c2 = a20 ^ a21 ^ ZKH_TOS(cv2), c3 = a30 ^ a31 ^ ZKH_TOS(cv3), c4 = a40 ^ a41 ^ ZKH_TOS(cv4);
                const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1);
                const __m128i e0 = ZKH_VX(cv0, ZKH_TOV(a00 ^ a01)), e1 = ZKH_VX(cv1, ZKH_TOV(a10 ^ a11)), e2 = ZKH_VX(cv2, ZKH_TOV(a20 ^ a21)), e3 = ZKH_VX(cv3, ZKH_TOV(a30 ^ a31)), e4 = ZKH_VX(cv4, ZKH_TOV(a40 ^ a41));
                const __m128i v0 = ZKH_VX(e4, ZKH_VROL(e1, 1)), v1 = ZKH_VX(e0, ZKH_VROL(e2, 1)), v2 = ZKH_VX(e1, ZKH_VROL(e3, 1)), v3 = ZKH_VX(e2, ZKH_VROL(e4, 1)), v4 = ZKH_VX(e3, ZKH_VROL(e0, 1));
                // rho + pi in the source's domain; bXY = b[x][y], plane y = 0, 1 scalar (s*), 2, 3, 4 vector (w*)
                const uint64_t s00 = a00 ^ d0;
                const uint64_t t13 = rotl(a01 ^ d0, 36);              // -> plane 3 (vector)
                const __m128i u21 = ZKH_VROL(ZKH_VX(a02, v0), 3);             // -> plane 1 (scalar)
                const __m128i w34 = ZKH_VROL(ZKH_VX(a03, v0), 41), w42 = ZKH_VROL(ZKH_VX(a04, v0), 18);
                const uint64_t t02 = rotl(a10 ^ d1, 1);               // -> plane 2 (vector)
                const uint64_t s10 = rotl(a11 ^ d1, 44);
                const __m128i w23 = ZKH_VROL(ZKH_VX(a12, v1), 10);
                const __m128i u31 = ZKH_VROL(ZKH_VX(a13, v1), 45);            // -> plane 1 (scalar)
                const __m128i w44 = ZKH_VROL(ZKH_VX(a14, v1), 2);
                const uint64_t t04 = rotl(a20 ^ d2, 62), t12 = rotl(a21 ^ d2, 6); // -> planes 4, 2 (vector)
                const __m128i u20 = ZKH_VROL(ZKH_VX(a22, v2), 43);            // -> plane 0 (scalar)
                const __m128i w33 = ZKH_VROL(ZKH_VX(a23, v2), 15);
                const __m128i u41 = ZKH_VROL(ZKH_VX(a24, v2), 61);            // -> plane 1 (scalar)
                const uint64_t s01 = rotl(a30 ^ d3, 28);
                const uint64_t t14 = rotl(a31 ^ d3, 55);              // -> plane 4 (vector)
                const __m128i w22 = ZKH_VROL(ZKH_VX(a32, v3), 25);
                const __m128i u30 = ZKH_VROL(ZKH_VX(a33, v3), 21);            // -> plane 0 (scalar)
                const __m128i w43 = ZKH_VROL(ZKH_VX(a34, v3), 56);
                const uint64_t t03 = rotl(a40 ^ d4, 27);              // -> plane 3 (vector)
                const uint64_t s11 = rotl(a41 ^ d4, 20);
                const __m128i w24 = ZKH_VROL(ZKH_VX(a42, v4), 39), w32 = ZKH_VROL(ZKH_VX(a43, v4), 8);
                const __m128i u40 = ZKH_VROL(ZKH_VX(a44, v4), 14);            // -> plane 0 (scalar)
                // the twelve crossings
                const uint64_t s20 = ZKH_TOS(u20), s30 = ZKH_TOS(u30), s40 = ZKH_TOS(u40), s21 = ZKH_TOS(u21), s31 = ZKH_TOS(u31), s41 = ZKH_TOS(u41);
                const __m128i w02 = ZKH_TOV(t02), w12 = ZKH_TOV(t12), w03 = ZKH_TOV(t03), w13 = ZKH_TOV(t13), w04 = ZKH_TOV(t04), w14 = ZKH_TOV(t14);
                // chi (+ iota)
                a00 = s00 ^ (~s10 & s20) ^ kRCh[r]; a10 = s10 ^ (~s20 & s30); a20 = s20 ^ (~s30 & s40); a30 = s30 ^ (~s40 & s00); a40 = s40 ^ (~s00 & s10);
                a01 = s01 ^ (~s11 & s21); a11 = s11 ^ (~s21 & s31); a21 = s21 ^ (~s31 & s41); a31 = s31 ^ (~s41 & s01); a41 = s41 ^ (~s01 & s11);
                a02 = ZKH_CHI(w02, w12, w22); a12 = ZKH_CHI(w12, w22, w32); a22 = ZKH_CHI(w22, w32, w42); a32 = ZKH_CHI(w32, w42, w02); a42 = ZKH_CHI(w42, w02, w12);
                a03 = ZKH_CHI(w03, w13, w23); a13 = ZKH_CHI(w13, w23, w33); a23 = ZKH_CHI(w23, w33, w43); a33 = ZKH_CHI(w33, w43, w03); a43 = ZKH_CHI(w43, w03, w13);
                a04 = ZKH_CHI(w04, w14, w24); a14 = ZKH_CHI(w14, w24, w34); a24 = ZKH_CHI(w24, w34, w44); a34 = ZKH_CHI(w34, w44, w04); a44 = ZKH_CHI(w44, w04, w14);
            }
        }
        s[0] = a00; s[1] = a10; s[2] = a20; s[3] = a30; s[4] = a40;
        s[5] = a01; s[6] = a11; s[7] = a21; s[8] = a31; s[9] = a41;
        ZKH_ST(s + 10, a02); ZKH_ST(s + 11, a12); ZKH_ST(s + 12, a22); ZKH_ST(s + 13, a32); ZKH_ST(s + 14, a42);
        ZKH_ST(s + 15, a03); ZKH_ST(s + 16, a13); ZKH_ST(s + 17, a23); ZKH_ST(s + 18, a33); ZKH_ST(s + 19, a43);
        ZKH_ST(s + 20, a04); ZKH_ST(s + 21, a14); ZKH_ST(s + 22, a24); ZKH_ST(s + 23, a34); ZKH_ST(s + 24, a44);
    }
#undef ZKH_X3
#undef ZKH_CHI
#undef ZKH_VX
#undef ZKH_VROL
#undef ZKH_LD
#undef ZKH_ST
#undef ZKH_TOV
#undef ZKH_TOS
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
        const uint64_t* const kRC = kRCh;
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
