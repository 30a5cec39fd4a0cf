// Keccak-f[1600], HYBRID: planes y = 0, 1 in general registers (BMI: ANDN, RORX), planes y = 2, 3, 4 one lane per xmm register
// (AVX-512VL: vprolq, vpternlogq) -- the two halves run on different execution ports of the core.  Per round 10 values cross for
// theta (5 column parities one way, 5 D values back) and 12 lanes cross for pi.
#include <immintrin.h>
#include <cstdint>
#include <cstring>
namespace khybB {
static const uint64_t kRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#define X3(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0x96)
#define CHI(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0xD2)
#define VX(a, b) _mm_xor_si128(a, b)
#define VROL(a, n) _mm_rol_epi64(a, n)
#define LD(p) _mm_loadl_epi64((const __m128i*)(p))
#define TOV(s) _mm_cvtsi64_si128((long long)(s))
#define TOS(v) ((uint64_t)_mm_cvtsi128_si64(v))
static inline uint64_t rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
static inline uint64_t ld64(const uint8_t* p) { uint64_t w; std::memcpy(&w, p, 8); return w; }
__attribute__((target("avx512f,avx512vl,bmi,bmi2"))) static void absorb_blocks(uint64_t* s, const uint8_t* data, size_t nblocks) {
    uint64_t a00 = s[0], a10 = s[1], a20 = s[2], a30 = s[3], a40 = s[4];
    uint64_t a01 = s[5], a11 = s[6], a21 = s[7], a31 = s[8], a41 = s[9];
    uint64_t a02 = s[10], a12 = s[11], a22 = s[12], a32 = s[13], a42 = s[14];
    __m128i a03 = LD(s + 15), a13 = LD(s + 16), a23 = LD(s + 17), a33 = LD(s + 18), a43 = LD(s + 19);
    __m128i a04 = LD(s + 20), a14 = LD(s + 21), a24 = LD(s + 22), a34 = LD(s + 23), a44 = LD(s + 24);
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
        a00 ^= ld64(data + 0); a10 ^= ld64(data + 8); a20 ^= ld64(data + 16); a30 ^= ld64(data + 24); a40 ^= ld64(data + 32);
        a01 ^= ld64(data + 40); a11 ^= ld64(data + 48); a21 ^= ld64(data + 56); a31 ^= ld64(data + 64); a41 ^= ld64(data + 72);
        a02 ^= ld64(data + 80); a12 ^= ld64(data + 88); a22 ^= ld64(data + 96); a32 ^= ld64(data + 104); a42 ^= ld64(data + 112);
        a03 = VX(a03, LD(data + 120)); a13 = VX(a13, LD(data + 128));
        for (int r = 0; r < 24; ++r) {
            const __m128i cv0 = VX(a03, a04), cv1 = VX(a13, a14), cv2 = VX(a23, a24), cv3 = VX(a33, a34), cv4 = VX(a43, a44);
            const uint64_t c0 = a00 ^ a01 ^ a02 ^ TOS(cv0), c1 = a10 ^ a11 ^ a12 ^ TOS(cv1), c2 = a20 ^ a21 ^ a22 ^ TOS(cv2), c3 = a30 ^ a31 ^ a32 ^ TOS(cv3),
                           c4 = a40 ^ a41 ^ a42 ^ TOS(cv4);
            const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1);
            const __m128i v0 = TOV(d0), v1 = TOV(d1), v2 = TOV(d2), v3 = TOV(d3), v4 = TOV(d4);
            const uint64_t s00 = a00 ^ d0;
            const uint64_t t13 = rotl(a01 ^ d0, 36);              // -> plane 3 (vector)
            const uint64_t s21 = rotl(a02 ^ d0, 3);
            const __m128i w34 = VROL(VX(a03, v0), 41);
            const __m128i u42 = VROL(VX(a04, v0), 18);            // -> plane 2 (scalar)
            const uint64_t s02 = rotl(a10 ^ d1, 1), s10 = rotl(a11 ^ d1, 44);
            const uint64_t t23 = rotl(a12 ^ d1, 10);              // -> plane 3 (vector)
            const __m128i u31 = VROL(VX(a13, v1), 45);            // -> plane 1 (scalar)
            const __m128i w44 = VROL(VX(a14, v1), 2);
            const uint64_t t04 = rotl(a20 ^ d2, 62);              // -> plane 4 (vector)
            const uint64_t s12 = rotl(a21 ^ d2, 6), s20 = rotl(a22 ^ d2, 43);
            const __m128i w33 = VROL(VX(a23, v2), 15);
            const __m128i u41 = VROL(VX(a24, v2), 61);            // -> plane 1 (scalar)
            const uint64_t s01 = rotl(a30 ^ d3, 28);
            const uint64_t t14 = rotl(a31 ^ d3, 55);              // -> plane 4 (vector)
            const uint64_t s22 = rotl(a32 ^ d3, 25);
            const __m128i u30 = VROL(VX(a33, v3), 21);            // -> plane 0 (scalar)
            const __m128i w43 = VROL(VX(a34, v3), 56);
            const uint64_t t03 = rotl(a40 ^ d4, 27);              // -> plane 3 (vector)
            const uint64_t s11 = rotl(a41 ^ d4, 20);
            const uint64_t t24 = rotl(a42 ^ d4, 39);              // -> plane 4 (vector)
            const __m128i u32 = VROL(VX(a43, v4), 8);             // -> plane 2 (scalar)
            const __m128i u40 = VROL(VX(a44, v4), 14);            // -> plane 0 (scalar)
            const uint64_t s42 = TOS(u42), s31 = TOS(u31), s41 = TOS(u41), s30 = TOS(u30), s32 = TOS(u32), s40 = TOS(u40);
            const __m128i w13 = TOV(t13), w23 = TOV(t23), w04 = TOV(t04), w14 = TOV(t14), w03 = TOV(t03), w24 = TOV(t24);
            a00 = s00 ^ (~s10 & s20) ^ kRC[r]; a10 = s10 ^ (~s20 & s30); a20 = s20 ^ (~s30 & s40); a30 = s30 ^ (~s40 & s00); a40 = s40 ^ (~s00 & s10);
            a01 = s01 ^ (~s11 & s21); a11 = s11 ^ (~s21 & s31); a21 = s21 ^ (~s31 & s41); a31 = s31 ^ (~s41 & s01); a41 = s41 ^ (~s01 & s11);
            a02 = s02 ^ (~s12 & s22); a12 = s12 ^ (~s22 & s32); a22 = s22 ^ (~s32 & s42); a32 = s32 ^ (~s42 & s02); a42 = s42 ^ (~s02 & s12);
            a03 = CHI(w03, w13, w23); a13 = CHI(w13, w23, w33); a23 = CHI(w23, w33, w43); a33 = CHI(w33, w43, w03); a43 = CHI(w43, w03, w13);
            a04 = CHI(w04, w14, w24); a14 = CHI(w14, w24, w34); a24 = CHI(w24, w34, w44); a34 = CHI(w34, w44, w04); a44 = CHI(w44, w04, w14);
        }
    }
#define ST(p, v) _mm_storel_epi64((__m128i*)(p), v)
    s[0] = a00; s[1] = a10; s[2] = a20; s[3] = a30; s[4] = a40;
    s[5] = a01; s[6] = a11; s[7] = a21; s[8] = a31; s[9] = a41;
    s[10] = a02; s[11] = a12; s[12] = a22; s[13] = a32; s[14] = a42;
    ST(s + 15, a03); ST(s + 16, a13); ST(s + 17, a23); ST(s + 18, a33); ST(s + 19, a43);
    ST(s + 20, a04); ST(s + 21, a14); ST(s + 22, a24); ST(s + 23, a34); ST(s + 24, a44);
}
#undef X3
#undef CHI
#undef VX
#undef VROL
#undef LD
#undef TOV
#undef TOS
#undef ST
} // namespace khybB
