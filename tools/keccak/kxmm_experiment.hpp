// Keccak-f[1600] with one lane per xmm register (AVX-512VL: 32 registers, vprolq, vpternlogq): scalar code in the vector register file
#pragma once
#include <immintrin.h>
#include <cstdint>
#include <cstring>
namespace kxmm {
static const uint64_t kRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#define X3(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0x96)
#define CHI(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0xD2)
#define ROL(a, n) _mm_rol_epi64(a, n)
#define LD(p) _mm_loadl_epi64((const __m128i*)(p))
__attribute__((target("avx512f,avx512vl"))) static void absorb_blocks(uint64_t* s, const uint8_t* data, size_t nblocks) {
    __m128i a00 = LD(s + 0), a10 = LD(s + 1), a20 = LD(s + 2), a30 = LD(s + 3), a40 = LD(s + 4);
    __m128i a01 = LD(s + 5), a11 = LD(s + 6), a21 = LD(s + 7), a31 = LD(s + 8), a41 = LD(s + 9);
    __m128i a02 = LD(s + 10), a12 = LD(s + 11), a22 = LD(s + 12), a32 = LD(s + 13), a42 = LD(s + 14);
    __m128i a03 = LD(s + 15), a13 = LD(s + 16), a23 = LD(s + 17), a33 = LD(s + 18), a43 = LD(s + 19);
    __m128i a04 = LD(s + 20), a14 = LD(s + 21), a24 = LD(s + 22), a34 = LD(s + 23), a44 = LD(s + 24);
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
        a00 = _mm_xor_si128(a00, LD(data + 0)); a10 = _mm_xor_si128(a10, LD(data + 8)); a20 = _mm_xor_si128(a20, LD(data + 16));
        a30 = _mm_xor_si128(a30, LD(data + 24)); a40 = _mm_xor_si128(a40, LD(data + 32));
        a01 = _mm_xor_si128(a01, LD(data + 40)); a11 = _mm_xor_si128(a11, LD(data + 48)); a21 = _mm_xor_si128(a21, LD(data + 56));
        a31 = _mm_xor_si128(a31, LD(data + 64)); a41 = _mm_xor_si128(a41, LD(data + 72));
        a02 = _mm_xor_si128(a02, LD(data + 80)); a12 = _mm_xor_si128(a12, LD(data + 88)); a22 = _mm_xor_si128(a22, LD(data + 96));
        a32 = _mm_xor_si128(a32, LD(data + 104)); a42 = _mm_xor_si128(a42, LD(data + 112));
        a03 = _mm_xor_si128(a03, LD(data + 120)); a13 = _mm_xor_si128(a13, LD(data + 128));
        for (int r = 0; r < 24; ++r) {
            const __m128i c0 = X3(X3(a00, a01, a02), a03, a04), c1 = X3(X3(a10, a11, a12), a13, a14), c2 = X3(X3(a20, a21, a22), a23, a24),
                          c3 = X3(X3(a30, a31, a32), a33, a34), c4 = X3(X3(a40, a41, a42), a43, a44);
            const __m128i d0 = _mm_xor_si128(c4, ROL(c1, 1)), d1 = _mm_xor_si128(c0, ROL(c2, 1)), d2 = _mm_xor_si128(c1, ROL(c3, 1)),
                          d3 = _mm_xor_si128(c2, ROL(c4, 1)), d4 = _mm_xor_si128(c3, ROL(c0, 1));
            const __m128i b00 = _mm_xor_si128(a00, d0);
            const __m128i b13 = ROL(_mm_xor_si128(a01, d0), 36), b21 = ROL(_mm_xor_si128(a02, d0), 3), b34 = ROL(_mm_xor_si128(a03, d0), 41), b42 = ROL(_mm_xor_si128(a04, d0), 18);
            const __m128i b02 = ROL(_mm_xor_si128(a10, d1), 1), b10 = ROL(_mm_xor_si128(a11, d1), 44), b23 = ROL(_mm_xor_si128(a12, d1), 10), b31 = ROL(_mm_xor_si128(a13, d1), 45), b44 = ROL(_mm_xor_si128(a14, d1), 2);
            const __m128i b04 = ROL(_mm_xor_si128(a20, d2), 62), b12 = ROL(_mm_xor_si128(a21, d2), 6), b20 = ROL(_mm_xor_si128(a22, d2), 43), b33 = ROL(_mm_xor_si128(a23, d2), 15), b41 = ROL(_mm_xor_si128(a24, d2), 61);
            const __m128i b01 = ROL(_mm_xor_si128(a30, d3), 28), b14 = ROL(_mm_xor_si128(a31, d3), 55), b22 = ROL(_mm_xor_si128(a32, d3), 25), b30 = ROL(_mm_xor_si128(a33, d3), 21), b43 = ROL(_mm_xor_si128(a34, d3), 56);
            const __m128i b03 = ROL(_mm_xor_si128(a40, d4), 27), b11 = ROL(_mm_xor_si128(a41, d4), 20), b24 = ROL(_mm_xor_si128(a42, d4), 39), b32 = ROL(_mm_xor_si128(a43, d4), 8), b40 = ROL(_mm_xor_si128(a44, d4), 14);
            a00 = _mm_xor_si128(CHI(b00, b10, b20), LD(&kRC[r])); a10 = CHI(b10, b20, b30); a20 = CHI(b20, b30, b40); a30 = CHI(b30, b40, b00); a40 = CHI(b40, b00, b10);
            a01 = CHI(b01, b11, b21); a11 = CHI(b11, b21, b31); a21 = CHI(b21, b31, b41); a31 = CHI(b31, b41, b01); a41 = CHI(b41, b01, b11);
            a02 = CHI(b02, b12, b22); a12 = CHI(b12, b22, b32); a22 = CHI(b22, b32, b42); a32 = CHI(b32, b42, b02); a42 = CHI(b42, b02, b12);
            a03 = CHI(b03, b13, b23); a13 = CHI(b13, b23, b33); a23 = CHI(b23, b33, b43); a33 = CHI(b33, b43, b03); a43 = CHI(b43, b03, b13);
            a04 = CHI(b04, b14, b24); a14 = CHI(b14, b24, b34); a24 = CHI(b24, b34, b44); a34 = CHI(b34, b44, b04); a44 = CHI(b44, b04, b14);
        }
    }
#define ST(p, v) _mm_storel_epi64((__m128i*)(p), v)
    ST(s + 0, a00); ST(s + 1, a10); ST(s + 2, a20); ST(s + 3, a30); ST(s + 4, a40);
    ST(s + 5, a01); ST(s + 6, a11); ST(s + 7, a21); ST(s + 8, a31); ST(s + 9, a41);
    ST(s + 10, a02); ST(s + 11, a12); ST(s + 12, a22); ST(s + 13, a32); ST(s + 14, a42);
    ST(s + 15, a03); ST(s + 16, a13); ST(s + 17, a23); ST(s + 18, a33); ST(s + 19, a43);
    ST(s + 20, a04); ST(s + 21, a14); ST(s + 22, a24); ST(s + 23, a34); ST(s + 24, a44);
}
#undef X3
#undef CHI
#undef ROL
#undef LD
#undef ST
} // namespace kxmm
