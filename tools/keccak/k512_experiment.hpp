// AVX-512 Keccak-f[1600]: one plane (row y: lanes x = 0..4) per zmm register
#pragma once
#include <immintrin.h>
#include <cstdint>
#include <cstring>
namespace k512 {
#define T512 __attribute__((target("avx512f"), always_inline)) static inline
static const uint64_t kRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
// rho offsets r[x][y], plane y holds x = 0..4
static const uint64_t kRho[5][8] = {{0, 1, 62, 28, 27, 0, 0, 0}, {36, 44, 6, 55, 20, 0, 0, 0}, {3, 10, 43, 25, 39, 0, 0, 0},
                                    {41, 45, 15, 21, 8, 0, 0, 0}, {18, 2, 61, 56, 14, 0, 0, 0}};
__attribute__((target("avx512f"))) static void absorb_blocks(uint64_t* s, const uint8_t* data, size_t nblocks) {
    const __mmask8 m5 = 0x1f;
    __m512i P0 = _mm512_maskz_loadu_epi64(m5, s), P1 = _mm512_maskz_loadu_epi64(m5, s + 5), P2 = _mm512_maskz_loadu_epi64(m5, s + 10),
            P3 = _mm512_maskz_loadu_epi64(m5, s + 15), P4 = _mm512_maskz_loadu_epi64(m5, s + 20);
    const __m512i R0 = _mm512_loadu_si512(kRho[0]), R1 = _mm512_loadu_si512(kRho[1]), R2 = _mm512_loadu_si512(kRho[2]),
                  R3 = _mm512_loadu_si512(kRho[3]), R4 = _mm512_loadu_si512(kRho[4]);
    const __m512i rotm1 = _mm512_setr_epi64(4, 0, 1, 2, 3, 5, 6, 7), rotp1 = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7),
                  rotp2 = _mm512_setr_epi64(2, 3, 4, 0, 1, 5, 6, 7);
    // pi: new plane y', slot x' = old plane x', lane (x' + 3 y') mod 5.  Two-source permutes: index bit 3 selects the second source.
    // pair (P0, P1) -> slots 0, 1; pair (P2, P3) -> slots 2, 3; P4 -> slot 4
    __m512i i01[5], i23[5], i4[5];
    for (int y = 0; y < 5; ++y) {
        uint64_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        a[0] = (0 + 3 * y) % 5;       // from P0
        a[1] = 8 + (1 + 3 * y) % 5;   // from P1
        b[2] = (2 + 3 * y) % 5;       // from P2
        b[3] = 8 + (3 + 3 * y) % 5;   // from P3
        c[4] = (4 + 3 * y) % 5;       // from P4
        i01[y] = _mm512_loadu_si512(a);
        i23[y] = _mm512_loadu_si512(b);
        i4[y] = _mm512_loadu_si512(c);
    }
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
        // rate = 17 lanes: planes 0..2 whole, plane 3 lanes 0, 1
        P0 = _mm512_xor_si512(P0, _mm512_maskz_loadu_epi64(m5, data));
        P1 = _mm512_xor_si512(P1, _mm512_maskz_loadu_epi64(m5, data + 40));
        P2 = _mm512_xor_si512(P2, _mm512_maskz_loadu_epi64(m5, data + 80));
        P3 = _mm512_xor_si512(P3, _mm512_maskz_loadu_epi64(0x03, data + 120));
        for (int r = 0; r < 24; ++r) {
            // theta
            const __m512i C = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(P0, P1, P2, 0x96), P3, P4, 0x96);
            const __m512i D = _mm512_xor_si512(_mm512_permutexvar_epi64(rotm1, C), _mm512_rol_epi64(_mm512_permutexvar_epi64(rotp1, C), 1));
            // rho
            const __m512i A0 = _mm512_rolv_epi64(_mm512_xor_si512(P0, D), R0), A1 = _mm512_rolv_epi64(_mm512_xor_si512(P1, D), R1),
                          A2 = _mm512_rolv_epi64(_mm512_xor_si512(P2, D), R2), A3 = _mm512_rolv_epi64(_mm512_xor_si512(P3, D), R3),
                          A4 = _mm512_rolv_epi64(_mm512_xor_si512(P4, D), R4);
            // pi + chi
#define PLANE(y, OUT)                                                                                                     \
    {                                                                                                                     \
        const __m512i t01 = _mm512_permutex2var_epi64(A0, i01[y], A1), t23 = _mm512_permutex2var_epi64(A2, i23[y], A3);   \
        const __m512i t4 = _mm512_permutexvar_epi64(i4[y], A4);                                                          \
        const __m512i B = _mm512_mask_blend_epi64(0x10, _mm512_mask_blend_epi64(0x0c, t01, t23), t4);                     \
        OUT = _mm512_ternarylogic_epi64(B, _mm512_permutexvar_epi64(rotp1, B), _mm512_permutexvar_epi64(rotp2, B), 0xD2); \
    }
            PLANE(0, P0) PLANE(1, P1) PLANE(2, P2) PLANE(3, P3) PLANE(4, P4)
#undef PLANE
            P0 = _mm512_xor_si512(P0, _mm512_maskz_set1_epi64(0x01, (long long)kRC[r]));
        }
    }
    _mm512_mask_storeu_epi64(s, m5, P0);
    _mm512_mask_storeu_epi64(s + 5, m5, P1);
    _mm512_mask_storeu_epi64(s + 10, m5, P2);
    _mm512_mask_storeu_epi64(s + 15, m5, P3);
    _mm512_mask_storeu_epi64(s + 20, m5, P4);
}
} // namespace k512
