// fq_dev.hpp -- BN254 Fq on gfx950: 8 x 32-bit limbs, Montgomery form (R = 2^256).
//
// The same 32 bytes ark-ff keeps per element (4 x u64 LE limbs of a*R mod p)
// viewed as 8 x u32, so tables move between host and device without conversion.
// Multiplication is the generated v_mad_u64_u32 block in fq_mul_asm.inc; add/sub are
// v_add_co/v_addc_co chains the compiler derives from __builtin_addc/__builtin_subc.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct alignas(16) fq {
    u32 v[8];
};

// FQ_P_LIMBS, FQ_2P_LIMBS, FQ_4P_LIMBS, FQ_R2_LIMBS, FQ_R_LIMBS, fq_mul(), fq_mul_uniform()
#include "fq_mul_asm.inc"

__device__ __forceinline__ fq fq_zero() {
    fq z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z.v[j] = 0;
    return z;
}

// 16-byte vector loads/stores: two global_load_dwordx4 per element
__device__ __forceinline__ fq fq_load(const fq* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fq r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
// streamed-once data: nontemporal hint keeps it from displacing the partial-sum lines in L2
__device__ __forceinline__ fq fq_load_stream(const fq* p) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
    u32x4 lo = __builtin_nontemporal_load(q), hi = __builtin_nontemporal_load(q + 1);
    fq r;
    r.v[0] = lo[0]; r.v[1] = lo[1]; r.v[2] = lo[2]; r.v[3] = lo[3];
    r.v[4] = hi[0]; r.v[5] = hi[1]; r.v[6] = hi[2]; r.v[7] = hi[3];
    return r;
}
__device__ __forceinline__ void fq_store(fq* p, const fq& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// streamed-once output: nontemporal hint (measured +2.5 % on the 16-stream fold pass, profiles/r01_mfbench_*)
__device__ __forceinline__ void fq_store_stream(fq* p, const fq& a) {
    u32x4* q = reinterpret_cast<u32x4*>(p);
    u32x4 lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
    __builtin_nontemporal_store(lo, q);
    __builtin_nontemporal_store(hi, q + 1);
}

// a - p if a >= p else a   (a < 2p)
__device__ __forceinline__ fq fq_reduce_once(const fq& a) {
    const u32 p[8] = FQ_P_LIMBS;
    u32 d[8];
    u32 br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = __builtin_subc(a.v[j], p[j], br, &br);
    fq o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = br ? a.v[j] : d[j];
    return o;
}

// (a + b) mod p, inputs < p
__device__ __forceinline__ fq fq_add(const fq& a, const fq& b) {
    fq s;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = __builtin_addc(a.v[j], b.v[j], c, &c);
    return fq_reduce_once(s); // a+b < 2^255: no carry out of limb 7
}

// (a - b) mod p, inputs < p
__device__ __forceinline__ fq fq_sub(const fq& a, const fq& b) {
    const u32 p[8] = FQ_P_LIMBS;
    u32 d[8];
    u32 br = 0, c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = __builtin_subc(a.v[j], b.v[j], br, &br);
    u32 mask = 0u - br;
    fq o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = __builtin_addc(d[j], p[j] & mask, c, &c);
    return o;
}

// a + (p - b) in (0, 2p): a valid (unreduced) multiplicand for fq_mul, 16 ops instead of 24
__device__ __forceinline__ fq fq_sub_lazy(const fq& a, const fq& b) {
    const u32 p[8] = FQ_P_LIMBS;
    u32 d[8];
    u32 br = 0, c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = __builtin_subc(p[j], b.v[j], br, &br);
    fq o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = __builtin_addc(a.v[j], d[j], c, &c);
    return o;
}

// a + b as a plain 256-bit integer, no reduction (callers keep the sum below 2^256: up to four elements < p)
__device__ __forceinline__ fq fq_add_plain(const fq& a, const fq& b) {
    fq s;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = __builtin_addc(a.v[j], b.v[j], c, &c);
    return s;
}

// 2a mod p
__device__ __forceinline__ fq fq_dbl(const fq& a) { return fq_add(a, a); }

// the fold of evaluation_form.rs:184:  y1 + (y2 - y1) * r   (r wave-uniform, Montgomery form)
__device__ __forceinline__ fq fq_fold(const fq& y1, const fq& y2, const fq& r) {
    fq d = fq_sub_lazy(y2, y1);
    fq t = fq_mul_uniform(d, r);
    return fq_add(y1, t);
}

// x mod p for any 256-bit x (2^256 < 6p): conditional subtraction of 4p, 2p, p
__device__ __forceinline__ fq fq_reduce_256(const fq& x) {
    const u32 p4[8] = FQ_4P_LIMBS;
    const u32 p2[8] = FQ_2P_LIMBS;
    fq a = x;
    {
        u32 d[8], br = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = __builtin_subc(a.v[j], p4[j], br, &br);
#pragma unroll
        for (int j = 0; j < 8; ++j) a.v[j] = br ? a.v[j] : d[j];
    }
    {
        u32 d[8], br = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = __builtin_subc(a.v[j], p2[j], br, &br);
#pragma unroll
        for (int j = 0; j < 8; ++j) a.v[j] = br ? a.v[j] : d[j];
    }
    return fq_reduce_once(a);
}

__device__ __forceinline__ fq fq_to_mont(const fq& canonical) {
    const fq r2 = {FQ_R2_LIMBS};
    return fq_mul(canonical, r2);
}
__device__ __forceinline__ fq fq_from_mont(const fq& a) {
    fq one = fq_zero();
    one.v[0] = 1;
    return fq_mul(a, one);
}

// ---- radix-2^28 carry-free accumulation ------------------------------------------------
// A weighted sum  sum_b a_b * w_b  (the m-variable fold written as one inner product with the
// eq weights) needs ONE Montgomery reduction, not one per product, if the 512-bit products are
// accumulated unreduced.  With 32-bit limbs every v_mad_u64_u32 needs its carry-out caught
// (2 VALU ops per limb product, fq_mul_asm.inc); with 28-bit limbs a limb product is < 2^56 and a
// u64 column takes many of them with no carry handling at all: 1 VALU op
// (v_mad_u64_u32 acc, a, w, acc) per limb product, written in plain C++.
// The reduction is Montgomery in radix 2^28 over 10 limbs, i.e. it divides by 2^280: the host
// supplies the weights as w * 2^280 / 2^256 so the result is again in R = 2^256 Montgomery form.
//
// Limb count.  Bits 224..255 are word 7 as it stands, and for a canonical element (< p < 2^254) that word is below
// 0x30644e73 < 0.76 * 2^30: it is used as ONE top limb, so a product is 9 x 9 = 81 limb products, not the 100 of a
// uniform 10-limb split (the limb products are what the fold passes spend their VALU time on).  Column bound: the
// widest column (k = 8) takes two terms with a top limb (< 2 * 0.76 * 2^58) and seven plain ones (< 7 * 2^56), i.e.
// < 13.1 * 2^56 per product, so 16 products stay below 210 * 2^56 < 2^64 -- the carry cadence the callers use.
// Both operands must be canonical; stored weights keep the 10 x 28-bit layout (limbs 8 and 9 are joined on use).
#define FQ28_MASK 0x0fffffffu

// 8 limbs of 28 bits and the top word whole; l[9] is unused (zero) -- the array keeps the stored 10-word layout
__device__ __forceinline__ void fq28_unpack(const fq& a, u32 (&l)[10]) {
    l[0] = a.v[0] & FQ28_MASK;
    l[1] = ((a.v[0] >> 28) | (a.v[1] << 4)) & FQ28_MASK;
    l[2] = ((a.v[1] >> 24) | (a.v[2] << 8)) & FQ28_MASK;
    l[3] = ((a.v[2] >> 20) | (a.v[3] << 12)) & FQ28_MASK;
    l[4] = ((a.v[3] >> 16) | (a.v[4] << 16)) & FQ28_MASK;
    l[5] = ((a.v[4] >> 12) | (a.v[5] << 20)) & FQ28_MASK;
    l[6] = ((a.v[5] >> 8) | (a.v[6] << 24)) & FQ28_MASK;
    l[7] = a.v[6] >> 4;
    l[8] = a.v[7];
    l[9] = 0;
}
// acc (20 u64 columns, weight 2^(28k)) += a * w ; a already unpacked, w: stored weight (10 x 28-bit limbs)
__device__ __forceinline__ void fq28_mac_limbs(u64 (&acc)[20], const u32 (&l)[10], const u32* __restrict__ w) {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const u32 wj = j < 8 ? w[j] : (w[8] | (w[9] << 28));
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i + j] += (u64)l[i] * wj;
    }
}
// the same from the packed element
__device__ __forceinline__ void fq28_mac(u64 (&acc)[20], const fq& a, const u32* __restrict__ w) {
    u32 l[10];
    fq28_unpack(a, l);
    fq28_mac_limbs(acc, l, w);
}

// acc += a * b, both operands unpacked by fq28_unpack (variable x variable: the degree-2 sums).  The operands may be
// plain sums of up to four canonical entries (< 4p < 2^256, top word anything): the one product of the two top words
// could pass 2^63, so it is split over columns 16 and 17 -- 82 limb products.  Column bound: k = 8 takes two terms
// with a top word (< 2 * 2^28 * 0.76 * 2^32 = 24.2 * 2^56) and seven plain ones, < 31.2 * 2^56 per product:
// carry at least every 8 products.
__device__ __forceinline__ void fq28_mac_vv(u64 (&acc)[20], const u32 (&a)[10], const u32 (&b)[10]) {
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (i + j < 16) acc[i + j] += (u64)a[i] * b[j];
    acc[16] += (u64)a[8] * (b[8] & FQ28_MASK);
    acc[17] += (u64)a[8] * (b[8] >> 28);
}
// every column back below 2^28 (the top one keeps the overflow): call at least every 16 accumulated products
// (every 8 with fq28_mac_vv)
__device__ __forceinline__ void fq28_carry(u64 (&acc)[20]) {
#pragma unroll
    for (int k = 0; k < 19; ++k) {
        acc[k + 1] += acc[k] >> 28;
        acc[k] &= FQ28_MASK;
    }
}

// Montgomery reduction radix 2^28: returns acc * 2^-280 mod p, fully reduced, as 8 x u32.
// Requires acc < p * 2^280 * (1 + small): columns stay below 2^64 for up to 16 accumulated products.
__device__ __forceinline__ fq fq28_redc(u64 (&acc)[20]) {
    const u32 p28[10] = FQ_P28_LIMBS;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const u32 m = (((u32)acc[k] & FQ28_MASK) * FQ_NINV28) & FQ28_MASK;
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[k + i] += (u64)m * p28[i];
        acc[k + 1] += acc[k] >> 28; // low 28 bits of column k are now zero
    }
    // normalise columns 10..19 to 28-bit limbs (value < 2p < 2^255: column 19 ends up < 2^4)
    u32 l[10];
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        u64 v = acc[10 + k] + carry;
        l[k] = (u32)v & FQ28_MASK;
        carry = v >> 28;
    }
    fq r;
    r.v[0] = l[0] | (l[1] << 28);
    r.v[1] = (l[1] >> 4) | (l[2] << 24);
    r.v[2] = (l[2] >> 8) | (l[3] << 20);
    r.v[3] = (l[3] >> 12) | (l[4] << 16);
    r.v[4] = (l[4] >> 16) | (l[5] << 12);
    r.v[5] = (l[5] >> 20) | (l[6] << 8);
    r.v[6] = (l[6] >> 24) | (l[7] << 4);
    r.v[7] = l[8] | (l[9] << 28);
    return fq_reduce_once(r);
}

// ---- reductions ----------------------------------------------------------------------
__device__ __forceinline__ fq fq_shfl_down(const fq& a, int off) {
    fq o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = __shfl_down(a.v[j], off, 64);
    return o;
}
// sum over the 64 lanes of a wave; lane 0 holds the result
__device__ __forceinline__ fq fq_wave_sum(fq a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a = fq_add(a, fq_shfl_down(a, off));
    return a;
}
