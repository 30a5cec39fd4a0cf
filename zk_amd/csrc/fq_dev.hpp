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
