// kernels_common.hpp -- device helpers, kernel templates and launcher helpers shared by kernels.hip (single-MLE path, tables) and
// deg2_kernels.hip (degree-2 / SumPolynomial branch).  Internal to the two translation units.
#pragma once
#include <algorithm>
#include <cstdlib>
#include "fq_dev.hpp"
#include "fq_host.hpp"
#include "kernels.h"

namespace zkk {

static constexpr int kThreads = 256;

// block-wide sums of NS accumulators; thread 0 of the block writes partials[blockIdx.x*NS + s]
template <int NS>
__device__ __forceinline__ void block_sum_store(fq (&acc)[NS], fq* __restrict__ partials) {
    __shared__ fq lds[NS][kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        fq w = fq_wave_sum(acc[s]);
        if (lane == 0) lds[s][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        fq t = lds[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) t = fq_add(t, lds[threadIdx.x][w]);
        fq_store(&partials[(size_t)blockIdx.x * NS + threadIdx.x], t);
    }
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// W.w[b] = eq(r,b) * 2^280 mod p as 10 x 28-bit limbs (made on the host), b < 2^M: kernel-argument weights of the small folds
template <int M> struct W28 { u32 w[1 << M][10]; };

// The eq weights of an m-variable pass (3 <= m <= 8) come from two half trees: eq over the top m/2 challenges (times 2^24: fq28_redc
// divides by 2^280, not 2^256) and eq over the rest, at most 16 entries each, and ONE product per weight joins them.  The trees are
// 30 host multiplications (make_eq_halves, ~1 us) and travel as the kernel's FIRST argument; until the end of round 3 every kernel
// built them itself from the challenges (m/2 dependent products and nine barriers at the head of every wide, tail and weight
// kernel: ~4 us on the critical path of each).  Same field elements either way.
struct EqHalves { fq half[2][16]; };
// entry idx (0..31: half[idx >> 4][idx & 15]) of the EqHalves that is the running kernel's FIRST argument, by a lane-indexed load from
// the kernel-argument segment (dynamic indexing of a by-value argument itself would be copied through scratch)
__device__ __forceinline__ fq eq_half_arg(int idx) {
    typedef const __attribute__((address_space(4))) u32x4* kptr;
    kptr p = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + 2 * idx;
    const u32x4 lo = p[0], hi = p[1];
    fq x;
#pragma unroll
    for (int k = 0; k < 4; ++k) { x.v[k] = lo[k]; x.v[4 + k] = hi[k]; }
    return x;
}
// thread b < 2^m of the workgroup gets eq(r, b) * 2^280 mod p as a plain integer (8 words, canonical); challenge l sits on bit m-1-l of b
__device__ __forceinline__ fq eq_weight_of_thread(int m) {
    const int ml = m - (m >> 1);
    if ((int)threadIdx.x < (1 << m)) return fq_mul(eq_half_arg((int)threadIdx.x >> ml), eq_half_arg(16 + ((int)threadIdx.x & ((1 << ml) - 1))));
    return fq_zero();
}
// wout[b * 10 + j] = limb j of eq(r, b) * 2^280 mod p, b < 2^m, by one workgroup of kThreads
__device__ __forceinline__ void eq_weights28_block(int m, u32* wout) {
    const fq w = eq_weight_of_thread(m);
    if ((int)threadIdx.x < (1 << m)) {
        u32 l28[10];
        fq28_unpack(w, l28);
#pragma unroll
        for (int j2 = 0; j2 < 10; ++j2) wout[threadIdx.x * 10 + j2] = l28[j2];
    }
}
// The wide pass with its 2^m read streams fed by LDS-DMA (nt policy), as k_block_sums_dma: a wave owns groups of 64
// consecutive outputs; the 2^m source rows of a group (row b = 64 entries at b * Lout + j, 2 KiB, two DMA instructions)
// run through the wave's ring of D slots, D - 1 rows in flight behind a counted vmcnt, the stream continuing across
// the wave's groups.  Lane l reads entry l of the row back from LDS and multiplies it into ITS output's unreduced
// accumulator with weight b (scalar loads).  No load buffers in registers, no barrier in the loop; the store of a
// finished group is followed by one vmcnt(0) (stores share the counter) -- once per 2^m rows.  `ring` is the kernel's
// only __shared__ object.  Lb (outputs per block) must be a multiple of 64.
template <bool BIG, int D>
__global__ void __launch_bounds__(kThreads) k_multifold_wide_dma(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout,
                                                                  const u32* __restrict__ wts, int m, int log_ob, fq* __restrict__ partials) {
    __shared__ u32x4 ring[kThreads / 64][D][128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wi = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob, j0 = (size_t)c * Lb;
    const unsigned ngroups = (unsigned)(Lb >> 6), gstride = gpb * (kThreads / 64);
    const unsigned nb = 1u << m;
    auto issue = [&](unsigned group, unsigned b, int slot) {
        const u32x4* p = reinterpret_cast<const u32x4*>(src + (size_t)b * Lout + j0 + ((size_t)group << 6)) + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)&ring[wave][slot][0], 16, 0, 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 64),
                                         (__attribute__((address_space(3))) void*)&ring[wave][slot][64], 16, 0, 2);
    };
    const unsigned g0 = wi * (kThreads / 64) + wave;
    unsigned ig = g0, ib = 0; // next row to issue
    int inflight = 0;
#pragma unroll
    for (int sl = 0; sl < D; ++sl)
        if (ig < ngroups) {
            issue(ig, ib, sl);
            ++inflight;
            if (++ib == nb) { ib = 0; ig += gstride; }
        }
    int slot = 0;
    fq run = fq_zero();
    for (unsigned g = g0; g < ngroups; g += gstride) {
        u64 acc[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0;
        const u32* __restrict__ w = wts; // wave-uniform: scalar loads
        for (unsigned b = 0; b < nb; ++b) {
            if (inflight == D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const u32x4 lo = ring[wave][slot][2 * lane], hi = ring[wave][slot][2 * lane + 1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // in registers before the slot is refilled
            --inflight;
            if (ig < ngroups) {
                issue(ig, ib, slot);
                ++inflight;
                if (++ib == nb) { ib = 0; ig += gstride; }
            }
            fq x;
#pragma unroll
            for (int k = 0; k < 4; ++k) { x.v[k] = lo[k]; x.v[4 + k] = hi[k]; }
            fq28_mac(acc, x, w);
            if ((b & 15) == 15) fq28_carry(acc); // every 16 products
            w += 10;
            slot = slot + 1 == D ? 0 : slot + 1;
        }
        const fq o = fq28_redc(acc);
        fq* out = dst + j0 + ((size_t)g << 6) + lane;
        if (BIG) fq_store_stream(out, o); else fq_store(out, o);
        // stores count in vmcnt too.  The counted waits would stay correct without this drain (outstanding loads <= the
        // counter), but measured without it the m = 2 passes of the degree-2 fold run 5 % slower and m = 8 the same
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        run = fq_add(run, o);
    }
    const fq wsum = fq_wave_sum(run);
    __syncthreads();
    fq* red = reinterpret_cast<fq*>(&ring[0][0][0]);
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        fq t = red[0];
#pragma unroll
        for (int k = 1; k < kThreads / 64; ++k) t = fq_add(t, red[k]);
        fq_store(&partials[blockIdx.x], t);
    }
}

static inline int grid_for(size_t work) {
    size_t b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}
static inline fq to_fq(const uint64_t r[4]) {
    fq f;
    for (int i = 0; i < 4; ++i) {
        f.v[2 * i] = (u32)r[i];
        f.v[2 * i + 1] = (u32)(r[i] >> 32);
    }
    return f;
}
// the two half trees of an m-variable pass's eq weights (EqHalves above): r = m challenges (Montgomery, 4 limbs each)
static inline EqHalves make_eq_halves(const uint64_t* r, int m, const uint64_t two24[4]) {
    EqHalves h{};
    const int mh = m >> 1, ml = m - mh; // 1 <= mh <= ml <= 4
    for (int side = 0; side < 2; ++side) {
        zkh::Fq t[16];
        t[0] = side ? zkh::one() : zkh::load(two24);
        const int nv = side ? ml : mh;
        for (int l = 0; l < nv; ++l) { // level l appends its variable as the new least significant bit
            const zkh::Fq x = zkh::load(r + 4 * ((side ? mh : 0) + l));
            for (int b = (1 << l) - 1; b >= 0; --b) {
                const zkh::Fq hi = zkh::mul(t[b], x);
                t[2 * b + 1] = hi;
                t[2 * b] = zkh::sub(t[b], hi);
            }
        }
        for (int b = 0; b < (1 << nv); ++b) h.half[side][b] = to_fq(t[b].l);
    }
    return h;
}

// grid for a pass whose work is cut into 2^log_ob blocks of Lb entries: a multiple of 2^log_ob
static inline int grid_blocks(size_t Lb, int log_ob) {
    size_t per = (Lb + kThreads - 1) / kThreads;
    size_t cap = (size_t)kMaxBlocks >> log_ob;
    if (per > cap) per = cap;
    if (per < 1) per = 1;
    return (int)(per << log_ob);
}
// four factor tables of the degree-2 branch
struct Tab4 { const fq* t[4]; };
struct Tab4w { fq* t[4]; };


// eq weights of a wide pass into wbuf (k_eq_weights28, kernels.hip): wbuf[b * 10 + j] = limb j of eq(r, b) * 2^280 mod p, b < 2^m
void launch_eq_weights28(const uint64_t* r, int m, const uint64_t two24[4], void* wbuf, hipStream_t st);
// the same weights as signed radix-256 digit tiles for the matrix-core folds (k_eq_weights_mfma, kernels.hip):
// wbuf[((bp * 4 + mt) * 64 + lane) * 4 + u], bp < 2^(m-1); every output of such a fold carries the constant kMfmaFoldFix
void launch_eq_weights_mfma(const uint64_t* r, int m, const uint64_t two24[4], void* wbuf, hipStream_t st);
// 128 (2^256 - 1)/255 - 2^27 ((2^512 - 1)/255) 2^-280 mod p (kernels.hip, "the wide pass on the matrix cores")
#define ZK_MFMA_FOLD_FIX {{0xe5f7fc63u, 0xf613786fu, 0x795bc6d2u, 0xbe62fc84u, 0x799d7fdeu, 0xed6bec43u, 0x8c88f1adu, 0x164bcf99u}}

} // namespace zkk
