// kernels_common.hpp -- device helpers, kernel templates and launcher helpers shared by kernels.hip (single-MLE path, tables) and
// deg2_kernels.hip (degree-2 / SumPolynomial branch).  Internal to the two translation units.
#pragma once
#include <algorithm>
#include <cstdlib>
#include "fq_dev.hpp"
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

struct ChalWide {
    fq r[kMaxFoldBatch];
    fq two24; // 2^24 in Montgomery form: eq * 2^256 -> eq * 2^280 (fq28_redc divides by 2^280)
};

// wout[b * 10 + j] = limb j of eq(r, b) * 2^280 mod p, b < 2^m (3 <= m <= 8); challenge l sits on bit m-1-l of b.
// By one workgroup of kThreads: the trees of the top m/2 and of the remaining variables are built side by side
// (threads 0-15 and 16-31; level l appends variable l as the new least significant bit; the top tree's root is 2^24,
// so every weight comes out multiplied by it) and one product per weight joins them: m/2 + 1 dependent products (5 at
// m = 8) instead of m + 1 -- this chain is on the critical path of every wide and tail pass.
__device__ __forceinline__ fq chal_pick(const ChalWide& ch, int idx) { // ch.r[idx] without dynamic indexing of kernel arguments
    fq r = ch.r[0];
#pragma unroll
    for (int k = 1; k < kMaxFoldBatch; ++k)
        if (idx == k) r = ch.r[k];
    return r;
}
// thread b < 2^m of the workgroup gets eq(r, b) * 2^280 mod p as a plain integer (8 words, canonical); every thread of the
// workgroup must call it (barriers inside)
__device__ __forceinline__ fq eq_weight_of_thread(const ChalWide& ch, int m) {
    __shared__ fq half[2][16];
    const int mh = m >> 1, ml = m - mh; // 1 <= mh <= ml <= 4
    const int t = threadIdx.x & 15, side = (threadIdx.x >> 4) & 1;
    const bool builder = threadIdx.x < 32;
    if (builder && t == 0) {
        const fq one = {FQ_R_LIMBS};
        half[side][0] = side ? one : ch.two24;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const bool act = builder && l < (side ? ml : mh) && t < (1 << l);
        fq v, hi;
        if (act) {
            v = half[side][t];
            hi = fq_mul(v, chal_pick(ch, side ? mh + l : l));
        }
        __syncthreads();
        if (act) {
            half[side][2 * t + 1] = hi;
            half[side][2 * t] = fq_sub(v, hi);
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < (1 << m)) return fq_mul(half[0][threadIdx.x >> ml], half[1][threadIdx.x & ((1 << ml) - 1)]);
    return fq_zero();
}
__device__ __forceinline__ void eq_weights28_block(const ChalWide& ch, int m, u32* wout) {
    const fq w = eq_weight_of_thread(ch, m);
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
