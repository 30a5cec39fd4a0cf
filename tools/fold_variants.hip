// tools/fold_variants.hip -- fold-pass kernel forms that were measured against the ones the library ships and lost
// (profiles/r01_mfbench_fold_pass_ablation.txt).  Kept out of zk_amd/csrc/kernels.hip since round 2;
// tools/mfbench.hip includes this file after kernels.hip so the ablations still build.  Not part of libzkmi.so.
#pragma once
namespace zkk {

// ---- k_multifold<M, DB> (register double-buffered fold tree, variant 2)
// src: Lout << M entries, dst: Lout entries.  DB: a second register buffer keeps the next step's
// 2^M element loads in flight under the current step's arithmetic.
template <int M, bool DB>
__global__ void __launch_bounds__(kThreads) k_multifold(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout, Chal<M> ch,
                                                         int log_ob, fq* __restrict__ partials) {
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wl = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob;
    const size_t j0 = (size_t)c * Lb;
    const size_t stride = (size_t)gpb * kThreads;
    fq acc[1] = {fq_zero()};
    size_t i = (size_t)wl * kThreads + threadIdx.x;
    if (DB) {
        fq A[1 << M], B[1 << M];
        if (i < Lb) mf_load<M>(A, src, Lout, j0 + i);
        while (i < Lb) {
            size_t n = i + stride;
            if (n < Lb) mf_load<M>(B, src, Lout, j0 + n);
            asm volatile("" ::: "memory"); // keep the prefetch above the arithmetic
            fq o = mf_tree<M>(A, ch);
            fq_store(dst + j0 + i, o);
            acc[0] = fq_add(acc[0], o);
            i = n;
            if (i >= Lb) break;
            n = i + stride;
            if (n < Lb) mf_load<M>(A, src, Lout, j0 + n);
            asm volatile("" ::: "memory");
            o = mf_tree<M>(B, ch);
            fq_store(dst + j0 + i, o);
            acc[0] = fq_add(acc[0], o);
            i = n;
        }
    } else {
        for (; i < Lb; i += stride) {
            fq A[1 << M];
            mf_load<M>(A, src, Lout, j0 + i);
            fq o = mf_tree<M>(A, ch);
            fq_store(dst + j0 + i, o);
            acc[0] = fq_add(acc[0], o);
        }
    }
    block_sum_store<1>(acc, partials);
}


// ---- k_multifold_roll (rolling window of loads, variants 4-6)
// Rolling-window form of the same inner product: every lane keeps D element loads in flight at all
// times (slot b % D is refilled with element b + D -- of the next output once b + D >= 2^M -- right after
// it has been unpacked), so the loads in flight no longer drop to one register buffer while the other is
// being multiplied, and the window, not 2 x G, sets the register cost.
// one element of the window, unrolled at the source level (the optimizer re-rolls a #pragma unroll loop here)
template <int M, int D, int B>
struct RollStep {
    static __device__ __forceinline__ void run(fq (&buf)[D], u64 (&acc)[20], const fq* __restrict__ src, size_t Lout, size_t cur, size_t nxt,
                                               const W28<M>& W) {
        constexpr int NB = 1 << M;
        u32 l[10];
        fq28_unpack(buf[B % D], l);
        if constexpr (B + D < NB) buf[B % D] = fq_load_stream(src + (size_t)(B + D) * Lout + cur);
        else buf[B % D] = fq_load_stream(src + (size_t)(B + D - NB) * Lout + nxt);
        asm volatile("" ::: "memory"); // keep the refill where it is issued
        fq28_mac_limbs(acc, l, W.w[B]);
        if constexpr (B + 1 < NB) RollStep<M, D, B + 1>::run(buf, acc, src, Lout, cur, nxt, W);
    }
};

template <int M, int D>
__global__ void __launch_bounds__(kThreads) k_multifold_roll(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout, W28<M> W,
                                                              int log_ob, fq* __restrict__ partials) {
    constexpr int NB = 1 << M;
    static_assert(NB % D == 0, "window must divide the batch");
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wl = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob;
    const size_t j0 = (size_t)c * Lb;
    const size_t stride = (size_t)gpb * kThreads;
    fq acc1[1] = {fq_zero()};
    size_t i = (size_t)wl * kThreads + threadIdx.x;
    fq buf[D];
    if (i < Lb) {
#pragma unroll
        for (int s = 0; s < D; ++s) buf[s] = fq_load_stream(src + (size_t)s * Lout + j0 + i);
    }
    while (i < Lb) {
        const size_t n = i + stride;
        const size_t nn = n < Lb ? n : i; // past the end: re-read this output's elements (unused) -- keeps the loop branch-free
        u64 acc[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0;
        RollStep<M, D, 0>::run(buf, acc, src, Lout, j0 + i, j0 + nn, W);
        fq o = fq28_redc(acc);
        fq_store(dst + j0 + i, o);
        acc1[0] = fq_add(acc1[0], o);
        i = n;
    }
    block_sum_store<1>(acc1, partials);
}


// ---- k_multifold_wide<BIG> (register-load form of the wide pass)
template <bool BIG> // BIG: the output is beyond the caches: streaming stores
__global__ void __launch_bounds__(kThreads) k_multifold_wide(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout,
                                                              const u32* __restrict__ wts, int m, int log_ob, fq* __restrict__ partials) {
    constexpr int G = 2;
    const int ng = (1 << m) / G; // m >= 3: ng is even
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wi = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob;
    const size_t j0 = (size_t)c * Lb;
    const unsigned stride = gpb * kThreads; // in-block indices fit 32 bits (the launcher checks): one register each
    const size_t gstep = (size_t)G * Lout;
    // the running sum of this lane's outputs lives in LDS: touched once per output (2^m elements apart), and the
    // 8 registers it would hold are the difference between 3 and 4 waves per SIMD
    __shared__ fq run[kThreads];
    run[threadIdx.x] = fq_zero();
    unsigned i = wi * kThreads + threadIdx.x;
    fq A[G], B[G];
    if (i < Lb) lazy_load<G>(A, src, Lout, j0 + i, 0);
    while (i < Lb) {
        const unsigned n = i + stride;
        u64 acc[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0;
        const fq* __restrict__ p = src + j0 + i;
        const u32* __restrict__ w = wts; // wave-uniform: scalar loads
        for (int g = 0; g < ng; g += 2) {
            lazy_load<G>(B, p + gstep, Lout, 0, 0);
#pragma unroll
            for (int b = 0; b < G; ++b) fq28_mac(acc, A[b], w + b * 10);
            if (g + 2 < ng) lazy_load<G>(A, p + 2 * gstep, Lout, 0, 0);
            else if (n < Lb) lazy_load<G>(A, src, Lout, j0 + n, 0);
#pragma unroll
            for (int b = 0; b < G; ++b) fq28_mac(acc, B[b], w + (G + b) * 10);
            // carry after every 16 products (the bound fq28_redc is written for): every column back below 2^28; the top one
            // keeps the overflow, which is tiny -- the whole sum is < 2^8 p^2.  Uniform branch.
            if ((g & 6) == 6) {
#pragma unroll
                for (int k = 0; k < 19; ++k) {
                    acc[k + 1] += acc[k] >> 28;
                    acc[k] &= FQ28_MASK;
                }
            }
            p += 2 * gstep;
            w += 2 * G * 10;
        }
        fq o = fq28_redc(acc);
        if (BIG) fq_store_stream(dst + j0 + i, o); else fq_store(dst + j0 + i, o);
        run[threadIdx.x] = fq_add(run[threadIdx.x], o);
        i = n;
    }
    fq acc1[1] = {run[threadIdx.x]};
    block_sum_store<1>(acc1, partials);
}


// ---- launcher of k_multifold_roll
template <int M, int D>
static int launch_multifold_roll_m(const void* src, void* dst, size_t Lout, const uint32_t* w28, int log_ob, void* partials, hipStream_t st) {
    W28<M> W;
    for (int b = 0; b < (1 << M); ++b)
        for (int j = 0; j < 10; ++j) W.w[b][j] = w28[b * 10 + j];
    int g = grid_blocks(Lout >> log_ob, log_ob);
    hipLaunchKernelGGL((k_multifold_roll<M, D>), dim3(g), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, Lout, W, log_ob, (fq*)partials);
    return g;
}

// ---- grid_blocks_wide
// grid for the wide pass: up to 2 * kMaxBlocks workgroups (16 per CU: the kernel runs 4 waves per SIMD)
static inline int grid_blocks_wide(size_t Lb, int log_ob) {
    size_t per = (Lb + kThreads - 1) / kThreads;
    size_t cap = (size_t)(2 * kMaxBlocks) >> log_ob; // 1x ... 8x kMaxBlocks measure the same (5.53 - 5.57 TB/s)
    if (per > cap) per = cap;
    if (per < 1) per = 1;
    return (int)(per << log_ob);
}

} // namespace zkk
