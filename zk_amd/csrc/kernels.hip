// kernels.hip -- gfx950 kernels of the multilinear sumcheck prover.
//
// Data layout in HBM: a table is N contiguous 32-byte elements (8 x u32 Montgomery limbs,
// index order = MultiLinearPolynomial::evaluation_points, variable 0 = most significant index
// bit; evaluation_form.rs:40-42).  The prover always binds variable 0, so a round pairs entry
// j with entry j + N/2 (prover.rs:29-41): two contiguous streams, one 32-byte element per lane,
// two global_load_dwordx4 each.
//
// All kernels are HBM-streaming integer kernels: no LDS tiling (no reuse), no MFMA.  Grids are
// capped at kMaxBlocks x 256 threads and grid-stride, block -> per-block partial sums, one
// small second-stage kernel folds the partials (fixed order: results are bit-reproducible).
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

// ---- round reduction, MLE branch (sumcheck/src/prover.rs:108-116) ----------------------
// partials[b*2+0] = sum over this block's share of T[0 .. half), [b*2+1] of T[half .. 2*half)
__global__ void __launch_bounds__(kThreads) k_round_sums(const fq* __restrict__ T, size_t half, fq* __restrict__ partials) {
    fq acc[2] = {fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < half; i += stride) {
        fq lo = fq_load_stream(T + i), hi = fq_load_stream(T + half + i);
        acc[0] = fq_add(acc[0], lo);
        acc[1] = fq_add(acc[1], hi);
    }
    block_sum_store<2>(acc, partials);
}

// ---- fused fold + next round's reduction ------------------------------------------------
// src has 4*q entries.  dst[i] = fold(src[i], src[i+2q]), dst[i+q] = fold(src[i+q], src[i+3q])
// (partially_evaluate((0, r)), evaluation_form.rs:163-192); the halves of dst are summed on the
// fly for the next round (prover.rs:113-116), so the folded table is never re-read.
struct Quad { fq a0, a1, b0, b1; };
__device__ __forceinline__ void load_quad(Quad& Q, const fq* __restrict__ src, size_t q, size_t i) {
    Q.a0 = fq_load_stream(src + i);
    Q.a1 = fq_load_stream(src + 2 * q + i);
    Q.b0 = fq_load_stream(src + q + i);
    Q.b1 = fq_load_stream(src + 3 * q + i);
}
__device__ __forceinline__ void fold_quad(const Quad& Q, fq* __restrict__ dst, size_t q, size_t i, const fq& r, fq (&acc)[2]) {
    fq o0 = fq_fold(Q.a0, Q.a1, r);
    fq o1 = fq_fold(Q.b0, Q.b1, r);
    fq_store(dst + i, o0);
    fq_store(dst + q + i, o1);
    acc[0] = fq_add(acc[0], o0);
    acc[1] = fq_add(acc[1], o1);
}
// Two register buffers (A, B) alternate so that the 8 x 16-byte loads of the NEXT grid-stride
// step are in flight while the current step's ~800 VALU instructions run: without it every
// step pays a full HBM round trip before it can start (measured 3.3 TB/s -> see DESIGN.md).
__global__ void __launch_bounds__(kThreads) k_fold_sums(const fq* __restrict__ src, fq* __restrict__ dst, size_t q, fq r,
                                                         fq* __restrict__ partials) {
    fq acc[2] = {fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * kThreads;
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    Quad A, B;
    if (i < q) load_quad(A, src, q, i);
    while (i < q) {
        size_t j = i + stride;
        if (j < q) load_quad(B, src, q, j);
        asm volatile("" ::: "memory"); // keep the prefetch above the arithmetic
        fold_quad(A, dst, q, i, r, acc);
        i = j;
        if (i >= q) break;
        j = i + stride;
        if (j < q) load_quad(A, src, q, j);
        asm volatile("" ::: "memory");
        fold_quad(B, dst, q, i, r, acc);
        i = j;
    }
    block_sum_store<2>(acc, partials);
}

// ---- plain fold of an arbitrary variable (partially_evaluate((v, r))) --------------------
// n_out = len/2 outputs; log_s = log2 of the pair stride 2^(n-1-v).  Output m pairs input
// i = ((m >> log_s) << (log_s+1)) | (m & (2^log_s - 1)) with i | 2^log_s  -- the order
// get_y1_y2_indexes (evaluation_form.rs:45-80) enumerates.
__global__ void __launch_bounds__(kThreads) k_fold_var(const fq* __restrict__ src, fq* __restrict__ dst, size_t n_out, int log_s, fq r) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    const size_t s = (size_t)1 << log_s;
    for (size_t m = (size_t)blockIdx.x * kThreads + threadIdx.x; m < n_out; m += stride) {
        size_t i = ((m >> log_s) << (log_s + 1)) | (m & (s - 1));
        fq y1 = fq_load(src + i), y2 = fq_load(src + i + s);
        fq_store(dst + m, fq_fold(y1, y2, r));
    }
}

// ---- second stage: fold per-block partials (fixed order) ---------------------------------
// out[s] = sum_b partials[b*ns + s], s < ns <= 3.  One block.
__global__ void __launch_bounds__(kThreads) k_reduce_partials(const fq* __restrict__ partials, int nblocks, int ns, fq* __restrict__ out) {
    fq acc[3] = {fq_zero(), fq_zero(), fq_zero()};
    for (int b = threadIdx.x; b < nblocks; b += kThreads) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < ns) acc[s] = fq_add(acc[s], fq_load(partials + (size_t)b * ns + s));
    }
    __shared__ fq lds[3][kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        fq w = fq_wave_sum(acc[s]);
        if (lane == 0) lds[s][wave] = w;
    }
    __syncthreads();
    if ((int)threadIdx.x < ns) {
        fq t = lds[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) t = fq_add(t, lds[threadIdx.x][w]);
        fq_store(out + threadIdx.x, t);
    }
}

// ---- plain sum (evaluation_sum, evaluation_form.rs:236-238) of a short table -------------
__global__ void __launch_bounds__(kThreads) k_sum_all(const fq* __restrict__ T, size_t n, fq* __restrict__ partials) {
    fq acc[1] = {fq_zero()};
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) acc[0] = fq_add(acc[0], fq_load_stream(T + i));
    block_sum_store<1>(acc, partials);
}

// ---- degree-2 round reduction for Sum[Prod[A,B], Prod[C,D]] (prover.rs:89-107) -----------
// For X in {0,1,2}: ev[X] = sum_j  A(X)_j*B(X)_j + C(X)_j*D(X)_j,  T(X)_j = lo_j + X*(hi_j - lo_j),
// lo = T[j], hi = T[j+half].  X = 2 needs only additions: 2*hi - lo.
struct Tab4 { const fq* t[4]; };
struct Tab4w { fq* t[4]; };

__device__ __forceinline__ void eval3_accumulate(fq (&acc)[3], const fq (&lo)[4], const fq (&hi)[4]) {
    fq x2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x2[k] = fq_sub(fq_dbl(hi[k]), lo[k]);
    acc[0] = fq_add(acc[0], fq_add(fq_mul(lo[0], lo[1]), fq_mul(lo[2], lo[3])));
    acc[1] = fq_add(acc[1], fq_add(fq_mul(hi[0], hi[1]), fq_mul(hi[2], hi[3])));
    acc[2] = fq_add(acc[2], fq_add(fq_mul(x2[0], x2[1]), fq_mul(x2[2], x2[3])));
}

__global__ void __launch_bounds__(kThreads) k_eval3(Tab4 T, size_t half, fq* __restrict__ partials) {
    fq acc[3] = {fq_zero(), fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < half; i += stride) {
        fq lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = fq_load_stream(T.t[k] + i);
            hi[k] = fq_load_stream(T.t[k] + half + i);
        }
        eval3_accumulate(acc, lo, hi);
    }
    block_sum_store<3>(acc, partials);
}

// fused: fold all four factor tables at r (prover.rs:131-139) and evaluate the NEXT round's
// three points from the folded values while they are still in registers.  src tables have 4*q
// entries, dst tables 2*q.
__global__ void __launch_bounds__(kThreads) k_fold4_eval3(Tab4 S, Tab4w D, size_t q, fq r, fq* __restrict__ partials) {
    fq acc[3] = {fq_zero(), fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < q; i += stride) {
        fq lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fq a0 = fq_load_stream(S.t[k] + i), a1 = fq_load_stream(S.t[k] + 2 * q + i);
            lo[k] = fq_fold(a0, a1, r);
            fq_store(D.t[k] + i, lo[k]);
            fq b0 = fq_load_stream(S.t[k] + q + i), b1 = fq_load_stream(S.t[k] + 3 * q + i);
            hi[k] = fq_fold(b0, b1, r);
            fq_store(D.t[k] + q + i, hi[k]);
        }
        eval3_accumulate(acc, lo, hi);
    }
    block_sum_store<3>(acc, partials);
}

// fold of four tables without the evaluation (last round / API use)
__global__ void __launch_bounds__(kThreads) k_fold4(Tab4 S, Tab4w D, size_t half, fq r) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < half; i += stride) {
#pragma unroll
        for (int k = 0; k < 4; ++k) fq_store(D.t[k] + i, fq_fold(fq_load_stream(S.t[k] + i), fq_load_stream(S.t[k] + half + i), r));
    }
}

// ---- format conversion / generation ------------------------------------------------------
// canonical little-endian 256-bit values (any value < 2^256) -> Montgomery table
__global__ void __launch_bounds__(kThreads) k_from_le32(const fq* __restrict__ in, fq* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
        fq_store(out + i, fq_to_mont(fq_reduce_256(fq_load(in + i))));
}
// to_bytes (evaluation_form.rs:225-234): Montgomery -> canonical LE32
__global__ void __launch_bounds__(kThreads) k_to_le32(const fq* __restrict__ in, fq* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
        fq_store(out + i, fq_from_mont(fq_load(in + i)));
}

__device__ __forceinline__ u64 splitmix64_at(u64 seed, u64 i) {
    u64 z = seed + i * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// synthetic table: counter-based, entry j depends on (seed, first + j) only
__global__ void __launch_bounds__(kThreads) k_generate(fq* __restrict__ out, u64 seed, u64 first, size_t n) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        u64 j = first + i;
        fq x;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            u64 v = splitmix64_at(seed, 4 * j + w + 1);
            x.v[2 * w] = (u32)v;
            x.v[2 * w + 1] = (u32)(v >> 32);
        }
        fq_store(out + i, fq_to_mont(fq_reduce_256(x)));
    }
}

// Montgomery-multiply throughput probe (int32-ALU roofline): iters dependent multiplies per lane
__global__ void __launch_bounds__(kThreads) k_modmul_probe(fq* __restrict__ io, int iters) {
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    fq a = fq_load(io + i), b = a;
    b.v[0] ^= 0x5a5a5a5au & 0x0fffffffu;
    b.v[7] &= 0x0fffffffu;
    a.v[7] &= 0x0fffffffu;
    for (int k = 0; k < iters; ++k) a = fq_mul(a, b);
    fq_store(io + i, a);
}

// ---- launchers ---------------------------------------------------------------------------
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

int launch_round_sums(const void* T, size_t half, void* partials, hipStream_t st) {
    int g = grid_for(half);
    hipLaunchKernelGGL(k_round_sums, dim3(g), dim3(kThreads), 0, st, (const fq*)T, half, (fq*)partials);
    return g;
}
int launch_fold_sums(const void* src, void* dst, size_t q, const uint64_t r[4], void* partials, hipStream_t st) {
    int g = grid_for(q);
    hipLaunchKernelGGL(k_fold_sums, dim3(g), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, q, to_fq(r), (fq*)partials);
    return g;
}
void launch_fold_var(const void* src, void* dst, size_t n_out, int log_s, const uint64_t r[4], hipStream_t st) {
    hipLaunchKernelGGL(k_fold_var, dim3(grid_for(n_out)), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, n_out, log_s, to_fq(r));
}
void launch_reduce_partials(const void* partials, int nblocks, int ns, void* out, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kThreads), 0, st, (const fq*)partials, nblocks, ns, (fq*)out);
}
int launch_sum_all(const void* T, size_t n, void* partials, hipStream_t st) {
    int g = grid_for(n);
    hipLaunchKernelGGL(k_sum_all, dim3(g), dim3(kThreads), 0, st, (const fq*)T, n, (fq*)partials);
    return g;
}
int launch_eval3(const void* const t[4], size_t half, void* partials, hipStream_t st) {
    Tab4 T;
    for (int k = 0; k < 4; ++k) T.t[k] = (const fq*)t[k];
    int g = grid_for(half);
    hipLaunchKernelGGL(k_eval3, dim3(g), dim3(kThreads), 0, st, T, half, (fq*)partials);
    return g;
}
int launch_fold4_eval3(const void* const s[4], void* const d[4], size_t q, const uint64_t r[4], void* partials, hipStream_t st) {
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = (fq*)d[k]; }
    int g = grid_for(q);
    hipLaunchKernelGGL(k_fold4_eval3, dim3(g), dim3(kThreads), 0, st, S, D, q, to_fq(r), (fq*)partials);
    return g;
}
void launch_fold4(const void* const s[4], void* const d[4], size_t half, const uint64_t r[4], hipStream_t st) {
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = (fq*)d[k]; }
    hipLaunchKernelGGL(k_fold4, dim3(grid_for(half)), dim3(kThreads), 0, st, S, D, half, to_fq(r));
}
void launch_from_le32(const void* in, void* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_from_le32, dim3(grid_for(n)), dim3(kThreads), 0, st, (const fq*)in, (fq*)out, n);
}
void launch_to_le32(const void* in, void* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_to_le32, dim3(grid_for(n)), dim3(kThreads), 0, st, (const fq*)in, (fq*)out, n);
}
void launch_generate(void* out, uint64_t seed, uint64_t first, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(n)), dim3(kThreads), 0, st, (fq*)out, seed, first, n);
}
void launch_modmul_probe(void* io, int blocks, int iters, hipStream_t st) {
    hipLaunchKernelGGL(k_modmul_probe, dim3(blocks), dim3(kThreads), 0, st, (fq*)io, iters);
}

} // namespace zkk
