// kernels.hip -- gfx950 kernels of the multilinear sumcheck prover: the single-MLE path (block sums, multi-variable folds) and the
// table operations.  The degree-2 (SumPolynomial) branch lives in deg2_kernels.hip; templates both use in kernels_common.hpp.
//
// Data layout in HBM: a table is N contiguous 32-byte elements (8 x u32 Montgomery limbs,
// index order = MultiLinearPolynomial::evaluation_points, variable 0 = most significant index
// bit; evaluation_form.rs:40-42).  The prover always binds variable 0, so a round pairs entry
// j with entry j + N/2 (prover.rs:29-41): two contiguous streams, one 32-byte element per lane,
// two global_load_dwordx4 each.
//
// The big passes are HBM-streaming integer kernels: every byte is read once per pass, so there is nothing to tile for reuse.  LDS
// is the landing zone of the LDS-DMA streams (per-wave rings) and of the byte-transposed operand reads; the matrix cores are used
// where a pass is a dense contraction once its operands are cut into bytes (k_multifold_wide_mfma here; k_sum16_mfma, k_cross64_mfma
// and k_fold3_cross64 in deg2_kernels.hip).
// Grids are capped and grid-stride, block -> per-block partial sums, one small second-stage kernel folds the partials (field
// addition is exact: results are bit-reproducible whatever the order).
#include <mutex>
#include "kernels_common.hpp"

namespace zkk {

thread_local unsigned g_forms = 0;

// ---- batched rounds: block sums + multi-variable fold -----------------------------------------
// The MLE-branch round polynomial (sumcheck/src/prover.rs:108-124) only needs the two half sums
// of the current table, and summation commutes with folding the TOP variables: if
//   S[b] = sum_j T[b * L/2^m + j]      (b < 2^m: the table cut into 2^m contiguous blocks)
// then the half sums of T folded at r_0..r_{k-1} equal the half sums of the 2^m-entry table S
// folded at the same points.  So one pass that produces S lets the host run m rounds (m Keccak
// challenges) on 2^m numbers, and ONE pass then binds all m variables at once:
//   T'[j] = fold_{r_{m-1}}( ... fold_{r_0}( T[b * L' + j] )_{b < 2^m} ... ),   L' = L / 2^m
// (partially_evaluate((0, r)) applied m times, evaluation_form.rs:163-192 -- the same field values
// in the same order; exact arithmetic makes the grouping irrelevant).  HBM traffic per batch is
// 32*(L + L') bytes instead of 48*L*(1 + 1/2 + ...): 128*N -> 85*N (m=2) / 73*N (m=3) for a prove.
//
// Work split shared by both kernels: the OUTPUT (or, for k_block_sums, the input) is cut into
// 2^log_ob contiguous blocks; workgroup w serves block w / gpb (gpb = gridDim.x >> log_ob) and
// grid-strides inside it, so every thread carries ONE accumulator whatever the batch size.

// partials[w] = this workgroup's share of the sum of block w / gpb of T (L entries)
__global__ void __launch_bounds__(kThreads) k_block_sums(const fq* __restrict__ T, size_t L, int log_ob, fq* __restrict__ partials) {
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wl = blockIdx.x % gpb;
    const size_t Lb = L >> log_ob;
    const fq* __restrict__ base = T + (size_t)c * Lb;
    const size_t stride = (size_t)gpb * kThreads;
    fq acc[1] = {fq_zero()};
    size_t i = (size_t)wl * kThreads + threadIdx.x;
    // two loads in flight per step
    for (; i + stride < Lb; i += 2 * stride) {
        fq a = fq_load_stream(base + i), b = fq_load_stream(base + i + stride);
        acc[0] = fq_add(acc[0], fq_add(a, b));
    }
    if (i < Lb) acc[0] = fq_add(acc[0], fq_load_stream(base + i));
    block_sum_store<1>(acc, partials);
}

// The same sums with the table streamed by LDS-DMA (global_load_lds_dwordx4: no VGPR destination) instead of register
// loads: MI355X_MICROARCH.md measures 6.4 - 6.8 TB/s for LDS-DMA streams against 6.0 - 6.1 for register sweeps.  Every
// wave runs its own ring of D rows (a row = 64 entries = 2 KiB = two DMA instructions, lane l reading entry l back
// from LDS), D - 1 rows in flight behind a counted vmcnt; no barrier in the loop (waves share nothing) and `ring` is
// the kernel's only __shared__ object (the final reduction reuses it).  Lb must be a multiple of 64.
template <int D, int AUX>
__global__ void __launch_bounds__(kThreads) k_block_sums_dma(const fq* __restrict__ T, size_t L, int log_ob, fq* __restrict__ partials) {
    __shared__ u32x4 ring[kThreads / 64][D][128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wl = blockIdx.x % gpb;
    const size_t Lb = L >> log_ob;
    const fq* __restrict__ base = T + (size_t)c * Lb;
    const size_t nrows = Lb >> 6, wstride = (size_t)gpb * (kThreads / 64);
    auto issue = [&](size_t r, int slot) {
        const u32x4* src = reinterpret_cast<const u32x4*>(base + (r << 6)) + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)&ring[wave][slot][0], 16, 0, AUX);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64),
                                         (__attribute__((address_space(3))) void*)&ring[wave][slot][64], 16, 0, AUX);
    };
    const size_t row0 = (size_t)wl * (kThreads / 64) + wave;
    size_t r_issue = row0;
    int inflight = 0;
#pragma unroll
    for (int sl = 0; sl < D; ++sl)
        if (r_issue < nrows) { issue(r_issue, sl); r_issue += wstride; ++inflight; }
    int slot = 0;
    fq acc = fq_zero();
    for (size_t r = row0; r < nrows; r += wstride) {
        // the oldest outstanding row sits in `slot`; leave the D - 1 younger ones in flight (drain only near the end)
        if (inflight == D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32x4 lo = ring[wave][slot][2 * lane], hi = ring[wave][slot][2 * lane + 1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the entry is in registers before its slot is refilled
        --inflight;
        if (r_issue < nrows) { issue(r_issue, slot); r_issue += wstride; ++inflight; }
        fq x;
#pragma unroll
        for (int w = 0; w < 4; ++w) { x.v[w] = lo[w]; x.v[4 + w] = hi[w]; }
        acc = fq_add(acc, x);
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    // block sum through the ring's memory (every DMA has landed: the loop ends on vmcnt(0))
    const fq wsum = fq_wave_sum(acc);
    __syncthreads();
    fq* red = reinterpret_cast<fq*>(&ring[0][0][0]);
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        fq t = red[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) t = fq_add(t, red[w]);
        fq_store(&partials[blockIdx.x], t);
    }
}

template <int M> struct Chal { fq r[M]; };

template <int M>
__device__ __forceinline__ void mf_load(fq (&in)[1 << M], const fq* __restrict__ src, size_t Lout, size_t j) {
#pragma unroll
    for (int b = 0; b < (1 << M); ++b) in[b] = fq_load_stream(src + (size_t)b * Lout + j);
}
// binds the top M variables: level l pairs (b, b + 2^(M-1-l)) with challenge l (variable 0 = MSB)
template <int M>
__device__ __forceinline__ fq mf_tree(fq (&in)[1 << M], const Chal<M>& ch) {
#pragma unroll
    for (int l = 0; l < M; ++l) {
        const int half = 1 << (M - 1 - l);
#pragma unroll
        for (int b = 0; b < half; ++b) in[b] = fq_fold(in[b], in[b + half], ch.r[l]);
    }
    return in[0];
}

// src: Lout << M entries, dst: Lout entries.  The fold TREE: statement by statement the M successive partially_evaluate calls
// of the reference -- kept as the independent cross-check of the inner-product forms below (zk_ctx_set_fold_batch variant 1)
// and as the m = 1 pass.
template <int M>
__global__ void __launch_bounds__(kThreads) k_multifold(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout, Chal<M> ch,
                                                         int log_ob, fq* __restrict__ partials) {
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wl = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob;
    const size_t j0 = (size_t)c * Lb;
    const size_t stride = (size_t)gpb * kThreads;
    fq acc[1] = {fq_zero()};
    for (size_t i = (size_t)wl * kThreads + threadIdx.x; i < Lb; i += stride) {
        fq A[1 << M];
        mf_load<M>(A, src, Lout, j0 + i);
        fq o = mf_tree<M>(A, ch);
        fq_store(dst + j0 + i, o);
        acc[0] = fq_add(acc[0], o);
    }
    block_sum_store<1>(acc, partials);
}

// Same pass with the tree replaced by ONE inner product with the eq weights
//   T'[j] = sum_b eq(r, b) * T[b * L' + j],   eq(r, b) = prod_l (b_l ? r_l : 1 - r_l), b_0 = MSB
// accumulated unreduced in radix 2^28 (fq_dev.hpp) and reduced once: (64 + 72/2^M) limb products per
// input entry instead of ~119-127 for the tree, which moves this kernel from ALU-bound to HBM-bound.
// W.w[b] = eq(r,b) * 2^280 mod p as 10 x 28-bit limbs (made on the host).  Same field values as the
// tree: both are exact.
// W28<M>: kernels_common.hpp

// Loads are software-pipelined through two register buffers of G elements: while group g is being
// multiplied into the accumulators, group g+1 (or group 0 of the next grid-stride step) is in flight.
template <int G>
__device__ __forceinline__ void lazy_load(fq (&buf)[G], const fq* __restrict__ src, size_t Lout, size_t j, int g) {
#pragma unroll
    for (int b = 0; b < G; ++b) buf[b] = fq_load_stream(src + (size_t)(g * G + b) * Lout + j);
    asm volatile("" ::: "memory"); // keep the loads where they are issued
}
template <int M, int G>
__device__ __forceinline__ void lazy_mac(u64 (&acc)[20], const fq (&buf)[G], const W28<M>& W, int g) {
#pragma unroll
    for (int b = 0; b < G; ++b) fq28_mac(acc, buf[b], W.w[g * G + b]);
}

template <int M, int G>
__global__ void __launch_bounds__(kThreads) k_multifold_lazy(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout, W28<M> W,
                                                              int log_ob, fq* __restrict__ partials) {
    constexpr int NB = 1 << M, NG = NB / G;
    static_assert(NB % G == 0 && (NG == 1 || NG % 2 == 0), "group count must be 1 or even");
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wl = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob;
    const size_t j0 = (size_t)c * Lb;
    const size_t stride = (size_t)gpb * kThreads;
    fq acc1[1] = {fq_zero()};
    const bool big = Lout >= ((size_t)1 << 22); // outputs beyond the caches: streaming stores; small ones stay cached for the next pass
    size_t i = (size_t)wl * kThreads + threadIdx.x;
    fq A[G], B[G];
    if (i < Lb) lazy_load<G>(A, src, Lout, j0 + i, 0);
    if (NG == 1) {
        // one group per output: alternate the buffers between consecutive grid-stride steps
        while (i < Lb) {
            size_t n = i + stride;
            if (n < Lb) lazy_load<G>(B, src, Lout, j0 + n, 0);
            u64 acc[20];
#pragma unroll
            for (int k = 0; k < 20; ++k) acc[k] = 0;
            lazy_mac<M, G>(acc, A, W, 0);
            fq o = fq28_redc(acc);
            if (big) fq_store_stream(dst + j0 + i, o); else fq_store(dst + j0 + i, o);
            acc1[0] = fq_add(acc1[0], o);
            i = n;
            if (i >= Lb) break;
            n = i + stride;
            if (n < Lb) lazy_load<G>(A, src, Lout, j0 + n, 0);
#pragma unroll
            for (int k = 0; k < 20; ++k) acc[k] = 0;
            lazy_mac<M, G>(acc, B, W, 0);
            o = fq28_redc(acc);
            if (big) fq_store_stream(dst + j0 + i, o); else fq_store(dst + j0 + i, o);
            acc1[0] = fq_add(acc1[0], o);
            i = n;
        }
    } else {
        while (i < Lb) {
            const size_t n = i + stride;
            u64 acc[20];
#pragma unroll
            for (int k = 0; k < 20; ++k) acc[k] = 0;
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                lazy_load<G>(B, src, Lout, j0 + i, g + 1);
                lazy_mac<M, G>(acc, A, W, g);
                if (g + 2 < NG) lazy_load<G>(A, src, Lout, j0 + i, g + 2);
                else if (n < Lb) lazy_load<G>(A, src, Lout, j0 + n, 0);
                lazy_mac<M, G>(acc, B, W, g + 1);
            }
            fq o = fq28_redc(acc);
            if (big) fq_store_stream(dst + j0 + i, o); else fq_store(dst + j0 + i, o);
            acc1[0] = fq_add(acc1[0], o);
            i = n;
        }
    }
    block_sum_store<1>(acc1, partials);
}

// Wide form of the same pass for 2^m = 8 .. 256 read streams (m up to kMaxFoldBatch).  The stores of a fold
// pass cost the same ~20 % of the HBM rate whether they are 1/2 or 1/16 of the traffic (tools/mfbench,
// profiles/r01_mfbench_fold_pass_ablation.txt); only at 1/256 does the pass run at the read-only rate, so the
// first pass over a large table binds eight variables.  2^m x 10 weight limbs no longer fit in kernel
// arguments: k_eq_weights28 (one workgroup, one product per weight from the host-built half trees) writes them into a
// small device buffer and the fold reads them with wave-uniform addresses, i.e. as scalar loads into SGPRs --
// no VGPRs and no LDS for the weights.  The group loop stays rolled (its body is 4 inner products), two
// elements per load buffer keep the kernel at 4 waves per SIMD (the arithmetic, 1.0 ms of a 1.5 ms pass, only
// hides under the loads with enough waves: 2 waves/SIMD measured 5.2 TB/s, 4 waves 5.6), and the columns are
// carried once per loop trip so that they stay below 2^64 however many products are summed.
// EqHalves (the kernel's FIRST argument, read through the kernel-argument segment), eq_weight_of_thread, eq_weights28_block: kernels_common.hpp
__global__ void __launch_bounds__(kThreads) k_eq_weights28(EqHalves, int m, u32* __restrict__ wout) { eq_weights28_block(m, wout); }

// k_multifold_wide_dma<BIG, D> (the LDS-DMA form of the wide pass; the degree-2 folds use it too): kernels_common.hpp

// The same pass for SMALL outputs (below 2^19 entries: a few waves per CU, time set by each lane's serial chain and by
// launch latency, tools/wide_g_sweep.sh): every workgroup derives the weights itself into LDS -- no separate
// k_eq_weights28 launch on the critical path -- and reads them as LDS broadcasts; 4 elements per load buffer.
__global__ void __launch_bounds__(kThreads) k_multifold_wide_lds(EqHalves, const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout, int m,
                                                              int log_ob, fq* __restrict__ partials) {
    constexpr int G = 4;
    __shared__ u32 wl[(1 << kMaxFoldBatch) * 10];
    const int nb = 1 << m, ng = nb / G; // m >= 3: ng is even
    eq_weights28_block(m, wl);
    __syncthreads();

    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wi = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob;
    const size_t j0 = (size_t)c * Lb;
    const size_t stride = (size_t)gpb * kThreads;
    const size_t gstep = (size_t)G * Lout;
    fq acc1[1] = {fq_zero()};
    const bool big = Lout >= ((size_t)1 << 22);
    size_t i = (size_t)wi * kThreads + threadIdx.x;
    fq A[G], B[G];
    if (i < Lb) lazy_load<G>(A, src, Lout, j0 + i, 0);
    while (i < Lb) {
        const size_t n = i + stride;
        u64 acc[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0;
        const fq* __restrict__ p = src + j0 + i;
        const u32* w = wl;
        for (int g = 0; g < ng; g += 2) {
            lazy_load<G>(B, p + gstep, Lout, 0, 0);
#pragma unroll
            for (int b = 0; b < G; ++b) fq28_mac(acc, A[b], w + b * 10);
            if (g + 2 < ng) lazy_load<G>(A, p + 2 * gstep, Lout, 0, 0);
            else if (n < Lb) lazy_load<G>(A, src, Lout, j0 + n, 0);
#pragma unroll
            for (int b = 0; b < G; ++b) fq28_mac(acc, B[b], w + (G + b) * 10);
            // carry after every 16 products (G = 4: every second trip), as above
            if (g & 2) {
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
        if (big) fq_store_stream(dst + j0 + i, o); else fq_store(dst + j0 + i, o);
        acc1[0] = fq_add(acc1[0], o);
        i = n;
    }
    block_sum_store<1>(acc1, partials);
}

// ---- the wide pass on the matrix cores ------------------------------------------------------------------------------------
// T'[j] = sum_b W_b * T[b * L' + j] (W_b = eq(r, b) * 2^280 mod p) is a dense contraction once the operands are cut into BYTES:
// column k (weight 256^k) of the unreduced product of output j is  sum_b sum_i T_b[j].byte[i] * W_b.digit[k - i]  -- a GEMM of
// the RAW table bytes (row j, K index (b, i)) against a banded Toeplitz matrix of the weights' radix-256 digits, 64 columns
// wide.  v_mfma_i32_16x16x64_i8 does 16 columns x 16 outputs x 64 K-values per instruction (16 cycles per SIMD): per 64-entry
// row 8 matrix instructions replace the ~106 vector instructions of k_multifold_wide_dma, whose VALU issue co-limits that pass.
// Signedness: the matrix cores multiply SIGNED bytes.  The weights are recoded once per pass into signed digits in [-128, 127]
// (k_eq_weights_mfma); a table byte a is presented as a - 128 (one v_xor_b32 per four bytes).  Then
//   sum a w = sum (a - 128) w + 128 sum w,
// and the second term does not depend on the table: over all columns it is 128 * ((2^256 - 1) / 255) * sum_b W_b, and since
// sum_b eq(r, b) = 1 that is a CONSTANT field element after the reduction.  The i32 accumulators start at beta = 2^27 (|column|
// <= 256 * 32 * 2^14 = 2^27), so every column comes out non-negative; beta's contribution is a constant too.  Both are folded into
// kfix = 128 (2^256 - 1)/255 - 2^27 ((2^512 - 1)/255) 2^-280 mod p, added to every output.  Exact integer arithmetic throughout:
// the outputs are the same field elements as every other form of the pass.
// Orientation: the weights are the A operand (M = product column), the table the B operand (N = output), so lane (g, n) of the
// result holds columns 16 mt + 4 g + r of output (nt, n): the 64 columns of one output sit in the four lanes n, n+16, n+32, n+48,
// which exchange 64-bit partial sums through LDS once per 64 outputs; one radix-2^28 Montgomery reduction per output, as before.
// K index of lane group g, byte t of its 16-byte fragment: table row b = 2 bp + (g >> 1), byte i = 16 (g & 1) + t -- the 16 bytes
// the lane reads from the row in LDS; the weight tiles are laid out to match (any K order works if A and B agree).
static constexpr int kMfmaTileWords = 1024; // u32 per b-pair: 4 column tiles x 64 lanes x 4 words

// tiles[((bp * 4 + mt) * 64 + lane) * 4 + u]: word u of lane's fragment of the weight tile (b-pair bp, columns 16 mt .. 16 mt + 15).
// Every workgroup derives all 2^m weights (one product each, from the host-built half trees) and writes its share of the tiles.
__global__ void __launch_bounds__(kThreads) k_eq_weights_mfma(EqHalves, int m, u32* __restrict__ tiles) {
    __shared__ signed char dig[1 << kMaxFoldBatch][32];
    const fq w = eq_weight_of_thread(m);
    if ((int)threadIdx.x < (1 << m)) { // signed radix-256 digits: the top byte of a canonical element is <= 0x30, so 32 digits do
        int carry = 0;
#pragma unroll
        for (int idx = 0; idx < 32; ++idx) {
            int v = (int)((w.v[idx >> 2] >> (8 * (idx & 3))) & 255u) + carry;
            carry = v >= 128;
            dig[threadIdx.x][idx] = (signed char)(v - (carry << 8));
        }
    }
    __syncthreads();
    const unsigned total = (1u << (m - 1)) * kMfmaTileWords;
    for (unsigned idx = blockIdx.x * kThreads + threadIdx.x; idx < total; idx += gridDim.x * kThreads) {
        const unsigned u = idx & 3, lane = (idx >> 2) & 63, mt = (idx >> 8) & 3, bp = idx >> 10;
        const unsigned g = lane >> 4, row = lane & 15, b = 2 * bp + (g >> 1);
        const int kg = (int)(16 * mt + row), i0 = (int)(16 * (g & 1) + 4 * u);
        u32 word = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int d = kg - (i0 + t);
            const u32 byte = (d >= 0 && d < 32) ? (u32)(unsigned char)dig[b][d] : 0u;
            word |= byte << (8 * t);
        }
        tiles[idx] = word;
    }
}

// Schedule.  The four waves of a workgroup walk the b-pairs in LOCKSTEP, each on its own groups of 64 outputs, so the weight
// tiles of a b-pair (4 KiB) are fetched once per workgroup: wave w brings tile mt = w by one LDS-DMA instruction into a shared
// ring of three steps, its own two table rows (4 KiB) by four into its private ring of NP steps, and everything a wave waits
// for is LDS-DMA in ONE in-order stream -- R(0) W(0) .. R(K-1) W(K-1) R(K) | step j: W(j+K) R(j+K+1) -- so the counted wait
// before step s leaves R(s+1) .. R(s+K) and W(s+1) .. W(s+K-1) (5K - 1 instructions; two more with the stores of a group that
// just ended) in flight.  One
// s_barrier per step makes the other waves' weight tiles visible and protects the shared ring (a wave passes barrier s only
// after every wave has finished reading step s-1).  A wave with fewer groups than its workgroup runs empty steps: the barrier
// count is the same for all four.  A finished group's partial sums are exchanged through the table slot the step just freed.
template <bool BIG, int K>
__global__ void __launch_bounds__(kThreads) k_multifold_wide_mfma(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout,
                                                                   const u32x4* __restrict__ tiles, int m, int log_ob, fq kfix,
                                                                   fq* __restrict__ partials) {
    constexpr int NP = K + 2, NW = K + 1; // step j issues W(j + K), R(j + K + 1)
    __shared__ u32x4 ring[kThreads / 64][NP][2][128]; // per wave: NP steps x 2 table rows x 2 KiB
    __shared__ u32x4 wring[NW][4][64];                // per workgroup: NW steps x 4 column tiles x 1 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, n = lane & 15;
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wi = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob, j0 = (size_t)c * Lb;
    const unsigned ngroups = (unsigned)(Lb >> 6), gstride = gpb * (kThreads / 64);
    const unsigned nbp = 1u << (m - 1);
    const unsigned first = wi * (kThreads / 64); // the workgroup's first group: wave 0's, which has the most
    const unsigned g0 = first + wave;
    const unsigned mysteps = (g0 < ngroups ? (ngroups - g0 + gstride - 1) / gstride : 0) * nbp;
    const unsigned wgsteps = (first < ngroups ? (ngroups - first + gstride - 1) / gstride : 0) * nbp;

    auto issue_rows = [&](unsigned s) { // R(s): rows 2 bp, 2 bp + 1 of step s's group; four DMA instructions
        const unsigned group = g0 + (s / nbp) * gstride, bp = s % nbp;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const u32x4* p = reinterpret_cast<const u32x4*>(src + (size_t)(2 * bp + r) * Lout + j0 + ((size_t)group << 6)) + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)&ring[wave][s % NP][r][0], 16, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 64),
                                             (__attribute__((address_space(3))) void*)&ring[wave][s % NP][r][64], 16, 0, 2);
        }
    };
    auto issue_tile = [&](unsigned s) { // W(s): this wave's quarter (column tile mt = wave) of step s's weight tiles; one DMA instruction
        const u32x4* p = tiles + ((size_t)(s % nbp) * 4 + wave) * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)&wring[s % NW][wave][0], 16, 0, 0);
    };

    v4i acc[4][4];
    fq run = fq_zero();
#pragma unroll
    for (int q = 0; q < K; ++q) {
        if ((unsigned)q < mysteps) issue_rows(q);
        if ((unsigned)q < wgsteps) issue_tile(q);
    }
    if ((unsigned)K < mysteps) issue_rows(K);
    bool stored = false;
    for (unsigned s = 0; s < wgsteps; ++s) {
        const bool active = s < mysteps; // wave-uniform
        if (active && s + K < mysteps) { // R(s+1) .. R(s+K) and W(s+1) .. W(s+K-1) may stay in flight: 5 K - 1 instructions
            if (stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * K + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * K - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the last two steps of a wave, and empty steps: drain
        }
        stored = false;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned bp = s % nbp;
        v4i T[4], W[4];
        if (active) {
            if (bp == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = v4i{1 << 27, 1 << 27, 1 << 27, 1 << 27}; // beta
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { // the table operand: 16 outputs x (2 rows x 32 bytes) per tile, bytes presented as a - 128
                const u32x4 raw = ring[wave][s % NP][g >> 1][2 * (16 * nt + n) + (g & 1)];
                T[nt] = v4i{(int)(raw[0] ^ 0x80808080u), (int)(raw[1] ^ 0x80808080u), (int)(raw[2] ^ 0x80808080u), (int)(raw[3] ^ 0x80808080u)};
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const u32x4 w4 = wring[s % NW][mt][lane];
                W[mt] = v4i{(int)w4[0], (int)w4[1], (int)w4[2], (int)w4[3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // operands in registers before their slots are refilled
        }
        if (s + K < wgsteps) issue_tile(s + K);
        const bool last = active && bp + 1 == nbp;
        if (active && !last && s + K + 1 < mysteps) issue_rows(s + K + 1);
        if (active) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[mt], T[nt], acc[mt][nt], 0, 0, 0);
        }
        if (last) { // the group is complete: 64 columns per output, spread over four lanes -> one field element per lane
            u64* x = reinterpret_cast<u64*>(&ring[wave][s % NP][0][0]); // the slot this step just read: 4 KiB of exchange space
            u64 S[16];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { // the 16 outputs of tile nt go to lane group g = nt
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) { // columns 16 mt + 4 g + r, r = 0..3: one 64-bit partial at word 4 mt + g
                    const v4i a = acc[mt][nt];
                    x[n * 17 + 4 * mt + g] = (u64)(u32)a[0] + ((u64)(u32)a[1] << 8) + ((u64)(u32)a[2] << 16) + ((u64)(u32)a[3] << 24);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-private exchange: program order, no barrier
                if (g == nt) {
#pragma unroll
                    for (int w = 0; w < 16; ++w) S[w] = x[n * 17 + w];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (s + K + 1 < mysteps) issue_rows(s + K + 1); // the slot is free again
            // V = sum_w S[w] 2^(32 w), S[w] < 2^53: 18 words, then 28-bit columns for the reduction
            u32 v[18];
            u64 cy = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                cy += S[w] & 0xffffffffull;
                v[w] = (u32)cy;
                cy = (cy >> 32) + (S[w] >> 32);
            }
            v[16] = (u32)cy;
            v[17] = (u32)(cy >> 32);
            u64 a28[20];
#pragma unroll
            for (int k = 0; k < 20; ++k) { // bits [28 k, 28 k + 28)
                const int bit = 28 * k, wd = bit >> 5, sh = bit & 31;
                const u64 two = (u64)v[wd] | (wd + 1 < 18 ? (u64)v[wd + 1] << 32 : 0);
                a28[k] = (two >> sh) & FQ28_MASK;
            }
            const fq o = fq_add(fq28_redc(a28), kfix);
            const unsigned group = g0 + (s / nbp) * gstride;
            fq* out = dst + j0 + ((size_t)group << 6) + lane; // lane (g, n) holds output 16 g + n = lane
            if (BIG) fq_store_stream(out, o); else fq_store(out, o);
            stored = true;
            run = fq_add(run, o);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const fq wsum = fq_wave_sum(run);
    __syncthreads();
    fq* red = reinterpret_cast<fq*>(&wring[0][0][0]);
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        fq t = red[0];
#pragma unroll
        for (int k = 1; k < kThreads / 64; ++k) t = fq_add(t, red[k]);
        fq_store(&partials[blockIdx.x], t);
    }
}

// The same pass for TINY outputs (a few thousand entries or fewer: the tail of every prove).  With one lane per output
// each lane walks its 2^m inputs alone and the pass costs 2^m / 8 dependent load round trips, so the tail used to be
// cut into m = 4 passes -- five of them (and five host round trips) between 2^20 entries and the end.  Here ONE WAVE
// forms one output: lane l takes inputs b = l, l + 64, ... (2^m / 64 of them, m = 6..8), the 64 unreduced partial
// sums are carried and added across the wave by a butterfly of shuffles, and a single reduction gives the output.
// Eight rounds per pass at the latency of one: 2^20 -> 2^12 -> 2^4 in two passes.  Weights per workgroup in LDS as
// in k_multifold_wide_lds (each lane reads ITS weights: stride-10 addresses, two-way bank conflicts at worst).
__global__ void __launch_bounds__(kThreads) k_multifold_coop(EqHalves, const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout, int m,
                                                              int log_ob, fq* __restrict__ partials) {
    __shared__ u32 wl[(1 << kMaxFoldBatch) * 10];
    eq_weights28_block(m, wl);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = 1 << (m - 6); // inputs per lane: 1, 2 or 4
    const unsigned gpb = gridDim.x >> log_ob;
    const unsigned c = blockIdx.x / gpb, wi = blockIdx.x % gpb;
    const size_t Lb = Lout >> log_ob;
    const size_t j0 = (size_t)c * Lb;
    fq acc1[1] = {fq_zero()};
    for (size_t i = (size_t)wi * (kThreads / 64) + wave; i < Lb; i += (size_t)gpb * (kThreads / 64)) { // wave-uniform
        fq x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < per) x[k] = fq_load(src + (size_t)(lane + 64 * k) * Lout + j0 + i);
        u64 acc[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < per) fq28_mac(acc, x[k], wl + (lane + 64 * k) * 10);
        fq28_carry(acc); // columns below 2^28 (+ the tiny top one): 64 of them add up below 2^35
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 20; ++k) {
                const u32 lo = __shfl_xor((u32)acc[k], off, 64), hi = __shfl_xor((u32)(acc[k] >> 32), off, 64);
                acc[k] += ((u64)hi << 32) | lo;
            }
        }
        const fq o = fq28_redc(acc); // every lane holds the same sums
        if (lane == 0) {
            fq_store(dst + j0 + i, o);
            acc1[0] = fq_add(acc1[0], o);
        }
    }
    block_sum_store<1>(acc1, partials);
}

// second stage for the kernels above: out[c] = sum of the gpb partials of block c (fixed order)
__global__ void __launch_bounds__(kThreads) k_reduce_groups(const fq* __restrict__ partials, int gpb, fq* __restrict__ out) {
    fq acc[1] = {fq_zero()};
    const fq* __restrict__ p = partials + (size_t)blockIdx.x * gpb;
    for (int b = threadIdx.x; b < gpb; b += kThreads) acc[0] = fq_add(acc[0], fq_load(p + b));
    block_sum_store<1>(acc, out);
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

// compute_quotient_remainder (evaluation_form.rs:280-301): the same pairing with a second output stream --
// quotient[m] = y2 - y1 (:294), remainder[m] = y1 + (y2 - y1) * r = partially_evaluate((v, r)) (:285)
__global__ void __launch_bounds__(kThreads) k_quot_rem_var(const fq* __restrict__ src, fq* __restrict__ quot, fq* __restrict__ rem, size_t n_out,
                                                            int log_s, fq r) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    const size_t s = (size_t)1 << log_s;
    for (size_t m = (size_t)blockIdx.x * kThreads + threadIdx.x; m < n_out; m += stride) {
        const size_t i = ((m >> log_s) << (log_s + 1)) | (m & (s - 1));
        const fq y1 = fq_load(src + i), y2 = fq_load(src + i + s);
        const fq d = fq_sub(y2, y1);
        fq_store(quot + m, d);
        fq_store(rem + m, fq_add(y1, fq_mul_uniform(d, r)));
    }
}
// minus (evaluation_form.rs:260-272): out[i] = in[i] - v
__global__ void __launch_bounds__(kThreads) k_minus(const fq* __restrict__ in, fq* __restrict__ out, size_t n, fq v) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) fq_store(out + i, fq_sub(fq_load(in + i), v));
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

// the switch from sharded to replicated passes: gathered local tables (rank-major) -> the remaining tables
__global__ void __launch_bounds__(kThreads) k_interleave(const fq* __restrict__ in, fq* __restrict__ out, unsigned world, unsigned ntab, size_t L) {
    const size_t total = (size_t)world * ntab * L, stride = (size_t)gridDim.x * kThreads;
    for (size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x; o < total; o += stride) {
        const unsigned q = (unsigned)(o % world);
        const size_t tj = o / world; // t * L + j
        fq_store(out + o, fq_load(in + (size_t)q * ntab * L + tj));
    }
}

// Contract check of a table handed over as Montgomery limbs: every entry must be fully reduced (< p), as ark-ff keeps them --
// the unreduced accumulations (fq28_*) size their carry cadence for canonical operands.  count[0] += entries >= p.
__global__ void __launch_bounds__(kThreads) k_count_noncanonical(const fq* __restrict__ T, size_t n, unsigned long long* __restrict__ count) {
    const u32 p[8] = FQ_P_LIMBS;
    unsigned bad = 0;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const fq a = fq_load_stream(T + i);
        u32 br = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) (void)__builtin_subc(a.v[j], p[j], br, &br);
        bad += br ? 0u : 1u; // no borrow: a >= p
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, (unsigned long long)bad);
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
// synthetic table: counter-based, local entry i is global entry first + i*step and
// depends on (seed, global index) only -- step = number of shards for a table sharded on its low bits
__global__ void __launch_bounds__(kThreads) k_generate(fq* __restrict__ out, u64 seed, u64 first, u64 step, size_t n) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        u64 j = first + i * step;
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

// Issue-rate probe for the int32-ALU roofline: four independent v_mad_u64_u32 chains per lane, 256 multiply-adds per
// loop trip, nothing else in the loop -- the ceiling every limb-product loop of this file runs against.
__global__ void __launch_bounds__(kThreads) k_mad_probe(u32* __restrict__ out, int iters) {
    const u32 t = threadIdx.x;
    u64 a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3;
    const u32 x = t | 1, y = (t << 7) | 5;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int j = 0; j < 64; ++j)
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                         : "v"(x), "v"(y)
                         : "vcc");
    }
    out[(size_t)blockIdx.x * kThreads + threadIdx.x] = (u32)(a0 + a1 + a2 + a3);
}

// ---- launchers ---------------------------------------------------------------------------
void launch_eq_weights28(const uint64_t* r, int m, const uint64_t two24[4], void* wbuf, hipStream_t st) {
    hipLaunchKernelGGL(k_eq_weights28, dim3(1), dim3(kThreads), 0, st, make_eq_halves(r, m, two24), m, (u32*)wbuf);
}
// the weight tiles of the matrix-core fold (k_eq_weights_mfma) for m challenges into wbuf: 2^(m-1) x kMfmaTileWords u32
void launch_eq_weights_mfma(const uint64_t* r, int m, const uint64_t two24[4], void* wbuf, hipStream_t st) {
    hipLaunchKernelGGL(k_eq_weights_mfma, dim3(1u << (m - 1)), dim3(kThreads), 0, st, make_eq_halves(r, m, two24), m, (u32*)wbuf);
}
// A pass cut into 2^log_ob output blocks leaves one partial sum per workgroup and a second stage adds the g >> log_ob partials of each
// block.  When every block has ONE workgroup (the small passes: 2^16 -> 2^8 entries, 2^24 -> 2^16 with 256 blocks ...) its partial IS the
// block's sum: the workgroups store straight into `out` and the second-stage launch (4 us on the critical path of a latency-bound pass)
// is dropped.
static inline void* sums_target(int g, int log_ob, void* partials, void* out) { return (g >> log_ob) == 1 ? out : partials; }
static void reduce_groups_unless_single(int g, int log_ob, const void* partials, void* out, hipStream_t st) {
    if ((g >> log_ob) > 1) hipLaunchKernelGGL(k_reduce_groups, dim3(1 << log_ob), dim3(kThreads), 0, st, (const fq*)partials, g >> log_ob, (fq*)out);
}
int launch_block_sums(const void* T, size_t L, int log_ob, void* partials, void* out, hipStream_t st, const DmaGates& gates) {
    const size_t Lb0 = L >> log_ob;
    // large tables: LDS-DMA stream with the nt policy, 6.7 vs 6.1 TB/s at 2^28 entries; it wins from 2^27 entries
    // (0.667 vs 0.722 ms), ties at 2^25 and loses at 2^26 and below (its 64 KiB of LDS per workgroup cap the waves per CU)
    if (Lb0 % 64 == 0 && gates.sums_from_log < 63 && L >= ((size_t)1 << std::max(gates.sums_from_log, 0))) {
        int g = grid_blocks(Lb0 / 64, log_ob);
        g_forms |= kFormSumsDma;
        hipLaunchKernelGGL((k_block_sums_dma<8, 2>), dim3(g), dim3(kThreads), 0, st, (const fq*)T, L, log_ob, (fq*)sums_target(g, log_ob, partials, out));
        reduce_groups_unless_single(g, log_ob, partials, out, st);
        return g;
    }
    int g = grid_blocks(L >> log_ob, log_ob);
    g_forms |= kFormSumsReg;
    hipLaunchKernelGGL(k_block_sums, dim3(g), dim3(kThreads), 0, st, (const fq*)T, L, log_ob, (fq*)sums_target(g, log_ob, partials, out));
    reduce_groups_unless_single(g, log_ob, partials, out, st);
    return g;
}
template <int M>
static int launch_multifold_m(const void* src, void* dst, size_t Lout, const uint64_t* r, int log_ob, void* partials, void* out, hipStream_t st) {
    Chal<M> ch;
    for (int l = 0; l < M; ++l) ch.r[l] = to_fq(r + 4 * l);
    int g = grid_blocks(Lout >> log_ob, log_ob);
    hipLaunchKernelGGL((k_multifold<M>), dim3(g), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, Lout, ch, log_ob, (fq*)sums_target(g, log_ob, partials, out));
    return g;
}
template <int M, int G>
static int launch_multifold_lazy_m(const void* src, void* dst, size_t Lout, const uint32_t* w28, int log_ob, void* partials, void* out, hipStream_t st) {
    W28<M> W;
    for (int b = 0; b < (1 << M); ++b)
        for (int j = 0; j < 10; ++j) W.w[b][j] = w28[b * 10 + j];
    int g = grid_blocks(Lout >> log_ob, log_ob);
    hipLaunchKernelGGL((k_multifold_lazy<M, G>), dim3(g), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, Lout, W, log_ob, (fq*)sums_target(g, log_ob, partials, out));
    return g;
}
int launch_multifold_lazy(int m, const void* src, void* dst, size_t Lout, const uint32_t* w28, int log_ob, void* partials, void* out,
                          hipStream_t st) {
    int g = 0;
    switch (m) {
        case 1: g = launch_multifold_lazy_m<1, 2>(src, dst, Lout, w28, log_ob, partials, out, st); break;
        case 2: g = launch_multifold_lazy_m<2, 4>(src, dst, Lout, w28, log_ob, partials, out, st); break;
        case 3: g = launch_multifold_lazy_m<3, 4>(src, dst, Lout, w28, log_ob, partials, out, st); break;
        case 4: g = launch_multifold_lazy_m<4, 4>(src, dst, Lout, w28, log_ob, partials, out, st); break;
        default: return -1;
    }
    reduce_groups_unless_single(g, log_ob, partials, out, st);
    return g;
}
int launch_multifold_wide(int m, const void* src, void* dst, size_t Lout, const uint64_t* r, const uint64_t two24[4], void* wbuf,
                          int lds_below_log, int coop_below_log, int log_ob, void* partials, void* out, hipStream_t st, const DmaGates& gates) {
    const bool stream_stores = gates.wide_stream_from_log < 63 && Lout >= ((size_t)1 << std::max(gates.wide_stream_from_log, 0));
    if (m < 3 || m > kMaxFoldBatch || (Lout >> log_ob) >= ((size_t)1 << 31)) return -1;
    const EqHalves ch = make_eq_halves(r, m, two24);
    if (coop_below_log > 0 && m >= 6 && Lout < ((size_t)1 << std::min(coop_below_log, 40))) { // tiny output: one wave per output
        size_t per = ((Lout >> log_ob) + 3) / 4, cap = (size_t)kMaxBlocks >> log_ob;
        if (per > cap) per = cap;
        if (per < 1) per = 1;
        const int g = (int)(per << log_ob);
        g_forms |= kFormCoop;
        hipLaunchKernelGGL(k_multifold_coop, dim3(g), dim3(kThreads), 0, st, ch, (const fq*)src, (fq*)dst, Lout, m, log_ob, (fq*)sums_target(g, log_ob, partials, out));
        reduce_groups_unless_single(g, log_ob, partials, out, st);
        return g;
    }
    // small outputs, and blocks that are not whole 64-entry rows (the LDS-DMA form needs them): weights per workgroup in LDS
    const bool rows = ((Lout >> log_ob) % 64) == 0;
    if (!rows || (lds_below_log > 0 && (lds_below_log >= 63 || Lout < ((size_t)1 << lds_below_log)))) {
        int g = grid_blocks(Lout >> log_ob, log_ob);
        g_forms |= kFormWideLds;
        hipLaunchKernelGGL(k_multifold_wide_lds, dim3(g), dim3(kThreads), 0, st, ch, (const fq*)src, (fq*)dst, Lout, m, log_ob, (fq*)sums_target(g, log_ob, partials, out));
        reduce_groups_unless_single(g, log_ob, partials, out, st);
        return g;
    }
    if (gates.wide_mfma_from_log < 63 && Lout >= ((size_t)1 << std::max(gates.wide_mfma_from_log, 0))) {
        // the contraction on the matrix cores: weight tiles by a small multi-workgroup kernel, then the LDS-DMA stream
        const fq kfix = {{0xe5f7fc63u, 0xf613786fu, 0x795bc6d2u, 0xbe62fc84u, 0x799d7fdeu, 0xed6bec43u, 0x8c88f1adu, 0x164bcf99u}};
        hipLaunchKernelGGL(k_eq_weights_mfma, dim3(1u << (m - 1)), dim3(kThreads), 0, st, ch, m, (u32*)wbuf); // one workgroup per b-pair's tiles
        size_t per = ((Lout >> log_ob) / 64 + 3) / 4, cap = (size_t)(256 * gates.wide_mfma_wg_per_cu) >> log_ob;
        if (per > cap) per = cap;
        if (per < 1) per = 1;
        const int g = (int)(per << log_ob);
        g_forms |= kFormWideMfma;
#define ZK_MFMA_LAUNCH(BIGV, KV)                                                                                                        \
    hipLaunchKernelGGL((k_multifold_wide_mfma<BIGV, KV>), dim3(g), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, Lout, (const u32x4*)wbuf, m, \
                       log_ob, kfix, (fq*)sums_target(g, log_ob, partials, out))
        const int depth = gates.wide_mfma_depth;
        if (stream_stores) {
            if (depth == 2) ZK_MFMA_LAUNCH(true, 2); else if (depth == 3) ZK_MFMA_LAUNCH(true, 3); else ZK_MFMA_LAUNCH(true, 4);
        } else {
            if (depth == 2) ZK_MFMA_LAUNCH(false, 2); else if (depth == 3) ZK_MFMA_LAUNCH(false, 3); else ZK_MFMA_LAUNCH(false, 4);
        }
#undef ZK_MFMA_LAUNCH
        reduce_groups_unless_single(g, log_ob, partials, out, st);
        return g;
    }
    // LDS-DMA streams (every large pass): 5.9 vs 5.5 TB/s for register loads at 2^28 entries; ring depth 3, 4, 6 alike.  The eq
    // weights come from a one-workgroup kernel and are read as scalar loads.
    hipLaunchKernelGGL(k_eq_weights28, dim3(1), dim3(kThreads), 0, st, ch, m, (u32*)wbuf);
    // one group of 64 outputs per wave-step; at most 1024 workgroups = 4 per CU (at 2^28 entries: exactly 4 groups per wave)
    size_t per = ((Lout >> log_ob) / 64 + 3) / 4, cap = (size_t)1024 >> log_ob;
    if (per > cap) per = cap;
    if (per < 1) per = 1;
    const int g = (int)(per << log_ob);
    g_forms |= stream_stores ? kFormWideDmaStream : kFormWideDma;
    if (stream_stores)
        hipLaunchKernelGGL((k_multifold_wide_dma<true, 4>), dim3(g), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, Lout, (const u32*)wbuf, m, log_ob,
                           (fq*)sums_target(g, log_ob, partials, out));
    else
        hipLaunchKernelGGL((k_multifold_wide_dma<false, 4>), dim3(g), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, Lout, (const u32*)wbuf, m, log_ob,
                           (fq*)sums_target(g, log_ob, partials, out));
    reduce_groups_unless_single(g, log_ob, partials, out, st);
    return g;
}
int launch_multifold(int m, const void* src, void* dst, size_t Lout, const uint64_t* r, int log_ob, void* partials, void* out, hipStream_t st) {
    int g = 0;
    switch (m) {
        case 1: g = launch_multifold_m<1>(src, dst, Lout, r, log_ob, partials, out, st); break;
        case 2: g = launch_multifold_m<2>(src, dst, Lout, r, log_ob, partials, out, st); break;
        case 3: g = launch_multifold_m<3>(src, dst, Lout, r, log_ob, partials, out, st); break;
        case 4: g = launch_multifold_m<4>(src, dst, Lout, r, log_ob, partials, out, st); break;
        default: return -1;
    }
    reduce_groups_unless_single(g, log_ob, partials, out, st);
    return g;
}
void launch_fold_var(const void* src, void* dst, size_t n_out, int log_s, const uint64_t r[4], hipStream_t st) {
    hipLaunchKernelGGL(k_fold_var, dim3(grid_for(n_out)), dim3(kThreads), 0, st, (const fq*)src, (fq*)dst, n_out, log_s, to_fq(r));
}
void launch_quot_rem_var(const void* src, void* quot, void* rem, size_t n_out, int log_s, const uint64_t r[4], hipStream_t st) {
    hipLaunchKernelGGL(k_quot_rem_var, dim3(grid_for(n_out)), dim3(kThreads), 0, st, (const fq*)src, (fq*)quot, (fq*)rem, n_out, log_s, to_fq(r));
}
void launch_minus(const void* in, void* out, size_t n, const uint64_t v[4], hipStream_t st) {
    hipLaunchKernelGGL(k_minus, dim3(grid_for(n)), dim3(kThreads), 0, st, (const fq*)in, (fq*)out, n, to_fq(v));
}
void launch_reduce_partials(const void* partials, int nblocks, int ns, void* out, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kThreads), 0, st, (const fq*)partials, nblocks, ns, (fq*)out);
}
int launch_sum_all(const void* T, size_t n, void* partials, hipStream_t st) {
    int g = grid_for(n);
    hipLaunchKernelGGL(k_sum_all, dim3(g), dim3(kThreads), 0, st, (const fq*)T, n, (fq*)partials);
    return g;
}
void launch_interleave(const void* in, void* out, int world, int ntab, size_t L, hipStream_t st) {
    hipLaunchKernelGGL(k_interleave, dim3(grid_for((size_t)world * ntab * L)), dim3(kThreads), 0, st, (const fq*)in, (fq*)out, (unsigned)world,
                       (unsigned)ntab, L);
}
void launch_count_noncanonical(const void* T, size_t n, void* count_u64, hipStream_t st) {
    hipLaunchKernelGGL(k_count_noncanonical, dim3(grid_for(n)), dim3(kThreads), 0, st, (const fq*)T, n, (unsigned long long*)count_u64);
}
void launch_from_le32(const void* in, void* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_from_le32, dim3(grid_for(n)), dim3(kThreads), 0, st, (const fq*)in, (fq*)out, n);
}
void launch_to_le32(const void* in, void* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_to_le32, dim3(grid_for(n)), dim3(kThreads), 0, st, (const fq*)in, (fq*)out, n);
}
void launch_generate(void* out, uint64_t seed, uint64_t first, uint64_t step, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(n)), dim3(kThreads), 0, st, (fq*)out, seed, first, step, n);
}
void launch_modmul_probe(void* io, int blocks, int iters, hipStream_t st) {
    hipLaunchKernelGGL(k_modmul_probe, dim3(blocks), dim3(kThreads), 0, st, (fq*)io, iters);
}
void launch_mad_probe(void* out, int blocks, int iters, hipStream_t st) {
    hipLaunchKernelGGL(k_mad_probe, dim3(blocks), dim3(kThreads), 0, st, (u32*)out, iters);
}

// ---- static scratch (zk_fold_block_sums_noalloc: the entry point that allocates nothing) ----------------------
// Every GPU the library runs on holds its own copy of this array, at its own address, until the process ends.  In host
// code the name `scratch` is only a stand-in for the symbol: the device address comes from hipGetSymbolAddress, taken
// once per device with that device current, and host code uses that pointer (kernel arguments) -- never the name.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

hipError_t static_scratch_ptr(void** out) {
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static void* ptrs[kMaxDevices] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        ptrs[dev] = p;
    }
    *out = ptrs[dev];
    return hipSuccess;
}

} // namespace zkk
