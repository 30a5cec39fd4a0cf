// gkr_kernels.hip -- gfx950 kernels of the GKR layer prover (gkr/src/prover.rs:24-154).
//
// The reference builds DENSE indicator tables add_i/mul_i of 2^(k_out + 2 k_in) entries
// (circuit.rs:82-123) and runs the degree-2 sumcheck over the 2^(2 k_in)-entry tables
// mul'(b,c), W(b)W(c), add'(b,c), W(b)+W(c).  Here the wiring stays a gate list.  With
//   wt[o]   = eq(rho, o)                              (layer 0, utils.rs:45-65)
//           = alpha*eq(r_b, o) + beta*eq(r_c, o)      (later layers, utils.rs:16-43)
// the folded predicates are  mul'(b,c) = sum_{mul gates g} wt[o_g] [b = l_g][c = r_g]  (add' alike), and
// the layer relation sums over c (first k_in rounds) resp. is evaluated at b = u (last k_in rounds) in
// closed form: the round polynomials are the SAME field elements as the dense computation yields
// (exact arithmetic), from tables of 2^k_in entries.  All kernels are gathers over small tables
// (<= a few MiB per layer): latency-, not bandwidth-bound.
#include "fq_dev.hpp"
#include "fq_host.hpp"
#include "gkr_kernels.h"

namespace zkk {

static constexpr int kT = 256;

static inline int grid1d(size_t n) {
    size_t b = (n + kT - 1) / kT;
    if (b < 1) b = 1;
    if (b > 65535) b = 65535;
    return (int)b;
}
static inline fq to_fq4(const uint64_t r[4]) {
    fq f;
    for (int i = 0; i < 4; ++i) {
        f.v[2 * i] = (u32)r[i];
        f.v[2 * i + 1] = (u32)(r[i] >> 32);
    }
    return f;
}
__device__ __forceinline__ fq fq_one_mont() {
    const fq one = {FQ_R_LIMBS};
    return one;
}

__global__ void __launch_bounds__(kT) k_circuit_layer(const fq* __restrict__ in, fq* __restrict__ out, const u32* __restrict__ left,
                                                      const u32* __restrict__ right, const uint8_t* __restrict__ op, size_t n_gates,
                                                      size_t n_out) {
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t g = (size_t)blockIdx.x * kT + threadIdx.x; g < n_out; g += stride) {
        fq o = fq_zero();
        if (g < n_gates) {
            const fq a = fq_load(in + left[g]), b = fq_load(in + right[g]);
            o = op[g] ? fq_mul(a, b) : fq_add(a, b);
        }
        fq_store(out + g, o);
    }
}

struct EqPts { fq x[kMaxEqVars]; fq scale; };
// up to four half tables in one launch (blockIdx.y = table): out[o] = scale * prod_m (bit (k-1-m) of o ? x[m] : 1 - x[m]).
// A thread's k factors are multiplied as two interleaved chains (even and odd m): the kernel is one wave-step deep, so its time
// is the dependent-product chain.
struct EqJobs { fq* out[4]; int k[4]; EqPts P[4]; };
__global__ void __launch_bounds__(kT) k_eq_small(EqJobs J) {
    const fq one = fq_one_mont();
    const int job = blockIdx.y, k = J.k[job];
    fq* __restrict__ out = J.out[job];
    const EqPts& P = J.P[job];
    const size_t n = (size_t)1 << k, stride = (size_t)gridDim.x * kT;
    for (size_t o = (size_t)blockIdx.x * kT + threadIdx.x; o < n; o += stride) {
        fq a = P.scale, b = one;
        bool b_used = false;
        for (int m = 0; m < k; m += 2) { // x[m] <-> bit (k-1-m) of o
            const fq xm = P.x[m];
            a = fq_mul(a, ((o >> (k - 1 - m)) & 1) ? xm : fq_sub(one, xm));
            if (m + 1 < k) {
                const fq xn = P.x[m + 1];
                const fq f = ((o >> (k - 2 - m)) & 1) ? xn : fq_sub(one, xn);
                b = b_used ? fq_mul(b, f) : f;
                b_used = true;
            }
        }
        fq_store(out + o, b_used ? fq_mul(a, b) : a);
    }
}

// The same tables for k <= 2 * kEqPairs variables from host-built TWO-variable tables: pair p covers x[2p], x[2p + 1] (bits k-1-2p and
// k-2-2p of o; a last pair of an odd k covers x[k-1] alone), its four entries eq((x[2p], x[2p+1]), (b, b')) at index 2b + b' cost the
// host one product, the scale sits in pair 0, and an entry of the table is a product of ceil(k/2) factors: 4 products per entry instead
// of 9-10 at k = 10 (the half tables of every GKR layer at 2^20 gates).  The tables are the kernel's FIRST argument and are read through the
// kernel-argument segment with a lane-dependent index (as eq_half_arg in kernels_common.hpp).  Same field elements as k_eq_small.
static constexpr int kEqPairs = 5;
struct EqPairJobs { fq tab[4][kEqPairs][4]; fq* out[4]; int k[4]; };
__device__ __forceinline__ fq eq_pair_arg(int idx) {
    typedef const __attribute__((address_space(4))) u32x4* kptr;
    kptr p = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + 2 * idx;
    const u32x4 lo = p[0], hi = p[1];
    fq x;
#pragma unroll
    for (int w = 0; w < 4; ++w) { x.v[w] = lo[w]; x.v[4 + w] = hi[w]; }
    return x;
}
__global__ void __launch_bounds__(kT) k_eq_small_pairs(EqPairJobs J) {
    const int job = blockIdx.y, k = J.k[job], np = (k + 1) >> 1;
    fq* __restrict__ out = J.out[job];
    const size_t n = (size_t)1 << k, stride = (size_t)gridDim.x * kT;
    for (size_t o = (size_t)blockIdx.x * kT + threadIdx.x; o < n; o += stride) {
        fq acc = fq_zero();
#pragma unroll
        for (int p = 0; p < kEqPairs; ++p) {
            if (p < np) {
                const int sh = k - 2 - 2 * p; // the pair's low bit; -1 for the single variable that ends an odd k
                const int bits = sh >= 0 ? (int)((o >> sh) & 3) : (int)((o & 1) << 1);
                const fq f = eq_pair_arg((job * kEqPairs + p) * 4 + bits);
                acc = p == 0 ? f : fq_mul(acc, f);
            }
        }
        fq_store(out + o, acc);
    }
}

__global__ void __launch_bounds__(kT) k_eq_outer(fq* __restrict__ out, size_t n, int log_lo, const fq* __restrict__ hiA,
                                                 const fq* __restrict__ loA, const fq* __restrict__ hiB, const fq* __restrict__ loB) {
    const size_t mask = ((size_t)1 << log_lo) - 1, stride = (size_t)gridDim.x * kT;
    for (size_t o = (size_t)blockIdx.x * kT + threadIdx.x; o < n; o += stride) {
        fq v = fq_mul(fq_load(hiA + (o >> log_lo)), fq_load(loA + (o & mask)));
        if (hiB) v = fq_add(v, fq_mul(fq_load(hiB + (o >> log_lo)), fq_load(loB + (o & mask))));
        fq_store(out + o, v);
    }
}

__global__ void __launch_bounds__(kT) k_fill_one(fq* __restrict__ out, size_t n) {
    const fq one = fq_one_mont();
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += stride) fq_store(out + i, one);
}

// Rows b = i * step + first, i < n_loc (step = 1, first = 0: the whole table; a rank of a sharded prove takes the rows that are
// `first` mod `step` and stores them at the local index i: the table is sharded on its LOW index bits, as sharded.cpp needs it)
__global__ void __launch_bounds__(kT) k_gkr_phase1(fq* __restrict__ A, fq* __restrict__ H2, const u32* __restrict__ lptr,
                                                   const u32* __restrict__ lidx, const u32* __restrict__ right,
                                                   const uint8_t* __restrict__ op, const fq* __restrict__ wt, const fq* __restrict__ W,
                                                   size_t n_loc, size_t step, size_t first) {
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n_loc; i += stride) {
        const size_t b = i * step + first;
        fq a = fq_zero(), h2 = fq_zero();
        for (u32 e = lptr[b]; e < lptr[b + 1]; ++e) {
            const u32 g = lidx[e]; // the gate's output index
            const fq w = fq_load(wt + g);
            const fq t = fq_mul(w, fq_load(W + right[g]));
            if (op[g]) a = fq_add(a, t);
            else { a = fq_add(a, w); h2 = fq_add(h2, t); }
        }
        fq_store(A + i, a);
        fq_store(H2 + i, h2);
    }
}

// With m(c), ad(c) the mul and add predicates at (rho, u, c) the layer relation over c is  m(c) wu W(c) + ad(c) (wu + W(c))
// (wu = W(u)); as polynomials in c that is  P(c) W(c) + Q(c) * 1  with  P = wu m + ad,  Q = wu ad  -- the same round polynomials in
// exact arithmetic, from three tables and a constant instead of four tables (phase 1 has that shape already: A W + H2 * 1).
__global__ void __launch_bounds__(kT) k_gkr_phase2(fq* __restrict__ P, fq* __restrict__ Q,
                                                   const u32* __restrict__ rptr, const u32* __restrict__ ridx,
                                                   const u32* __restrict__ left, const uint8_t* __restrict__ op,
                                                   const fq* __restrict__ wt, const fq* __restrict__ Eu, fq wu,
                                                   size_t n_loc, size_t step, size_t first) {
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n_loc; i += stride) {
        const size_t c = i * step + first;
        fq m = fq_zero(), ad = fq_zero();
        for (u32 e = rptr[c]; e < rptr[c + 1]; ++e) {
            const u32 g = ridx[e];
            const fq t = fq_mul(fq_load(wt + g), fq_load(Eu + left[g]));
            if (op[g]) m = fq_add(m, t);
            else ad = fq_add(ad, t);
        }
        fq_store(P + i, fq_add(fq_mul(m, wu), ad));
        fq_store(Q + i, fq_mul(ad, wu));
    }
}

// out[i] = in[i * step + first], i < n_loc: a rank's shard of a replicated table
__global__ void __launch_bounds__(kT) k_stride_gather(fq* __restrict__ out, const fq* __restrict__ in, size_t n_loc, size_t step, size_t first) {
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n_loc; i += stride) fq_store(out + i, fq_load(in + i * step + first));
}

// verifier side: add'(u,v) and mul'(u,v) from the gate list, sum_g wt[o_g] * eq(u, l_g) * eq(v, r_g) split by gate type
// (what `new_addi_b_c.evaluate(&next_evaluation_values)` yields on the dense tables, gkr/src/verifier.rs:67-73).
// partials[2*block + 0] = add share, [2*block + 1] = mul share
__global__ void __launch_bounds__(kT) k_gkr_pred_eval(const u32* __restrict__ left, const u32* __restrict__ right, const uint8_t* __restrict__ op,
                                                      const fq* __restrict__ wt, const fq* __restrict__ Eu, const fq* __restrict__ Ev, size_t n_gates,
                                                      fq* __restrict__ partials) {
    fq acc[2] = {fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t g = (size_t)blockIdx.x * kT + threadIdx.x; g < n_gates; g += stride) {
        const fq t = fq_mul(fq_mul(fq_load(wt + g), fq_load(Eu + left[g])), fq_load(Ev + right[g]));
        if (op[g]) acc[1] = fq_add(acc[1], t);
        else acc[0] = fq_add(acc[0], t);
    }
    __shared__ fq lds[2][kT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fq w = fq_wave_sum(acc[s]);
        if (lane == 0) lds[s][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        fq t = lds[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < kT / 64; ++w) t = fq_add(t, lds[threadIdx.x][w]);
        fq_store(partials + (size_t)blockIdx.x * 2 + threadIdx.x, t);
    }
}

__global__ void __launch_bounds__(kT) k_gate_poly(fq* __restrict__ out, const u32* __restrict__ left, const u32* __restrict__ right,
                                                  const uint8_t* __restrict__ op, size_t n_gates, int k_in, int want_op) {
    const fq one = fq_one_mont();
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t g = (size_t)blockIdx.x * kT + threadIdx.x; g < n_gates; g += stride)
        if ((int)op[g] == want_op) fq_store(out + ((((g << k_in) | left[g]) << k_in) | right[g]), one); // get_bit_idx, circuit.rs:66-74
}

__global__ void __launch_bounds__(kT) k_w_op(fq* __restrict__ out, const fq* __restrict__ wb, const fq* __restrict__ wc, size_t L, int log_L,
                                             int mul) {
    const size_t n = L * L, stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += stride) {
        const fq a = fq_load(wb + (i >> log_L)), b = fq_load(wc + (i & (L - 1)));
        fq_store(out + i, mul ? fq_mul(a, b) : fq_add(a, b));
    }
}

__global__ void __launch_bounds__(kT) k_scalar_mul(fq* __restrict__ out, const fq* __restrict__ in, fq s, size_t n) {
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += stride) fq_store(out + i, fq_mul_uniform(fq_load(in + i), s));
}
__global__ void __launch_bounds__(kT) k_add(fq* __restrict__ out, const fq* __restrict__ a, const fq* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * kT;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += stride) fq_store(out + i, fq_add(fq_load(a + i), fq_load(b + i)));
}

// ---- launchers -------------------------------------------------------------------------------
void launch_circuit_layer(const void* in, void* out, const uint32_t* left, const uint32_t* right, const uint8_t* op, size_t n_gates,
                          size_t n_out, hipStream_t st) {
    hipLaunchKernelGGL(k_circuit_layer, dim3(grid1d(n_out)), dim3(kT), 0, st, (const fq*)in, (fq*)out, left, right, op, n_gates, n_out);
}
void launch_eq_small_multi(int njobs, void* const out[], const int k[], const uint64_t* const x[], const uint64_t* const scale[], hipStream_t st) {
    bool pairs = njobs >= 1 && njobs <= 4;
    for (int j = 0; j < njobs && j < 4; ++j) pairs = pairs && k[j] >= 1 && k[j] <= 2 * kEqPairs;
    if (pairs) { // short halves: two-variable tables from the host, ceil(k/2) - 1 products per entry
        EqPairJobs J{};
        int kmax = 0;
        const zkh::Fq one = zkh::one();
        for (int j = 0; j < njobs; ++j) {
            J.out[j] = (fq*)out[j];
            J.k[j] = k[j];
            kmax = k[j] > kmax ? k[j] : kmax;
            const zkh::Fq sc = zkh::load(scale[j]);
            for (int p = 0; p < (k[j] + 1) / 2; ++p) {
                zkh::Fq e[4];
                const zkh::Fq a = zkh::load(x[j] + 4 * (2 * p));
                if (2 * p + 1 < k[j]) {
                    const zkh::Fq b = zkh::load(x[j] + 4 * (2 * p + 1)), ab = zkh::mul(a, b);
                    e[3] = ab;
                    e[2] = zkh::sub(a, ab);                                   // a (1 - b)
                    e[1] = zkh::sub(b, ab);                                   // (1 - a) b
                    e[0] = zkh::sub(zkh::sub(one, a), e[1]);                  // (1 - a)(1 - b) = 1 - a - (b - ab)
                } else { // the single last variable: index 2 * bit
                    e[0] = zkh::sub(one, a);
                    e[2] = a;
                    e[1] = e[3] = zkh::Fq{{0, 0, 0, 0}};
                }
                for (int i = 0; i < 4; ++i) J.tab[j][p][i] = to_fq4((p == 0 ? zkh::mul(e[i], sc) : e[i]).l);
            }
        }
        hipLaunchKernelGGL(k_eq_small_pairs, dim3(grid1d((size_t)1 << kmax), njobs), dim3(kT), 0, st, J);
        return;
    }
    EqJobs J{};
    int kmax = 0;
    for (int j = 0; j < njobs && j < 4; ++j) {
        J.out[j] = (fq*)out[j];
        J.k[j] = k[j];
        for (int m = 0; m < kMaxEqVars; ++m) J.P[j].x[m] = m < k[j] ? to_fq4(x[j] + 4 * m) : to_fq4(scale[j]);
        J.P[j].scale = to_fq4(scale[j]);
        kmax = k[j] > kmax ? k[j] : kmax;
    }
    hipLaunchKernelGGL(k_eq_small, dim3(grid1d((size_t)1 << kmax), njobs), dim3(kT), 0, st, J);
}
void launch_eq_small(void* out, int k, const uint64_t* x, const uint64_t scale[4], hipStream_t st) {
    void* const o[1] = {out};
    const int kk[1] = {k};
    const uint64_t* const xx[1] = {x};
    const uint64_t* const ss[1] = {scale};
    launch_eq_small_multi(1, o, kk, xx, ss, st);
}
void launch_eq_outer(void* out, size_t n, int log_lo, const void* hiA, const void* loA, const void* hiB, const void* loB, hipStream_t st) {
    hipLaunchKernelGGL(k_eq_outer, dim3(grid1d(n)), dim3(kT), 0, st, (fq*)out, n, log_lo, (const fq*)hiA, (const fq*)loA, (const fq*)hiB,
                       (const fq*)loB);
}
void launch_fill_one(void* out, size_t n, hipStream_t st) { hipLaunchKernelGGL(k_fill_one, dim3(grid1d(n)), dim3(kT), 0, st, (fq*)out, n); }
void launch_gkr_phase1(void* A, void* H2, const uint32_t* lptr, const uint32_t* lidx, const uint32_t* right, const uint8_t* op,
                       const void* wt, const void* W, size_t n_loc, size_t step, size_t first, hipStream_t st) {
    hipLaunchKernelGGL(k_gkr_phase1, dim3(grid1d(n_loc)), dim3(kT), 0, st, (fq*)A, (fq*)H2, lptr, lidx, right, op, (const fq*)wt, (const fq*)W,
                       n_loc, step, first);
}
void launch_gkr_phase2(void* P, void* Q, const uint32_t* rptr, const uint32_t* ridx, const uint32_t* left, const uint8_t* op,
                       const void* wt, const void* Eu, const uint64_t wu[4], size_t n_loc, size_t step, size_t first, hipStream_t st) {
    hipLaunchKernelGGL(k_gkr_phase2, dim3(grid1d(n_loc)), dim3(kT), 0, st, (fq*)P, (fq*)Q, rptr, ridx, left, op, (const fq*)wt,
                       (const fq*)Eu, to_fq4(wu), n_loc, step, first);
}
void launch_stride_gather(void* out, const void* in, size_t n_loc, size_t step, size_t first, hipStream_t st) {
    hipLaunchKernelGGL(k_stride_gather, dim3(grid1d(n_loc)), dim3(kT), 0, st, (fq*)out, (const fq*)in, n_loc, step, first);
}
int launch_gkr_pred_eval(const uint32_t* left, const uint32_t* right, const uint8_t* op, const void* wt, const void* Eu, const void* Ev,
                         size_t n_gates, void* partials, hipStream_t st) {
    int g = grid1d(n_gates);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_gkr_pred_eval, dim3(g), dim3(kT), 0, st, left, right, op, (const fq*)wt, (const fq*)Eu, (const fq*)Ev, n_gates,
                       (fq*)partials);
    return g;
}
void launch_gate_poly(void* out, size_t n_total, const uint32_t* left, const uint32_t* right, const uint8_t* op, size_t n_gates, int k_in,
                      int want_op, hipStream_t st) {
    (void)hipMemsetAsync(out, 0, n_total * sizeof(fq), st);
    if (n_gates) hipLaunchKernelGGL(k_gate_poly, dim3(grid1d(n_gates)), dim3(kT), 0, st, (fq*)out, left, right, op, n_gates, k_in, want_op);
}
void launch_w_op(void* out, const void* wb, const void* wc, size_t L, int mul, hipStream_t st) {
    int log_L = 0;
    while (((size_t)1 << log_L) < L) ++log_L;
    hipLaunchKernelGGL(k_w_op, dim3(grid1d(L * L)), dim3(kT), 0, st, (fq*)out, (const fq*)wb, (const fq*)wc, L, log_L, mul);
}
void launch_scalar_mul(void* out, const void* in, const uint64_t s[4], size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_scalar_mul, dim3(grid1d(n)), dim3(kT), 0, st, (fq*)out, (const fq*)in, to_fq4(s), n);
}
void launch_add(void* out, const void* a, const void* b, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_add, dim3(grid1d(n)), dim3(kT), 0, st, (fq*)out, (const fq*)a, (const fq*)b, n);
}

} // namespace zkk
