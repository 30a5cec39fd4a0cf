// gkr_kernels.h -- launchers of the GKR layer-prover kernels (gkr_kernels.hip).
// Device pointers to 32-byte Montgomery elements unless a name says otherwise; u32 arrays are
// gate records / CSR indices built once per circuit on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace zkk {

static constexpr int kMaxEqVars = 16; // variables of one eq half-table (full tables: up to 2 x 16)

// evaluate_at_input (circuit.rs:37-59), one layer: out[g] = in[left[g]] (+|*) in[right[g]] for g < n_gates,
// 0 for n_gates <= g < n_out
void launch_circuit_layer(const void* in, void* out, const uint32_t* left, const uint32_t* right, const uint8_t* op,
                          size_t n_gates, size_t n_out, hipStream_t st);
// out[o] = scale * eq(x, o), o < 2^k, x[0] belongs to the most significant bit of o (k <= kMaxEqVars);
// x: k elements as 4 x u64 Montgomery limbs each
void launch_eq_small(void* out, int k, const uint64_t* x, const uint64_t scale[4], hipStream_t st);
// up to four such half tables in ONE launch (they are independent; each is a chain of dependent products)
void launch_eq_small_multi(int njobs, void* const out[], const int k[], const uint64_t* const x[], const uint64_t* const scale[], hipStream_t st);
// out[o] = hiA[o >> log_lo] * loA[o & (2^log_lo - 1)]  (+ hiB[..] * loB[..] when hiB != NULL), o < n
void launch_eq_outer(void* out, size_t n, int log_lo, const void* hiA, const void* loA, const void* hiB, const void* loB,
                     hipStream_t st);
void launch_fill_one(void* out, size_t n, hipStream_t st);
// phase 1 (the k_in "b" variables): for b = i * step + first, i < n_loc (step 1, first 0: every b; a rank of a sharded prove takes
// the b that are `first` mod `step`: tables sharded on their low index bits), over the gates whose LEFT input is b
// (CSR: gate ids lidx[lptr[b] .. lptr[b+1])):
//   A[i]  = sum_{mul gates} wt[o] * W[right]  +  sum_{add gates} wt[o]
//   H2[i] = sum_{add gates} wt[o] * W[right]
void launch_gkr_phase1(void* A, void* H2, const uint32_t* lptr, const uint32_t* lidx, const uint32_t* right, const uint8_t* op,
                       const void* wt, const void* W, size_t n_loc, size_t step, size_t first, hipStream_t st);
// phase 2 (the k_in "c" variables), b bound to u: for c = i * step + first, i < n_loc, over the gates whose RIGHT input is c, with
// m = sum_{mul gates} wt[o] * Eu[left] and ad = sum_{add gates} wt[o] * Eu[left]:
//   P[i] = wu * m + ad,  Q[i] = wu * ad     (the relation m wu W(c) + ad (wu + W(c)) written as P W + Q * 1)
void launch_gkr_phase2(void* P, void* Q, const uint32_t* rptr, const uint32_t* ridx, const uint32_t* left, const uint8_t* op,
                       const void* wt, const void* Eu, const uint64_t wu[4], size_t n_loc, size_t step, size_t first, hipStream_t st);
// out[i] = in[i * step + first], i < n_loc
void launch_stride_gather(void* out, const void* in, size_t n_loc, size_t step, size_t first, hipStream_t st);
// verifier: per-block shares (2 per block: add, mul) of sum_g wt[o_g] eq(u,l_g) eq(v,r_g); returns the block count (<= 1024)
int launch_gkr_pred_eval(const uint32_t* left, const uint32_t* right, const uint8_t* op, const void* wt, const void* Eu, const void* Ev,
                         size_t n_gates, void* partials, hipStream_t st);
// the reference's dense tables (small circuits / API parity)
// get_gate_poly (circuit.rs:82-123): out has n_out << 2*k_in entries, zero except 1 at (o << 2k_in | l << k_in | r)
void launch_gate_poly(void* out, size_t n_total, const uint32_t* left, const uint32_t* right, const uint8_t* op, size_t n_gates,
                      int k_in, int want_op, hipStream_t st);
// operate_w (evaluation_form.rs:82-112): out[i] = wb[i / L] (+|*) wc[i % L], L = len, i < L*L
void launch_w_op(void* out, const void* wb, const void* wc, size_t L, int mul, hipStream_t st);
// scalar_mul (:151-153) and _add (:241-258)
void launch_scalar_mul(void* out, const void* in, const uint64_t s[4], size_t n, hipStream_t st);
void launch_add(void* out, const void* a, const void* b, size_t n, hipStream_t st);

} // namespace zkk
