// kernels.h -- host-callable launchers of the gfx950 kernels in kernels.hip and deg2_kernels.hip.
// All pointers are device pointers to 32-byte Montgomery elements; `r` is a challenge as
// 4 x u64 Montgomery limbs (passed to the kernel by value: it lands in SGPRs).
// Launchers that produce per-block partial sums return the number of blocks launched;
// `partials` must hold kMaxBlocks * 3 elements.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace zkk {

// 256 CUs x 8 blocks of 256 threads: enough waves to cover HBM latency, few enough partials
// that the second stage stays a single block (cdna_hip_programming.md, Guideline 11)
static constexpr int kMaxBlocks = 2048;

static constexpr int kMaxFoldBatch = 8;  // rounds bound per pass (k_multifold_wide)
static constexpr int kMaxLazyBatch = 4;  // ... by the kernel-argument-weight forms (k_multifold<M>, k_multifold_lazy<M,G>)
// Which form of a pass a launcher picked, OR-ed into g_forms (per host thread) since the caller last cleared it: the
// parity tests force every form onto oracle-sized tables through the zk_ctx_set_* thresholds and assert that the form
// under test is the one that ran (zk_ctx_forms_used).
enum Form : unsigned {
    kFormSumsDma = 1u << 0,       // k_block_sums_dma
    kFormSumsReg = 1u << 1,       // k_block_sums
    kFormWideDma = 1u << 2,       // k_multifold_wide_dma<false>
    kFormWideDmaStream = 1u << 3, // k_multifold_wide_dma<true> (streaming stores)
    kFormWideReg = 1u << 4,       // (retired: the register-load wide pass lives in tools/fold_variants.hip)
    kFormWideLds = 1u << 5,       // k_multifold_wide_lds
    kFormCoop = 1u << 6,          // k_multifold_coop
    kFormFold2Dma = 1u << 7,      // m = 2 LDS-DMA fold of the degree-2 branch
    kFormFold2Reg = 1u << 8,      // k_fold2_tables
    kFormSum9Lazy = 1u << 9,      // k_sum9_lazy
    kFormSum9 = 1u << 10,         // k_sum9<YSPLIT>
    kFormFold2Sum9 = 1u << 11,    // k_fold2_sum9 (fold fused with the next nine sums)
    kFormWideMfma = 1u << 12,     // k_multifold_wide_mfma (the wide pass on the matrix cores)
    kFormSum9Mfma = 1u << 13,     // k_sum16_mfma (the nine sums on the matrix cores)
    kFormCross64 = 1u << 14,      // k_cross64_mfma (three rounds per pass: 8 x 8 cross sums per term on the matrix cores)
    kFormFold3Dma = 1u << 15,     // m = 3 LDS-DMA fold of the degree-2 branch (three rounds per pass)
    kFormFold3Cross64 = 1u << 16, // k_fold3_cross64 (that fold fused with the next level's cross sums)
    kFormFold3Cross64Mfma = 1u << 17, // k_fold3_cross64_mfma (the same sweep with the fold on the matrix cores)
    kFormFoldSum27 = 1u << 18,    // k_fold_sum27 (small tables: fold + the 27 sums of the next three rounds)
    kFormFold3Reg = 1u << 19,     // k_fold3_tables (small tables: the three-variable fold of all four tables in one launch)
    kFormPwq = 1u << 20,          // k_fold_sum27<M, PWQ = true>: the product-plus-linear-term form of the GKR phases (the constant-one factor is never read)
    kFormSum27First = 1u << 21,   // k_fold_sum27<0, .>: the first pass of a prove as 27 sums (three rounds) instead of nine
};
extern thread_local unsigned g_forms;
// Size thresholds of the LDS-DMA forms (log2 of entries; defaults = the measured break-even points)
struct DmaGates {
    int sums_from_log = 27;        // block sums by LDS-DMA from tables of 2^k entries
    int wide_stream_from_log = 22; // wide fold: streaming (nontemporal) stores from 2^k outputs
    int fold2_from_log = 22;       // degree-2 two-variable fold by LDS-DMA from 2^k outputs per table
    int wide_mfma_from_log = 15;   // wide fold on the matrix cores (k_multifold_wide_mfma) from 2^k outputs; 64 = never
    int wide_mfma_wg_per_cu = 1;   // its grid: workgroups per CU (1 and 2 measure alike; LDS admits two at depth 2)
    int sum9_mfma_from_log = 22;   // nine sums on the matrix cores (k_sum16_mfma) from 2^k quarter-entries (measured break-even); 64 = never
    int wide_mfma_depth = 2;       // its pipeline depth K (2..4): table rows of K steps in flight per wave
    int deg2_triple_from_log = 20; // degree-2 prover: three rounds per pass (k_cross64_mfma + m = 3 fold) on tables of >= 2^k entries; 64 = never.
                                   // A pass that FOLLOWS a three-round pass (the fused fold + cross sums) keeps the form down to results of
                                   // 2^(k-1) entries.  Relations of the product + linear shape (the GKR phases) take these sweeps one size class
                                   // later (k + 1): their 27-sum passes move three tables instead of four (GKR layers 10.0 ms at 21, 10.3 at 20).
                                   // 20 since the raw-digit second stage (21 before): 4 x 2^22 0.526 -> 0.481 ms, 4 x 2^20 0.286 -> 0.273
    int deg2_fused_below_log = 20; // degree-2 prover: fold + nine sums in one kernel (k_fold2_sum9) for folded tables below 2^k entries; 0 = never
    int deg2_sum27_first_below_log = 19; // degree-2 prover: the FIRST pass as 27 sums (k_fold_sum27<0>: three rounds before the first fold) for tables below
                                         // 2^k entries; 0 = never (measured: 4 x 2^18 prove 0.224 -> 0.199 ms, 2^14 0.129 -> 0.122; at 2^20 the nine-sum pass wins)
    int deg2_pwq_first_below_log = 20;   // the same threshold for relations F G + H * 1 (ones3: the GKR phases), whose 27-sum passes run in the
                                         // product-plus-linear form; 0 = that form off (GKR config 5 layers: 11.35 ms off, 10.17 with 20, 10.65 with 21)
    int deg2_sum27_below_log = 20; // degree-2 prover: THREE rounds per pass on small tables too (k_fold_sum27) while the folded tables have fewer
                                   // than 2^k entries; 0 = never (measured: 4 x 2^16 .. 2^24 proves 4-14 % faster with 20, 21 slower at 2^23)
    int deg2_fold3_fused = 8;      // three-round passes: fold fused with the next level's cross sums; 8 = the fold on the matrix cores too
                                   // (k_fold3_cross64_mfma, ring of 4 steps; 12..14: 2..4 steps), 3..5 = DMA slots per wave of the VALU-fold form (k_fold3_cross64), 0 = off
    int deg2_triple_ring = 4;      // its LDS ring: 3 or 4 tiles of 32 KiB
};
// out[b] (b < 2^log_ob) = sum of the b-th contiguous block of T (L entries); partials is scratch
int launch_block_sums(const void* T, size_t L, int log_ob, void* partials, void* out, hipStream_t st, const DmaGates& gates = DmaGates());
// dst (Lout entries) = src (Lout << m entries) with its top m variables bound to r[0..m) (4 limbs each);
// out[c] (c < 2^log_ob) = sum of the c-th contiguous block of dst.  The fold tree (m <= kMaxLazyBatch).
int launch_multifold(int m, const void* src, void* dst, size_t Lout, const uint64_t* r, int log_ob, void* partials, void* out, hipStream_t st);
// the same pass as one inner product with the eq weights: w28[b*10 + j] = limb j (28 bits) of
// eq(r, b) * 2^280 mod p, b < 2^m (b's most significant bit belongs to r[0])
int launch_multifold_lazy(int m, const void* src, void* dst, size_t Lout, const uint32_t* w28, int log_ob, void* partials, void* out,
                          hipStream_t st);
// the same pass for 3 <= m <= kMaxFoldBatch: weights derived on the device from the challenges r[0..m); two24 = 2^24
// (Montgomery).  Outputs below 2^lds_below_log entries: per workgroup in LDS (one launch); larger ones: by a one-workgroup
// kernel into wbuf (device, kWideWeightWords u32), read as scalar loads (blocks that are not whole 64-entry rows take the LDS
// form whatever their size).  partials must hold kMaxBlocks elements
static constexpr int kWideWeightWords = (1 << (kMaxFoldBatch - 1)) * 1024; // the matrix-core form's weight tiles (512 KiB); the scalar-load form uses 2^m x 10 words of it
// Outputs below 2^coop_below_log entries (m >= 6): one wave per output (k_multifold_coop); 0 = never.
int launch_multifold_wide(int m, const void* src, void* dst, size_t Lout, const uint64_t* r, const uint64_t two24[4], void* wbuf,
                          int lds_below_log, int coop_below_log, int log_ob, void* partials, void* out, hipStream_t st,
                          const DmaGates& gates = DmaGates());
void launch_fold_var(const void* src, void* dst, size_t n_out, int log_s, const uint64_t r[4], hipStream_t st);
// quot[m] = y2 - y1, rem[m] = y1 + (y2 - y1) r over the pairs of variable v (log_s = n - 1 - v), as launch_fold_var
void launch_quot_rem_var(const void* src, void* quot, void* rem, size_t n_out, int log_s, const uint64_t r[4], hipStream_t st);
void launch_minus(const void* in, void* out, size_t n, const uint64_t v[4], hipStream_t st);
void launch_reduce_partials(const void* partials, int nblocks, int ns, void* out, hipStream_t st);
int launch_sum_all(const void* T, size_t n, void* partials, hipStream_t st);
int launch_eval3(const void* const t[4], size_t half, void* partials, hipStream_t st);
int launch_fold4_eval3(const void* const s[4], void* const d[4], size_t q, const uint64_t r[4], void* partials, hipStream_t st);
void launch_fold4(const void* const s[4], void* const d[4], size_t half, const uint64_t r[4], hipStream_t st);
// degree-2 branch, two rounds per pass (deg2_kernels.hip): out[3Y + t] (9 elements) = the nine sums of four tables of 4q
// entries; partials must hold kSum9PartialElems elements
static constexpr int kMaxBlocks9 = 1024;
static constexpr size_t kSum9PartialElems = (size_t)160 * kMaxBlocks9 + 256; // elements `partials` must hold for launch_sum9 / launch_cross64
// fix24 = 2^24 (Montgomery); tables of at least 2^lazy_from_log quarter-entries take the unreduced-accumulation kernel
// (lazy_from_log < 0: never)
// Returns 0 when out holds the nine sums; with allow_u, 1 when it holds the nine U sums of k_sum9_lazy instead (out[3 b + a]; the
// caller applies k_sum9_from_u's linear maps on the host: one launch less on the critical path of a latency-bound pass)
// pwq: the relation is t[0] t[1] + t[2] * 1 and the lazy sweep runs its linear term as additions (t[3] is not read)
int launch_sum9(const void* const t[4], size_t q, const uint64_t fix24[4], int lazy_from_log, void* partials, void* out, hipStream_t st,
                const DmaGates& gates = DmaGates(), bool allow_u = false, bool pwq = false);
// bind the two top variables of four tables of 4*outlen entries (w28: the 4 eq weights as for launch_multifold_lazy)
// into d[] (outlen entries each)
// wbuf (device, >= 40 u32) and partials (>= 1024 elements) are scratch for the large-table form; either may be null (then never used)
void launch_fold2_tables(const void* const s[4], void* const d[4], size_t outlen, const uint32_t* w28, void* wbuf, void* partials, hipStream_t st,
                         const DmaGates& gates = DmaGates());
// degree-2 branch, THREE rounds per pass: out[term * 64 + 8 e + e'] (128 elements) = sum_j F_e[j] G_e'[j] of term (F, G) = (t[2 term],
// t[2 term + 1]), tables of 8 q entries, entry e of quarter-index j at e q + j; needs cross64_fits(q); ring = 3 or 4 LDS tiles
bool cross64_fits(size_t q);
// raw (device, cross64_raw_bytes(q) bytes, or null): the sweep leaves each workgroup's 64 cross sums as signed integer digits there and the
// second stage adds them over the workgroups before the one product per digit; null = every workgroup forms its field elements itself
size_t cross64_raw_bytes(size_t q);
void launch_cross64(const void* const t[4], size_t q, int ring, void* partials, void* raw, void* out, hipStream_t st);
// bind the three top variables of four tables of 8 * outlen entries at r3[0..3) (4 limbs each) into d[] (outlen entries each, a
// multiple of 64); wbuf (>= 80 u32) and partials (>= 1024 elements) are scratch
void launch_fold3_tables(const void* const s[4], void* const d[4], size_t outlen, const uint64_t* r3, const uint64_t two24[4], void* wbuf,
                         void* partials, hipStream_t st, const DmaGates& gates = DmaGates());
// the three-variable fold of all four tables in ONE launch with the eight eq weights as kernel arguments (w28: 8 x 10 limbs): any outlen >= 1
void launch_fold3_tables_small(const void* const s[4], void* const d[4], size_t outlen, const uint32_t* w28, void* gather, hipStream_t st);
// launch_fold2_tables followed by launch_sum9 on its result, as one kernel + the second stage (small, cache-resident tables; outlen >= 4)
// Returns 0 when out holds the nine sums.  With at most host_groups_max workgroups in the launch (out must then hold 9 elements per
// workgroup, normally pinned host memory) there is no second stage: returns G = the number of workgroups, out[g * 9 + s] = workgroup
// g's share of sum s, and the caller adds the G shares.
int launch_fold2_sum9(const void* const s[4], void* const d[4], size_t outlen, const uint32_t* w28, void* partials, void* out, hipStream_t st,
                      int host_groups_max = 0);
// the m-variable fold (m = 2 or 3, w28 = the 2^m eq weights) of all four tables and the 27 sums T[9a + 3b + c] of the RESULT that serve
// the next THREE rounds (k_fold_sum27: small tables, outlen a multiple of 16).  Returns like launch_fold2_sum9 with 27 elements per
// workgroup (G > 0: out[g * 27 + i] = workgroup g's share), 0 when out holds the 27 sums, < 0 for a shape it does not take.
int launch_fold_sum27(int m, const void* const s[4], void* const d[4], size_t outlen, const uint32_t* w28, void* partials, void* out,
                      hipStream_t st, int host_groups_max, bool pwq);
// launch_fold3_tables followed by launch_cross64 on d[], in one sweep: the folded tables are not re-read (needs cross64_fits(outlen / 8));
// ring = LDS-DMA slots per wave (3, 4 or 5); raw as for launch_cross64 with q = outlen / 8 (used by the matrix-core form of the fold)
void launch_fold3_cross64(const void* const s[4], void* const d[4], size_t outlen, const uint64_t* r3, const uint64_t two24[4], void* wbuf,
                          int ring, void* partials, void* raw, void* out, hipStream_t st, const DmaGates& gates = DmaGates());
// out[k*n + i] = t[k][i], k < 4, i < n
void launch_gather4(const void* const t[4], size_t n, void* out, hipStream_t st);
// out[(t * L + j) * world + q] = in[(q * ntab + t) * L + j]: the ranks' gathered local tables (rank-major) -> the remaining
// tables, entry j * world + q = rank q's entry j (the table is sharded on its LOW index bits)
void launch_interleave(const void* in, void* out, int world, int ntab, size_t L, hipStream_t st);
// *count_u64 (device, zeroed by the caller) += number of entries of T that are not fully reduced (>= p)
void launch_count_noncanonical(const void* T, size_t n, void* count_u64, hipStream_t st);
void launch_from_le32(const void* in, void* out, size_t n, hipStream_t st);
void launch_to_le32(const void* in, void* out, size_t n, hipStream_t st);
void launch_generate(void* out, uint64_t seed, uint64_t first, uint64_t step, size_t n, hipStream_t st);
void launch_modmul_probe(void* io, int blocks, int iters, hipStream_t st);
// out: blocks * 256 u32; every lane issues iters * 256 v_mad_u64_u32
void launch_mad_probe(void* out, int blocks, int iters, hipStream_t st);

// ---- static scratch of the entry point that allocates nothing (zk_fold_block_sums_noalloc) ----
// One file-scope __device__ array per GPU (kernels.hip): [0, kStaticPartialsBytes) per-workgroup partial sums of a fold
// pass, then the eq weights of a wide pass, then one element for the block sum of a pass that asks for none.
static constexpr size_t kStaticPartialsBytes = (size_t)2 * kMaxBlocks * 32;
static constexpr size_t kStaticWeightsBytes = (size_t)kWideWeightWords * sizeof(uint32_t);
static constexpr size_t kStaticScratchBytes = kStaticPartialsBytes + kStaticWeightsBytes + 256;
// Device address of the CURRENT device's copy of the array (hipGetSymbolAddress, once per device, cached).
hipError_t static_scratch_ptr(void** out);

} // namespace zkk
