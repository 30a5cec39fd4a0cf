/*
 * zkmi.h -- C ABI of libzkmi.so: the MI355X (gfx950) multilinear sumcheck prover.
 *
 * Drop-in boundary for ONE path of the webdott/zk Rust workspace: the per-round
 * round-polynomial reduction + fold of a multilinear-extension table over BN254 Fq
 * (SURVEY.md section 8).  The reference has no FFI of its own: each entry
 * point below names the Rust item (file:line under /root/reference) whose work it
 * replaces; INTEGRATION.md shows the `extern "C"` shim crate a maintainer would add.
 *
 * Conventions
 *  - Field elements cross the boundary as 4 x u64 little-endian limbs of a*R mod p
 *    (R = 2^256, fully reduced) -- ark-ff's in-memory form of `Fp256<MontBackend>` --
 *    unless a name says le32 (canonical value, 32 little-endian bytes =
 *    `into_bigint().to_bytes_le()`).  Limbs that are not fully reduced (value >= p) are outside the
 *    contract, as they are for ark-ff: the kernels' unreduced accumulations size their carry cadence for
 *    reduced operands (zk_amd/csrc/fq_dev.hpp), and results for other inputs are unspecified.
 *  - Every function returns a zk_status; where the reference panics, a non-zero code is
 *    returned instead (the Rust shim turns it back into panic!).  Verifiers' `bool`
 *    results come back through an int* out-parameter.
 *  - All output buffers are caller-allocated.  Handles are opaque, not thread-safe; one
 *    prove in flight per context.  Tables handed to a prover are left unmodified.
 *  - There is NO CPU fallback: every compute entry point needs a gfx950 device.
 */
#ifndef ZKMI_H
#define ZKMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built -fvisibility=hidden: only this ABI is exported */
#endif

typedef enum zk_status {
    ZK_OK = 0,
    ZK_E_NOT_POW2 = 1,     /* evaluation_form.rs:23-27  "length is not a power of two" */
    ZK_E_POINTS_LEN = 2,   /* evaluation_form.rs:197-199, product_polynomial.rs:32-34 */
    ZK_E_LEN_MISMATCH = 3, /* product_polynomial.rs:15-19, sum_polynomial.rs:15-19, evaluation_form.rs:244-246 */
    ZK_E_LAYER_OOB = 4,    /* circuit.rs:83-85, 131-133 */
    ZK_E_DEGREE = 5,       /* sum polynomial not 2 terms x 2 factors (sumcheck/src/prover.rs:90-107) */
    ZK_E_ALLOC = 6,
    ZK_E_HIP = 7,          /* a HIP runtime call failed; zk_last_error() has the text */
    ZK_E_ARG = 8,          /* NULL pointer / bad argument */
    ZK_E_NO_DEVICE = 9,    /* no gfx950 device: the library refuses to run (no CPU path) */
    ZK_E_COMM = 10,        /* collective (RCCL) failure */
    ZK_E_TOO_LARGE = 11    /* input beyond the maximum of the entry point that allocates nothing */
} zk_status;

typedef struct zk_ctx zk_ctx;               /* one GPU: device, stream, scratch */
typedef struct zk_mle zk_mle;               /* MultiLinearPolynomial<T> as a device-resident table */
typedef struct zk_transcript zk_transcript; /* fiat_shamir::Transcript<T> (host) */

const char *zk_strerror(int status);
const char *zk_last_error(const zk_ctx *ctx); /* text of the last ZK_E_HIP / ZK_E_COMM on this context */
const char *zk_version(void);

/* ---- host field helpers: the few ark-ff operations a caller needs around the prover ------- */
void zk_fq_from_u64(uint64_t v, uint64_t out_mont[4]);                      /* T::from(u64) */
void zk_fq_to_le32(const uint64_t a_mont[4], uint8_t out[32]);              /* into_bigint().to_bytes_le() */
void zk_fq_from_le32_mod_order(const uint8_t in[32], uint64_t out_mont[4]); /* from_le_bytes_mod_order, transcript.rs:35 */
void zk_fq_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void zk_fq_sub(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void zk_fq_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
/* UnivariatePolynomial::evaluate (dense_coefficient_form.rs:20-34), coefficients ascending */
void zk_unipoly_evaluate(const uint64_t *coeffs, size_t n, const uint64_t x[4], uint64_t out[4]);

/* ---- context ------------------------------------------------------------------------- */
int zk_ctx_create(int device_id, zk_ctx **out);
/* the same on a stream the caller owns (e.g. the current stream of the host framework that also
 * runs the collectives), so kernels and collectives order on one stream without extra syncs */
int zk_ctx_create_on_stream(int device_id, void *hip_stream, zk_ctx **out);
void zk_ctx_destroy(zk_ctx *ctx); /* tables of this context may be freed afterwards, not used */
/* the HIP stream (hipStream_t) every kernel of this context is launched on */
void *zk_ctx_stream(zk_ctx *ctx);
int zk_ctx_synchronize(zk_ctx *ctx);

/* ---- fiat_shamir::Transcript (fiat_shamir/src/transcript.rs:11-42), host-side -------- */
zk_transcript *zk_transcript_new(void);                                    /* Transcript::new   :12-17 */
zk_transcript *zk_transcript_clone(const zk_transcript *t);                /* the sponge state is plain data */
void zk_transcript_free(zk_transcript *t);
int zk_transcript_append(zk_transcript *t, const uint8_t *data, size_t len); /* append         :20-22 */
int zk_transcript_sample_challenge(zk_transcript *t, uint64_t out_mont[4]);  /* sample_challenge :28-36 */
int zk_transcript_sample_n_challenges(zk_transcript *t, size_t n, uint64_t *out_mont); /* :39-41 */
int zk_transcript_peek_digest(const zk_transcript *t, uint8_t out[32]);    /* Keccak256(all bytes so far) */
/* The host Keccak-f[1600] exists in three builds of the round function -- baseline x86-64; BMI1/BMI2 (ANDN, RORX); a hybrid that keeps planes
 * 0, 1 in general registers and planes 2, 3, 4 one lane per xmm register (AVX-512VL) -- one chosen per process by the CPU's features.
 * Runs every available build on nblocks pseudo-random rate blocks: 0 = all states equal the baseline build's, 1 = one differs. */
int zk_transcript_selftest(size_t nblocks);

/* ---- MultiLinearPolynomial (polynomials/src/multilinear_polynomial/evaluation_form.rs) */
/* MultiLinearPolynomial::new :22-32 -- copies `len` elements host -> device.  The limbs must be fully reduced (< p, as
 * ark-ff keeps them): an upload with any entry >= p is refused with ZK_E_ARG (one extra sweep of the table; switch the
 * check off with zk_ctx_set_validate(ctx, 0)).  zk_mle_wrap_device does not check -- call zk_mle_count_noncanonical on
 * memory of unknown origin. */
int zk_mle_upload_mont(zk_ctx *ctx, const uint64_t *limbs, size_t len, zk_mle **out);
int zk_ctx_set_validate(zk_ctx *ctx, int on);                                /* default on */
int zk_mle_count_noncanonical(const zk_mle *m, uint64_t *count_out);         /* entries >= p; 0 for a valid table */
int zk_mle_upload_le32(zk_ctx *ctx, const uint8_t *bytes, size_t len, zk_mle **out);
/* borrow `len` elements of device memory the caller owns (Montgomery limbs); never freed by the library */
int zk_mle_wrap_device(zk_ctx *ctx, void *device_ptr, size_t len, zk_mle **out);
/* synthetic table generated on device: entry j = splitmix64 stream `seed`,
 * draws 4(first+j)+1 .. +4, little-endian 256 bits, mod p, to Montgomery.  No reference counterpart. */
int zk_mle_generate(zk_ctx *ctx, uint64_t seed, uint64_t first, size_t len, zk_mle **out);
/* as zk_mle_generate, local entry j = global entry first + j*step (step = number of shards) */
int zk_mle_generate_strided(zk_ctx *ctx, uint64_t seed, uint64_t first, uint64_t step, size_t len, zk_mle **out);
/* shard `rank` of `world` (power of two) of a host table of global_len entries: local entry j =
 * global entry j*world + rank -- the table is sharded on the variables bound LAST, so
 * every fold pair (j, j + N/2) of evaluation_form.rs:61-79 stays on one GPU */
int zk_mle_upload_mont_shard(zk_ctx *ctx, const uint64_t *limbs, size_t global_len, int rank, int world, zk_mle **out);
int zk_mle_clone(const zk_mle *src, zk_mle **out);                          /* #[derive(Clone)] :16 */
void zk_mle_free(zk_mle *m);
size_t zk_mle_len(const zk_mle *m);
void *zk_mle_device_ptr(const zk_mle *m);
int zk_mle_download_mont(const zk_mle *m, uint64_t *out_limbs);             /* get_evaluation_points :155-157 */
int zk_mle_to_le32(const zk_mle *m, uint8_t *out_bytes);                    /* to_bytes :225-234 (32*len bytes) */
/* partially_evaluate((var, r)) :163-192 with the pairing of get_y1_y2_indexes :45-80 -> new table of len/2 */
int zk_mle_partially_evaluate(const zk_mle *m, size_t var, const uint64_t r_mont[4], zk_mle **out);
/* compute_quotient_remainder(&divisor, variable_index) :280-301 (the division step of the multilinear-KZG opening,
 * kzg/src/multilinear/prover.rs:77): quotient[k] = y2 - y1 over the pairs of get_y1_y2_indexes(var) in their order (a
 * Vec<T> of len/2 elements, returned as a table), remainder = partially_evaluate((var, divisor)).  One pass, two outputs. */
int zk_mle_quotient_remainder(const zk_mle *m, size_t var, const uint64_t divisor_mont[4], zk_mle **quotient_out,
                              zk_mle **remainder_out);
int zk_mle_minus(const zk_mle *m, const uint64_t v_mont[4], zk_mle **out);  /* minus :260-272: F(x) - V */
/* evaluate(&[Option<T>]) :194-223; has[i] != 0 means Some(vals[i]).  npts must equal log2(len). */
int zk_mle_evaluate(const zk_mle *m, const uint8_t *has, const uint64_t *vals_mont, size_t npts, zk_mle **out);
int zk_mle_evaluation_sum(const zk_mle *m, uint64_t out_mont[4]);           /* evaluation_sum :236-238 */
/* the MLE-branch round reduction of generate_round_polys (sumcheck/src/prover.rs:108-116):
 * out[0..4) = sum of the first half, out[4..8) = sum of the second half.  len >= 2. */
int zk_mle_round_sums(const zk_mle *m, uint64_t out_mont[8]);

/* ---- sumcheck prover (sumcheck/src/prover.rs) ----------------------------------------- */
/* SumcheckProver::generate_sumcheck_proof :158-173 when `transcript` is NULL or
 * absorb_table != 0 (a fresh transcript absorbs to_bytes() of the table first); with a
 * caller transcript and absorb_table == 0 it is generate_round_polys :45-155 on the MLE branch.
 *   claim_out      SumCheckProof.initial_claim_sum
 *   coeffs_out     round_polys, n x 2 coefficients ascending (n = log2 len)
 *   challenges_out n challenges (the reference discards them; may be NULL) */
int zk_sumcheck_prove_mle(zk_ctx *ctx, const zk_mle *poly, int absorb_table, zk_transcript *transcript,
                          uint64_t claim_out[4], uint64_t *coeffs_out, uint64_t *challenges_out);
/* SumcheckProver::generate_proof_for_partial_verify :175-190 for
 * SumPolynomial[ ProductPolynomial[t0,t1], ProductPolynomial[t2,t3] ] -- the only shape the
 * reference's degree-2 branch accepts and the GKR layer relation.
 *   coeffs_out n x 3 coefficients, challenges_out n random points.  The claim is copied into
 *   the proof unchecked, as in the reference (:185). */
int zk_sumcheck_prove_sum2x2(zk_ctx *ctx, zk_mle *const tables[4], const uint64_t claim_mont[4],
                             zk_transcript *transcript, uint64_t *coeffs_out, uint64_t *challenges_out);
/* The same entry point for SumPolynomial[ ProductPolynomial[f, g], ProductPolynomial[h, ones] ] with `ones` the table of ones of
 * the same length: the relation f g + h, i.e. both phases of a GKR layer (gkr/src/prover.rs:55-147 builds exactly such four tables).
 * tables = {f, g, h}.  Byte-identical proofs and transcript to zk_sumcheck_prove_sum2x2 on {f, g, h, ones}; the small-table passes
 * run in their product-plus-linear form, which never reads the constant factor (see zk_ctx_set_deg2_first_triple).  The table of
 * ones the general kernels need is allocated and freed inside the call. */
int zk_sumcheck_prove_prod_plus_linear(zk_ctx *ctx, zk_mle *const tables[3], const uint64_t claim_mont[4],
                                       zk_transcript *transcript, uint64_t *coeffs_out, uint64_t *challenges_out);

/* ---- the same round loop in steps, for a table sharded over several GPUs ----------------
 * generate_round_polys (sumcheck/src/prover.rs:45-155, MLE branch) cut at the points where ranks
 * must exchange: device passes (zk_shard_*, asynchronous on the context stream) and the host part
 * (zk_sumcheck_rounds_from_block_sums, no GPU needed).  zk_sumcheck_prove_mle is these same calls
 * with world = 1.  The host framework all-gathers the 2^m block sums of every rank between the two
 * (RCCL over xGMI; zk_amd/sharded.py uses torch.distributed), every rank runs the identical host part.
 *
 * sums: [world][2^m] elements, rank-major; the reduction runs m rounds on sum_q sums[q]:
 *   coeffs_out m x 2, challenges_out m (may be NULL), total_out = sum of everything (may be NULL; the
 *   first batch's total is SumCheckProof.initial_claim_sum), folded_out[world] (may be NULL) = each
 *   rank's own 2^m sums folded at the m challenges -- once a local table is down to 2^m entries these
 *   are the entries of the remaining log2(world)-variable table, which is finished by one more call
 *   with world = 1, m = log2(world). */
typedef struct zk_shard zk_shard;
int zk_sumcheck_rounds_from_block_sums(zk_transcript *t, const uint64_t *sums, int world, int m, uint64_t *coeffs_out,
                                       uint64_t *challenges_out, uint64_t *folded_out, uint64_t total_out[4]);
int zk_shard_begin(zk_ctx *ctx, const zk_mle *local_table, zk_shard **out); /* the table is not modified */
void zk_shard_end(zk_shard *s);
int zk_shard_vars_left(const zk_shard *s);
/* how many rounds (m, 1..8) a pass should bind once `after_binding` more variables of the current local table
 * are bound (0: the next pass; the m of the pass being issued: the m_next to give zk_shard_fold): the context's
 * wide batch while the table is large, its default batch after that (zk_ctx_set_fold_batch /
 * zk_ctx_set_wide_fold).  Any m <= vars_left is valid for zk_shard_block_sums / zk_shard_fold; this is the
 * schedule zk_sumcheck_prove_mle uses.  0 when no variable is left, -1 on a bad argument. */
int zk_shard_next_batch(const zk_shard *s, int after_binding);
/* dst_dev (device, 2^vars_left elements) = the current local table (asynchronous).  For the switch from sharded
 * passes to replicated ones: once the local tables are small, the ranks all-gather them (entry j of rank q is
 * entry j*world + q of the remaining table) and every rank finishes with zk_sumcheck_prove_mle(absorb_table = 0)
 * on the gathered table and the running transcript -- no further exchange. */
int zk_shard_copy_table(zk_shard *s, void *dst_dev);
/* sums_dev_out (device, 2^m elements) = sums of the 2^m contiguous blocks of the current local table */
int zk_shard_block_sums(zk_shard *s, int m, void *sums_dev_out);
/* bind the top m variables at r[0..m) (m x partially_evaluate((0, r)), evaluation_form.rs:163-192) and
 * leave the 2^m_next block sums of the result in sums_dev_out (m_next = 0: none) */
int zk_shard_fold(zk_shard *s, int m, const uint64_t *r_mont, int m_next, void *sums_dev_out);

/* zk_shard_fold for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same
 * pass, same kernels, same results as zk_shard_fold with the default schedule knobs: dst_dev (len >> m elements) =
 * src_dev (len elements, a power of two) with its top m variables bound at r_mont[0..m) (1 <= m <= 8), and
 * sums_dev_out (2^m_next elements, device; may be NULL when m_next = 0) = the block sums of dst_dev.  No context: the
 * pass runs on the calling thread's current device, asynchronously on hip_stream.  It calls no hipMalloc / hipFree /
 * allocating routine: its scratch (per-workgroup partial sums, the weight tiles of a wide pass; 640 KiB) is a file-scope
 * __device__ array that every GPU the library runs on holds until the process ends.  len is limited to
 * 2^ZK_NOALLOC_MAX_LOG_LEN entries (ZK_E_TOO_LARGE beyond).  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (two streams, two threads) race -- serialise them.  Pointers must be 16-byte aligned; the src, dst and
 * sums ranges must not intersect (ZK_E_ARG: the kernels stream src while they store dst).  The FIRST call on a device takes
 * the array's address with hipGetSymbolAddress, which is not capture-safe: before capturing a graph, make one call on that
 * device outside the capture (or create any zk_ctx there: zk_ctx_create takes the address too). */
#define ZK_NOALLOC_MAX_LOG_LEN 30
int zk_fold_block_sums_noalloc(void *hip_stream, const void *src_dev, size_t len, int m, const uint64_t *r_mont,
                               void *dst_dev, int m_next, void *sums_dev_out);

/* The SumPolynomial branch (prover.rs:89-107, :131-139) in steps, for four factor tables sharded the same
 * way.  Per round every rank contributes its share of the three evaluations (3 x 32 B, all-gathered by
 * the host framework); zk_sumcheck_round_from_evals sums the shares and runs the host part of the round.
 * After the last local round each rank holds one entry per table: these rows form four log2(world)-
 * variable tables that zk_sumcheck_prove_sum2x2_host finishes on the host. */
typedef struct zk_shard4 zk_shard4;
int zk_shard4_begin(zk_ctx *ctx, zk_mle *const tables[4], zk_shard4 **out); /* tables are not modified */
void zk_shard4_end(zk_shard4 *s);
size_t zk_shard4_len(const zk_shard4 *s);                  /* entries left per local table */
int zk_shard4_eval(zk_shard4 *s, void *ev_dev_out);        /* 3 elements: shares of ev(0), ev(1), ev(2) */
/* fold all four tables at r; out_dev gets the next round's 3 shares, or -- when one entry per table is
 * left -- those 4 entries (table order) */
int zk_shard4_fold(zk_shard4 *s, const uint64_t r_mont[4], void *out_dev);
/* evals: [world][3] elements; coeffs_out: the round polynomial (3 coefficients); host only */
int zk_sumcheck_round_from_evals(zk_transcript *t, const uint64_t *evals, int world, uint64_t coeffs_out[12],
                                 uint64_t challenge_out[4]);
/* Three rounds per step (degree-2 branch; see zk_ctx_set_deg2_triple).  zk_shard4_can_triple: 1 if the context's thresholds give the
 * current tables a three-round pass.  zk_shard4_cross64: cross_dev_out (device, 128 elements) = this shard's share of
 * P[term][e][e'] = sum_j F_e[j] G_e'[j] (e, e' < 8: the three top variables; tables of >= 512 entries, whole 64-entry eighths).
 * zk_sumcheck_rounds_from_cross64: sums the ranks' shares ([world][128]) and runs THREE rounds of prover.rs:89-135 on them
 * (coeffs_out 3 x 3, challenges_out 3 elements).  zk_shard4_fold3: binds the three variables (prover.rs:131-139 three times) and, when
 * next_dev_out is not NULL, leaves there what the next step needs: *next_rounds_out = 3 (128 cross sums), 2 (nine sums) or 0. */
int zk_shard4_can_triple(const zk_shard4 *s);
int zk_shard4_cross64(zk_shard4 *s, void *cross_dev_out);
int zk_sumcheck_rounds_from_cross64(zk_transcript *t, const uint64_t *cross, int world, uint64_t coeffs_out[36], uint64_t challenges_out[12]);
int zk_shard4_fold3(zk_shard4 *s, const uint64_t r3[12], void *next_dev_out, int *next_rounds_out);
/* Two rounds per exchange (the schedule zk_sumcheck_prove_sum2x2 uses on one GPU): each rank's share of the nine
 * sums S[3Y + t] (t: f0*g0, f1*g1, df*dg with the top variable at 0 / 1 / difference; Y in {0,1,2} the value of
 * the second variable; summed over the rest and both product terms -- linear in the entries, so shares add)
 * determines the round polynomials of BOTH next rounds (prover.rs:89-107 twice); one 4-to-1 fold binds both.
 *   zk_shard4_sum9: 9 elements (needs >= 4 entries per local table)
 *   zk_sumcheck_round_pair_from_sum9: sums [world][9]; coeffs_out 2 x 3 coefficients, challenges_out 2; host only
 *   zk_shard4_fold2: bind both at r2[0], r2[1]; sums_dev_out (may be NULL) gets the next nine sums if >= 4 entries
 *     per table remain
 *   zk_shard4_copy_tables: dst_dev (4 x len elements, table after table) = the current local tables, for the
 *     switch to replicated passes (see zk_shard_copy_table); every rank then finishes with zk_sumcheck_prove_sum2x2. */
int zk_shard4_sum9(zk_shard4 *s, void *sums_dev_out);
int zk_sumcheck_round_pair_from_sum9(zk_transcript *t, const uint64_t *sums, int world, uint64_t coeffs_out[24],
                                     uint64_t challenges_out[8]);
int zk_shard4_fold2(zk_shard4 *s, const uint64_t r2_mont[8], void *sums_dev_out);
int zk_shard4_copy_tables(zk_shard4 *s, void *dst_dev);
/* generate_round_polys, SumPolynomial branch, on four HOST tables of len entries (4 x len elements,
 * table-major); coeffs_out log2(len) x 3, challenges_out log2(len) (may be NULL); host only */
int zk_sumcheck_prove_sum2x2_host(zk_transcript *t, const uint64_t *tables, size_t len, uint64_t *coeffs_out,
                                  uint64_t *challenges_out);

/* ---- sharded provers: the loops above inside the library, the exchange included --------------------------------
 * One process per GPU.  A context becomes rank `rank` of `world` (a power of two) by one of
 *   zk_comm_init_rccl: RCCL over xGMI.  One rank calls zk_comm_unique_id and hands the 128 bytes to the others by any
 *     means it has (the launcher's rendezvous, a file, MPI); every rank then calls zk_comm_init_rccl (collective:
 *     ncclCommInitRank).  The all-gathers run on the context's stream (ncclAllGather).  librccl.so.1 is loaded on
 *     first use -- the copy already in the process if the host framework loaded one -- so single-GPU users never map it.
 *   zk_comm_init_host: the caller's own all-gather over HOST buffers (MPI_Allgather shaped: every rank's `bytes` at
 *     send -> recv[world][bytes], rank order; return 0 on success).  The library stages through pinned memory.
 * Failures of either come back as ZK_E_COMM with the text in zk_last_error.  The reference has no counterpart
 * (single thread); what stays identical is generate_round_polys (sumcheck/src/prover.rs:45-155):
 * every rank returns the same claim, coefficients and challenges as the unsharded prover on the whole table.
 *   local / tables: this rank's shard(s) (zk_mle_upload_mont_shard / zk_mle_generate_strided: local entry j =
 *     global entry j * world + rank)
 *   gather_log: local tables of at most 2^gather_log entries are all-gathered and every rank finishes alone
 *     (replicated, no further exchange; recommended: 13); negative: sharded to the end
 *   zk_sumcheck_prove_mle_sharded: absorb_table / transcript as zk_sumcheck_prove_mle (the absorbed byte stream is
 *     the WHOLE table's to_bytes(), gathered chunk by chunk); coeffs_out (log2(len) + log2(world)) x 2
 *   zk_sumcheck_prove_sum2x2_sharded: as zk_sumcheck_prove_sum2x2; coeffs_out (log2(len) + log2(world)) x 3 */
#define ZK_COMM_ID_BYTES 128
typedef int (*zk_allgather_host_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int zk_comm_unique_id(uint8_t out[ZK_COMM_ID_BYTES]);
int zk_comm_init_rccl(zk_ctx *ctx, const uint8_t id[ZK_COMM_ID_BYTES], int rank, int world);
int zk_comm_init_host(zk_ctx *ctx, int rank, int world, zk_allgather_host_fn fn, void *user);
void zk_comm_destroy(zk_ctx *ctx); /* also done by zk_ctx_destroy */
int zk_comm_rank(const zk_ctx *ctx);  /* -1 without a communicator */
int zk_comm_world(const zk_ctx *ctx); /* 0 without a communicator */
int zk_sumcheck_prove_mle_sharded(zk_ctx *ctx, const zk_mle *local, int absorb_table, zk_transcript *transcript,
                                  int gather_log, uint64_t claim_out[4], uint64_t *coeffs_out, uint64_t *challenges_out);
int zk_sumcheck_prove_sum2x2_sharded(zk_ctx *ctx, zk_mle *const tables[4], const uint64_t claim_mont[4],
                                     zk_transcript *transcript, int gather_log, uint64_t *coeffs_out,
                                     uint64_t *challenges_out);

/* ---- sumcheck verifier (sumcheck/src/verifier.rs) -- the parity harness ---------------- */
/* partial_verify :29-62 (host only). */
int zk_sumcheck_partial_verify(const uint64_t claim_mont[4], const uint64_t *coeffs, size_t n_rounds,
                               size_t coeffs_per_round, zk_transcript *transcript, int *ok_out,
                               uint64_t final_claim_out[4], uint64_t *challenges_out);
/* verify_proof :64-94: table absorb + partial_verify + the oracle check (device evaluate). */
int zk_sumcheck_verify_mle(zk_ctx *ctx, const zk_mle *poly, const uint64_t claim_mont[4], const uint64_t *coeffs,
                           size_t n_rounds, int *ok_out);

/* ---- MultiLinearPolynomial helpers the GKR prover uses (evaluation_form.rs) -------------- */
int zk_mle_w_add(const zk_mle *w_b, const zk_mle *w_c, zk_mle **out);       /* w_add :303-308 via operate_w :82-112: out[i] = w_b[i / L] + w_c[i % L] */
int zk_mle_w_mul(const zk_mle *self, const zk_mle *other, zk_mle **out);    /* w_mul(&self, other) :310-312 = operate_w(other, self): out[i] = other[i / L] * self[i % L] */
int zk_mle_scalar_mul(const zk_mle *m, const uint64_t s_mont[4], zk_mle **out); /* scalar_mul :151-153 */
int zk_mle_add(const zk_mle *a, const zk_mle *b, zk_mle **out);             /* _add / Add :241-258 */

/* ---- arithmetic_circuit::Circuit (arithmetic_circuit/src/circuit.rs) ---------------------- */
typedef struct zk_circuit zk_circuit;
/* Circuit::new :15-20.  gates: for every layer (layers[0] is next to the inputs) its gates in order,
 * each as three u64 (left, right, op) with op 0 = Add, 1 = Mul (gate.rs:7-22); layer_len[i] = gates in layer i.
 * The wiring is kept as gate lists + per-layer CSR indices on the device (never as dense tables). */
int zk_circuit_create(zk_ctx *ctx, const uint64_t *gates, const size_t *layer_len, size_t n_layers, zk_circuit **out);
void zk_circuit_free(zk_circuit *c);
size_t zk_circuit_layer_count(const zk_circuit *c);                         /* get_layer_count :149-151 */
/* proof shape: output_len = len(W_0); rounds_per_layer[i] (i counted from the OUTPUT layer, n_layers
 * entries) = 2*k_in of that layer = rounds of its sumcheck */
int zk_circuit_shape(const zk_circuit *c, size_t *output_len, size_t *rounds_per_layer);
/* evaluate_at_input :37-59 -> n_layers+1 new tables, layers_out[0] = the inputs, widths padded as :43 */
int zk_circuit_evaluate(zk_circuit *c, const zk_mle *inputs, zk_mle **layers_out);
/* get_add_i (op 0) / get_mul_i (op 1) :82-146 as the reference's DENSE 2^(k_out+2k_in) table; layer_idx
 * counts from the output.  For small circuits and parity tests: the prover below never builds it. */
int zk_circuit_gate_poly(zk_circuit *c, size_t layer_idx, int op, zk_mle **out);

/* ---- GKRProver::generate_proof (gkr/src/prover.rs:24-154) --------------------------------- */
/* GKRProof (gkr_protocol.rs:8-26) comes back as
 *   output_poly_out  output_len elements            (GKRProof.output_poly = W_0)
 *   w_evals_out      (n_layers-1) x 2 elements      (w_polys_evals: (W_i(r_b), W_i(r_c)), layers 1..)
 *   claims_out       n_layers elements              (sumcheck_proofs[i].initial_claim_sum)
 *   coeffs_out       sum_i rounds_per_layer[i] x 3  (sumcheck_proofs[i].round_polys, layer after layer)
 *   layer_ms_out     optional, n_layers + 2: device time per layer (HIP events), then the host wall time of
 *                    the circuit evaluation and of the serial Keccak absorb of W_0 (prover.rs:46)
 * Per layer the add_i/mul_i predicates folded at the previous challenges (utils.rs:16-65) are built
 * on the device from the gate list, never as dense tables; round polynomials are identical field
 * elements.  Needs 2^k_in == len(W_{i+1}) on every layer (the reference panics otherwise). */
int zk_gkr_prove(zk_ctx *ctx, zk_circuit *circuit, zk_transcript *transcript, const zk_mle *inputs,
                 uint64_t *output_poly_out, uint64_t *w_evals_out, uint64_t *claims_out, uint64_t *coeffs_out,
                 double *layer_ms_out);
/* The same proof from the ranks of the context's communicator (zk_comm_init_*; SURVEY.md 8e: "phase tables sharded on low bits; gate
 * list partitioned").  Every rank passes the same circuit and inputs (replicated: the layer tables are small) and gets the same
 * GKRProof; what is sharded is each layer's degree-2 sumcheck (gkr/src/prover.rs:128-141): rank q builds the folded-predicate tables
 * for the indices that are q mod world and the ranks run zk_sumcheck_prove_sum2x2_sharded's loop on them.  gather_log >= 0 as there
 * (the last rounds of each phase run replicated on the gathered tables, which is where W(u), W(v) of prover.rs:87-99 come from).
 * Layers with fewer than two entries per rank run replicated.  ZK_E_COMM without a communicator.  A GKR layer at 2^20 gates is bound
 * by launch latency, not by bytes: this entry point exists for layers too large for one GPU's latency budget, it is not faster at
 * BASELINE config 5. */
int zk_gkr_prove_sharded(zk_ctx *ctx, zk_circuit *circuit, zk_transcript *transcript, const zk_mle *inputs, int gather_log,
                         uint64_t *output_poly_out, uint64_t *w_evals_out, uint64_t *claims_out, uint64_t *coeffs_out,
                         double *layer_ms_out);

/* GKRVerifier::verify_proof (gkr/src/verifier.rs:21-119); proof arrays as zk_gkr_prove returns them.  The
 * predicates are evaluated from the gate list (the reference rebuilds the dense tables, :39-40).  *ok_out =
 * 1 accepted, 0 rejected. */
int zk_gkr_verify(zk_ctx *ctx, zk_circuit *circuit, zk_transcript *transcript, const zk_mle *inputs,
                  const uint64_t *output_poly, size_t output_len, const uint64_t *w_evals, const uint64_t *claims,
                  const uint64_t *coeffs, int *ok_out);
/* INHERITED LIMITATION.  Like the reference (verifier.rs:63-64: partial_verify starts from
 * sumcheck_proofs[i].initial_claim_sum as sent), zk_gkr_verify takes every layer's claim from the proof: it never checks
 * that layer 0's claim is output_poly(rho), nor that layer i's is alpha W_i(r_b) + beta W_i(r_c).  A prover can therefore
 * absorb a false output_poly and still be accepted -- the default stays bit-compatible with the reference, callers that rely
 * on soundness switch the binding on: with zk_ctx_set_gkr_strict_claims(ctx, 1) the verifier recomputes both and rejects
 * a mismatch (honest proofs are accepted in both modes, with the same transcript). */
int zk_ctx_set_gkr_strict_claims(zk_ctx *ctx, int on);                       /* default off */

/* ---- tuning / measurement hooks (no reference counterpart) ---------------------------- */
/* Rounds bound per device pass by the single-MLE prover on small tables (1..8, default 4; passes of 5..8 rounds always use
 * the wide kernels) and the form of the passes of up to 4 rounds: 0 = ONE inner product with the eq weights and one
 * Montgomery reduction per output (default); 1 = the fold tree, i.e. the reference's successive partially_evaluate calls
 * statement by statement -- slower, kept as the independent cross-check.  Proof bytes do not depend on either. */
int zk_ctx_set_fold_batch(zk_ctx *ctx, int rounds_per_pass, int variant);
/* Passes over a LARGE table bind up to `rounds` variables at once (5..8, default 8; 0 = off): as many as leave
 * the folded table at least 2^min_log_out entries (default 16), when that is `rounds` itself or at least two more
 * than the default batch.  At 2^8 read streams per store stream the pass runs at the read-only HBM rate
 * (fold-pass ceiling).  lds_weights_below_log (default 15; 0 = never, 64 = always): wide passes with
 * fewer than 2^k outputs derive their weights per workgroup in LDS (one launch, latency matters), larger ones
 * read them from a small device buffer as scalar loads (4 waves per SIMD).  Proof bytes do not depend on any of it. */
int zk_ctx_set_wide_fold(zk_ctx *ctx, int rounds, int min_log_out, int lds_weights_below_log);
/* The tail of a prove: tables of at most 2^max_log_table entries (default 16; 0 = off) bind eight rounds per pass, and
 * passes whose output has fewer than 2^out_below_log entries (default 13) form each output with one wave instead of
 * one lane -- eight rounds at the latency of one.  Proof bytes do not depend on it. */
int zk_ctx_set_tail_fold(zk_ctx *ctx, int max_log_table, int out_below_log);
/* Rounds bound per device pass by the degree-2 prover: 2 (default; nine sums serve two rounds) or 1 (one
 * evaluation triple per pass).  Proof bytes do not depend on it. */
int zk_ctx_set_deg2_rounds_per_pass(zk_ctx *ctx, int rounds);
/* Nine-sum sweeps over tables of at least 2^from_log quarter-entries (a quarter of the entries per table) accumulate
 * their products unreduced with one reduction per lane (default 18; negative = never).  Proof bytes do not depend on it. */
int zk_ctx_set_deg2_lazy_sums(zk_ctx *ctx, int from_log);
/* Size thresholds (log2 of entries; 64 = never) of the LDS-DMA forms of the big passes, whose defaults are the measured
 * break-even points: block sums stream through LDS-DMA rings from tables of 2^sums_from_log entries (default 27); wide
 * fold passes use streaming stores from 2^wide_stream_from_log outputs (default 22); the degree-2 two-variable fold runs as
 * m = 2 LDS-DMA passes from 2^fold2_from_log outputs per table (default 22).  The parity tests lower them so that the
 * kernels the benchmark numbers come from run on oracle-sized tables.  Proof bytes do not depend on any of it. */
int zk_ctx_set_dma_gates(zk_ctx *ctx, int sums_from_log, int wide_stream_from_log, int fold2_from_log);
/* Wide fold passes with at least 2^from_log outputs (default 15; 64 = never) run their contraction on the matrix cores
 * (v_mfma_i32_16x16x64_i8 over the raw table bytes and signed radix-256 digits of the eq weights) instead
 * of the vector ALU; workgroups_per_cu (1..8) sizes its grid, depth (2..4) is the number of steps (4 KiB of table rows per
 * wave each) kept in flight (defaults 1 and 2).  Proof bytes do not depend on it. */
int zk_ctx_set_wide_mfma(zk_ctx *ctx, int from_log, int workgroups_per_cu, int depth);
/* Which kernel forms the launches of the calling thread took since the last reset (bit mask, ZK_FORM_*): lets a test
 * assert that the form it forced is the one that ran. */
enum { ZK_FORM_SUMS_DMA = 1 << 0, ZK_FORM_SUMS_REG = 1 << 1, ZK_FORM_WIDE_DMA = 1 << 2, ZK_FORM_WIDE_DMA_STREAM = 1 << 3,
       ZK_FORM_WIDE_REG = 1 << 4, ZK_FORM_WIDE_LDS = 1 << 5, ZK_FORM_COOP = 1 << 6, ZK_FORM_FOLD2_DMA = 1 << 7,
       ZK_FORM_FOLD2_REG = 1 << 8, ZK_FORM_SUM9_LAZY = 1 << 9, ZK_FORM_SUM9 = 1 << 10, ZK_FORM_FOLD2_SUM9 = 1 << 11, ZK_FORM_WIDE_MFMA = 1 << 12, ZK_FORM_SUM9_MFMA = 1 << 13,
       ZK_FORM_CROSS64 = 1 << 14, ZK_FORM_FOLD3_DMA = 1 << 15, ZK_FORM_FOLD3_CROSS64 = 1 << 16 };
unsigned zk_ctx_forms_used(zk_ctx *ctx, int reset);
/* Nine-sum sweeps over tables of at least 2^from_log quarter-entries (64 = never) run on the matrix cores: the sixteen cross sums
 * of a term's entries as 32 x 32 byte-product matrices (v_mfma_i32_32x32x32_i8 with the contraction over the quarter-index,
 * operands through ds_read_b64_tr_b8), mapped to the nine sums exactly.  Proof bytes do not depend on it. */
int zk_ctx_set_deg2_mfma_sums(zk_ctx *ctx, int from_log);
/* Degree-2 prover (SumPolynomial branch, sumcheck/src/prover.rs:89-107, :131-139): passes over four tables of at least 2^from_log
 * entries (64 = never; tables of whole 64-entry eighths only) serve THREE rounds each: one sweep forms the 8 x 8 cross sums
 * sum_j F_e[j] G_e'[j] of both terms on the matrix cores, the host runs three rounds on them, one fold binds the three variables.
 * ring = 3 or 4 LDS tiles in the sweep.  Default from_log 20; relations of the product + linear shape (zk_sumcheck_prove_prod_plus_linear,
 * the GKR phases) take these sweeps from 2^(from_log + 1) entries.  An execution schedule: proof bytes do not depend on it. */
int zk_ctx_set_deg2_triple(zk_ctx *ctx, int from_log, int ring);
/* Three-round passes whose successor serves three rounds too: the three-variable fold and the next level's cross sums are ONE sweep
 * (the folded rows go to HBM and into the LDS tile of the matrix stage, so the folded tables are not re-read).
 * dma_slots = 8 (default): the fold's contraction runs on the matrix cores as well (k_fold3_cross64_mfma; 12..14 = the same with a
 * ring of 2..4 steps per wave); 3..5: the vector-ALU fold (k_fold3_cross64) with that many LDS-DMA slots per wave; 0 = separate fold
 * and sums sweeps.  Proof bytes do not depend on it. */
int zk_ctx_set_deg2_fold3_fused(zk_ctx *ctx, int dma_slots);
/* Degree-2 prover, two rounds per pass: while the folded tables have fewer than 2^below_log entries (0 = never) the two-variable fold
 * (prover.rs:131-139, twice) and the nine sums of its result (the next two rounds' prover.rs:89-107) are one kernel -- GKR layer
 * tables are cache-resident and a pass is bound by launches and per-lane chains, not by bytes.  Proof bytes do not depend on it. */
int zk_ctx_set_deg2_fused(zk_ctx *ctx, int below_log);
/* Degree-2 prover, three-round sweeps (k_cross64_mfma, k_fold3_cross64_mfma): on (default) = every workgroup leaves its 64 cross sums
 * as signed integer digits in a context-owned device buffer (64 KiB per workgroup pair, made on first use, grow-only) and the second
 * stage adds the digits over the workgroups before the one field product per digit; off = every workgroup forms its field elements
 * itself (the form of rounds 2 and 3a).  Integer addition is exact: proof bytes do not depend on it. */
int zk_ctx_set_deg2_raw_cross(zk_ctx *ctx, int on);
/* Degree-2 prover, THREE rounds per pass on small tables: while the folded tables have fewer than 2^below_log entries (default 20; 0 = never) a
 * pass folds the variables of the rounds just run (prover.rs:131-139) and forms the 27 sums that determine the next three round
 * polynomials (prover.rs:89-107, three times) in one kernel -- a pass on a cache-resident table costs its host round trip, so it
 * should serve as many rounds as one launch can.  No counterpart in the reference; proof bytes do not depend on it. */
int zk_ctx_set_deg2_small_triple(zk_ctx *ctx, int below_log);
/* Degree-2 prover, the FIRST pass as 27 sums (three rounds before the first fold, k_fold_sum27<0>) for tables below 2^below_log
 * entries (default 19; 0 = never: the nine-sum pass serves two rounds).  linear_below_log (default 20) is the same threshold for relations
 * of the shape F G + H * 1 -- both phases of a GKR layer (gkr/src/prover.rs:55-147 builds them as four dense tables) -- which the GKR
 * prover runs through the product-plus-linear form of the 27-sum kernels: the constant factor is neither read nor folded and the two
 * terms share one wave population.  linear_below_log = 0 turns that form off altogether (the general four-table kernels run).  No
 * counterpart in the reference; proof bytes do not depend on either value. */
int zk_ctx_set_deg2_first_triple(zk_ctx *ctx, int below_log, int linear_below_log);
enum { ZK_K_BLOCK_SUMS = 0, ZK_K_MULTIFOLD = 1, ZK_K_FOLD = 2, ZK_K_EVAL3 = 3, ZK_K_FOLD4_EVAL3 = 4, ZK_K_SUM9 = 5, ZK_K_FOLD2X4_SUM9 = 6, ZK_K_MULTIFOLD_WIDE = 7, ZK_K_MULTIFOLD_TAIL = 8,
       ZK_K_COUNT = 10 };
/* per-kernel HIP-event timing of the launches made by the provers on this context */
int zk_prof_enable(zk_ctx *ctx, int on);
int zk_prof_reset(zk_ctx *ctx);
/* synchronize the stream and account the launches the zk_shard_* calls recorded (the provers do this themselves) */
int zk_prof_collect(zk_ctx *ctx);
/* totals since the last reset: launches, elapsed ms (HIP events on the context stream) and
 * ALGORITHMIC bytes (block sums 32 B/entry; m-round fold 32 B per input entry + 32 B per
 * output entry; degree-2 kernels 32 B (eval, sum9) / 48 B (fold+eval) per input entry per table, two-round
 * fold 32 B per input entry + 32 B per output entry per table) */
int zk_prof_read(zk_ctx *ctx, int kernel, uint64_t *launches, double *ms, uint64_t *alg_bytes);
/* Montgomery-multiply throughput probe: `iters` dependent multiplies per lane on `blocks` x 256
 * lanes; returns elapsed ms.  Used for the int32-ALU roofline figure. */
int zk_bench_modmul(zk_ctx *ctx, int blocks, int iters, double *ms_out);
/* v_mad_u64_u32 issue-rate probe: blocks x 256 lanes, iters x 256 independent multiply-adds per lane; elapsed ms.
 * lane-MADs per second = blocks * 256 * iters * 256 / time: the int32-ALU ceiling of the limb-product loops. */
int zk_bench_mad(zk_ctx *ctx, int blocks, int iters, double *ms_out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ZKMI_H */
