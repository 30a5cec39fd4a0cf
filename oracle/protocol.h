/*
 * oracle/protocol.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * CPU restatement of the reference's protocol layer:
 *   Transcript            fiat_shamir/src/transcript.rs:6-42
 *   SumcheckProver        sumcheck/src/prover.rs:17-191
 *   SumcheckVerifier      sumcheck/src/verifier.rs:9-95
 *   SumCheckProof         sumcheck/src/sumcheck_protocol.rs:5-9
 *   Circuit / Gate        arithmetic_circuit/src/circuit.rs:9-152, gate.rs:7-22
 *   GKRProver / Verifier  gkr/src/prover.rs:24-154, gkr/src/verifier.rs:21-119,
 *                         gkr/src/utils.rs:5-65, gkr/src/gkr_protocol.rs:8-26
 */
#ifndef ORACLE_PROTOCOL_H
#define ORACLE_PROTOCOL_H
#include "keccak.h"
#include "poly.h"

/* ---- Transcript ---- */
typedef struct { keccak256_t hasher; } transcript_t;
void transcript_new(transcript_t *t);                                       /* :12-17 */
void transcript_append(transcript_t *t, const uint8_t *data, size_t len);   /* :20-22 */
void transcript_sample_challenge(transcript_t *t, fq_t *out);               /* :28-36 */

/* ---- SumCheckProof ---- */
typedef struct {
    fq_t initial_claim_sum;
    unipoly_t *round_polys;
    size_t n_rounds;
} sumcheck_proof_t;
void sumcheck_proof_free(sumcheck_proof_t *p);

/* ComposedPolynomial (prover.rs:12-15): exactly one of the two is non-NULL */
int sumcheck_generate_round_polys(const sumpoly_t *sum_poly, const mle_t *multi_poly,
                                  transcript_t *transcript, unipoly_t **round_polys,
                                  fq_t **challenges, size_t *n_rounds);     /* prover.rs:45-155 */
int sumcheck_generate_proof(sumcheck_proof_t *proof, const mle_t *init_poly); /* prover.rs:158-173 */
int sumcheck_generate_proof_for_partial_verify(sumcheck_proof_t *proof, fq_t **random_points,
                                               const fq_t *initial_claim_sum, const sumpoly_t *init_poly,
                                               transcript_t *transcript);   /* prover.rs:175-190 */

/* verifier.rs:29-62: returns ok flag; *final_claim and challenges (n_rounds, caller frees) */
int sumcheck_partial_verify(const sumcheck_proof_t *proof, transcript_t *transcript,
                            fq_t *final_claim, fq_t **challenges, size_t *n_challenges);
int sumcheck_verify_proof(const mle_t *initial_polynomial, const sumcheck_proof_t *proof); /* :64-94 */

/* ---- Circuit ---- */
enum { GATE_ADD = 0, GATE_MUL = 1 };
typedef struct { size_t left, right; int op; } gate_t;           /* gate.rs:7-22 */
typedef struct { gate_t **layers; size_t *layer_len; size_t n_layers; } circuit_t; /* layers[0] is next to the inputs */

int circuit_new(circuit_t *c, const gate_t *const *layers, const size_t *layer_len, size_t n_layers);
void circuit_free(circuit_t *c);
/* circuit.rs:37-59: evals[0] = inputs, evals[k] = output of layers[k-1]; n_layers+1 entries */
int circuit_evaluate_at_input(const circuit_t *c, const fq_t *inputs, size_t n_inputs, mle_t **evals);
int circuit_get_gate_poly(mle_t *out, const circuit_t *c, size_t layer_idx, int op); /* :82-123 */
int circuit_get_w_i(mle_t *out, size_t layer_idx, const mle_t *evals, size_t n_evals); /* :126-138 */

/* ---- GKR ---- */
typedef struct {
    mle_t output_poly;
    fq_t *w_evals;            /* (wb, wc) pairs, 2 * n_w_evals entries */
    size_t n_w_evals;
    sumcheck_proof_t *sumcheck_proofs;
    size_t n_proofs;
} gkr_proof_t;
void gkr_proof_free(gkr_proof_t *p);
int gkr_generate_proof(gkr_proof_t *proof, const circuit_t *c, transcript_t *t,
                       const fq_t *inputs, size_t n_inputs);               /* gkr/src/prover.rs:24-154 */
int gkr_verify_proof(const fq_t *inputs, size_t n_inputs, const circuit_t *c, transcript_t *t,
                     const gkr_proof_t *proof);                            /* gkr/src/verifier.rs:21-119 */

#endif
