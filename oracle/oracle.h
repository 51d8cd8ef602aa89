/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference's hot path: field/curve arithmetic that the reference
 * takes from the un-vendored crates amcl_wrapper 0.1.5 / amcl, the Merlin 1.x transcript, and the
 * inner-product argument of src/ipp.rs.  PARITY UNPINNED with respect to the reference itself (no Rust
 * toolchain, crates absent, reference tests hold no known-answer vectors: SURVEY.md F2/F4/F5, 8c);
 * pinned instead by tests/golden/ (*.json; Python-int vectors from oracle/gen_golden.py, public curve
 * KATs, the Merlin conformance vector).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Byte formats are the C ABI's BP_FMT_LE (include/bpmsm.h): field elements little-endian canonical,
 * 4*limbs32 bytes; points x||y; the all-zero point encoding is the identity.
 * curve: 0 = BLS12-381, 1 = AMCL BN254 (Nogami).  Return codes follow include/bpmsm.h (0 ok, 2 bad
 * argument, 3 verification failed).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int orc_fp_bytes(int curve);   /* 48 / 32 */
int orc_fr_bytes(int curve);   /* 32 / 32 */
int orc_modbytes(int curve);   /* amcl MODBYTES: 48 / 32 */

/* which: 0 = Fp, 1 = Fr; op: 0 add, 1 sub, 2 mul, 3 inverse of a (b ignored; inv(0) = 0) */
int orc_field_op(int curve, int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out);
int orc_g1_on_curve(int curve, const uint8_t* p);
int orc_g1_generator(int curve, uint8_t* out);
int orc_g1_add(int curve, const uint8_t* p, const uint8_t* q, uint8_t* out);
int orc_g1_mul(int curve, const uint8_t* k, const uint8_t* p, uint8_t* out);
int orc_g1_binary_scalar_mul(int curve, const uint8_t* p, const uint8_t* h, const uint8_t* r1, const uint8_t* r2, uint8_t* out);
int orc_g1_fixed_base_batch(int curve, const uint8_t* ks, size_t n, int nthreads, uint8_t* out_points);
/* G1::from_msg_hash (amcl_wrapper) and get_generators (src/utils/mod.rs:16-23): msg -> point, x || y little-endian */
int orc_g1_from_msg_hash(int curve, const uint8_t* msg, size_t len, uint8_t* out);
int orc_get_generators(int curve, const uint8_t* prefix, size_t prefix_len, uint64_t first, size_t n, int nthreads, uint8_t* out);
int orc_g1_to_amcl(int curve, const uint8_t* p, uint8_t* out /* 2*modbytes+1 */);

/* algo: 0 naive, 1 Strauss wNAF-5 single thread (reference-like), 2 Pippenger with nthreads */
int orc_msm(int curve, int algo, const uint8_t* points, const uint8_t* scalars, size_t n, int nthreads, uint8_t* out);
int orc_msm_timed(int curve, int algo, const uint8_t* points, const uint8_t* scalars, size_t n, int nthreads, uint8_t* out, double* seconds);
int orc_fr_inner(int curve, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);

/* seeded inputs: splitmix64, rejection sampling of bit_length(r)-bit draws */
int orc_random_scalars(int curve, uint64_t seed, size_t n, uint8_t* out);

/* Merlin transcript (opaque, 208 bytes) */
size_t orc_transcript_size(void);
void orc_transcript_new(void* t, const uint8_t* label, size_t label_len);
void orc_transcript_append_message(void* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len);
void orc_transcript_challenge_bytes(void* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len);
int orc_transcript_commit_point(int curve, void* t, const char* label, const uint8_t* p);
int orc_transcript_challenge_scalar(int curve, void* t, const char* label, uint8_t* out);

/* src/ipp.rs:35-202; L_out/R_out receive lg2(n) points */
int orc_ipp_create(int curve, void* transcript, const uint8_t* Q, const uint8_t* G_factors, const uint8_t* H_factors,
                   const uint8_t* G, const uint8_t* H, const uint8_t* a, const uint8_t* b, size_t n,
                   uint8_t* L_out, uint8_t* R_out, uint8_t* a_out, uint8_t* b_out);
/* src/ipp.rs:204-260 */
int orc_ipp_verify(int curve, void* transcript, size_t n, const uint8_t* G_factors, const uint8_t* H_factors,
                   const uint8_t* P, const uint8_t* Q, const uint8_t* G, const uint8_t* H,
                   const uint8_t* a, const uint8_t* b, const uint8_t* L, const uint8_t* R, size_t lg_n);
/* src/ipp.rs:262-315 */
int orc_ipp_verification_scalars(int curve, void* transcript, const uint8_t* L, const uint8_t* R, size_t lg_n, size_t n,
                                 uint8_t* u_sq, uint8_t* u_inv_sq, uint8_t* s);

/* worker threads for the MSMs and the fold loop inside orc_ipp_* / orc_r1cs_* (results are independent of k; default 1) */
void orc_set_threads(int k);
int orc_transcript_commit_scalar(int curve, void* t, const char* label, const uint8_t* x_le32);

/* R1CS layer, constraint systems without deferred constraints (oracle/orc_r1cs_tmpl.h):
 *   Prover::prove   src/r1cs/prover.rs:322-593      Verifier::verify   src/r1cs/verifier.rs:267-457
 * Terms: constraint index, kind (0/1/2 MultiplierLeft/Right/Output, 3 Committed, 4 One), variable index, 32-byte LE coefficient.
 * The transcript already holds r1cs_domain_sep and one commit_point("V") per committed value.  blindings = i, o, s, t1, t3,
 * t4, t5, t6.  Proof bytes as include/bpmsm.h lays them out: 11 points | t_x t_x_blinding e_blinding | L[lg] R[lg] | a b.
 * Return 0 ok / accepted, 1 not enough generators, 2 bad argument, 3 verification failed. */
int orc_r1cs_prove(int curve, void* transcript, size_t n_terms, const uint32_t* term_constraint, const uint8_t* term_kind, const uint32_t* term_index,
                   const uint8_t* coeff, size_t n_constraints, size_t n, size_t m, const uint8_t* g, const uint8_t* h, const uint8_t* G,
                   const uint8_t* H, size_t ngens, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO, const uint8_t* v_blinding,
                   const uint8_t* sL, const uint8_t* sR, const uint8_t* blindings, uint8_t* proof_out);
int orc_r1cs_verify(int curve, void* transcript, size_t n_terms, const uint32_t* term_constraint, const uint8_t* term_kind, const uint32_t* term_index,
                    const uint8_t* coeff, size_t n_constraints, size_t n, size_t m, const uint8_t* V, const uint8_t* proof, size_t proof_len,
                    const uint8_t* g, const uint8_t* h, const uint8_t* G, const uint8_t* H, size_t ngens, const uint8_t* rnd);
int orc_r1cs_flattened_constraints(int curve, size_t n_terms, const uint32_t* term_constraint, const uint8_t* term_kind, const uint32_t* term_index,
                                   const uint8_t* coeff, size_t n_constraints, size_t n, size_t m, const uint8_t* z, uint8_t* wL, uint8_t* wR, uint8_t* wO,
                                   uint8_t* wV, uint8_t* wc);
#ifdef __cplusplus
}
#endif
#endif
