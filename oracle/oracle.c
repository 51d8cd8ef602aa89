/* oracle/oracle.c -- TEST INFRASTRUCTURE ONLY.  See oracle.h for scope, pinning and who may load this. */
#define _GNU_SOURCE
#include "oracle.h"
#include "orc_merlin.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---------------- threads ----------------
 * orc_set_threads(k): worker threads used by the MSMs inside the IPP / R1CS restatements and by the IPP fold loop
 * (each i of src/ipp.rs:115-130,181-188 is independent).  The reference is single-threaded; threads only shorten the
 * checker's run time, results do not depend on k. */
static int orc_threads = 1;
typedef struct { void (*fn)(size_t, size_t, void*); void* arg; size_t lo, hi; } orc_pf_job;
static void* orc_pf_worker(void* p) { orc_pf_job* j = (orc_pf_job*)p; j->fn(j->lo, j->hi, j->arg); return NULL; }
static void orc_parallel_for(size_t n, void (*fn)(size_t, size_t, void*), void* arg) {
    int k = orc_threads;
    if (k > 64) k = 64;
    if ((size_t)k > n / 4) k = (int)(n / 4);
    if (k <= 1) { fn(0, n, arg); return; }
    pthread_t th[64]; orc_pf_job jobs[64];
    for (int t = 0; t < k; t++) {
        jobs[t] = (orc_pf_job){fn, arg, n * (size_t)t / k, n * (size_t)(t + 1) / k};
        pthread_create(&th[t], NULL, orc_pf_worker, &jobs[t]);
    }
    for (int t = 0; t < k; t++) pthread_join(th[t], NULL);
}

/* ---------------- field instances ---------------- */
#define NL 6
#define F(x) fp381_##x
#include "orc_field_tmpl.h"
#undef NL
#undef F

#define NL 4
#define F(x) fr381_##x
#include "orc_field_tmpl.h"
#undef NL
#undef F

#define NL 4
#define F(x) fp254_##x
#include "orc_field_tmpl.h"
#undef NL
#undef F

#define NL 4
#define F(x) fr254_##x
#include "orc_field_tmpl.h"
#undef NL
#undef F

/* ---------------- curve instances ---------------- */
#define C(x) bls381_##x
#define FP(x) fp381_##x
#define FR(x) fr381_##x
#define FP_NL 6
#define FR_NL 4
#define FP_LE_BYTES 48
#define FR_LE_BYTES 32
#define MODBYTES 48
#define COFACTOR_WORDS {0x8c00aaab0000aaabULL, 0x396c8c005555e156ULL, 0, 0}   /* (x-1)^2/3 */
#include "orc_curve_tmpl.h"
#include "orc_ipp_tmpl.h"
#include "orc_r1cs_tmpl.h"
#include "orc_api_tmpl.h"
#undef C
#undef FP
#undef FR
#undef FP_NL
#undef FR_NL
#undef FP_LE_BYTES
#undef FR_LE_BYTES
#undef MODBYTES
#undef COFACTOR_WORDS

#define C(x) bn254_##x
#define FP(x) fp254_##x
#define FR(x) fr254_##x
#define FP_NL 4
#define FR_NL 4
#define FP_LE_BYTES 32
#define FR_LE_BYTES 32
#define MODBYTES 32
#define COFACTOR_WORDS {1, 0, 0, 0}
#include "orc_curve_tmpl.h"
#include "orc_ipp_tmpl.h"
#include "orc_r1cs_tmpl.h"
#include "orc_api_tmpl.h"
#undef C
#undef FP
#undef FR
#undef FP_NL
#undef FR_NL
#undef FP_LE_BYTES
#undef FR_LE_BYTES
#undef MODBYTES
#undef COFACTOR_WORDS

/* ---------------- constants (public parameters; checked against tests/golden/curves.json) ------------- */
static void hex_to_words(uint64_t* out, int nwords, const char* hex) {
    memset(out, 0, nwords * 8);
    size_t len = strlen(hex);
    for (size_t i = 0; i < len; i++) {
        char ch = hex[len - 1 - i];
        uint64_t v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : ch - 'A' + 10;
        if (i / 16 < (size_t)nwords) out[i / 16] |= v << (4 * (i % 16));
    }
}

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_all(void) {
    uint64_t p[6], r[6], gx[6], gy[6];
    hex_to_words(p, 6, "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab");
    hex_to_words(r, 6, "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001");
    hex_to_words(gx, 6, "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb");
    hex_to_words(gy, 6, "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1");
    bls381_api_init(p, r, 4, gx, gy);
    /* AMCL "BN254" (Nogami, u = -(2^62+2^55+1)); G = (p-1, 1); SURVEY F8 */
    hex_to_words(p, 6, "2523648240000001ba344d80000000086121000000000013a700000000000013");
    hex_to_words(r, 6, "2523648240000001ba344d8000000007ff9f800000000010a10000000000000d");
    hex_to_words(gx, 6, "2523648240000001ba344d80000000086121000000000013a700000000000012");
    hex_to_words(gy, 6, "1");
    bn254_api_init(p, r, 2, gx, gy);
}
#define INIT() pthread_once(&g_once, init_all)
#define DISPATCH(call_bls, call_bn) do { INIT(); switch (curve) { case 0: return call_bls; case 1: return call_bn; default: return 2; } } while (0)

int orc_fp_bytes(int curve) { return curve == 0 ? 48 : 32; }
int orc_fr_bytes(int curve) { (void)curve; return 32; }
int orc_modbytes(int curve) { return curve == 0 ? 48 : 32; }

int orc_field_op(int curve, int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    DISPATCH(bls381_api_field_op(which, op, a, b, out), bn254_api_field_op(which, op, a, b, out));
}
int orc_g1_on_curve(int curve, const uint8_t* p) { DISPATCH(bls381_api_on_curve(p), bn254_api_on_curve(p)); }
int orc_g1_generator(int curve, uint8_t* out) {
    INIT();
    if (curve == 0) { bls381_aff_to_le(out, &bls381_GEN); return 0; }
    if (curve == 1) { bn254_aff_to_le(out, &bn254_GEN); return 0; }
    return 2;
}
int orc_g1_add(int curve, const uint8_t* p, const uint8_t* q, uint8_t* out) { DISPATCH(bls381_api_g1_add(p, q, out), bn254_api_g1_add(p, q, out)); }
int orc_g1_mul(int curve, const uint8_t* k, const uint8_t* p, uint8_t* out) { DISPATCH(bls381_api_g1_mul(k, p, out), bn254_api_g1_mul(k, p, out)); }
int orc_g1_binary_scalar_mul(int curve, const uint8_t* p, const uint8_t* h, const uint8_t* r1, const uint8_t* r2, uint8_t* out) {
    DISPATCH(bls381_api_binary_scalar_mul(p, h, r1, r2, out), bn254_api_binary_scalar_mul(p, h, r1, r2, out));
}
int orc_g1_fixed_base_batch(int curve, const uint8_t* ks, size_t n, int nthreads, uint8_t* out) {
    DISPATCH(bls381_api_fixed_base_batch(ks, n, nthreads, out), bn254_api_fixed_base_batch(ks, n, nthreads, out));
}
int orc_g1_to_amcl(int curve, const uint8_t* p, uint8_t* out) {
    INIT();
    if (curve == 0) { bls381_aff_t a; bls381_aff_from_le(&a, p); bls381_aff_to_amcl(out, &a, 48); return 0; }
    if (curve == 1) { bn254_aff_t a; bn254_aff_from_le(&a, p); bn254_aff_to_amcl(out, &a, 32); return 0; }
    return 2;
}
int orc_g1_from_msg_hash(int curve, const uint8_t* msg, size_t len, uint8_t* out) {
    DISPATCH(bls381_api_from_msg_hash(msg, len, out), bn254_api_from_msg_hash(msg, len, out));
}
int orc_get_generators(int curve, const uint8_t* prefix, size_t prefix_len, uint64_t first, size_t n, int nthreads, uint8_t* out) {
    DISPATCH(bls381_api_get_generators(prefix, prefix_len, first, n, nthreads, out), bn254_api_get_generators(prefix, prefix_len, first, n, nthreads, out));
}
int orc_msm(int curve, int algo, const uint8_t* points, const uint8_t* scalars, size_t n, int nthreads, uint8_t* out) {
    DISPATCH(bls381_api_msm(algo, points, scalars, n, nthreads, out), bn254_api_msm(algo, points, scalars, n, nthreads, out));
}
int orc_msm_timed(int curve, int algo, const uint8_t* points, const uint8_t* scalars, size_t n, int nthreads, uint8_t* out, double* seconds) {
    DISPATCH(bls381_api_msm_timed(algo, points, scalars, n, nthreads, out, seconds), bn254_api_msm_timed(algo, points, scalars, n, nthreads, out, seconds));
}
int orc_fr_inner(int curve, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
    DISPATCH(bls381_api_fr_inner(a, b, n, out), bn254_api_fr_inner(a, b, n, out));
}

int orc_random_scalars(int curve, uint64_t seed, size_t n, uint8_t* out) {
    INIT();
    const uint64_t* mod; int bits;
    if (curve == 0) { mod = fr381_P.mod; bits = fr381_P.bits; } else if (curve == 1) { mod = fr254_P.mod; bits = fr254_P.bits; } else return 2;
    uint64_t s = seed;
    for (size_t i = 0; i < n; i++) {
        uint64_t v[4];
        for (;;) {
            for (int w = 0; w < 4; w++) {
                s += 0x9E3779B97F4A7C15ULL;
                uint64_t z = s;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
                v[w] = z ^ (z >> 31);
            }
            if (bits < 256) v[3] &= (1ULL << (bits - 192)) - 1;
            int lt = 0;
            for (int w = 3; w >= 0; w--) { if (v[w] < mod[w]) { lt = 1; break; } if (v[w] > mod[w]) break; }
            if (lt) break;
        }
        for (int j = 0; j < 32; j++) out[i * 32 + j] = (uint8_t)(v[j / 8] >> (8 * (j % 8)));
    }
    return 0;
}

size_t orc_transcript_size(void) { return sizeof(orc_transcript); }
void orc_transcript_new(void* t, const uint8_t* label, size_t label_len) { orc_transcript_init((orc_transcript*)t, label, label_len); }
void orc_transcript_append_message(void* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    orc_transcript_append((orc_transcript*)t, label, label_len, msg, msg_len);
}
void orc_transcript_challenge_bytes(void* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len) {
    orc_transcript_challenge((orc_transcript*)t, label, label_len, out, out_len);
}
int orc_transcript_commit_point(int curve, void* t, const char* label, const uint8_t* p) {
    DISPATCH(bls381_api_commit_point((orc_transcript*)t, label, p), bn254_api_commit_point((orc_transcript*)t, label, p));
}
int orc_transcript_challenge_scalar(int curve, void* t, const char* label, uint8_t* out) {
    DISPATCH(bls381_api_challenge_scalar((orc_transcript*)t, label, out), bn254_api_challenge_scalar((orc_transcript*)t, label, out));
}

int orc_ipp_create(int curve, void* tr, const uint8_t* Q, const uint8_t* Gf, const uint8_t* Hf, const uint8_t* G, const uint8_t* H,
                   const uint8_t* a, const uint8_t* b, size_t n, uint8_t* L_out, uint8_t* R_out, uint8_t* a_out, uint8_t* b_out) {
    DISPATCH(bls381_api_ipp_create((orc_transcript*)tr, Q, Gf, Hf, G, H, a, b, n, L_out, R_out, a_out, b_out),
             bn254_api_ipp_create((orc_transcript*)tr, Q, Gf, Hf, G, H, a, b, n, L_out, R_out, a_out, b_out));
}
int orc_ipp_verify(int curve, void* tr, size_t n, const uint8_t* Gf, const uint8_t* Hf, const uint8_t* P, const uint8_t* Q,
                   const uint8_t* G, const uint8_t* H, const uint8_t* a, const uint8_t* b, const uint8_t* L, const uint8_t* R, size_t lg_n) {
    DISPATCH(bls381_api_ipp_verify((orc_transcript*)tr, n, Gf, Hf, P, Q, G, H, a, b, L, R, lg_n),
             bn254_api_ipp_verify((orc_transcript*)tr, n, Gf, Hf, P, Q, G, H, a, b, L, R, lg_n));
}
int orc_ipp_verification_scalars(int curve, void* tr, const uint8_t* L, const uint8_t* R, size_t lg_n, size_t n,
                                 uint8_t* u_sq, uint8_t* u_inv_sq, uint8_t* s) {
    DISPATCH(bls381_api_verification_scalars((orc_transcript*)tr, L, R, lg_n, n, u_sq, u_inv_sq, s),
             bn254_api_verification_scalars((orc_transcript*)tr, L, R, lg_n, n, u_sq, u_inv_sq, s));
}

void orc_set_threads(int k) { orc_threads = k < 1 ? 1 : k; }

int orc_r1cs_prove(int curve, void* tr, size_t n_terms, const uint32_t* term_constraint, const uint8_t* term_kind, const uint32_t* term_index,
                   const uint8_t* coeff, size_t n_constraints, size_t n, size_t m, const uint8_t* g, const uint8_t* h, const uint8_t* G,
                   const uint8_t* H, size_t ngens, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO, const uint8_t* v_blinding,
                   const uint8_t* sL, const uint8_t* sR, const uint8_t* blindings, uint8_t* proof_out) {
    DISPATCH(bls381_api_r1cs_prove((orc_transcript*)tr, n_terms, term_constraint, term_kind, term_index, coeff, n_constraints, n, m, g, h, G, H, ngens, aL, aR,
                                   aO, v_blinding, sL, sR, blindings, proof_out),
             bn254_api_r1cs_prove((orc_transcript*)tr, n_terms, term_constraint, term_kind, term_index, coeff, n_constraints, n, m, g, h, G, H, ngens, aL, aR,
                                  aO, v_blinding, sL, sR, blindings, proof_out));
}
int orc_r1cs_verify(int curve, void* tr, size_t n_terms, const uint32_t* term_constraint, const uint8_t* term_kind, const uint32_t* term_index,
                    const uint8_t* coeff, size_t n_constraints, size_t n, size_t m, const uint8_t* V, const uint8_t* proof, size_t proof_len,
                    const uint8_t* g, const uint8_t* h, const uint8_t* G, const uint8_t* H, size_t ngens, const uint8_t* rnd) {
    DISPATCH(bls381_api_r1cs_verify((orc_transcript*)tr, n_terms, term_constraint, term_kind, term_index, coeff, n_constraints, n, m, V, proof, proof_len, g, h,
                                    G, H, ngens, rnd),
             bn254_api_r1cs_verify((orc_transcript*)tr, n_terms, term_constraint, term_kind, term_index, coeff, n_constraints, n, m, V, proof, proof_len, g, h,
                                   G, H, ngens, rnd));
}
int orc_r1cs_flattened_constraints(int curve, size_t n_terms, const uint32_t* term_constraint, const uint8_t* term_kind, const uint32_t* term_index,
                                   const uint8_t* coeff, size_t n_constraints, size_t n, size_t m, const uint8_t* z, uint8_t* wL, uint8_t* wR, uint8_t* wO,
                                   uint8_t* wV, uint8_t* wc) {
    DISPATCH(bls381_api_r1cs_flatten(n_terms, term_constraint, term_kind, term_index, coeff, n_constraints, n, m, z, wL, wR, wO, wV, wc),
             bn254_api_r1cs_flatten(n_terms, term_constraint, term_kind, term_index, coeff, n_constraints, n, m, z, wL, wR, wO, wV, wc));
}
int orc_transcript_commit_scalar(int curve, void* t, const char* label, const uint8_t* x) {
    DISPATCH(bls381_api_commit_scalar((orc_transcript*)t, label, x), bn254_api_commit_scalar((orc_transcript*)t, label, x));
}
