/* include/bpmsm.h -- C ABI of the MI355X-native MSM / inner-product-argument engine (libbpmsm.so).
 *
 * This is the drop-in boundary for the hot path of lovesh/bulletproofs-amcl.  The reference has NO FFI or
 * plugin seam (it calls amcl_wrapper's Rust types by static dispatch, e.g. /root/reference src/ipp.rs:7-9), so
 * each entry point below names the amcl_wrapper / crate call it replaces; a Rust shim that binds these is shown
 * in INTEGRATION.md.  Conventions kept from the call sites:
 *   - every function returns an int status (no exceptions cross the boundary):
 *       BP_OK                0
 *       BP_ERR_LENGTH        1   length mismatch        <-> amcl_wrapper ValueError (callers .unwrap(): ipp.rs:91,104,158,170,253)
 *       BP_ERR_ARG           2   bad argument           <-> assert!/assert_eq! panics in create_ipp (ipp.rs:48-55)
 *       BP_ERR_VERIFY        3   verification failed    <-> R1CSError::VerificationError (ipp.rs:258,272,275; errors.rs:7-28)
 *       BP_ERR_DEVICE        4   HIP runtime error / no GPU.  There is NO CPU fallback: without a gfx950 device
 *                                every compute entry point fails with this code.
 *   - inputs are borrowed, outputs are caller-owned buffers; opaque handles are owned by the library until *_free.
 *   - one bp_ctx per host thread (the reference is single-threaded; `cargo test` runs tests on parallel threads,
 *     so contexts share no mutable state).  Each ctx owns one HIP stream and its scratch workspace.
 *
 * Byte formats (fmt):
 *   BP_FMT_LE   0   canonical little-endian: field element = 4*limbs32 bytes (48 for BLS12-381 Fp, 32 otherwise);
 *                   G1 point = x || y; the all-zero encoding is the identity.  Scalars are ALWAYS 32-byte LE, < r.
 *   BP_FMT_AMCL 1   amcl ECP::tobytes(compress=false): 0x04 || X || Y, MODBYTES big-endian each
 *                   (97 B BLS12-381 / 65 B BN254); identity = 04 || 0 || 1.  [UNVERIFIED-RECALL, SURVEY 8c]
 */
#ifndef BPMSM_H
#define BPMSM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BP_OK 0
#define BP_ERR_LENGTH 1
#define BP_ERR_ARG 2
#define BP_ERR_VERIFY 3
#define BP_ERR_DEVICE 4

#define BP_CURVE_BLS12_381 0 /* Cargo feature bls381 (default), Cargo.toml:22-23 */
#define BP_CURVE_BN254 1     /* Cargo feature bn254 -> AMCL "BN254" (Nogami), Cargo.toml:24, SURVEY F8 */

#define BP_FMT_LE 0
#define BP_FMT_AMCL 1

typedef struct bp_ctx bp_ctx;
typedef struct bp_g1vec bp_g1vec; /* amcl_wrapper::group_elem_g1::G1Vector, device resident */
typedef struct bp_frvec bp_frvec; /* amcl_wrapper::field_elem::FieldElementVector, device resident */

typedef struct {
    int curve_id;
    int fp_bytes;  /* BP_FMT_LE field-element bytes: 48 / 32 */
    int fr_bytes;  /* 32 */
    int modbytes;  /* amcl MODBYTES: 48 / 32 */
    int fr_bits;   /* 255 / 254 */
    uint8_t p_le[48];
    uint8_t r_le[32];
    uint8_t gen_le[96]; /* generator, BP_FMT_LE (first 2*fp_bytes bytes used) */
} bp_curve_info;

/* Library / build identification; never touches the GPU. */
const char* bp_version(void);
int bp_curve_params(int curve_id, bp_curve_info* out);
/* Number of visible HIP devices (0 if none); never fails. */
int bp_device_count(void);

/* ---- context ---------------------------------------------------------------------------------------------- */
/* Selects curve (compile-time limb geometry behind a runtime id; the reference selects by Cargo feature,
 * Cargo.toml:22-27) and the HIP device ordinal.  Fails with BP_ERR_DEVICE if there is no usable GPU. */
int bp_ctx_create(int curve_id, int device_ordinal, bp_ctx** out);
int bp_ctx_destroy(bp_ctx* ctx);
/* Run this context's kernels on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the
 * context's own stream. */
int bp_ctx_set_stream(bp_ctx* ctx, void* hip_stream);
/* Block until everything queued on the context's stream has finished. */
int bp_ctx_synchronize(bp_ctx* ctx);
/* Pippenger window width in bits (1..16) for subsequent MSMs; 0 = choose from n (default). */
int bp_ctx_set_window_bits(bp_ctx* ctx, int c);

/* ---- G1Vector --------------------------------------------------------------------------------------------- */
/* G1Vector::from(Vec<G1>) : host bytes -> HBM (converted to packed Montgomery affine on the device). */
int bp_g1vec_upload(bp_ctx* ctx, const uint8_t* points, size_t n, int fmt, bp_g1vec** out);
/* G1Vector::with_capacity / new(n): n identity points. */
int bp_g1vec_alloc(bp_ctx* ctx, size_t n, bp_g1vec** out);
int bp_g1vec_download(bp_ctx* ctx, const bp_g1vec* v, size_t offset, size_t n, int fmt, uint8_t* out);
int bp_g1vec_free(bp_g1vec* v);
size_t bp_g1vec_len(const bp_g1vec* v);
/* Raw HBM pointer and byte stride of the resident vector (2*fp_bytes per point), for callers that manage device
 * memory themselves (torch tensors, RCCL buffers). */
void* bp_g1vec_device_ptr(bp_g1vec* v);
/* Non-owning view over caller-owned HBM already in the resident layout (as produced by bp_g1vec_device_ptr). */
int bp_g1vec_wrap_device(bp_ctx* ctx, void* device_ptr, size_t n, bp_g1vec** out);
/* out[i] = k[i] * G.  Batched form of `&G1::generator() * &FieldElement` (src/utils/mod.rs:34); used to build
 * synthetic generator vectors on the device. */
int bp_g1vec_fixed_base_mul(bp_ctx* ctx, const bp_frvec* k, bp_g1vec** out);
/* out[i] = k[i] * p[i]   (`&G1 * &FieldElement`, src/r1cs/prover.rs:358,423,550), batched. */
int bp_g1vec_scalar_mul(bp_ctx* ctx, const bp_g1vec* p, const bp_frvec* k, bp_g1vec** out);

/* ---- FieldElementVector ------------------------------------------------------------------------------------- */
int bp_frvec_upload(bp_ctx* ctx, const uint8_t* scalars_le32, size_t n, bp_frvec** out);
int bp_frvec_alloc(bp_ctx* ctx, size_t n, bp_frvec** out); /* zeros */
int bp_frvec_download(bp_ctx* ctx, const bp_frvec* v, size_t offset, size_t n, uint8_t* out_le32);
int bp_frvec_free(bp_frvec* v);
size_t bp_frvec_len(const bp_frvec* v);
void* bp_frvec_device_ptr(bp_frvec* v);
int bp_frvec_wrap_device(bp_ctx* ctx, void* device_ptr, size_t n, bp_frvec** out);
/* ---- multi-scalar multiplication (the headline path) -------------------------------------------------------- */
/* G1Vector::multi_scalar_mul_var_time(&scalars) (src/ipp.rs:251-253, tests :372,:471),
 * G1Vector::inner_product_var_time_with_ref_vecs (src/ipp.rs:91,104,158,170; src/r1cs/verifier.rs:451) and
 * G1Vector::inner_product_const_time (src/r1cs/prover.rs:358,423 -- a constant-time request has no meaning for
 * this engine and runs the same kernels):   out = sum_i scalars[i] * points[i]   as BP_FMT_LE affine bytes.
 * Pippenger bucket method, signed windows, hand-written gfx950 kernels; bit-exact (canonical affine) against the
 * CPU oracle.  BP_ERR_LENGTH if the two lengths differ (amcl_wrapper's ValueError). */
int bp_msm_g1(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars, uint8_t* out_le);
/* Same over sub-ranges [poff, poff+n) x [soff, soff+n) (the reference slices G[0..n1], src/r1cs/prover.rs:343-344).
 * BP_ERR_LENGTH if a range overruns its vector. */
int bp_msm_g1_range(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n,
                    uint8_t* out_le);
/* Two-stage form used when the index range is sharded over several GPUs (one process per GPU).
 * Stage 1 (device): each rank runs the bucket pipeline on its own slice and leaves its W per-window bucket sums
 * ("window records", un-normalised XYZZ, bp_msm_record_bytes() each, bp_msm_window_records(ctx, n) of them) in a
 * caller-owned HBM buffer.  The caller all-gathers the N ranks' records over RCCL (point addition is not an RCCL
 * reduction op; N*W*192 B is latency-bound).  Stage 2 (bp_msm_g1_finish): one D2H copy of `sets`
 * record sets, per-window sum, the serial 2^(c w) fold and the affine normalisation, giving BP_FMT_LE bytes.
 * All ranks must use the same n_per_set (or the same bp_ctx_set_window_bits) so that window geometry agrees. */
size_t bp_msm_window_records(bp_ctx* ctx, size_t n);
size_t bp_msm_record_bytes(int curve_id);
int bp_msm_g1_windows(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n,
                      void* device_out);
int bp_msm_g1_finish(bp_ctx* ctx, const void* device_records, size_t sets, size_t n_per_set, uint8_t* out_le);

/* Timing of the last bp_msm_* call on this context, measured with HIP events on the context's stream.
 * ms[0] = whole device pipeline, ms[1..] = per stage (digits+count, scan, scatter, tasks, accumulate, reduce); returns the
 * number of entries written (<= cap). */
int bp_msm_last_timing(bp_ctx* ctx, float* ms, int cap);
int bp_ctx_enable_timing(bp_ctx* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* BPMSM_H */
