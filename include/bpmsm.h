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
/* libbpmsm.so is built with -fvisibility=hidden: exactly the functions declared between this push and the pop at the end of the
 * header are exported (tests/test_capi_cpu.py compares `nm -D` with this file). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
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
/* Where bp_msm_g1 / bp_msm_g1_range fold the W window sums into the result (sum_w 2^off_w S_w, a serial chain of ~255
 * doublings): 0 (default) on the host in ~0.13 ms; 1 on the device by a single lane (~2 ms; nothing but the final
 * 2*fp_bytes leaves HBM).  Same bytes either way. */
int bp_ctx_set_device_tail(bp_ctx* ctx, int on);
/* Pippenger window width in bits (2..16) for subsequent MSMs; 0 = choose from n (default). */
int bp_ctx_set_window_bits(bp_ctx* ctx, int c);
/* Engineering knobs of the MSM pipeline, per context, VALIDATED when set (BP_ERR_ARG otherwise).  Until round 2 these were
 * environment variables trusted as they were (a tile that is not a multiple of 256 silently dropped scalars); since round 3 the
 * library reads no environment variable that can change a result (BP_VERBOSE / BP_TRACE only add diagnostics on stderr).
 * value 0 = automatic (the default).  Sibling contexts (R1CS commitments in flight) and bp_msm_g1_multi shards inherit them. */
#define BP_TUNE_TILE 1        /* scalars per block of the binning passes: a multiple of 256 in [256, 16384] */
#define BP_TUNE_REDUCE_M 2    /* buckets per bucket-reduce thread: a power of two in [1, 16384] */
#define BP_TUNE_TASK_TARGET 3 /* number of tasks the accumulate kernel aims at: [1024, 2^28] */
#define BP_TUNE_SMALL_MSM 4   /* 1 (default) / 0: single-launch path for n <= 1536 terms (and, inside an inner-product proof of 16 .. 4096
                               * generators, for its rounds of up to 8193 terms over precomputed digit multiples) */
#define BP_TUNE_TAIL_CHAINS 5 /* host tail: independent Horner walks on helper threads, 1 .. 16 (0: 4 when a fold has >= 48 records, 8 / 16 for several shards' sets, else 1) */
#define BP_TUNE_GLV 7         /* 0 (default): that compaction and the rounds after it -- and every round of a proof of 64 .. 4096 generators -- split
                               * each scalar into two 128-bit halves with the curve's endomorphism (both curves: half the Horner chain, half the
                               * windows per launch, half the host tail); 1 = off.  Proof bytes do not depend on it. */
#define BP_TUNE_COMPACT_AT 6  /* inner-product prover (bp_ipp_create, round API): live length at which the folded generators are materialised once
                               * and the remaining rounds run as single launches over their digit multiples, instead of a full-size paired MSM in
                               * every round (/root/reference src/ipp.rs:181-188 folds G, H every round; this is that fold, done once).  0 = automatic
                               * (4096, for proofs of >= 8192 generators), 1 = never, else a power of two in [16, 4096]: every longer proof compacts
                               * there.  Proof bytes do not depend on it. */
#define BP_TUNE_VERIFY_TABLES 8 /* verifiers (bp_ipp_verify, bp_ipp_verify_batch, bp_r1cs_verify*): n >= 2 = when BOTH generator vectors carry a window table
                               * of the same width (bp_g1vec_precompute) and the statement has at least n generators per vector, the [G | H] part of the
                               * verification MSM (/root/reference src/ipp.rs:244-253, src/r1cs/verifier.rs:431-451) runs over those tables, the proof's
                               * own points as a second, small MSM beside it; the context then keeps the side-by-side table of the generators it last
                               * verified against (2 W n rows) until bp_ctx_drop_verify_table / bp_ctx_destroy.  0 (default) = never: on MI355X the
                               * plain MSM is as fast or faster (its 16 passes over 2n points stay in L2, the table's rows do not: IPP 2^16 verify 1.09
                               * against 1.13 ms).  Accept / reject do not depend on it. */
int bp_ctx_set_tuning(bp_ctx* ctx, int knob, long value);
/* Vectors and temporaries come from a per-context caching pool (hipMalloc / hipFree per proof cost more than the kernels
 * of a small proof; blocks are recycled in stream order).  bp_ctx_trim returns the cached blocks to the driver. */
int bp_ctx_trim(bp_ctx* ctx);

/* ---- G1Vector --------------------------------------------------------------------------------------------- */
/* G1Vector::from(Vec<G1>) : host bytes -> HBM (converted to packed Montgomery affine on the device).
 * Every point is VALIDATED as amcl's G1::from_bytes does: BP_ERR_ARG if a coordinate is not canonical (>= p) or the point
 * is not on y^2 = x^3 + b (the all-zero identity encoding is accepted).  The verifiers upload proof points through the same
 * check and report BP_ERR_VERIFY.  bp_g1vec_wrap_device is the unchecked door for memory the caller already trusts. */
int bp_g1vec_upload(bp_ctx* ctx, const uint8_t* points, size_t n, int fmt, bp_g1vec** out);
/* G1Vector::with_capacity / new(n): n identity points. */
int bp_g1vec_alloc(bp_ctx* ctx, size_t n, bp_g1vec** out);
int bp_g1vec_download(bp_ctx* ctx, const bp_g1vec* v, size_t offset, size_t n, int fmt, uint8_t* out);
int bp_g1vec_free(bp_g1vec* v);
size_t bp_g1vec_len(const bp_g1vec* v);
/* Raw HBM pointer and byte stride of the resident vector (2*fp_bytes per point), for callers that manage device
 * memory themselves (torch tensors, RCCL buffers).
 * LIFETIME: the library orders every use of a vector on its owner's stream and recycles freed blocks through the context's pool
 * without synchronising.  A pointer obtained here leaves that order (a torch / RCCL stream, a bp_g1vec_wrap_device view on
 * ANOTHER context), so from this call on bp_g1vec_free of the owner waits for the whole device before the block can be reused
 * (the behaviour of hipFree).  A view must still not be used after its owner has been freed. */
void* bp_g1vec_device_ptr(bp_g1vec* v);
/* Non-owning view over caller-owned HBM already in the resident layout (as produced by bp_g1vec_device_ptr). */
int bp_g1vec_wrap_device(bp_ctx* ctx, void* device_ptr, size_t n, bp_g1vec** out);
/* Window-multiples table of a resident vector, OPT-IN (round 3): rows 2^(c w) P_i for every window w are built once
 * (one launch: c doublings per window per point, one inversion per point) and kept with the vector until it is freed or
 * bp_g1vec_drop_table.  From then on every MSM over the WHOLE vector (bp_msm_g1, _begin, _pair, and the IPP / R1CS calls that
 * take it as G or H) sorts on the bucket alone: all windows of a scalar share one set of 2^(c-1) buckets, the bucket reduce
 * runs over one window instead of ~16 and the host tail shrinks from ~255 doublings to c.  For the generators of a proof
 * system -- public parameters reused by every proof (/root/reference src/r1cs/prover.rs:347-362, src/ipp.rs:91,104,158,170).
 * Results are bit-identical with and without a table.  window_bits: 2..16, 0 = chosen from n; memory = ceil((fr_bits+1)/c) x the
 * vector.  The caller must not modify the vector's points afterwards (bp_g1vec_device_ptr writers): the table would be stale.
 * Round 4: when window_bits divides 64 (16, 8, 4, 2) the call also builds the vector's COMPACTION TABLE -- the affine digit
 * multiples m 2^(64 k) P_i, m = 1 .. 8, k < 4 (32 more rows per point) -- which lets the inner-product prover materialise its folded
 * generators with a Horner chain of 60 doublings instead of 252 (BP_TUNE_COMPACT_AT).  ctx must be on the vector's device
 * (BP_ERR_ARG otherwise); the call returns when the tables are complete, so any context may use them afterwards. */
int bp_g1vec_precompute(bp_ctx* ctx, bp_g1vec* v, int window_bits);
int bp_g1vec_drop_table(bp_g1vec* v);
/* window width, number of windows and bytes of the vector's tables (window multiples + compaction table; all 0 when it has none) */
int bp_g1vec_table_info(const bp_g1vec* v, int* window_bits, int* windows, size_t* bytes);
/* The side-by-side table [G | H] a context keeps for its verifiers (BP_TUNE_VERIFY_TABLES): generators per vector and bytes held (zeros when
 * there is none), and giving it back to the pool before the context goes (the next verification over tables rebuilds it). */
int bp_ctx_verify_table_info(const bp_ctx* ctx, size_t* n, size_t* bytes);
int bp_ctx_drop_verify_table(bp_ctx* ctx);
/* out[i] = k[i] * G.  Batched form of `&G1::generator() * &FieldElement` (src/utils/mod.rs:34); used to build
 * synthetic generator vectors on the device. */
int bp_g1vec_fixed_base_mul(bp_ctx* ctx, const bp_frvec* k, bp_g1vec** out);
/* Batched `commit_to_field_element(g, h, m, r)` = `g.binary_scalar_mul(h, m, r)` (src/r1cs/prover.rs:123,496-500):
 * out[i] = k1[i] * g + k2[i] * h for fixed points g, h (x || y little-endian) and one scalar pair per commitment. */
int bp_g1vec_commit_pairs(bp_ctx* ctx, const uint8_t* g_le, const uint8_t* h_le, const bp_frvec* k1, const bp_frvec* k2, bp_g1vec** out);
/* Hash to G1, batched: out[i] = `G1::from_msg_hash(message i)` (amcl_wrapper; the reference calls it at
 * src/utils/mod.rs:20 and for the `g`, `h` of every gadget test, e.g. src/r1cs/gadgets/bound_check.rs:202-203).
 * Message i = msgs[offsets[i] .. offsets[i+1]); offsets has n + 1 entries, offsets[0] == 0.
 * Map (restated from amcl's published `ECP::mapit`, see bp_hash.cuh): x = BE(SHAKE256(msg)[0..MODBYTES)) mod p,
 * try-and-increment on x, the even square root, multiplication by the cofactor. */
int bp_g1vec_from_msg_hash(bp_ctx* ctx, const uint8_t* msgs, const uint64_t* offsets, size_t n, bp_g1vec** out);
/* `get_generators(prefix, n)` (src/utils/mod.rs:16-23): out[i] = from_msg_hash(prefix || decimal(first + i)), i < n.
 * The reference counts from 1 (first = 1); `first` lets ranks generate disjoint index ranges.  The messages are built
 * on the device: nothing but the prefix crosses PCIe. */
int bp_get_generators(bp_ctx* ctx, const uint8_t* prefix, size_t prefix_len, uint64_t first, size_t n, bp_g1vec** out);
/* out[i] = k[i] * p[i]   (`&G1 * &FieldElement`, src/r1cs/prover.rs:358,423,550), batched. */
int bp_g1vec_scalar_mul(bp_ctx* ctx, const bp_g1vec* p, const bp_frvec* k, bp_g1vec** out);
/* Compressed points (the proof structs derive Serialize / Deserialize, src/ipp.rs:13, src/r1cs/proof.rs:24).
 * Wire form OF THIS BUILD, bp_g1_compressed_bytes() = 1 + MODBYTES bytes per point: tag 0x02 (y even) / 0x03 (y odd) || X
 * big-endian; tag 0x00 || zeros = identity.  (amcl's own compressed bytes cannot be checked in this pipeline, so the format
 * is not claimed to be amcl's; the uncompressed BP_FMT_AMCL is what the transcript uses.)
 * bp_g1vec_decompress recomputes y = sqrt(x^3 + b) on the device and returns BP_ERR_ARG for an x >= p, an x that is not an
 * abscissa of the curve, an unknown tag or a non-zero identity encoding. */
size_t bp_g1_compressed_bytes(int curve_id);
int bp_g1vec_compress(bp_ctx* ctx, const bp_g1vec* v, size_t offset, size_t n, uint8_t* out);
int bp_g1vec_decompress(bp_ctx* ctx, const uint8_t* in, size_t n, bp_g1vec** out);

/* ---- FieldElementVector ------------------------------------------------------------------------------------- */
/* Scalars must be canonical (< r): BP_ERR_ARG otherwise (the window recoding of the MSM assumes it; a value >= r would wrap
 * silently).  bp_frvec_wrap_device is unchecked. */
int bp_frvec_upload(bp_ctx* ctx, const uint8_t* scalars_le32, size_t n, bp_frvec** out);
int bp_frvec_alloc(bp_ctx* ctx, size_t n, bp_frvec** out); /* zeros */
int bp_frvec_download(bp_ctx* ctx, const bp_frvec* v, size_t offset, size_t n, uint8_t* out_le32);
/* dst[dst_off .. dst_off + n) = src[src_off .. src_off + n), device to device on the context's stream (concatenations such as
 * [a_L | a_R | blinding] for commit_to_field_element_vectors, src/r1cs/prover.rs:346-361, without a trip through the host). */
int bp_frvec_copy(bp_ctx* ctx, bp_frvec* dst, size_t dst_off, const bp_frvec* src, size_t src_off, size_t n);
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
/* Asynchronous form of bp_msm_g1: _begin queues the device pipeline (and the D2H of the window sums) on the context's
 * stream and returns immediately; _end waits and finishes on the host.  One MSM in flight per context; two contexts give
 * two MSMs in flight from one host thread, which hides the latency-bound bucket reduce of one behind the accumulate of
 * the other (+15-25 % throughput at n = 2^20).  BP_ERR_ARG if _end is called with nothing pending; the
 * vectors passed to _begin must stay alive and unmodified until _end returns. */
int bp_msm_g1_begin(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars);
int bp_msm_g1_end(bp_ctx* ctx, uint8_t* out_le);
/* Two scalar vectors over the SAME points in one pipeline pass (twice the windows, one set of launches, one D2H):
 * out1 = <scalars1, points>, out2 = <scalars2, points>.  The IPP prover's L and R of a round are such a pair
 * (src/ipp.rs:148-170), as are commitments that share the generator vector (src/r1cs/prover.rs:347-362). */
int bp_msm_g1_pair(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars1, const bp_frvec* scalars2, uint8_t* out1_le, uint8_t* out2_le);
/* Two-stage form used when the index range is sharded over several GPUs (one process per GPU).
 * Stage 1 (device): each rank runs the bucket pipeline on its own slice and leaves its partial sums ("records":
 * un-normalised XYZZ points with a power-of-two weight each, bp_msm_record_bytes() bytes, bp_msm_window_records(ctx, n) of them
 * including the header) in a caller-owned HBM buffer.  The caller all-gathers the N ranks' records over RCCL (point addition is not an RCCL
 * reduction op; N*W*192 B is latency-bound).  Stage 2 (bp_msm_g1_finish): one D2H copy of `sets`
 * record sets, per-window sum, the serial 2^(c w) fold and the affine normalisation, giving BP_FMT_LE bytes.
 * A record block is W window records followed by ONE header record naming the geometry (c, W, widths) that produced it;
 * bp_msm_window_records counts both.  bp_msm_g1_finish recomputes the geometry from n_per_set (and the context's
 * bp_ctx_set_window_bits) and returns BP_ERR_ARG if any set's header disagrees -- ranks whose shard sizes differ (index
 * ranges differ by one) must therefore fix a common width first: bp_ctx_set_window_bits(ctx, c) with the c of
 * bp_msm_geometry(curve, largest shard, 0, ...), and pass that shard size as n_per_set.
 * bp_msm_g1_windows returns after the records are complete in device_out (it synchronises the context's stream), so they
 * can be handed to a collective on any other stream; the caller must in turn complete the gather before bp_msm_g1_finish. */
size_t bp_msm_window_records(bp_ctx* ctx, size_t n);
size_t bp_msm_record_bytes(int curve_id);
int bp_msm_g1_windows(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n,
                      void* device_out);
int bp_msm_g1_finish(bp_ctx* ctx, const void* device_records, size_t sets, size_t n_per_set, uint8_t* out_le);
/* ---- 2-D sharding (round 4): index range x WINDOW GROUP.  The index-range split leaves every shard with all W windows, i.e. with the
 * full bucket reduce, sort and host tail however short its slice is; a shard may instead take the windows [w_first, w_first + w_count)
 * of the same recoding over a LONGER slice (N = 8 as 2 x 4: half of the points, 4 of 16 windows each).  Its block holds
 * block_records records: the group's tail records (bp_msm_window_records_subset - 1 of them), zero padding, and the geometry header --
 * which names the window group -- as the LAST record.  The context's window width must be fixed (bp_ctx_set_window_bits) when the
 * slices differ in length.  bp_msm_g1_finish_blocks folds any mix of blocks: every index range must be covered by groups that
 * together hold all W windows exactly once (checked as far as the blocks can tell: every window must occur in the same number of
 * blocks); headers that do not fit the caller's geometry, window groups that do not tile the windows: BP_ERR_ARG.  /root/reference call site: multi_scalar_mul_var_time, src/ipp.rs:251-253. */
size_t bp_msm_window_records_subset(bp_ctx* ctx, size_t n, int w_first, int w_count);
int bp_msm_g1_windows_subset(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n, int w_first, int w_count,
                             size_t block_records, void* device_out);
int bp_msm_g1_finish_blocks(bp_ctx* ctx, const void* device_records, size_t n_blocks, size_t block_records, size_t n_per_set, uint8_t* out_le);
int bp_msm_g1_finish_blocks_host(int curve_id, const void* host_records, size_t n_blocks, size_t block_records, size_t n_per_set, int window_bits, uint8_t* out_le);
int bp_msm_record_positions_subset(int curve_id, size_t n, int window_bits, int w_first, int w_count, int* nrec_out, uint16_t* pos_out);
int bp_msm_record_header_subset(int curve_id, size_t n, int window_bits, int w_first, int w_count, void* record_out);
/* Stage 2 on HOST memory, no GPU needed (an aggregator that only receives record blocks): same validation and fold.
 * window_bits: 0 = the width chosen from n_per_set, else the common width the ranks fixed. */
int bp_msm_g1_finish_host(int curve_id, const void* host_records, size_t sets, size_t n_per_set, int window_bits, uint8_t* out_le);
/* The window geometry an MSM of n terms uses (host arithmetic): width c, W windows, per-window widths cw[W] and bit offsets
 * off[W], and the recoding bias (window w of k + bias, minus 2^(cw-1) - 1, is the signed digit of window w).  Any output
 * pointer may be NULL. */
int bp_msm_geometry(int curve_id, size_t n, int window_bits, int* c_out, int* W_out, uint8_t* cw_out, uint16_t* off_out, uint8_t* bias_le32);
/* A record block holds nrec records and its header; record r carries weight 2^pos[r]:  result = sum_r 2^pos[r] * record[r].
 * Since round 3 a window contributes one record for its plain weighted sum plus one per bit of the reduce-thread index (bit planes:
 * 13 per window at n = 2^20); the single-launch path for n <= 1536 -- taken only when no window width is fixed -- has one per window (two above 512 terms).
 * nrec = bp_msm_window_records() - 1; pos_out needs room for 4096 entries. */
int bp_msm_record_positions(int curve_id, size_t n, int window_bits, int* nrec_out, uint16_t* pos_out);
/* Record-block helpers for hosts that build or check blocks themselves (tests, aggregators): an affine point as a window
 * record, and the header record of the geometry above.  Host arithmetic. */
int bp_msm_record_from_affine(int curve_id, const uint8_t* point_le, void* record_out);
int bp_msm_record_header(int curve_id, size_t n, int window_bits, void* record_out);
/* The whole sharded MSM from ONE host thread without torch / RCCL (a Rust caller of G1Vector::multi_scalar_mul_var_time,
 * src/ipp.rs:251-253, has neither): shard i = (points[i], scalars[i]) is resident with ctxs[i] -- contexts on different
 * devices of the node, or several on one device.  All shards run concurrently with one common window width; each device
 * copies its W window sums (W x 192 B) to pinned host memory and the caller's thread folds the n_shards sets.  That gather
 * IS the "reduce" of the partial sums: n_shards x ~40 KiB of tail records, latency-bound; point addition is not an RCCL op
 * (SURVEY F9).  NOTE (SURVEY 8b asked for an in-library RCCL communicator): this entry point deliberately stages the records through
 * PINNED HOST MEMORY -- one D2H copy per device -- because the fold that consumes them runs on the host anyway; the RCCL path
 * (all_gather of the same record blocks in HBM, bp_msm_g1_windows + bp_msm_g1_finish) is what a one-process-per-GPU host uses
 * (bench.py, bulletproofs-amcl_amd/sharding.py). */
int bp_msm_g1_multi(bp_ctx* const* ctxs, const bp_g1vec* const* points, const bp_frvec* const* scalars, size_t n_shards, uint8_t* out_le);

/* Timing of the last bp_msm_* call on this context, measured with HIP events on the context's stream.
 * ms[0] = whole device pipeline, ms[1..] = per stage (digits+count, scan, scatter, tasks, accumulate, reduce); returns the
 * number of entries written (<= cap). */
int bp_msm_last_timing(bp_ctx* ctx, float* ms, int cap);
int bp_ctx_enable_timing(bp_ctx* ctx, int on);

/* ---- FieldElementVector kernels --------------------------------------------------------- */
/* FieldElementVector::inner_product (src/ipp.rs:77,78,145,146; src/utils/vector_poly.rs:44-50,82-87):
 * out = sum_i a[aoff+i] * b[boff+i] mod r, 32-byte LE.  BP_ERR_LENGTH if a range overruns its vector. */
int bp_fr_inner_product(bp_ctx* ctx, const bp_frvec* a, size_t aoff, const bp_frvec* b, size_t boff, size_t n, uint8_t* out_le32);
/* FieldElementVector::hadamard_product (src/ipp.rs:81,82,94,95): BP_ERR_LENGTH on unequal lengths. */
int bp_fr_hadamard(bp_ctx* ctx, const bp_frvec* a, const bp_frvec* b, bp_frvec** out);
/* FieldElementVector::scaled_by (src/r1cs/verifier.rs:416). */
int bp_fr_scaled_by(bp_ctx* ctx, const bp_frvec* a, const uint8_t* s_le32, bp_frvec** out);
/* out[i] = the two 128-bit halves of a[i] under the curve's GLV endomorphism, as the inner-product prover's kernels split their scalars
 * (a[i] = s1 + s2 LAMBDA mod r, phi(x, y) = (BETA x, y) = LAMBDA (x, y)): 16 bytes s1, 16 bytes s2, little-endian.  BLS12-381: LAMBDA = z^2 - 1,
 * both halves plain numbers; BN254: LAMBDA = 36u^4 - 1 mod r (190 bits), s2 is a number mod 2^128 that is NEGATIVE when >= 2^66.  Exposed so
 * that the split can be checked on its own (tests/test_gpu_ipp.py); no reference counterpart (amcl_wrapper multiplies without it). */
int bp_fr_glv_split(bp_ctx* ctx, const bp_frvec* a, bp_frvec** out);
/* FieldElementVector::new_vandermonde_vector(e, n) = [1, e, e^2, ...] (src/ipp.rs:348; src/r1cs/prover.rs:463). */
int bp_fr_vandermonde(bp_ctx* ctx, const uint8_t* e_le32, size_t n, bp_frvec** out);
/* VecPoly3::special_inner_product (src/utils/vector_poly.rs:79-97): lhs = (0, l1, l2, l3), rhs = (r0, r1, 0, r3);
 * writes t1..t6 (6 x 32-byte LE).  lhs[0] and rhs[2] are not read.  One fused pass instead of nine inner products. */
int bp_vecpoly3_special_inner_product(bp_ctx* ctx, const bp_frvec* const lhs[4], const bp_frvec* const rhs[4], uint8_t* out_t1_to_t6);
/* VecPoly1::inner_product (src/utils/vector_poly.rs:36-53): Poly2 (t0, t1, t2), 3 x 32-byte LE. */
int bp_vecpoly1_inner_product(bp_ctx* ctx, const bp_frvec* const l[2], const bp_frvec* const r[2], uint8_t* out_t0_t1_t2);
/* VecPoly1::eval / VecPoly3::eval (src/utils/vector_poly.rs:55-62, 99-106): out[i] = sum_d p[d][i] x^d, degree 1 or 3. */
int bp_vecpoly_eval(bp_ctx* ctx, const bp_frvec* const* p, int degree, const uint8_t* x_le32, bp_frvec** out);
/* FieldElement::inverse (src/ipp.rs:113,179); host arithmetic, no GPU needed; inverse of 0 is 0. */
int bp_fr_inverse(int curve_id, const uint8_t* in_le32, uint8_t* out_le32);
/* FieldElement::random() (src/r1cs/verifier.rs:392; the provers' blindings, prover.rs:337-341): n uniform non-zero scalars
 * from the operating system's generator (getrandom).  Host only. */
int bp_fr_random(int curve_id, uint8_t* out_le32, size_t n);
/* 1 if 0 < x < r (canonical, non-zero), else 0. */
int bp_fr_is_canonical_nonzero(int curve_id, const uint8_t* x_le32);

/* ---- transcript: merlin::Transcript + TranscriptProtocol (src/transcript.rs:12-61); host only -------------------- */
typedef struct bp_transcript bp_transcript;
int bp_transcript_new(const uint8_t* label, size_t label_len, bp_transcript** out);            /* Transcript::new(label) */
int bp_transcript_free(bp_transcript* t);
int bp_transcript_append_message(bp_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len);
int bp_transcript_append_u64(bp_transcript* t, const uint8_t* label, size_t label_len, uint64_t x);
int bp_transcript_challenge_bytes(bp_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len);
/* commit_point: append_message(label, G1::to_bytes())          (src/transcript.rs:51-53); point given as BP_FMT_LE */
int bp_transcript_commit_point(bp_transcript* t, int curve_id, const char* label, const uint8_t* point_le);
/* n commit_point calls with the same label (the "V" commitments of Prover::commit / Verifier::commit, prover.rs:118-127): one call. */
int bp_transcript_commit_points(bp_transcript* t, int curve_id, const char* label, const uint8_t* points_le, size_t n);
/* commit_scalar: append_message(label, FieldElement::to_bytes()) (src/transcript.rs:47-49) */
int bp_transcript_commit_scalar(bp_transcript* t, int curve_id, const char* label, const uint8_t* scalar_le32);
/* challenge_scalar: MODBYTES challenge bytes -> FieldElement::from (src/transcript.rs:55-60) */
int bp_transcript_challenge_scalar(bp_transcript* t, int curve_id, const char* label, uint8_t* out_le32);

/* ---- inner-product argument ------------------------------------------------------------ */
/* Device-resident prover state: working copies of G, H, a, b (src/ipp.rs:57-60) stay in HBM for all lg n rounds;
 * per round only L, R go to the host (for the transcript) and u, u^-1 come back.  For a host that owns its own
 * transcript (the Rust crate):  state_create; while len > 1 { round -> L,R; [host: commit, challenge]; fold(u, u^-1) };
 * state_finish -> a, b.
 * state_create enforces create_ipp's assertions (src/ipp.rs:48-55): n a power of two, all six lengths equal ->
 * BP_ERR_ARG otherwise. */
typedef struct bp_ipp_state bp_ipp_state;
/* Prover strategy for states created afterwards on this context.  0 (default): the generators are never folded --
 * each round's L and R are MSMs over the original resident [G | H | Q] with per-generator coefficient vectors (four Fr
 * multiplications per generator per round).  1: the reference's shape -- G and H are folded in place every round by a
 * batched G1::binary_scalar_mul kernel (src/ipp.rs:119,125,185,187), a 255-step serial chain per element.  Both give
 * bit-identical L, R, a, b.
 * Mode 0 keeps per-state tables beside [G | H | Q]: for 16 <= n <= 4096 the multiples 1P .. 8P of the 2n + 1 points (8 x 192 B per
 * point, one launch inside state_create) so that a round is ONE kernel launch; above that, when G and H carry window tables
 * (bp_g1vec_precompute), their concatenation.  Freed with the state. */
int bp_ctx_set_ipp_fold_generators(bp_ctx* ctx, int on);
int bp_ipp_state_create(bp_ctx* ctx, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* Q_le, const bp_frvec* G_factors,
                        const bp_frvec* H_factors, const bp_frvec* a, const bp_frvec* b, bp_ipp_state** out);
size_t bp_ipp_state_len(const bp_ipp_state* st);
/* L = <a_L(.Gf_R), G_R> + <b_R(.Hf_L), H_L> + c_L Q ;  R = <a_R(.Gf_L), G_L> + <b_L(.Hf_R), H_R> + c_R Q
 * (src/ipp.rs:77-104 first round, :145-170 later rounds), BP_FMT_LE. */
int bp_ipp_round(bp_ipp_state* st, uint8_t* L_le, uint8_t* R_le);
/* The fold of src/ipp.rs:115-130 / :181-188 (scalar fold + G1::binary_scalar_mul per element); halves the length. */
int bp_ipp_fold(bp_ipp_state* st, const uint8_t* u_le32, const uint8_t* u_inv_le32);
int bp_ipp_state_finish(bp_ipp_state* st, uint8_t* a_le32, uint8_t* b_le32); /* needs len == 1 */
int bp_ipp_state_free(bp_ipp_state* st);

/* IPP::create_ipp (src/ipp.rs:35-202) with this library's transcript: L_out / R_out receive lg n BP_FMT_LE points
 * (InnerProductArgumentProof{L, R, a, b}, src/ipp.rs:13-20). */
int bp_ipp_create(bp_ctx* ctx, bp_transcript* t, const uint8_t* Q_le, const bp_frvec* G_factors, const bp_frvec* H_factors, const bp_g1vec* G,
                  const bp_g1vec* H, const bp_frvec* a, const bp_frvec* b, uint8_t* L_out, uint8_t* R_out, size_t* lg_n_out,
                  uint8_t* a_out_le32, uint8_t* b_out_le32);
/* IPP::create_ipp with the generators SHARDED BY INDEX RANGE over several contexts / devices: shard i = (G[i], H[i],
 * G_factors[i], H_factors[i]) is resident with ctxs[i] and covers the next G[i]->n indices; the sizes add up to n (a power of two).
 * a, b arrive as host scalars (n x 32 bytes, canonical) and are replicated.  Every round each shard runs its slice of the L / R
 * MSMs with one common window width; the record sets are gathered through pinned host memory and folded on the calling thread
 * (as bp_msm_g1_multi).  Same proof bytes as bp_ipp_create.  Pays only for n >= 2^18 or with window tables on every shard's G, H
 * (DESIGN.md section 6); exists so that a generator set too large for one device, or already distributed, needs no gathering. */
int bp_ipp_create_multi(bp_ctx* const* ctxs, size_t n_shards, bp_transcript* t, const uint8_t* Q_le, const bp_frvec* const* G_factors,
                        const bp_frvec* const* H_factors, const bp_g1vec* const* G, const bp_g1vec* const* H, const uint8_t* a_le32,
                        const uint8_t* b_le32, size_t n, uint8_t* L_out, uint8_t* R_out, size_t* lg_n_out, uint8_t* a_out_le32,
                        uint8_t* b_out_le32);
/* IPP::verify_ipp (src/ipp.rs:204-260): BP_OK, or BP_ERR_VERIFY (R1CSError::VerificationError) when the single
 * (1 + 2n + 2 lg n)-term MSM differs from P or when lg_n >= 32 / n != 2^lg_n (src/ipp.rs:269-276). */
int bp_ipp_verify(bp_ctx* ctx, bp_transcript* t, size_t n, const bp_frvec* G_factors, const bp_frvec* H_factors, const uint8_t* P_le,
                  const uint8_t* Q_le, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* a_le32, const uint8_t* b_le32, const uint8_t* L_le,
                  const uint8_t* R_le, size_t lg_n);
/* Batch verification: m proofs of the same length n over the same generators and factors are accepted
 * together iff  sum_j w_j * (G^(a_j s_j) H^(b_j / s_j) Q_j^(a_j b_j) - P_j - sum_k (u_jk^2 L_jk + u_jk^-2 R_jk)) == O,
 * i.e. ONE MSM of 2n + m (2 lg n + 2) terms instead of m MSMs of 2n + 2 lg n + 1 -- the random-linear-combination
 * argument the reference's verifier already uses inside one proof (src/r1cs/verifier.rs:392) applied across proofs.
 * weights_le32: NULL (the library draws m fresh scalars with bp_fr_random), or m canonical NON-ZERO scalars chosen by the
 * caller AFTER the proofs are fixed, from a source the provers cannot predict (BP_ERR_ARG for a zero / non-canonical
 * weight); a forged proof then passes with probability ~ 1/r.  BP_ERR_VERIFY does
 * not say which proof failed: fall back to bp_ipp_verify per proof.  Each transcript is advanced exactly as by
 * bp_ipp_verify. */
typedef struct bp_ipp_proof_ref {
    bp_transcript* transcript;
    const uint8_t* P_le;      /* commitment, x || y little-endian */
    const uint8_t* Q_le;
    const uint8_t* a_le32;
    const uint8_t* b_le32;
    const uint8_t* L_le;      /* lg_n points */
    const uint8_t* R_le;
} bp_ipp_proof_ref;
int bp_ipp_verify_batch(bp_ctx* ctx, size_t n, size_t lg_n, const bp_frvec* G_factors, const bp_frvec* H_factors, const bp_g1vec* G,
                        const bp_g1vec* H, const bp_ipp_proof_ref* proofs, size_t m, const uint8_t* weights_le32);
/* IPP::verification_scalars (src/ipp.rs:262-315): u_sq / u_inv_sq (lg_n scalars each) and s (n scalars), 32-byte LE.
 * Host arithmetic (used by the R1CS verifier, src/r1cs/verifier.rs:354); no GPU needed. */
int bp_ipp_verification_scalars(int curve_id, bp_transcript* t, const uint8_t* L_le, const uint8_t* R_le, size_t lg_n, size_t n, uint8_t* u_sq,
                                uint8_t* u_inv_sq, uint8_t* s);

/* ---- R1CS vector pipeline either side of the IPP ---------------------------------------------- */
/* flattened_constraints (src/r1cs/prover.rs:142-184; src/r1cs/verifier.rs:149-193): wL, wR, wO (length n), wV (length
 * m) and the constant wc = sums of z^(q+1) * coeff over the terms of constraint q (wV and wc subtracted).  The constraint
 * system is fixed per circuit, so its terms are regrouped once into a plan; each proof evaluates the plan for its z.
 * Term t: constraint term_constraint[t] (0-based, any order), variable kind term_kind[t], index term_index[t] (< n for
 * the multiplier kinds, < m for BP_VAR_COMMITTED, ignored for BP_VAR_ONE), coefficient coeff_le32 + 32 t. */
enum { BP_VAR_MUL_LEFT = 0, BP_VAR_MUL_RIGHT = 1, BP_VAR_MUL_OUTPUT = 2, BP_VAR_COMMITTED = 3, BP_VAR_ONE = 4 };
typedef struct bp_r1cs_plan bp_r1cs_plan;
int bp_r1cs_plan_create(bp_ctx* ctx, size_t n_terms, const uint32_t* term_constraint, const uint8_t* term_kind, const uint32_t* term_index,
                        const uint8_t* coeff_le32, size_t n_constraints, size_t n, size_t m, bp_r1cs_plan** out);
int bp_r1cs_plan_free(bp_r1cs_plan* plan);
/* out = {wL, wR, wO, wV} (new vectors); wc_le32 may be NULL (the prover does not need the constant, prover.rs:176-178). */
int bp_r1cs_flattened_constraints(bp_ctx* ctx, const bp_r1cs_plan* plan, const uint8_t* z_le32, bp_frvec* out[4], uint8_t* wc_le32);
/* The two callers of everything above, as host orchestration inside the library (bp_capi_r1cs.hip, written against this
 * header): `Prover::prove` (src/r1cs/prover.rs:323-560) and `Verifier::verify` (src/r1cs/verifier.rs:265-452) for
 * single-phase constraint systems (no randomised second phase: A_I2 = A_O2 = S2 = identity).  The transcript must already
 * hold what `Prover::new` / `commit` put there (r1cs_domain_sep, one commit_point("V") per committed value).
 * Proof layout (bp_r1cs_proof_bytes): A_I1 A_O1 S1 A_I2 A_O2 S2 T_1 T_3 T_4 T_5 T_6 (points, x || y LE) | t_x t_x_blinding
 * e_blinding | L[lg] R[lg] | a b, lg = log2 of the padded gate count.
 * prove: a_L, a_R, a_O, s_L, s_R of length n (gates), v_blinding of length m (may be NULL when m = 0); blindings_le32 = eight
 * scalars i, o, s, t1, t3, t4, t5, t6 (the reference draws them from its RNG, prover.rs:337-341,490-494).  G, H need at
 * least padded-n points (BP_ERR_LENGTH = R1CSError::InvalidGeneratorsLength).
 * verify: r_le32 = the verifier's random weight (verifier.rs:392): pass NULL and the library draws it (bp_fr_random), as the
 * reference does; an explicit value is for reproducible tests and must be canonical and non-zero (BP_ERR_ARG: r = 0 would
 * drop the t(x) / constraint check from the combined MSM).  BP_ERR_VERIFY if the combined MSM is not the identity or a proof
 * point is not a point of the curve. */
size_t bp_r1cs_proof_bytes(int curve_id, size_t n);
/* The same proof with every point compressed (11 + 2 lg points at 1 + MODBYTES bytes, the five scalars unchanged):
 * 2 267 instead of 4 288 bytes for BLS12-381 at 2^16 gates.  decompress returns BP_ERR_VERIFY for a point that does not decode. */
size_t bp_r1cs_proof_compressed_bytes(int curve_id, size_t n);
int bp_r1cs_proof_compress(bp_ctx* ctx, size_t n, const uint8_t* proof, size_t proof_len, uint8_t* out, size_t out_cap);
int bp_r1cs_proof_decompress(bp_ctx* ctx, size_t n, const uint8_t* in, size_t in_len, uint8_t* proof_out, size_t proof_cap);
int bp_r1cs_prove(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                  const uint8_t* h_le, const bp_frvec* a_L, const bp_frvec* a_R, const bp_frvec* a_O, const bp_frvec* v_blinding, const bp_frvec* s_L,
                  const bp_frvec* s_R, const uint8_t* blindings_le32, uint8_t* proof_out, size_t proof_cap);
int bp_r1cs_verify(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                   const uint8_t* h_le, const uint8_t* V_le, size_t n, size_t m, const uint8_t* proof, size_t proof_len, const uint8_t* r_le32);
/* Two-phase (randomised) constraint systems: Prover::prove / Verifier::verify split at create_randomized_constraints
 * (src/r1cs/prover.rs:298-319, verifier.rs:245-263), where the reference runs the callbacks registered with
 * specify_randomized_constraints (constraint_system.rs:77-99).  The caller plays that role between the two calls:
 *   bp_r1cs_prove_begin    appends "m", commits A_I1 / A_O1 / S1 of the n1 first-phase multipliers (vectors may be NULL when
 *                          n1 = 0) and the "r1cs-2phase" domain separator; blindings3 = i_blinding1, o_blinding1, s_blinding1.
 *                          phase1_out (bp_r1cs_phase1_bytes() bytes, caller memory) carries what _finish needs.
 *   ... the callbacks: RandomizedConstraintSystem::challenge_scalar(label) = bp_transcript_challenge_scalar(t, curve, label, ..);
 *       allocate second-phase multipliers and constraints; build the plan of the COMPLETE system (bp_r1cs_plan_create) ...
 *   bp_r1cs_prove_finish   a_L, a_R, a_O, s_L, s_R now hold all n = n1 + n2 entries (the first n1 as given to _begin); commits
 *                          A_I2 / A_O2 / S2 over G[n1..n), H[n1..n) (prover.rs:385-434) and finishes the proof.
 *                          blindings8 = i_blinding2, o_blinding2, s_blinding2, t_1, t_3, t_4, t_5, t_6 blindings.
 *   bp_r1cs_verify_begin   transcript only: "m", A_I1, A_O1, S1, the domain separator (verifier.rs:276-287, 253).
 *   bp_r1cs_verify_finish  the rest of Verifier::verify for n1 first-phase multipliers out of n; r_le32 as in bp_r1cs_verify.
 * The single-phase bp_r1cs_prove / bp_r1cs_verify are these with n2 = 0 and the "r1cs-1phase" separator. */
size_t bp_r1cs_phase1_bytes(void);
int bp_r1cs_prove_begin(bp_ctx* ctx, bp_transcript* t, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* h_le, size_t m, const bp_frvec* a_L1,
                        const bp_frvec* a_R1, const bp_frvec* a_O1, const bp_frvec* s_L1, const bp_frvec* s_R1, const uint8_t* blindings3_le32,
                        uint8_t* phase1_out, size_t phase1_cap);
int bp_r1cs_prove_finish(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                         const uint8_t* h_le, const uint8_t* phase1, const bp_frvec* a_L, const bp_frvec* a_R, const bp_frvec* a_O,
                         const bp_frvec* v_blinding, const bp_frvec* s_L, const bp_frvec* s_R, const uint8_t* blindings8_le32, uint8_t* proof_out,
                         size_t proof_cap);
int bp_r1cs_verify_begin(bp_transcript* t, int curve_id, size_t m, const uint8_t* proof, size_t proof_len);
int bp_r1cs_verify_finish(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                          const uint8_t* h_le, const uint8_t* V_le, size_t n1, size_t n, size_t m, const uint8_t* proof, size_t proof_len,
                          const uint8_t* r_le32);
/* Prover, src/r1cs/prover.rs:465-486.  in = {a_L, a_R, a_O, s_L, s_R, wL, wR, wO} (equal lengths n; wL.. are the
 * flattened constraints, computed on the host); out = {l1, l2, l3, r0, r1, r3}: the non-zero coefficient vectors of
 * l(X) = l1 X + l2 X^2 + l3 X^3 and r(X) = r0 + r1 X + r3 X^3 (feed bp_vecpoly3_special_inner_product / bp_vecpoly_eval). */
int bp_r1cs_prover_polys(bp_ctx* ctx, const bp_frvec* const in[8], const uint8_t* y_le32, bp_frvec* out[6]);
/* Prover, src/r1cs/prover.rs:526-563.  out = {l_vec, r_vec, G_factors, H_factors}, all of length padded_n:
 * l_vec = l(x) | 0,  r_vec = r(x) | -y^i,  G_factors = 1 (i < n1) | u,  H_factors = y^-i * G_factors. */
int bp_r1cs_ipp_inputs(bp_ctx* ctx, const bp_frvec* l_eval, const bp_frvec* r_eval, const uint8_t* y_le32, const uint8_t* u_le32, size_t n1,
                       size_t padded_n, bp_frvec* out[4]);
/* Verifier, src/r1cs/verifier.rs:342-390.  Replays IPP::verification_scalars on the transcript (L, R of the proof),
 * writes u_j^2 / u_j^-2 (lg_n x 32 B each) and returns the G and H scalars of the single verification MSM:
 * g_i = u_or_1 (x y^-i wR_i - a s_i),  h_i = u_or_1 (y^-i (x wL_i + wO_i - b s_(N-1-i)) - 1),  wL/wR/wO (length n) = 0
 * beyond n.  BP_ERR_VERIFY if lg_n >= 32 or padded_n != 2^lg_n. */
int bp_r1cs_verifier_scalars(bp_ctx* ctx, bp_transcript* t, const uint8_t* L_le, const uint8_t* R_le, size_t lg_n, size_t padded_n, size_t n1,
                             const bp_frvec* wL, const bp_frvec* wR, const bp_frvec* wO, const uint8_t* y_inv_le32, const uint8_t* x_le32,
                             const uint8_t* u_le32, const uint8_t* a_le32, const uint8_t* b_le32, uint8_t* u_sq_out, uint8_t* u_inv_sq_out,
                             bp_frvec** g_scalars, bp_frvec** h_scalars);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BPMSM_H */
