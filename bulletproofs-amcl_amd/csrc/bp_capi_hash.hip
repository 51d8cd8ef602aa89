// bp_capi_hash.hip -- C ABI for hash-to-G1 (include/bpmsm.h: bp_g1vec_from_msg_hash, bp_get_generators).
// Kernels: bp_hash.cuh.  Reference call sites: src/utils/mod.rs:16-23 (get_generators) and G1::from_msg_hash in the
// gadget tests (e.g. src/r1cs/gadgets/bound_check.rs:200-203).
#include <vector>

#include "bp_internal.hpp"
#include "bp_hash.cuh"

using namespace bp;

namespace {

template <class C> SqrtExp sqrt_exponent() {   // (p + 1) / 4 from the modulus words
    using W = typename C::Fp::Words;
    constexpr int NW = C::Fp::NW;
    uint32_t t[12] = {};
    uint64_t carry = 1;
    for (int i = 0; i < NW; i++) { uint64_t v = (uint64_t)W::MODW[i] + carry; t[i] = (uint32_t)v; carry = v >> 32; }
    SqrtExp e{};
    for (int i = 0; i < NW; i++) e.w[i] = (t[i] >> 2) | (i + 1 < NW ? t[i + 1] << 30 : (uint32_t)carry << 30);
    return e;
}

template <class C>
int launch_hash(bp_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offs, uint32_t prefix_len, uint64_t first, size_t n,
                unsigned long long* d_next, void* out) {
    static_assert((C::Fp::Words::MODW[0] & 3) == 3, "sqrt by (p+1)/4 needs p = 3 mod 4");
    const SqrtExp e = sqrt_exponent<C>();
    // search: at most two resident blocks per CU's worth of lanes; each lane pulls messages until the counter passes n
    size_t want = (n + kHashBlock - 1) / kHashBlock;
    unsigned grid = (unsigned)(want < 512 ? want : 512);
    HIPCHK(hipMemsetAsync(d_next, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_hash_search<C>, dim3(grid), dim3(kHashBlock), 0, ctx->stream, d_bytes, d_offs, prefix_len, first, n, e, d_next,
                       (AffPacked<C>*)out);
    HIPCHK(hipGetLastError());
    BP_TRACE_SYNC(ctx, "k_hash_search<C>");
    if (!C::COFACTOR_IS_ONE) {
        hipLaunchKernelGGL(k_clear_cofactor<C>, dim3((unsigned)want), dim3(kHashBlock), 0, ctx->stream, n, e, (AffPacked<C>*)out);
        HIPCHK(hipGetLastError());
        BP_TRACE_SYNC(ctx, "k_clear_cofactor<C>");
    }
    return BP_OK;
}

// bytes (+ optional offsets) -> device scratch, launch, wait (the staging buffers are freed on return)
int hash_common(bp_ctx* ctx, const uint8_t* bytes, size_t nbytes, const uint64_t* offs, uint64_t first, size_t n, bp_g1vec** out) {
    int rc = bp_g1vec_alloc(ctx, n, out);
    if (rc) return rc;
    if (n == 0) return BP_OK;
    // staging: [next-message counter (8 B) | message bytes]; offsets separately (8-byte aligned)
    PoolBlock b_stage, b_offs;
    auto fail = [&](int code) { bp_g1vec_free(*out); *out = nullptr; return code; };
    if (!b_stage.alloc(ctx, 8 + (nbytes ? nbytes : 1))) return fail(BP_ERR_DEVICE);
    void *d_stage = b_stage.p, *d_offs = nullptr;
    uint8_t* d_bytes = (uint8_t*)d_stage + 8;
    if (nbytes && hipMemcpyAsync(d_bytes, bytes, nbytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(BP_ERR_DEVICE);
    if (offs) {
        if (!b_offs.alloc(ctx, (n + 1) * sizeof(uint64_t))) return fail(BP_ERR_DEVICE);
        d_offs = b_offs.p;
        if (hipMemcpyAsync(d_offs, offs, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(BP_ERR_DEVICE);
    }
    if (ctx->curve == BP_CURVE_BLS12_381) rc = launch_hash<Bls381>(ctx, d_bytes, (const uint64_t*)d_offs, (uint32_t)nbytes, first, n, (unsigned long long*)d_stage, (*out)->d);
    else rc = launch_hash<Bn254>(ctx, d_bytes, (const uint64_t*)d_offs, (uint32_t)nbytes, first, n, (unsigned long long*)d_stage, (*out)->d);
    if (rc == BP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = BP_ERR_DEVICE;
    if (rc) return fail(rc);
    return BP_OK;
}

template <class C>
int compress_impl(bp_ctx* ctx, const bp_g1vec* v, size_t offset, size_t n, uint8_t* out) {
    const size_t per = C::MODBYTES + 1;
    PoolBlock stage;
    if (!stage.alloc(ctx, n * per)) return BP_ERR_DEVICE;
    hipLaunchKernelGGL(k_g1_compress<C>, dim3((unsigned)((n + kHashBlock - 1) / kHashBlock)), dim3(kHashBlock), 0, ctx->stream,
                       (const AffPacked<C>*)v->d + offset, n, (uint8_t*)stage.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, stage.p, n * per, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BP_OK;
}

template <class C>
int decompress_impl(bp_ctx* ctx, const uint8_t* in, size_t n, bp_g1vec* out) {
    const size_t per = C::MODBYTES + 1;
    PoolBlock stage;
    if (!stage.alloc(ctx, n * per)) return BP_ERR_DEVICE;
    int rc;
    if ((rc = ctx->flags.reserve(ctx, 64))) return rc;
    uint32_t* flag = (uint32_t*)ctx->flags.p;
    uint32_t host_flag = 0;
    HIPCHK(hipMemsetAsync(flag, 0, 4, ctx->stream));
    HIPCHK(hipMemcpyAsync(stage.p, in, n * per, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_g1_decompress<C>, dim3((unsigned)((n + kHashBlock - 1) / kHashBlock)), dim3(kHashBlock), 0, ctx->stream, (const uint8_t*)stage.p, n,
                       sqrt_exponent<C>(), (AffPacked<C>*)out->d, flag);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&host_flag, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return host_flag ? BP_ERR_ARG : BP_OK;
}

}  // namespace

extern "C" {

size_t bp_g1_compressed_bytes(int curve_id) { return curve_id == BP_CURVE_BLS12_381 ? Bls381::MODBYTES + 1 : curve_id == BP_CURVE_BN254 ? Bn254::MODBYTES + 1 : 0; }

int bp_g1vec_compress(bp_ctx* ctx, const bp_g1vec* v, size_t offset, size_t n, uint8_t* out) {
    return bp_guard([&]() -> int {
    if (!ctx || !v || (!out && n)) return BP_ERR_ARG;
    if (offset > v->n || n > v->n - offset) return BP_ERR_LENGTH;
    if (n == 0) return BP_OK;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    return ctx->curve == BP_CURVE_BLS12_381 ? compress_impl<Bls381>(ctx, v, offset, n, out) : compress_impl<Bn254>(ctx, v, offset, n, out);
    });
}

int bp_g1vec_decompress(bp_ctx* ctx, const uint8_t* in, size_t n, bp_g1vec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out || (!in && n)) return BP_ERR_ARG;
    *out = nullptr;
    int rc = bp_g1vec_alloc(ctx, n, out);
    if (rc || n == 0) return rc;
    rc = ctx->curve == BP_CURVE_BLS12_381 ? decompress_impl<Bls381>(ctx, in, n, *out) : decompress_impl<Bn254>(ctx, in, n, *out);
    if (rc) { bp_g1vec_free(*out); *out = nullptr; }
    return rc;
    });
}

// R1CS proof <-> its compressed wire form: every point of the layout of bp_r1cs_prove as 1 + MODBYTES bytes, the five scalars
// unchanged.  11 + 2 lg points: 49 instead of 96 bytes each for BLS12-381 (4 288 -> 2 267 bytes at 2^16 gates).
static size_t r1cs_points(size_t n) { size_t p = 1, lg = 0; while (p < n) { p <<= 1; lg++; } return 11 + 2 * lg; }

size_t bp_r1cs_proof_compressed_bytes(int curve_id, size_t n) {
    if (!curve_ok(curve_id) || n == 0) return 0;
    return r1cs_points(n) * bp_g1_compressed_bytes(curve_id) + 5 * 32;
}

int bp_r1cs_proof_compress(bp_ctx* ctx, size_t n, const uint8_t* proof, size_t proof_len, uint8_t* out, size_t out_cap) {
    return bp_guard([&]() -> int {
    if (!ctx || !proof || !out || n == 0) return BP_ERR_ARG;
    const size_t pb = 2 * (size_t)fp_bytes_of(ctx->curve), np = r1cs_points(n), cb = bp_g1_compressed_bytes(ctx->curve);
    if (proof_len != np * pb + 5 * 32 || out_cap < np * cb + 5 * 32) return BP_ERR_LENGTH;
    // layout: 11 points | 3 scalars | 2 lg points | 2 scalars  ->  gather the points, compress, interleave again
    std::vector<uint8_t> pts(np * pb), comp(np * cb);
    memcpy(pts.data(), proof, 11 * pb);
    memcpy(pts.data() + 11 * pb, proof + 11 * pb + 96, (np - 11) * pb);
    bp_g1vec* v = nullptr;
    int rc = bp_g1vec_upload(ctx, pts.data(), np, BP_FMT_LE, &v);
    if (rc) return rc;
    rc = bp_g1vec_compress(ctx, v, 0, np, comp.data());
    bp_g1vec_free(v);
    if (rc) return rc;
    memcpy(out, comp.data(), 11 * cb);
    memcpy(out + 11 * cb, proof + 11 * pb, 96);
    memcpy(out + 11 * cb + 96, comp.data() + 11 * cb, (np - 11) * cb);
    memcpy(out + np * cb + 96, proof + np * pb + 96, 64);
    return BP_OK;
    });
}

int bp_r1cs_proof_decompress(bp_ctx* ctx, size_t n, const uint8_t* in, size_t in_len, uint8_t* proof_out, size_t proof_cap) {
    return bp_guard([&]() -> int {
    if (!ctx || !in || !proof_out || n == 0) return BP_ERR_ARG;
    const size_t pb = 2 * (size_t)fp_bytes_of(ctx->curve), np = r1cs_points(n), cb = bp_g1_compressed_bytes(ctx->curve);
    if (in_len != np * cb + 5 * 32 || proof_cap < np * pb + 5 * 32) return BP_ERR_LENGTH;
    std::vector<uint8_t> comp(np * cb), pts(np * pb);
    memcpy(comp.data(), in, 11 * cb);
    memcpy(comp.data() + 11 * cb, in + 11 * cb + 96, (np - 11) * cb);
    bp_g1vec* v = nullptr;
    int rc = bp_g1vec_decompress(ctx, comp.data(), np, &v);
    if (rc) return rc == BP_ERR_ARG ? BP_ERR_VERIFY : rc;      // a proof that does not decode is a proof that does not verify
    rc = bp_g1vec_download(ctx, v, 0, np, BP_FMT_LE, pts.data());
    bp_g1vec_free(v);
    if (rc) return rc;
    memcpy(proof_out, pts.data(), 11 * pb);
    memcpy(proof_out + 11 * pb, in + 11 * cb, 96);
    memcpy(proof_out + 11 * pb + 96, pts.data() + 11 * pb, (np - 11) * pb);
    memcpy(proof_out + np * pb + 96, in + np * cb + 96, 64);
    return BP_OK;
    });
}

int bp_g1vec_from_msg_hash(bp_ctx* ctx, const uint8_t* msgs, const uint64_t* offsets, size_t n, bp_g1vec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out || (n && !offsets)) return BP_ERR_ARG;
    *out = nullptr;
    size_t total = 0;
    if (n) {
        if (offsets[0] != 0) return BP_ERR_ARG;
        for (size_t i = 0; i < n; i++) {
            if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0x7fffffffull) return BP_ERR_ARG;
        }
        total = (size_t)offsets[n];
        if (total && !msgs) return BP_ERR_ARG;
    }
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    return hash_common(ctx, msgs, total, offsets, 0, n, out);
    });
}

int bp_get_generators(bp_ctx* ctx, const uint8_t* prefix, size_t prefix_len, uint64_t first, size_t n, bp_g1vec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out || (prefix_len && !prefix) || prefix_len > 0x7fffffffull) return BP_ERR_ARG;
    *out = nullptr;
    if (n && first + (n - 1) < first) return BP_ERR_ARG;   // the counter must not wrap
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    return hash_common(ctx, prefix, prefix_len, nullptr, first, n, out);
    });
}

}  // extern "C"
