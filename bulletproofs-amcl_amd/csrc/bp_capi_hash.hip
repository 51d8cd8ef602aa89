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

}  // namespace

extern "C" {

int bp_g1vec_from_msg_hash(bp_ctx* ctx, const uint8_t* msgs, const uint64_t* offsets, size_t n, bp_g1vec** out) {
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
}

int bp_get_generators(bp_ctx* ctx, const uint8_t* prefix, size_t prefix_len, uint64_t first, size_t n, bp_g1vec** out) {
    if (!ctx || !out || (prefix_len && !prefix) || prefix_len > 0x7fffffffull) return BP_ERR_ARG;
    *out = nullptr;
    if (n && first + (n - 1) < first) return BP_ERR_ARG;   // the counter must not wrap
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    return hash_common(ctx, prefix, prefix_len, nullptr, first, n, out);
}

}  // extern "C"
