// bp_internal.hpp -- private declarations shared by the translation units of libbpmsm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/bpmsm.h"
#include "bp_kernels.cuh"

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            if (getenv("BP_VERBOSE")) fprintf(stderr, "[bpmsm] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return BP_ERR_DEVICE;                                                                            \
        }                                                                                                    \
    } while (0)

// ------------------------------------------------------------------------------------------------ handles
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return BP_OK;
        if (p) { if (hipFree(p) != hipSuccess) return BP_ERR_DEVICE; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return BP_ERR_DEVICE; }
        cap = want;
        return BP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct bp_ctx {
    int curve = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int c_override = 0;
    bool timing = false;
    bool pending = false;               // bp_msm_g1_begin issued, bp_msm_g1_end not yet called
    size_t pending_n = 0;
    bool device_tail = false;           // fold the window sums on the device (one lane) instead of on the host
    bool ipp_fold_generators = false;   // IPP prover: fold G/H each round (reference shape) instead of MSMs over the originals
    // MSM workspace
    DevBuf fixed_base_table;            // d * 2^(4j) * G, built on first use (bp_g1vec_fixed_base_mul)
    bool fixed_base_ready = false;
    DevBuf count, cursor, block_sums, idx, code, tile_hist, tmp_code, tmp_idx, ntasks, task_off, order, t_start, t_len, tsum, heavy, heavy_chunks, meta, partial, window_sum, scratch;
    void* host_pinned = nullptr;
    size_t host_pinned_cap = 0;
    hipEvent_t ev[8] = {};
    bool ev_ready = false;
    float last_ms[8] = {};
    int last_ms_n = 0;
};

// Handles remember their device ordinal so that they can be freed after their context is gone
// (hipFree synchronises with outstanding work on the device by itself).
struct bp_g1vec {
    bp_ctx* ctx;
    void* d;
    size_t n;
    bool owned;
    int device;
};
struct bp_frvec {
    bp_ctx* ctx;
    void* d;
    size_t n;
    bool owned;
    int device;
};

static inline int fp_bytes_of(int curve) { return curve == BP_CURVE_BLS12_381 ? 48 : 32; }
static inline bool curve_ok(int curve) { return curve == BP_CURVE_BLS12_381 || curve == BP_CURVE_BN254; }

static int host_pinned_reserve(bp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->host_pinned_cap) return BP_OK;
    if (ctx->host_pinned) { HIPCHK(hipHostFree(ctx->host_pinned)); ctx->host_pinned = nullptr; ctx->host_pinned_cap = 0; }
    HIPCHK(hipHostMalloc(&ctx->host_pinned, bytes + 4096, hipHostMallocDefault));
    ctx->host_pinned_cap = bytes + 4096;
    return BP_OK;
}


// BP_TRACE=1: synchronise and log after every stage (debugging aid; off by default).
#define BP_TRACE_SYNC(ctx_, what)                                                                      \
    do {                                                                                               \
        static const bool on_ = getenv("BP_TRACE") != nullptr;                                         \
        if (on_) {                                                                                     \
            hipError_t te_ = hipStreamSynchronize((ctx_)->stream);                                     \
            fprintf(stderr, "[bpmsm trace] %s -> %s\n", what, hipGetErrorString(te_));                 \
            fflush(stderr);                                                                            \
        }                                                                                              \
    } while (0)

// MSM over raw resident device arrays (n > 0 or n == 0 -> identity); defined in bp_capi.hip.
int bp_internal_msm(bp_ctx* ctx, const void* points, const void* scalars, size_t n, uint8_t* out_le);
// nnz = non-zero scalars per set if known (0: assume n)
int bp_internal_msm2(bp_ctx* ctx, const void* points, const void* scalars1, const void* scalars2, size_t n, uint8_t* out1_le, uint8_t* out2_le,
                     size_t nnz);
int bp_internal_set_device(const bp_ctx* ctx);
