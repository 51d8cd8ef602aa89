// bp_internal.hpp -- private declarations shared by the translation units of libbpmsm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/bpmsm.h"
#include "bp_kernels.cuh"

// Environment variables: the library reads exactly two, both DIAGNOSTIC (they add stderr output / synchronisation, never change a
// result): BP_VERBOSE (name the failing HIP call) and BP_TRACE (synchronise and log after every pipeline stage).  Everything that
// can change what the pipeline does is a validated per-context knob (bp_ctx_set_tuning, include/bpmsm.h).
static inline bool bp_verbose_on() { static const bool on = getenv("BP_VERBOSE") != nullptr; return on; }
static inline bool bp_trace_on() { static const bool on = getenv("BP_TRACE") != nullptr; return on; }
// BP_PROFILE (diagnostic, like BP_TRACE): wall-clock accumulators around the host-side steps of a proof, printed to stderr by
// bp_ipp_create.  Slots: see bp_prof_names in bp_capi_ipp.hip.
static inline bool bp_profile_on() { static const bool on = getenv("BP_PROFILE") != nullptr; return on; }
#include <chrono>
struct BpProf {
    double acc[12] = {};
    std::chrono::steady_clock::time_point t0;
    void start() { if (bp_profile_on()) t0 = std::chrono::steady_clock::now(); }
    void lap(int slot) {
        if (!bp_profile_on()) return;
        auto t1 = std::chrono::steady_clock::now();
        acc[slot] += std::chrono::duration<double, std::micro>(t1 - t0).count();
        t0 = t1;
    }
};
inline BpProf& bp_prof() { static thread_local BpProf p; return p; }

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            if (bp_verbose_on()) fprintf(stderr, "[bpmsm] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return BP_ERR_DEVICE;                                                                            \
        }                                                                                                    \
    } while (0)

struct bp_ctx;

// ------------------------------------------------------------------------------------------------ handles
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    inline int reserve(bp_ctx* ctx, size_t bytes);   // defined after bp_ctx: a failed hipMalloc gives the pool's cached blocks back and retries once
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Per-context caching allocator for the vectors and temporaries of the proof path.  hipMalloc costs tens of microseconds and
// hipFree synchronises the whole device; a proof creates dozens of short-lived vectors (bp_frvec_alloc / bp_g1vec_alloc, the
// IPP state, the verifier's term arrays), which at n = 64 cost more than the kernels.  Blocks are recycled by size class and
// never returned to the driver before the pool dies.  Safe because every use of a context's memory is ordered on the context's
// stream: a block handed out again is only touched by work queued after the work that last used it (bp_ctx_set_stream
// synchronises when the stream changes).  The pool is reference-counted: handles may outlive their context.
struct DevPool {
    int device = 0;
    std::mutex mu;
    std::atomic<long> refs{1};
    std::map<size_t, std::vector<void*>> free_;   // capacity -> blocks
    size_t cached_bytes = 0;
    static size_t size_class(size_t bytes) {
        if (bytes < 256) return 256;
        if (bytes <= ((size_t)1 << 20)) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
        return (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);      // large vectors: whole MiB, exact reuse
    }
    void* get(size_t bytes, size_t* cap) {
        const size_t c = size_class(bytes);
        *cap = c;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_.find(c);
            if (it != free_.end() && !it->second.empty()) { void* p = it->second.back(); it->second.pop_back(); cached_bytes -= c; refs++; return p; }
        }
        void* p = nullptr;
        if (hipMalloc(&p, c) != hipSuccess) {
            (void)hipGetLastError();                                              // the failure must not surface at the next kernel launch's error check
            trim();                                                               // give cached blocks back and retry once
            if (hipMalloc(&p, c) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        refs++;
        return p;
    }
    void put(void* p, size_t cap) {
        if (!p) return;
        bool dead;
        {
            std::lock_guard<std::mutex> lk(mu);
            free_[cap].push_back(p);
            cached_bytes += cap;
            dead = --refs == 0;
        }
        if (dead) destroy();
    }
    void trim() {
        std::lock_guard<std::mutex> lk(mu);
        int prev = -1;
        (void)hipGetDevice(&prev);                                                // a late put() from a handle that outlived its context runs on
        (void)hipSetDevice(device);                                               // the caller's thread: leave its current device as it was
        for (auto& kv : free_) for (void* p : kv.second) (void)hipFree(p);
        free_.clear();
        cached_bytes = 0;
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    }
    void release() { if (--refs == 0) destroy(); }     // the context's own reference
    void destroy() { trim(); delete this; }
};

#include "bp_hostpool.hpp"

// Engineering knobs of the MSM pipeline (bp_ctx_set_tuning): 0 = automatic.  Every value is validated when it is set.
struct bp_tuning {
    uint32_t tile = 0;          // scalars per binning block: a multiple of 256 in [256, 16384]
    uint32_t reduce_m = 0;      // buckets per bucket-reduce thread: a power of two in [1, 16384]
    uint64_t task_target = 0;   // task count the accumulate aims at: [1024, 2^28]
    bool small_msm = true;      // single-launch path for n <= 512
    uint32_t compact_at = 0;    // inner-product prover: live length at which the folded generators are materialised (0 automatic, 1 never)
    bool glv = true;            // ... and whether that compaction (and the rounds after it) split the scalars with the GLV endomorphism (BLS12-381)
    uint32_t verify_tables = 0; // verifiers: generators per vector from which the [G | H] part of the check runs over the vectors' window tables (0 = never)
};

struct bp_ctx {
    int curve = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int c_override = 0;
    uint32_t ipp_n0 = 0, ipp_live = 0;   // set around the paired MSM of an inner-product round: generators per vector and the round's live length (IppSparse)
    int win_first = 0, win_count = 0;   // window group of the MSM being queued (bp_msm_g1_windows_subset sets and clears them); 0 = all windows
    bp_tuning tuning;
    bool timing = false;
    bool pending = false;               // bp_msm_g1_begin issued, bp_msm_g1_end not yet called
    size_t pending_n = 0;
    bool device_tail = false;           // fold the window sums on the device (one lane) instead of on the host
    bool ipp_fold_generators = false;   // IPP prover: fold G/H each round (reference shape) instead of MSMs over the originals
    // MSM workspace
    DevBuf fixed_base_table;            // d * 2^(4j) * G, built on first use (bp_g1vec_fixed_base_mul)
    bool fixed_base_ready = false;
    DevBuf count, cursor, block_sums, idx, code, tile_hist, tmp_idx, ntasks, task_off, order, t_start, t_len, tsum, heavy, heavy_chunks, meta, partial, window_sum, scratch, huge, negbits;
    void* host_pinned = nullptr;
    size_t host_pinned_cap = 0;
    void* stage = nullptr;              // page-locked ring for small host -> device copies of LIBRARY-made data that must not wait for the
    size_t stage_cap = 0, stage_cur = 0;   // stream (bp_internal_frvec_upload_trusted): a slice is reused only after a synchronisation
    hipEvent_t ev[8] = {};              // stage boundaries of the last MSM (created on first use, only when timing is on)
    bool ev_ready = false;
    float last_ms[8] = {};
    int last_ms_n = 0;
    DevPool* pool = nullptr;            // vectors and temporaries (see DevPool)
    HostWorker worker;
    HostPool tail_pool;                 // helper threads of the host tail (independent Horner chains)
    int tail_chains = 0;                // 0 = automatic (4 chains when a fold has >= 48 records), 1 = single chain
    bp_ctx* helper[2] = {nullptr, nullptr};   // lazily created sibling contexts (own stream + workspace) for independent MSMs in flight
    hipEvent_t ev_fork = nullptr;             // "everything queued on this context so far" for the siblings' streams
    DevBuf flags;                       // 64 B of device error flags (point / scalar validation)
    // geometry of the MSM queued by bp_msm_g1_begin (consumed by _end; bp_ctx_set_window_bits in between cannot disturb it)
    int pending_nrec = 0;
    uint16_t pending_rpos[bp::kMaxRecords] = {};
    // the verifiers' window table of [G[offG .. +n) | H[offH .. +n)] (bp_internal_gh_table): built from the two vectors' tables on the first
    // verification and kept until other generators (other table ids / ranges) are verified against or the context is destroyed
    struct bp_g1table* gh_table = nullptr;
    uint64_t gh_id[2] = {0, 0};
    size_t gh_off[2] = {0, 0}, gh_n = 0;
};

inline int DevBuf::reserve(bp_ctx* ctx, size_t bytes) {
    if (bytes <= cap) return BP_OK;
    if (p) { if (hipFree(p) != hipSuccess) return BP_ERR_DEVICE; p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx && ctx->pool) ctx->pool->trim();       // gigabytes of cached vector blocks must not make a larger workspace fail
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return BP_ERR_DEVICE; }
    }
    cap = want;
    return BP_OK;
}

// RAII block from a context's pool (temporaries inside one call)
struct PoolBlock {
    DevPool* pool = nullptr;
    void* p = nullptr;
    size_t cap = 0;
    PoolBlock() = default;
    PoolBlock(const PoolBlock&) = delete;
    PoolBlock& operator=(const PoolBlock&) = delete;
    bool alloc(bp_ctx* ctx, size_t bytes) { release(); pool = ctx->pool; p = pool->get(bytes ? bytes : 1, &cap); return p != nullptr; }
    void release() { if (p) pool->put(p, cap); p = nullptr; }
    ~PoolBlock() { release(); }
};

// Handles keep a reference on their context's pool, so they can be freed after the context is gone (the block is then
// released to the driver together with the pool).
struct bp_g1vec {
    bp_ctx* ctx;
    void* d;
    size_t n;
    bool owned;
    int device;
    DevPool* pool = nullptr;   // owned blocks come from (and return to) this pool
    size_t cap = 0;
    // bp_g1vec_device_ptr was called: the raw pointer may be in use on streams this library does not know (torch, RCCL, a view on
    // another context), so freeing the vector waits for the whole device first (what hipFree used to do) instead of relying on the
    // owner's stream order.  Vectors that never leave the library keep the synchronisation-free path.
    bool exported = false;
    struct bp_g1table* table = nullptr;   // window-multiples table built by bp_g1vec_precompute (owned; freed with the vector)
    // a view over points [tview_off, tview_off + n) of a vector that owns a table (library-internal views: the R1CS prover's
    // G[0..pn)): lets bp_internal_table_concat copy the rows it needs; never freed through the view
    const struct bp_g1table* tview = nullptr;
    size_t tview_off = 0;
    // compaction table (round 4; built by bp_g1vec_precompute when the window width divides 64): affine digit multiples
    // m 2^(64 k) P_i, m = 1 .. 8, k < 4 -- what the inner-product prover's generator compaction (bp_compact.cuh) needs to cut its Horner
    // chain from 252 to 60 doublings.  Owned like `table`; `cview` is the view form (same offset as tview_off).
    struct bp_g1table* ctable = nullptr;
    const struct bp_g1table* cview = nullptr;
};
struct bp_frvec {
    bp_ctx* ctx;
    void* d;
    size_t n;
    bool owned;
    int device;
    DevPool* pool = nullptr;
    size_t cap = 0;
    bool exported = false;     // see bp_g1vec
};

// Window-multiples table of a resident vector (bp_g1vec_precompute): rows[w * n + i] = 2^(c w) P_i, affine, packed.
struct bp_g1table {
    DevPool* pool = nullptr;
    int device = 0;
    void* d = nullptr;       // W * n packed affine rows (window 0 = a copy of the vector itself)
    size_t cap = 0;
    size_t n = 0;
    int c = 0, W = 0;
    // digits == true: NOT window multiples but the digit multiples of the single-launch small MSM (k_small_msm): rows[(m - 1) * n + i]
    // = m P_i for m = 1 .. 2^(c-1), packed lazy XYZZ (W = 2^(c-1) rows).  Built by the inner-product state for its [G | H | Q] when a
    // round is <= kSmallMsmMax terms: the lanes then load their digit's multiple instead of computing it (library-internal).
    bool digits = false;
    bool affine = false;     // digits only: the rows are canonical AFFINE points (AffPacked) instead of lazy XYZZ (bp_compact.cuh: batch conversion)
    bool glv = false;        // digits only: the table of a GLV-split set (bp_compact.cuh: k_glv_table_rows; rows[((m - 1) * 2 + half) * n + t], 16 multiples):
                             // only k_small_msm_glv reads it; the generic MSM paths ignore such a table
    int K = 1;               // digits only: sub-rows per point (compaction table: K = 4, rows[(m - 1) * K * n + k * n + i] = m 2^(64 k) P_i)
    uint64_t id = 0;         // window tables: unique per build (never reused), so that a cache keyed on it cannot mistake a rebuilt table for the old one
};
extern "C" void bp_internal_table_free(bp_g1table* t);
int bp_internal_digit_table_build(bp_ctx* ctx, const void* points, size_t n, bp_g1table** out);
// compaction table of a vector from its window-multiples table (bp_capi_ipp.hip); *out stays NULL when the width does not divide 64
int bp_internal_ctable_build(bp_ctx* ctx, const bp_g1table* wt, bp_g1table** out);

// No exception crosses the C ABI: every multi-line `int bp_*` entry point runs its body inside this guard -- a
// std::bad_alloc from a std::vector in the host orchestration (or anything else) becomes BP_ERR_DEVICE.
template <class F> static inline int bp_guard(F&& body) noexcept {
    try { return body(); } catch (...) { return BP_ERR_DEVICE; }
}

static inline int fp_bytes_of(int curve) { return curve == BP_CURVE_BLS12_381 ? 48 : 32; }
static inline bool curve_ok(int curve) { return curve == BP_CURVE_BLS12_381 || curve == BP_CURVE_BN254; }

static int host_pinned_reserve(bp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->host_pinned_cap) return BP_OK;
    if (ctx->host_pinned) { HIPCHK(hipHostFree(ctx->host_pinned)); ctx->host_pinned = nullptr; ctx->host_pinned_cap = 0; }
    HIPCHK(hipHostMalloc(&ctx->host_pinned, bytes + 4096, hipHostMallocDefault));
    ctx->host_pinned_cap = bytes + 4096;
    return BP_OK;
}


// BP_TRACE=1: synchronise and log after every stage (debugging aid; off by default; changes timing, never results).
#define BP_TRACE_SYNC(ctx_, what)                                                                      \
    do {                                                                                               \
        if (bp_trace_on()) {                                                                           \
            hipError_t te_ = hipStreamSynchronize((ctx_)->stream);                                     \
            fprintf(stderr, "[bpmsm trace] %s -> %s\n", what, hipGetErrorString(te_));                 \
            fflush(stderr);                                                                            \
        }                                                                                              \
    } while (0)

// MSM over raw resident device arrays (n > 0 or n == 0 -> identity); defined in bp_capi.hip.
int bp_internal_msm(bp_ctx* ctx, const void* points, const void* scalars, size_t n, uint8_t* out_le, const bp_g1table* tb = nullptr);
// window-multiples table over n resident points (pool block; bp_internal_table_free)
int bp_internal_table_build(bp_ctx* ctx, const void* points, size_t n, int c, bp_g1table** out);
// table of [G[offG..+n) | H[offH..+n) | extra] out of the tables of G and H; *out = NULL when they have none (bp_capi.hip)
int bp_internal_table_concat(bp_ctx* ctx, const bp_g1vec* G, size_t offG, const bp_g1vec* H, size_t offH, size_t n, const uint8_t* extra_le, bp_g1table** out);
// sibling contexts for independent MSMs in flight (bp_capi.hip)
extern "C" bp_ctx* bp_internal_helper(bp_ctx* ctx, int k);
extern "C" int bp_internal_fork(bp_ctx* ctx, bp_ctx* sibling);
// nnz = non-zero scalars per set if known (0: assume n)
int bp_internal_msm2(bp_ctx* ctx, const void* points, const void* scalars1, const void* scalars2, size_t n, uint8_t* out1_le, uint8_t* out2_le,
                     size_t nnz, const bp_g1table* tb = nullptr);
int bp_internal_set_device(const bp_ctx* ctx);
// n canonical scalars the LIBRARY produced (challenges, their powers, blinding combinations) -> a resident vector, without the
// canonicity check and without waiting for the stream (the bytes are staged in the context's page-locked ring)
extern "C" int bp_internal_frvec_upload_trusted(bp_ctx* ctx, const uint8_t* le32, size_t n, bp_frvec** out);
extern "C" int bp_internal_stage_h2d(bp_ctx* ctx, const void* src, size_t bytes, void* dst);      // library-made host bytes -> device, through the pinned ring (no wait)
// a few two-term commitments k1 g + k2 h on the host (bp_capi.hip)
// The verifiers' single check  out = <xsc, xpts> + <gh_sc, [G[0 .. n) | H[0 .. n)]>  with the [G | H] part over the vectors' window tables
// (merged-window pipeline on ctx's stream) and the nx other terms (proof points, commitments) as their own MSM on a sibling stream, one
// host fold over both record sets.  *done = false (and nothing queued) when BP_TUNE_VERIFY_TABLES is off (the default) or n is below its
// limit, when G or H has no table, or the widths differ: the caller then runs the plain MSM over the concatenated vector.
// whether bp_internal_msm_extras_gh will run over the tables for these generators (builds / finds the context's [G | H] table)
int bp_internal_gh_ready(bp_ctx* ctx, const bp_g1vec* G, const bp_g1vec* H, size_t n, bool* yes);
int bp_internal_msm_extras_gh(bp_ctx* ctx, const void* xpts, const void* xsc, size_t nx, const void* gh_sc, const bp_g1vec* G, const bp_g1vec* H, size_t n,
                              uint8_t* out_le, bool* done);
int bp_internal_host_mul2(bp_ctx* ctx, const uint8_t* g_le, const uint8_t* h_le, const uint8_t* k1_le32, const uint8_t* k2_le32, int count, uint8_t* const* out_le);
// building blocks of the sharded inner-product argument (bp_capi.hip)
int bp_internal_pair_width(bp_ctx* ctx, size_t n, size_t nnz);
int bp_internal_msm2_begin(bp_ctx* ctx, const void* pts, const void* sc1, const void* sc2, size_t n, int c, size_t nnz, const bp_g1table* tb, int* nrec_out,
                           uint16_t* rpos_out);
int bp_internal_fold_sets(bp_ctx* ctx, int nfolds, const void* const* rec, size_t sets, int nrec, const uint16_t* const* pos, uint8_t* const* out_le);
