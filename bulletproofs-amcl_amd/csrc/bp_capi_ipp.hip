// bp_capi_ipp.hip -- C ABI (include/bpmsm.h) for the FieldElementVector helpers, the host transcript and the
// inner-product argument.  Host orchestration mirrors /root/reference src/ipp.rs:35-315 and src/transcript.rs:29-61;
// all per-element work runs in the kernels of bp_ipp.cuh, the MSMs in the bucket pipeline of bp_capi.hip.
#include <errno.h>
#include <sys/random.h>
#include <new>
#include <vector>

#include "bp_internal.hpp"
#include "bp_ipp.cuh"
#include "bp_compact.cuh"
#include "bp_merlin.hpp"
#include "bp_host_tail.hpp"

using namespace bp;

struct bp_transcript {
    Transcript t;
    bp_transcript(const uint8_t* label, size_t len) : t(label, len) {}
};

struct bp_ipp_state {
    bp_ctx* ctx;
    size_t n0, n;            // original / current length
    bool first;              // G_factors / H_factors still to be applied (src/ipp.rs:68-136)
    void *G, *H, *a, *b;     // working copies (src/ipp.rs:57-60)
    void *gf, *hf;           // factors (first round only)
    void *Q;                 // one resident point
    void *pts_tmp, *sc_tmp;  // MSM terms of L and R
    void *cLR;               // c_L, c_R
    void *partial;           // inner-product block partials
    // default mode (generators never folded, see bp_ipp.cuh): resident [G | H | Q], coefficient vectors, L/R scalars
    bool fold_generators;
    void *Pall, *cG, *cH, *sL, *sR;
    bp_g1table* table;       // window multiples of [G | H | Q] when G and H carry tables (bp_g1vec_precompute): every round's MSM is merged-window
    // generator compaction (bp_compact.cuh): once the live length has shrunk to compact_at the folded generators are materialised and
    // the remaining rounds run as single-launch rounds over THEIR digit multiples (n0, Pall, cG, cH, table then describe the compacted set)
    bool glv;                // the compaction (and the rounds after it) work on GLV-split scalars (no compaction tables, BP_TUNE_GLV)
    size_t compact_at;       // 0: this proof never compacts
    bool compacted;
    const bp_g1table *ctG, *ctH;   // compaction tables of G and H (bp_g1vec_precompute) when both have one: nothing to build, Horner chain of 60 doublings
    size_t ctG_off, ctH_off;
    void* Daff;              // otherwise: affine digit multiples m P_i (m = 1 .. 8) of the 2 n0 originals [G | H]: needs no challenge, queued at creation
    hipEvent_t ev_side;      // ... on a sibling stream: recorded behind that work
    bool side_pending;       // ev_side not yet waited for by the context's stream
    int device;
    // every buffer above is a block of the context's pool (recycled, no hipMalloc / hipFree per proof)
    DevPool* pool;
    std::vector<std::pair<void*, size_t>>* blocks;
    bool take(void** out, size_t bytes) {
        size_t cap = 0;
        void* p = pool->get(bytes ? bytes : 1, &cap);
        if (!p) return false;
        blocks->push_back({p, cap});
        *out = p;
        return true;
    }
};

// flattened_constraints plan: the circuit's terms grouped by destination (CSR), resident on the device
struct bp_r1cs_plan {
    int device, curve;
    size_t n, m, nq, nterms, ndest, nheavy;
    void *seg, *tq, *coeff, *heavy;     // uint32[ndest + 1], uint32[nterms], ScalarWords[nterms], uint32[nheavy]
    size_t nchunks = 0;                 // heavy destinations cut into chunks of <= kFlattenChunk terms (k_r1cs_flatten_heavy)
    void *chunk = nullptr, *hfirst = nullptr;   // uint32[2 nchunks] (first, end term of a chunk), uint32[nheavy + 1] (first chunk of a heavy destination)
};

namespace {

constexpr unsigned kInnerBlocks = 256;
constexpr uint32_t kFlattenLightMax = 64;   // destinations with more terms get a block (k_r1cs_flatten_heavy)

inline unsigned blocks_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// ---- host-side Fr helpers (shared templates, host build) ----------------------------------------------------
template <class F> Fe<F> fr_from_le(const uint8_t* le32) {
    uint32_t w[8];
    memcpy(w, le32, 32);
    return fe_to_mont<F>(fe_unpack_words<F>(w));
}
template <class F> void fr_to_le(const Fe<F>& x_mont, uint8_t* le32) {
    uint32_t w[8];
    fe_pack_words<F>(w, fe_from_mont<F>(x_mont));
    memcpy(le32, w, 32);
}
template <class F> ScalarWords fr_mont_words(const Fe<F>& x_mont) {
    ScalarWords s;
    fe_pack_words<F>(s.w, x_mont);
    return s;
}
// x^-1 on the host in ~3 us (64-bit limbs, binary extended Euclid: bp_host_tail.hpp) instead of ~40 us (x^(r-2) with the 30-bit device
// templates): one per inner-product round sits between the round's MSM and the next round's first kernel.  0 -> 0 as before.
template <class F> Fe<F> fr_inv_fast(const Fe<F>& x_mont) {
    if (fe_is_zero(x_mont)) return x_mont;
    static const host::F64<F> f;
    constexpr int N = host::F64<F>::N;
    uint32_t w[F::NW];
    fe_pack_words<F>(w, fe_from_mont<F>(x_mont));                 // plain x
    uint64_t a[N] = {}, r[N], one[N] = {};
    for (int i = 0; i < F::NW; i++) a[i / 2] |= (uint64_t)w[i] << (32 * (i & 1));
    one[0] = 1;
    f.inverse(r, a);                                              // takes x = (x / R) R, returns (x / R)^-1 R = x^-1 R^2
    f.mul(r, r, one);                                             // x^-1 R
    f.mul(r, r, one);                                             // x^-1
    for (int i = 0; i < F::NW; i++) w[i] = (uint32_t)(r[i / 2] >> (32 * (i & 1)));
    return fe_to_mont<F>(fe_unpack_words<F>(w));
}

// FieldElement::from(&[u8; MODBYTES]): big-endian integer mod r (src/transcript.rs:55-60)  [UNVERIFIED-RECALL]
template <class F> Fe<F> fr_from_be_reduce(const uint8_t* be, int nbytes) {
    Fe<F> acc = fe_zero<F>();
    Fe<F> c256 = fe_zero<F>();
    c256.v[0] = 256;
    c256 = fe_to_mont<F>(c256);
    for (int i = 0; i < nbytes; i++) {
        Fe<F> d = fe_zero<F>();
        d.v[0] = be[i];
        acc = fe_add(fe_mul(acc, c256), fe_to_mont<F>(d));
    }
    return acc;
}

// amcl byte formats used by the transcript ([UNVERIFIED-RECALL])
void point_le_to_amcl(const uint8_t* le, int fb, uint8_t* out /* 2 fb + 1 */) {
    memset(out, 0, 2 * fb + 1);
    out[0] = 0x04;
    bool z = true;
    for (int k = 0; k < 2 * fb; k++) if (le[k]) { z = false; break; }
    if (z) { out[2 * fb] = 1; return; }
    for (int k = 0; k < fb; k++) { out[fb - k] = le[k]; out[2 * fb - k] = le[fb + k]; }
}

template <class C>
struct Ipp {
    using F = typename C::Fr;
    using Fp = typename C::Fp;
    static constexpr size_t kPt = sizeof(AffPacked<C>);
    static constexpr int kFb = 4 * Fp::NW;

    static int inner(bp_ctx* ctx, const ScalarWords* a, const ScalarWords* b, size_t n, ScalarWords* d_partial, ScalarWords* d_out) {
        unsigned g = blocks_for(n);
        if (g > kInnerBlocks) g = kInnerBlocks;
        if (g == 0) g = 1;
        hipLaunchKernelGGL(k_fr_inner<C>, dim3(g), dim3(kBlock), 0, ctx->stream, a, b, n, d_partial);
        hipLaunchKernelGGL(k_fr_inner_final<C>, dim3(1), dim3(kBlock), 0, ctx->stream, d_partial, g, d_out);
        HIPCHK(hipGetLastError());
        return BP_OK;
    }


    // ---- generator compaction (bp_compact.cuh) ----
    // out[i] = affine form of in[i] for i < n, one field inversion for the whole array.  device_root: the inversion runs on one GPU lane
    // (~0.4 ms of latency, no host round trip: for batches queued ahead of their use); otherwise the host inverts the root (~3 us + a sync).
    // `hold` != nullptr: the batch runs on stream `s` (NOT the context's) with the root inverted on the device, and its scratch blocks
    // stay with the state until it is freed (the pool recycles in the order of the context's stream only).
    static int batch_to_affine(bp_ctx* ctx, const XyzzPacked<C>* in, size_t n, AffPacked<C>* out, hipStream_t s, bp_ipp_state* hold) {
        if (n == 0) return BP_OK;
        constexpr size_t kChunk = (size_t)kBlock * kBaiMidPer * kBaiTile;       // what one middle block covers (4 M elements): larger arrays in chunks, an inversion each
        if (n > kChunk) {
            for (size_t o = 0; o < n; o += kChunk) {
                int rc = batch_to_affine(ctx, in + o, n - o < kChunk ? n - o : kChunk, out + o, s, hold);
                if (rc) return rc;
            }
            return BP_OK;
        }
        const size_t nb = (n + kBaiTile - 1) / kBaiTile;
        PoolBlock b_prod, b_inv;
        void *p_prod = nullptr, *p_inv = nullptr;
        if (hold) {
            if (!hold->take(&p_prod, (nb + 1) * sizeof(FePacked<Fp>)) || !hold->take(&p_inv, nb * sizeof(FePacked<Fp>))) return BP_ERR_DEVICE;
        } else {
            if (!b_prod.alloc(ctx, (nb + 1) * sizeof(FePacked<Fp>)) || !b_inv.alloc(ctx, nb * sizeof(FePacked<Fp>))) return BP_ERR_DEVICE;
            p_prod = b_prod.p; p_inv = b_inv.p;
        }
        auto* bprod = (FePacked<Fp>*)p_prod;
        auto* binv = (FePacked<Fp>*)p_inv;
        FePacked<Fp> zero;
        memset(&zero, 0, sizeof zero);
        hipLaunchKernelGGL(k_bai_block_products<C>, dim3((unsigned)nb), dim3(kBlock), 0, s, in, n, bprod);
        if (hold) {
            hipLaunchKernelGGL((k_bai_middle<C, 2>), dim3(1), dim3(kBlock), 0, s, (const FePacked<Fp>*)bprod, (uint32_t)nb, zero, bprod + nb, binv);
        } else {
            hipLaunchKernelGGL((k_bai_middle<C, 0>), dim3(1), dim3(kBlock), 0, s, (const FePacked<Fp>*)bprod, (uint32_t)nb, zero, bprod + nb, binv);
            HIPCHK(hipGetLastError());
            int rc = host_pinned_reserve(ctx, sizeof(FePacked<Fp>));
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(ctx->host_pinned, bprod + nb, sizeof(FePacked<Fp>), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            FePacked<Fp> root;
            memcpy(&root, ctx->host_pinned, sizeof root);
            const FePacked<Fp> rinv = fe_pack(fr_inv_fast<Fp>(fe_unpack_words<Fp>(root.w)));      // (a value < 2p: the conversion out of Montgomery form reduces it)
            hipLaunchKernelGGL((k_bai_middle<C, 1>), dim3(1), dim3(kBlock), 0, s, (const FePacked<Fp>*)bprod, (uint32_t)nb, rinv, bprod + nb, binv);
        }
        hipLaunchKernelGGL(k_bai_finish<C>, dim3((unsigned)nb), dim3(kBlock), 0, s, in, n, (const FePacked<Fp>*)binv, out);
        HIPCHK(hipGetLastError());
        return BP_OK;
    }

    // the XYZZ digit multiples of a table (bp_internal_digit_table_build) -> affine rows, in place of the old block
    static int digit_table_to_affine(bp_ctx* ctx, bp_g1table* t) {
        if (!t || !t->digits || t->affine) return BP_OK;
        const size_t rows = (size_t)t->W * t->n;
        size_t cap = 0;
        void* d = ctx->pool->get(rows * kPt, &cap);
        if (!d) return BP_ERR_DEVICE;
        int rc = batch_to_affine(ctx, (const XyzzPacked<C>*)t->d, rows, (AffPacked<C>*)d, ctx->stream, nullptr);
        if (rc) { ctx->pool->put(d, cap); return rc; }
        ctx->pool->put(t->d, t->cap);            // recycled in stream order: the conversion above is queued before any later user
        t->d = d; t->cap = cap; t->affine = true;
        return BP_OK;
    }

    // The 16 affine digit multiples of [G | H | Q] for k_small_msm_glv: a proof of 1024 .. 4096 generators (no compaction) runs ALL its rounds
    // over scalars split in two halves (half the windows per launch, half the doublings in every host tail) when the curve has the split.
    // affine = false (proofs below 1024 generators): the rows stay packed lazy XYZZ -- no batch inversion, k_small_msm_glv<C, true>
    static int glv_round_table(bp_ipp_state* st, bool affine) {
        bp_ctx* ctx = st->ctx;
        const size_t m = 2 * st->n0 + 1, rows = kGlvRows;
        bp_g1table* t = new (std::nothrow) bp_g1table();
        if (!t) return BP_ERR_DEVICE;
        t->pool = ctx->pool; t->device = ctx->device; t->n = m; t->c = kGlvBits; t->W = (int)rows; t->digits = true; t->affine = affine; t->glv = true;
        t->d = ctx->pool->get(rows * m * (affine ? kPt : sizeof(XyzzPacked<C>)), &t->cap);
        PoolBlock b_mx;
        if (!t->d || (affine && !b_mx.alloc(ctx, rows * m * sizeof(XyzzPacked<C>)))) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
        if (rows * m <= kDigitTableParMax)
            hipLaunchKernelGGL(k_digit_table_build_par<C>, dim3((unsigned)((rows * m + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const AffPacked<C>*)st->Pall,
                               (uint32_t)m, (XyzzPacked<C>*)(affine ? b_mx.p : t->d), (uint32_t)rows);
        else
            hipLaunchKernelGGL(k_digit_table_build<C>, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream, (const AffPacked<C>*)st->Pall, (uint32_t)m,
                               (XyzzPacked<C>*)(affine ? b_mx.p : t->d), (uint32_t)rows);
        if (hipGetLastError() != hipSuccess) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
        if (affine) {
            int rc = batch_to_affine(ctx, (const XyzzPacked<C>*)b_mx.p, rows * m, (AffPacked<C>*)t->d, ctx->stream, nullptr);
            if (rc) { bp_internal_table_free(t); return rc; }
        }
        st->table = t;
        return BP_OK;
    }

    // Daff[(m - 1) * npts + i] = m P_i (affine), m = 1 .. 8, over the npts = 2 n0 originals [G | H] of the state.  Needs no challenge:
    // queued at state creation on a SIBLING stream (bp_internal_helper), where it fills the gaps of the latency-bound first rounds;
    // compact() makes the context's stream wait for ev_side.
    static int build_original_multiples(bp_ipp_state* st) {
        bp_ctx* ctx = st->ctx;
        const size_t npts = 2 * st->n0, rows = (size_t)1 << (kSmallDigitBits - 1);
        if (npts >= ((size_t)1 << 31)) return BP_ERR_ARG;
        bp_ctx* side = bp_internal_helper(ctx, 0);
        if (!side) return BP_ERR_DEVICE;
        void* tmp = nullptr;
        if (!st->take(&tmp, rows * npts * sizeof(XyzzPacked<C>)) || !st->take(&st->Daff, rows * npts * kPt)) return BP_ERR_DEVICE;
        int rc = bp_internal_fork(ctx, side);
        if (rc) return rc;
        if (!st->ev_side) HIPCHK(hipEventCreateWithFlags(&st->ev_side, hipEventDisableTiming));
        hipLaunchKernelGGL(k_digit_table_build<C>, dim3((unsigned)((npts + 63) / 64)), dim3(64), 0, side->stream, (const AffPacked<C>*)st->Pall, (uint32_t)npts,
                           (XyzzPacked<C>*)tmp, (uint32_t)rows);
        HIPCHK(hipGetLastError());
        rc = batch_to_affine(ctx, (const XyzzPacked<C>*)tmp, rows * npts, (AffPacked<C>*)st->Daff, side->stream, st);
        HIPCHK(hipEventRecord(st->ev_side, side->stream));
        st->side_pending = true;
        return rc;
    }

    // ---- the same over GLV-split scalars (BLS12-381): 26 windows of 5 bits over 2 T sub-terms per output, a Horner chain of 125
    // doublings, and a digit table of [G' | H' | Q] with 16 multiples of P and of phi(P) for k_small_msm_glv
    static int compact_glv(bp_ipp_state* st) {
        bp_ctx* ctx = st->ctx;
        hipStream_t s = ctx->stream;
        const size_t n0 = st->n0, nj = st->n, nout = 2 * nj, m = nout + 1, rows = kGlvRows;
        int rc;
        PoolBlock b_part, b_wsum, b_S, b_mx;
        if (!b_part.alloc(ctx, (size_t)2 * kGlvCWin * nout * sizeof(XyzzPacked<C>)) || !b_wsum.alloc(ctx, (size_t)kGlvCWin * nout * sizeof(XyzzPacked<C>)) ||
            !b_S.alloc(ctx, m * sizeof(XyzzPacked<C>)) || !b_mx.alloc(ctx, rows * m * sizeof(XyzzPacked<C>)))
            return BP_ERR_DEVICE;
        // c_G, c_H split into halves (sL / sR are free between rounds and large enough)
        hipLaunchKernelGGL(k_glv_decompose<C>, dim3(blocks_for(n0)), dim3(kBlock), 0, s, (const ScalarWords*)st->cG, (const ScalarWords*)st->cH, n0,
                           (ScalarWords*)st->sL, (ScalarWords*)st->sR);
        const AffPacked<C>* DG = (const AffPacked<C>*)st->Daff;
        hipLaunchKernelGGL(k_compact_window_sums_glv<C>, dim3((unsigned)((nout + kBlock - 1) / kBlock), (unsigned)kGlvCWin, 2u), dim3(kBlock), 0, s, DG, DG + n0, 2 * n0,
                           (const ScalarWords*)st->sL, (const ScalarWords*)st->sR, (uint32_t)n0, (uint32_t)nj, (uint32_t)nout, (XyzzPacked<C>*)b_part.p);
        hipLaunchKernelGGL(k_compact_merge_halves<C>, dim3((unsigned)((nout + kBlock - 1) / kBlock), (unsigned)kGlvCWin), dim3(kBlock), 0, s, (const XyzzPacked<C>*)b_part.p,
                           (uint32_t)nout, (uint32_t)nout, kGlvCWin, (XyzzPacked<C>*)b_wsum.p);
        BP_TRACE_SYNC(ctx, "k_compact_window_sums_glv");
        hipLaunchKernelGGL(k_compact_horner<C>, dim3((unsigned)((nout + kHornerQuads / 2 - 1) / (kHornerQuads / 2))), dim3(4 * kHornerQuads), 0, s,
                           (const XyzzPacked<C>*)b_wsum.p, (uint32_t)nout, (uint32_t)nout, kGlvCWin, kGlvCBits, (const AffPacked<C>*)st->Q, (XyzzPacked<C>*)b_S.p);
        BP_TRACE_SYNC(ctx, "k_compact_horner");
        hipLaunchKernelGGL(k_digit_table_build_xyzz<C>, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, (const XyzzPacked<C>*)b_S.p, (uint32_t)m, (uint32_t)rows,
                           (XyzzPacked<C>*)b_mx.p);
        HIPCHK(hipGetLastError());
        bp_g1table* t = new (std::nothrow) bp_g1table();
        if (!t) return BP_ERR_DEVICE;
        t->pool = ctx->pool; t->device = ctx->device; t->n = m; t->c = kGlvBits; t->W = (int)rows; t->digits = true; t->affine = true; t->glv = true;
        t->d = ctx->pool->get(rows * m * kPt, &t->cap);
        if (!t->d) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
        if ((rc = batch_to_affine(ctx, (const XyzzPacked<C>*)b_mx.p, rows * m, (AffPacked<C>*)t->d, s, nullptr))) { bp_internal_table_free(t); return rc; }
        hipLaunchKernelGGL(k_fr_fill_one, dim3(blocks_for(nj)), dim3(kBlock), 0, s, (ScalarWords*)st->cG, (ScalarWords*)st->cH, nj);
        if (hipGetLastError() != hipSuccess) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
        BP_TRACE_SYNC(ctx, "compaction (glv): digit table");
        if (st->table) bp_internal_table_free(st->table);
        st->table = t;
        st->Pall = t->d;                 // the first m rows (multiple 1) are [G' | H' | Q] themselves
        st->n0 = nj;
        st->compacted = true;
        return BP_OK;
    }

    // L, R of a round after a GLV compaction (the round's scalars arrive split: k_ipp_round_prep<C, true>): k_small_msm_glv leaves 26 x splits records per
    // scalar set at bit positions 5 w, the host folds two tails of <= 125 doublings
    static int msm2_glv(bp_ipp_state* st, uint8_t* L_le, uint8_t* R_le) {
        bp_ctx* ctx = st->ctx;
        hipStream_t s = ctx->stream;
        const size_t m = 2 * st->n0 + 1;
        const unsigned splits = st->n0 + 1 > 512 ? 4u : 2u;         // blocks per (window, set): half of them per scalar half (n0 + 1 participating terms)
        const int R1 = kGlvWin * (int)splits;
        int rc;
        bp_prof().lap(0);
        if ((rc = ctx->window_sum.reserve(ctx, (size_t)2 * R1 * sizeof(XyzzPacked<C>)))) return rc;
        if ((rc = host_pinned_reserve(ctx, (size_t)2 * R1 * sizeof(XyzzPacked<C>)))) return rc;
        const IppSparse sp = st->n >= 2 ? IppSparse{(uint32_t)st->n0, (uint32_t)st->n, (uint32_t)(st->n / 2)} : IppSparse{0, 0, 0};
        if (st->table->affine)
            hipLaunchKernelGGL((k_small_msm_glv<C, false>), dim3(kGlvWin, 2, splits), dim3(kBlock), 0, s, (const ScalarWords*)st->sL, (const ScalarWords*)st->sR, (uint32_t)m,
                               (const void*)st->table->d, (XyzzPacked<C>*)ctx->window_sum.p, sp);
        else
            hipLaunchKernelGGL((k_small_msm_glv<C, true>), dim3(kGlvWin, 2, splits), dim3(kBlock), 0, s, (const ScalarWords*)st->sL, (const ScalarWords*)st->sR, (uint32_t)m,
                               (const void*)st->table->d, (XyzzPacked<C>*)ctx->window_sum.p, sp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, ctx->window_sum.p, (size_t)2 * R1 * sizeof(XyzzPacked<C>), hipMemcpyDeviceToHost, s));
        bp_prof().lap(1);
        HIPCHK(hipStreamSynchronize(s));
        bp_prof().lap(2);
        uint16_t pos[kGlvWin * 4];
        for (int r = 0; r < R1; r++) pos[r] = (uint16_t)(kGlvBits * (r / (int)splits));
        const XyzzPacked<C>* rec = (const XyzzPacked<C>*)ctx->host_pinned;
        const void* recs[2] = {rec, rec + R1};
        const uint16_t* poss[2] = {pos, pos};
        uint8_t* outs[2] = {L_le, R_le};
        rc = bp_internal_fold_sets(ctx, 2, recs, 1, R1, poss, outs);
        bp_prof().lap(3);
        return rc;
    }

    // The folded generators of the current round, materialised (called by fold() when the live length reaches compact_at):
    // afterwards n0 = the live length, Pall = [G' | H' | Q] (affine), c_G = c_H = 1 and table = the affine digit multiples of Pall.
    static int compact(bp_ipp_state* st) {
        bp_ctx* ctx = st->ctx;
        hipStream_t s = ctx->stream;
        const size_t n0 = st->n0, nj = st->n, nout = 2 * nj, m = nout + 1, rows = (size_t)1 << (kSmallDigitBits - 1);
        int rc;
        const bool tabled = st->ctG && st->ctH;
        if (!tabled && !st->Daff && (rc = build_original_multiples(st))) return rc;
        if (st->side_pending) { HIPCHK(hipStreamWaitEvent(s, st->ev_side, 0)); st->side_pending = false; }
        if constexpr (C::HAS_GLV) { if (st->glv && !tabled) return compact_glv(st); }
        const int lgK = tabled ? 2 : 0;                                                  // tables: a scalar is 4 sub-scalars of 64 bits over the rows 2^(64 k) P
        const int nwin = ((C::Fr::BITS + 1 + kSmallDigitBits - 1) / kSmallDigitBits) >> lgK;      // 64 (16) windows of 4 bits
        const AffPacked<C>*DG, *DH;
        size_t drowsG, drowsH, ksG, ksH;
        if (tabled) {
            DG = (const AffPacked<C>*)st->ctG->d + st->ctG_off; drowsG = (size_t)st->ctG->K * st->ctG->n; ksG = st->ctG->n;
            DH = (const AffPacked<C>*)st->ctH->d + st->ctH_off; drowsH = (size_t)st->ctH->K * st->ctH->n; ksH = st->ctH->n;
        } else {
            DG = (const AffPacked<C>*)st->Daff; DH = DG + n0; drowsG = drowsH = 2 * n0; ksG = ksH = 0;
        }
        ScalarWords bias;
        for (int k = 0; k < 8; k++) bias.w[k] = 0x77777777u;                             // digit = nibble - 7
        PoolBlock b_wsum, b_S, b_mx;
        if (!b_wsum.alloc(ctx, (size_t)nwin * nout * sizeof(XyzzPacked<C>)) || !b_S.alloc(ctx, m * sizeof(XyzzPacked<C>)) ||
            !b_mx.alloc(ctx, rows * m * sizeof(XyzzPacked<C>)))
            return BP_ERR_DEVICE;
        hipLaunchKernelGGL(k_compact_window_sums<C>, dim3((unsigned)((nout + kBlock - 1) / kBlock), (unsigned)nwin), dim3(kBlock), 0, s, DG, DH, drowsG, drowsH, ksG, ksH,
                           (const ScalarWords*)st->cG, (const ScalarWords*)st->cH, (uint32_t)n0, (uint32_t)nj, lgK, nwin, bias, (uint32_t)nout, (XyzzPacked<C>*)b_wsum.p);
        BP_TRACE_SYNC(ctx, "k_compact_window_sums");
        hipLaunchKernelGGL(k_compact_horner<C>, dim3((unsigned)((nout + kHornerQuads / 2 - 1) / (kHornerQuads / 2))), dim3(4 * kHornerQuads), 0, s,
                           (const XyzzPacked<C>*)b_wsum.p, (uint32_t)nout, (uint32_t)nout, nwin, kSmallDigitBits, (const AffPacked<C>*)st->Q, (XyzzPacked<C>*)b_S.p);
        BP_TRACE_SYNC(ctx, "k_compact_horner");
        // the table of the rounds that follow: with the GLV split (BLS12-381) 16 multiples for k_small_msm_glv, else 8 for k_small_msm
        bool glv_rounds = false;
        if constexpr (C::HAS_GLV) glv_rounds = ctx->tuning.glv;
        const size_t rrows = glv_rounds ? (size_t)kGlvRows : rows;
        if (glv_rounds && !b_mx.alloc(ctx, rrows * m * sizeof(XyzzPacked<C>))) return BP_ERR_DEVICE;
        hipLaunchKernelGGL(k_digit_table_build_xyzz<C>, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, (const XyzzPacked<C>*)b_S.p, (uint32_t)m, (uint32_t)rrows, (XyzzPacked<C>*)b_mx.p);
        HIPCHK(hipGetLastError());
        bp_g1table* t = new (std::nothrow) bp_g1table();
        if (!t) return BP_ERR_DEVICE;
        t->pool = ctx->pool; t->device = ctx->device; t->n = m; t->c = glv_rounds ? kGlvBits : kSmallDigitBits; t->W = (int)rrows; t->digits = true; t->affine = true;
        t->glv = glv_rounds;
        t->d = ctx->pool->get(rrows * m * kPt, &t->cap);
        if (!t->d) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
        if ((rc = batch_to_affine(ctx, (const XyzzPacked<C>*)b_mx.p, rrows * m, (AffPacked<C>*)t->d, s, nullptr))) { bp_internal_table_free(t); return rc; }
        BP_TRACE_SYNC(ctx, "compaction: digit multiples");
        hipLaunchKernelGGL(k_fr_fill_one, dim3(blocks_for(nj)), dim3(kBlock), 0, s, (ScalarWords*)st->cG, (ScalarWords*)st->cH, nj);
        HIPCHK(hipGetLastError());
        if (st->table) bp_internal_table_free(st->table);
        st->table = t;
        st->Pall = t->d;                 // the first m rows (multiple 1) are [G' | H' | Q] themselves, affine
        st->n0 = nj;
        st->compacted = true;
        return BP_OK;
    }

    static int round(bp_ipp_state* st, uint8_t* L_le, uint8_t* R_le) {
        bp_ctx* ctx = st->ctx;
        size_t h = st->n / 2;
        auto* a = (ScalarWords*)st->a; auto* b = (ScalarWords*)st->b;
        auto* cLR = (ScalarWords*)st->cLR;
        int rc;
        unsigned g = blocks_for(h);
        if (g > kInnerBlocks / 2) g = kInnerBlocks / 2;
        if (g == 0) g = 1;
        if (!st->fold_generators) {
            // c_L = <a_L, b_R>, c_R = <a_R, b_L> (src/ipp.rs:77-78, 145-146) and the round's L / R scalars: two launches (k_ipp_round_prep / _final)
            size_t m = 2 * st->n0 + 1;
            unsigned gx = blocks_for(st->n0);                 // the scalar row: a lane per generator (up to 1024 blocks, then grid-stride)
            if (gx > 1024) gx = 1024;
            if (gx < g) gx = g;
            bool split = false;
            if constexpr (C::HAS_GLV) split = st->table && st->table->glv;
            if (split) {
                if constexpr (C::HAS_GLV) {
                    hipLaunchKernelGGL((k_ipp_round_prep<C, true>), dim3(gx, 3), dim3(kBlock), 0, ctx->stream, a, b, (const ScalarWords*)st->cG, (const ScalarWords*)st->cH, st->n0,
                                       st->n, g, (ScalarWords*)st->partial, (ScalarWords*)st->sL, (ScalarWords*)st->sR);
                    hipLaunchKernelGGL((k_ipp_round_final<C, true>), dim3(2), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)st->partial, g, cLR, st->n0, (ScalarWords*)st->sL,
                                       (ScalarWords*)st->sR);
                }
            } else {
                hipLaunchKernelGGL((k_ipp_round_prep<C, false>), dim3(gx, 3), dim3(kBlock), 0, ctx->stream, a, b, (const ScalarWords*)st->cG, (const ScalarWords*)st->cH, st->n0,
                                   st->n, g, (ScalarWords*)st->partial, (ScalarWords*)st->sL, (ScalarWords*)st->sR);
                hipLaunchKernelGGL((k_ipp_round_final<C, false>), dim3(2), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)st->partial, g, cLR, st->n0, (ScalarWords*)st->sL,
                                   (ScalarWords*)st->sR);
            }
            HIPCHK(hipGetLastError());
            BP_TRACE_SYNC(ctx, "ipp round scalars");
            if constexpr (C::HAS_GLV) { if (split) return msm2_glv(st, L_le, R_le); }
            ctx->ipp_n0 = (uint32_t)st->n0; ctx->ipp_live = (uint32_t)st->n;       // single-launch rounds walk the participating terms only
            const int rcm = bp_internal_msm2(ctx, st->Pall, st->sL, st->sR, m, L_le, R_le, st->n0 + 1, st->table);   // both sums in one pipeline pass
            ctx->ipp_n0 = 0; ctx->ipp_live = 0;
            return rcm;
        }
        {   // reference-shaped mode
            hipLaunchKernelGGL(k_fr_inner2<C>, dim3(g, 2), dim3(kBlock), 0, ctx->stream, a, b + h, a + h, b, h, (ScalarWords*)st->partial);
            hipLaunchKernelGGL(k_fr_inner2_final<C>, dim3(2), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)st->partial, g, cLR);
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_ipp_pack_round<C>, dim3(blocks_for(h)), dim3(kBlock), 0, ctx->stream, (const AffPacked<C>*)st->G,
                           (const AffPacked<C>*)st->H, a, b, st->first ? (const ScalarWords*)st->gf : nullptr,
                           st->first ? (const ScalarWords*)st->hf : nullptr, (const AffPacked<C>*)st->Q, cLR, h, (AffPacked<C>*)st->pts_tmp,
                           (ScalarWords*)st->sc_tmp);
        HIPCHK(hipGetLastError());
        BP_TRACE_SYNC(ctx, "ipp inner+pack");
        size_t blk = 2 * h + 1;
        if ((rc = bp_internal_msm(ctx, st->pts_tmp, st->sc_tmp, blk, L_le))) return rc;
        if ((rc = bp_internal_msm(ctx, (const uint8_t*)st->pts_tmp + blk * kPt, (const uint8_t*)st->sc_tmp + blk * 32, blk, R_le))) return rc;
        return BP_OK;
    }

    static int fold(bp_ipp_state* st, const uint8_t* u_le, const uint8_t* uinv_le) {
        bp_ctx* ctx = st->ctx;
        size_t h = st->n / 2;
        Fe<F> u = fr_from_le<F>(u_le), ui = fr_from_le<F>(uinv_le);
        if (!st->fold_generators) {
            hipLaunchKernelGGL(k_ipp_fold_scalars<C>, dim3(blocks_for(st->n0)), dim3(kBlock), 0, ctx->stream, (ScalarWords*)st->a, (ScalarWords*)st->b,
                               (ScalarWords*)st->cG, (ScalarWords*)st->cH, fr_mont_words<F>(u), fr_mont_words<F>(ui), st->n0, (size_t)0, st->n);
            HIPCHK(hipGetLastError());
            BP_TRACE_SYNC(ctx, "ipp fold scalars");
            st->n = h;
            st->first = false;
            if (st->compact_at && !st->compacted && st->n == st->compact_at) return compact(st);
            return BP_OK;
        }
        BP_TRACE_SYNC(ctx, "ipp fold: launching");
        hipLaunchKernelGGL(k_ipp_fold<C>, dim3(blocks_for(2 * h)), dim3(kBlock), 0, ctx->stream, (AffPacked<C>*)st->G, (AffPacked<C>*)st->H,
                           (ScalarWords*)st->a, (ScalarWords*)st->b, st->first ? (const ScalarWords*)st->gf : nullptr,
                           st->first ? (const ScalarWords*)st->hf : nullptr, fr_mont_words<F>(u), fr_mont_words<F>(ui), h);
        HIPCHK(hipGetLastError());
        BP_TRACE_SYNC(ctx, "ipp fold");
        st->n = h;
        st->first = false;
        return BP_OK;
    }

    // ---- transcript protocol (src/transcript.rs:29-61) ----
    static void commit_point(Transcript& t, const char* label, const uint8_t* p_le) {
        uint8_t buf[2 * kFb + 1];
        point_le_to_amcl(p_le, kFb, buf);
        t.append_message((const uint8_t*)label, strlen(label), buf, sizeof buf);
    }
    static void commit_scalar(Transcript& t, const char* label, const uint8_t* s_le) {
        uint8_t buf[C::MODBYTES];
        memset(buf, 0, sizeof buf);
        for (int k = 0; k < 32; k++) buf[C::MODBYTES - 1 - k] = s_le[k];     // MODBYTES big-endian
        t.append_message((const uint8_t*)label, strlen(label), buf, sizeof buf);
    }
    static Fe<F> challenge_scalar(Transcript& t, const char* label) {
        uint8_t buf[C::MODBYTES];
        t.challenge_bytes((const uint8_t*)label, strlen(label), buf, sizeof buf);
        return fr_from_be_reduce<F>(buf, C::MODBYTES);
    }
    static void ipp_domain_sep(Transcript& t, uint64_t n) {
        t.append_message((const uint8_t*)"dom-sep", 7, (const uint8_t*)"ipp v1", 6);
        t.append_u64((const uint8_t*)"n", 1, n);
    }

    // IPP::create_ipp, src/ipp.rs:35-202
    static int create(bp_ipp_state* st, Transcript& t, uint8_t* L_out, uint8_t* R_out, size_t* lg_n_out, uint8_t* a_out, uint8_t* b_out) {
        ipp_domain_sep(t, st->n);                                                       // :62
        size_t k = 0;
        int rc;
        BpProf& pf = bp_prof();
        if (bp_profile_on()) for (double& x : pf.acc) x = 0;
        while (st->n != 1) {                                                            // :68, :138
            uint8_t* L = L_out + k * 2 * kFb;
            uint8_t* R = R_out + k * 2 * kFb;
            pf.start();
            if ((rc = round(st, L, R))) return rc;                                      // :77-104 / :145-170
            pf.start();
            commit_point(t, "L", L);                                                    // :106-107 / :172-173
            commit_point(t, "R", R);
            Fe<F> u = challenge_scalar(t, "u");                                         // :112 / :178
            Fe<F> ui = fr_inv_fast<F>(u);                                               // :113 / :179
            uint8_t ub[32], uib[32];
            fr_to_le<F>(u, ub); fr_to_le<F>(ui, uib);
            pf.lap(4);
            const bool will_compact = st->compact_at && !st->compacted && st->n / 2 == st->compact_at;
            if ((rc = fold(st, ub, uib))) return rc;                                    // :115-130 / :181-188
            pf.lap(will_compact ? 6 : 5);
            k++;
        }
        if (lg_n_out) *lg_n_out = k;
        HIPCHK(hipMemcpyAsync(a_out, st->a, 32, hipMemcpyDeviceToHost, st->ctx->stream));   // :196-201
        HIPCHK(hipMemcpyAsync(b_out, st->b, 32, hipMemcpyDeviceToHost, st->ctx->stream));
        HIPCHK(hipStreamSynchronize(st->ctx->stream));
        if (bp_profile_on())
            fprintf(stderr, "[bpmsm profile] ipp n0=%zu rounds=%zu us: scalar-kernel launch %.0f  msm launch %.0f  sync-wait %.0f  host-tails %.0f  transcript+inverse %.0f  fold-launch %.0f  compaction(host side) %.0f\n",
                    (size_t)1 << k, k, pf.acc[0], pf.acc[1], pf.acc[2], pf.acc[3], pf.acc[4], pf.acc[5], pf.acc[6]);
        return BP_OK;
    }

    // IPP::verification_scalars, src/ipp.rs:262-315 (host: O(lg n) transcript work + O(n) Fr products)
    // FieldElement::batch_invert (src/ipp.rs:295): Montgomery's trick, one inversion + 3 products per element;
    // zero stays zero (as the Fermat inverse does).
    static void batch_invert(const std::vector<Fe<F>>& in, std::vector<Fe<F>>& out) {
        size_t k = in.size();
        out.assign(k, fe_zero<F>());
        std::vector<Fe<F>> prefix(k);
        Fe<F> acc = fe_one<F>();
        for (size_t i = 0; i < k; i++) { prefix[i] = acc; if (!fe_is_zero(in[i])) acc = fe_mul(acc, in[i]); }
        Fe<F> inv = fr_inv_fast<F>(acc);
        for (size_t i = k; i-- > 0;) {
            if (fe_is_zero(in[i])) continue;
            out[i] = fe_mul(inv, prefix[i]);
            inv = fe_mul(inv, in[i]);
        }
    }

    // the transcript half of verification_scalars: challenges u_j in creation order (src/ipp.rs:269-288)
    static int verification_challenges(Transcript& t, const uint8_t* L_le, const uint8_t* R_le, size_t lg_n, size_t n, std::vector<Fe<F>>& ch) {
        if (lg_n >= 32) return BP_ERR_VERIFY;                                           // :269-273
        if (n != ((size_t)1 << lg_n)) return BP_ERR_VERIFY;                             // :274-276
        ipp_domain_sep(t, n);                                                           // :278
        ch.resize(lg_n);
        for (size_t j = 0; j < lg_n; j++) {                                             // :283-288
            commit_point(t, "L", L_le + j * 2 * kFb);
            commit_point(t, "R", R_le + j * 2 * kFb);
            ch[j] = challenge_scalar(t, "u");
        }
        return BP_OK;
    }

    static int verification_scalars(Transcript& t, const uint8_t* L_le, const uint8_t* R_le, size_t lg_n, size_t n, std::vector<Fe<F>>& ch,
                                    std::vector<Fe<F>>& ch_inv) {
        int rc = verification_challenges(t, L_le, R_le, lg_n, n, ch);
        if (rc) return rc;
        batch_invert(ch, ch_inv);                                                       // :295
        return BP_OK;
    }

    // IPP::verify_ipp, src/ipp.rs:204-260
    static int verify(bp_ctx* ctx, Transcript& t, size_t n, const bp_frvec* Gf, const bp_frvec* Hf, const uint8_t* P_le, const uint8_t* Q_le,
                      const bp_g1vec* G, const bp_g1vec* H, const uint8_t* a_le, const uint8_t* b_le, const uint8_t* L_le, const uint8_t* R_le,
                      size_t lg_n) {
        std::vector<Fe<F>> ch, ch_inv;
        int rc = verification_scalars(t, L_le, R_le, lg_n, n, ch, ch_inv);              // :218
        if (rc) return rc;
        size_t m = 1 + 2 * n + 2 * lg_n;
        // G and H precomputed: their terms run over the window tables, [Q | L | R] as a small MSM beside them
        bool tabled = false;
        if ((rc = bp_internal_gh_ready(ctx, G, H, n, &tabled))) return rc;
        const size_t nx = 1 + 2 * lg_n;
        // what the host made -- challenges (Montgomery form), the proof's points, the head / tail scalars -- travels as ONE staged copy
        const size_t o_raw = (2 * lg_n + 1) * 32, o_ends = o_raw + nx * 2 * kFb, in_bytes = o_ends + nx * 32;
        PoolBlock b_pts, b_sc, b_in;              // recycled through the context's pool (no hipMalloc / hipFree per proof)
        if (!b_pts.alloc(ctx, (tabled ? nx : m) * kPt) || !b_sc.alloc(ctx, (m + (tabled ? nx : 0)) * 32) || !b_in.alloc(ctx, in_bytes)) return BP_ERR_DEVICE;
        void *pts = b_pts.p, *sc = b_sc.p, *chd = b_in.p, *raw = (uint8_t*)b_in.p + o_raw;
        const ScalarWords* d_ends = (const ScalarWords*)((uint8_t*)b_in.p + o_ends);
        if ((rc = ctx->flags.reserve(ctx, 64))) return rc;
        uint32_t* flag = (uint32_t*)ctx->flags.p;     // Q, L, R come from the proof: validated (on the curve, canonical)
        uint32_t host_flag = 0;
        std::vector<uint8_t> hin(in_bytes);
        ScalarWords* hch = (ScalarWords*)hin.data();                                     // challenges for the per-element products
        for (size_t j = 0; j < lg_n; j++) { hch[j] = fr_mont_words<F>(ch[j]); hch[lg_n + j] = fr_mont_words<F>(ch_inv[j]); }
        memset(&hch[2 * lg_n], 0, 32);
        Fe<F> a = fr_from_le<F>(a_le), b = fr_from_le<F>(b_le);
        // points Q, L_vec, R_vec -> resident form                                       :244-249
        uint8_t* hraw = hin.data() + o_raw;
        memcpy(hraw, Q_le, 2 * kFb);
        if (lg_n) { memcpy(hraw + 2 * kFb, L_le, lg_n * 2 * kFb); memcpy(hraw + (1 + lg_n) * 2 * kFb, R_le, lg_n * 2 * kFb); }
        // head and tail scalars: a*b, -u_j^2, -u_j^-2 (canonical)                      :234-242
        uint8_t* ends = hin.data() + o_ends;
        { uint32_t w[8]; fe_pack_words<F>(w, fe_from_mont<F>(fe_mul(a, b))); memcpy(ends, w, 32); }
        for (size_t j = 0; j < lg_n; j++) {
            uint32_t w[8];
            fe_pack_words<F>(w, fe_from_mont<F>(fe_neg(fe_sqr(ch[j])))); memcpy(ends + (1 + j) * 32, w, 32);
            fe_pack_words<F>(w, fe_from_mont<F>(fe_neg(fe_sqr(ch_inv[j])))); memcpy(ends + (1 + lg_n + j) * 32, w, 32);
        }
        hipStream_t s = ctx->stream;
        if (hipMemsetAsync(flag, 0, 4, s) != hipSuccess) return BP_ERR_DEVICE;
        if ((rc = bp_internal_stage_h2d(ctx, hin.data(), in_bytes, b_in.p))) return rc;
        if (tabled) {
            // [Q | L | R] -> pts[0 .. nx), their scalars = d_ends; the generators' scalars -> sc[1 .. 1 + 2n)
            hipLaunchKernelGGL(k_points_to_resident<C>, dim3(blocks_for(nx)), dim3(kBlock), 0, s, (const uint32_t*)raw, nx, (AffPacked<C>*)pts, flag);
            if (hipMemcpyAsync(&host_flag, flag, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return BP_ERR_DEVICE;
            hipLaunchKernelGGL(k_ipp_verify_terms<C>, dim3(blocks_for(n)), dim3(kBlock), 0, s, (const AffPacked<C>*)G->d, (const AffPacked<C>*)H->d,
                               (const ScalarWords*)Gf->d, (const ScalarWords*)Hf->d, (const ScalarWords*)chd, (const ScalarWords*)chd + lg_n, (int)lg_n,
                               fr_mont_words<F>(a), fr_mont_words<F>(b), n, (AffPacked<C>*)nullptr, (ScalarWords*)sc, (const ScalarWords*)nullptr, (ScalarWords*)nullptr);
            if (hipGetLastError() != hipSuccess) return BP_ERR_DEVICE;
            uint8_t expect[2 * kFb];
            bool done = false;
            rc = bp_internal_msm_extras_gh(ctx, pts, d_ends, nx, (const uint8_t*)sc + 32, G, H, n, expect, &done);   // synchronises both streams
            if (rc) return rc;
            if (!done) return BP_ERR_DEVICE;                                                // bp_internal_gh_ready said yes
            if (host_flag) return BP_ERR_VERIFY;
            return memcmp(expect, P_le, 2 * kFb) == 0 ? BP_OK : BP_ERR_VERIFY;
        }
        // Q -> pts[0]; L,R -> pts[1+2n ..]
        hipLaunchKernelGGL(k_points_to_resident<C>, dim3(1), dim3(kBlock), 0, s, (const uint32_t*)raw, (size_t)1, (AffPacked<C>*)pts, flag);
        if (lg_n)
            hipLaunchKernelGGL(k_points_to_resident<C>, dim3(blocks_for(2 * lg_n)), dim3(kBlock), 0, s, (const uint32_t*)((uint8_t*)raw + 2 * kFb),
                               2 * lg_n, (AffPacked<C>*)pts + 1 + 2 * n, flag);
        if (hipMemcpyAsync(&host_flag, flag, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return BP_ERR_DEVICE;
        hipLaunchKernelGGL(k_ipp_verify_terms<C>, dim3(blocks_for(n)), dim3(kBlock), 0, s, (const AffPacked<C>*)G->d, (const AffPacked<C>*)H->d,
                           (const ScalarWords*)Gf->d, (const ScalarWords*)Hf->d, (const ScalarWords*)chd, (const ScalarWords*)chd + lg_n, (int)lg_n,
                           fr_mont_words<F>(a), fr_mont_words<F>(b), n, (AffPacked<C>*)pts, (ScalarWords*)sc, d_ends, (ScalarWords*)sc + 1 + 2 * n);
        if (hipGetLastError() != hipSuccess) return BP_ERR_DEVICE;
        uint8_t expect[2 * kFb];
        rc = bp_internal_msm(ctx, pts, sc, m, expect);                                  // :251-253
        if (hipStreamSynchronize(s) != hipSuccess) rc = rc ? rc : BP_ERR_DEVICE;
        if (rc) return rc;
        if (host_flag) return BP_ERR_VERIFY;                                            // a proof point that is not a curve point
        return memcmp(expect, P_le, 2 * kFb) == 0 ? BP_OK : BP_ERR_VERIFY;              // :255-259
    }

    // m proofs, same n / generators / factors: one MSM of 2n + m (2 lg n + 2) terms that must be the identity.
    static int verify_batch(bp_ctx* ctx, size_t n, size_t lg_n, const bp_frvec* Gf, const bp_frvec* Hf, const bp_g1vec* G, const bp_g1vec* H,
                            const bp_ipp_proof_ref* proofs, size_t m, const uint8_t* weights_le32) {
        const size_t per = 2 * lg_n + 2;                 // Q, L.., R.., P
        const size_t total = 2 * n + m * per;
        std::vector<ScalarWords> hch(m * lg_n ? m * lg_n : 1), hchi(m * lg_n ? m * lg_n : 1), hwa(m), hwb(m), tail(m * per);
        std::vector<uint8_t> hraw(m * per * 2 * kFb);
        std::vector<Fe<F>> all_ch, all_inv;
        all_ch.reserve(m * lg_n);
        for (size_t p = 0; p < m; p++) {
            std::vector<Fe<F>> ch;
            int rc = verification_challenges(proofs[p].transcript->t, proofs[p].L_le, proofs[p].R_le, lg_n, n, ch);
            if (rc) return rc;
            all_ch.insert(all_ch.end(), ch.begin(), ch.end());
        }
        batch_invert(all_ch, all_inv);                   // one field inversion for the whole batch
        for (size_t p = 0; p < m; p++) {
            const bp_ipp_proof_ref& pr = proofs[p];
            const Fe<F>* ch = all_ch.data() + p * lg_n;
            const Fe<F>* ch_inv = all_inv.data() + p * lg_n;
            Fe<F> w = fr_from_le<F>(weights_le32 + 32 * p), a = fr_from_le<F>(pr.a_le32), b = fr_from_le<F>(pr.b_le32);
            Fe<F> wa = fe_mul(w, a), wb = fe_mul(w, b);
            hwa[p] = fr_mont_words<F>(wa);
            hwb[p] = fr_mont_words<F>(wb);
            ScalarWords* tl = &tail[p * per];
            uint8_t* rw = &hraw[p * per * 2 * kFb];
            auto put = [&](size_t k, const Fe<F>& x_mont) { uint32_t wd[8]; fe_pack_words<F>(wd, fe_from_mont<F>(x_mont)); memcpy(tl[k].w, wd, 32); };
            put(0, fe_mul(wa, b));                                                       // w a b   on Q
            memcpy(rw, pr.Q_le, 2 * kFb);
            for (size_t j = 0; j < lg_n; j++) {
                hch[p * lg_n + j] = fr_mont_words<F>(ch[j]);
                hchi[p * lg_n + j] = fr_mont_words<F>(ch_inv[j]);
                put(1 + j, fe_neg(fe_mul(w, fe_sqr(ch[j]))));                            // -w u_j^2   on L_j
                put(1 + lg_n + j, fe_neg(fe_mul(w, fe_sqr(ch_inv[j]))));                 // -w u_j^-2  on R_j
            }
            if (lg_n) { memcpy(rw + 2 * kFb, pr.L_le, lg_n * 2 * kFb); memcpy(rw + (1 + lg_n) * 2 * kFb, pr.R_le, lg_n * 2 * kFb); }
            put(1 + 2 * lg_n, fe_neg(w));                                                // -w        on P
            memcpy(rw + (1 + 2 * lg_n) * 2 * kFb, pr.P_le, 2 * kFb);
        }
        const size_t nch = hch.size();
        bool tabled = false;
        { int rct = bp_internal_gh_ready(ctx, G, H, n, &tabled); if (rct) return rct; }
        PoolBlock b_pts, b_sc, b_chd, b_raw;
        if (!b_pts.alloc(ctx, (tabled ? m * per : total) * kPt) || !b_sc.alloc(ctx, total * 32) || !b_chd.alloc(ctx, (2 * nch + 2 * m) * 32) || !b_raw.alloc(ctx, hraw.size()))
            return BP_ERR_DEVICE;
        void *pts = b_pts.p, *sc = b_sc.p, *chd = b_chd.p, *raw = b_raw.p;
        auto cleanup = [&]() {};
        { int rcf = ctx->flags.reserve(ctx, 64); if (rcf) return rcf; }
        uint32_t* flag = (uint32_t*)ctx->flags.p;
        uint32_t host_flag = 0;
        ScalarWords *d_ch = (ScalarWords*)chd, *d_chi = d_ch + nch, *d_wa = d_chi + nch, *d_wb = d_wa + m;
        hipStream_t s = ctx->stream;
        bool ok = hipMemsetAsync(flag, 0, 4, s) == hipSuccess &&
                  hipMemcpyAsync(d_ch, hch.data(), nch * 32, hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync(d_chi, hchi.data(), nch * 32, hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync(d_wa, hwa.data(), m * 32, hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync(d_wb, hwb.data(), m * 32, hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync(raw, hraw.data(), hraw.size(), hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync((uint8_t*)sc + 2 * n * 32, tail.data(), m * per * 32, hipMemcpyHostToDevice, s) == hipSuccess;
        if (!ok) { cleanup(); return BP_ERR_DEVICE; }
        AffPacked<C>* xp = (AffPacked<C>*)pts + (tabled ? 0 : 2 * n);      // the proofs' own points: alone in the buffer when [G | H] run over their tables
        hipLaunchKernelGGL(k_points_to_resident<C>, dim3(blocks_for(m * per)), dim3(kBlock), 0, s, (const uint32_t*)raw, m * per, xp, flag);
        if (hipMemcpyAsync(&host_flag, flag, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return BP_ERR_DEVICE;
        hipLaunchKernelGGL(k_ipp_verify_terms_batch<C>, dim3(blocks_for(n)), dim3(kBlock), 0, s, (const AffPacked<C>*)G->d, (const AffPacked<C>*)H->d,
                           (const ScalarWords*)Gf->d, (const ScalarWords*)Hf->d, d_ch, d_chi, d_wa, d_wb, (int)lg_n, m, n,
                           tabled ? (AffPacked<C>*)nullptr : (AffPacked<C>*)pts, (ScalarWords*)sc);
        if (hipGetLastError() != hipSuccess) { cleanup(); return BP_ERR_DEVICE; }
        uint8_t got[2 * kFb];
        int rc;
        if (tabled) {
            bool done = false;
            rc = bp_internal_msm_extras_gh(ctx, xp, (const uint8_t*)sc + 2 * n * 32, m * per, sc, G, H, n, got, &done);
            if (rc == BP_OK && !done) rc = BP_ERR_DEVICE;
        } else {
            rc = bp_internal_msm(ctx, pts, sc, total, got);
        }
        if (hipStreamSynchronize(s) != hipSuccess) rc = rc ? rc : BP_ERR_DEVICE;
        if (rc) return rc;
        if (host_flag) return BP_ERR_VERIFY;
        for (size_t k = 0; k < 2 * kFb; k++) if (got[k]) return BP_ERR_VERIFY;          // identity = all-zero bytes
        return BP_OK;
    }
};

#define IPP_DISPATCH(curve_, expr)                                         \
    do {                                                                   \
        if ((curve_) == BP_CURVE_BLS12_381) { using I = Ipp<Bls381>; return expr; } \
        else { using I = Ipp<Bn254>; return expr; }                        \
    } while (0)

int alloc_frvec(bp_ctx* ctx, size_t n, bp_frvec** out) { return bp_frvec_alloc(ctx, n, out); }

template <class C>
static int r1cs_prover_polys_impl(bp_ctx* ctx, const bp_frvec* const in[8], const uint8_t* y_le32, size_t n, bp_frvec* const outv[6]) {
    using F = typename C::Fr;
    Fe<F> y = fr_from_le<F>(y_le32), yi = fe_inv<F>(y);
    auto I = [&](int k) { return (const ScalarWords*)in[k]->d; };
    auto O = [&](int k) { return (ScalarWords*)outv[k]->d; };
    hipLaunchKernelGGL(k_r1cs_prover_polys<C>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, I(0), I(1), I(2), I(3), I(4), I(5), I(6), I(7),
                       fr_mont_words<F>(y), fr_mont_words<F>(yi), n, O(0), O(1), O(2), O(3), O(4), O(5));
    HIPCHK(hipGetLastError());
    return BP_OK;
}

template <class C>
static int r1cs_ipp_inputs_impl(bp_ctx* ctx, const bp_frvec* l_eval, const bp_frvec* r_eval, const uint8_t* y_le32, const uint8_t* u_le32, size_t n1,
                                size_t padded_n, bp_frvec* const outv[4]) {
    using F = typename C::Fr;
    Fe<F> y = fr_from_le<F>(y_le32), yi = fe_inv<F>(y), u = fr_from_le<F>(u_le32);
    hipLaunchKernelGGL(k_r1cs_ipp_inputs<C>, dim3(blocks_for(padded_n)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)l_eval->d,
                       (const ScalarWords*)r_eval->d, fr_mont_words<F>(y), fr_mont_words<F>(yi), fr_mont_words<F>(u), l_eval->n, n1, padded_n,
                       (ScalarWords*)outv[0]->d, (ScalarWords*)outv[1]->d, (ScalarWords*)outv[2]->d, (ScalarWords*)outv[3]->d);
    HIPCHK(hipGetLastError());
    return BP_OK;
}

template <class C>
static int r1cs_verifier_scalars_impl(bp_ctx* ctx, Transcript& t, const uint8_t* L_le, const uint8_t* R_le, size_t lg_n, size_t padded_n, size_t n1,
                                      const bp_frvec* wL, const bp_frvec* wR, const bp_frvec* wO, const uint8_t* yinv_le, const uint8_t* x_le,
                                      const uint8_t* u_le, const uint8_t* a_le, const uint8_t* b_le, uint8_t* u_sq, uint8_t* u_inv_sq, bp_frvec* g_sc,
                                      bp_frvec* h_sc) {
    using F = typename C::Fr;
    std::vector<Fe<F>> ch, ch_inv;
    int rc = Ipp<C>::verification_scalars(t, L_le, R_le, lg_n, padded_n, ch, ch_inv);     // IPP::verification_scalars, verifier.rs:354-360
    if (rc) return rc;
    std::vector<ScalarWords> hch(2 * lg_n + 1);
    for (size_t j = 0; j < lg_n; j++) {
        hch[j] = fr_mont_words<F>(ch[j]);
        hch[lg_n + j] = fr_mont_words<F>(ch_inv[j]);
        fr_to_le<F>(fe_sqr(ch[j]), u_sq + 32 * j);
        fr_to_le<F>(fe_sqr(ch_inv[j]), u_inv_sq + 32 * j);
    }
    if ((rc = ctx->scratch.reserve(ctx, (2 * lg_n + 1) * 32))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->scratch.p, hch.data(), (2 * lg_n + 1) * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_r1cs_verifier_scalars<C>, dim3(blocks_for(padded_n)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)wL->d,
                       (const ScalarWords*)wR->d, (const ScalarWords*)wO->d, (const ScalarWords*)ctx->scratch.p, (const ScalarWords*)ctx->scratch.p + lg_n,
                       (int)lg_n, fr_mont_words<F>(fr_from_le<F>(yinv_le)), fr_mont_words<F>(fr_from_le<F>(x_le)), fr_mont_words<F>(fr_from_le<F>(u_le)),
                       fr_mont_words<F>(fr_from_le<F>(a_le)), fr_mont_words<F>(fr_from_le<F>(b_le)), wL->n, n1, padded_n, (ScalarWords*)g_sc->d,
                       (ScalarWords*)h_sc->d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));   // hch is a host temporary
    return BP_OK;
}


}  // namespace

template <class C>
static int flattened_constraints_impl(bp_ctx* ctx, const bp_r1cs_plan* p, const uint8_t* z_le32, bp_frvec* outv[4], uint8_t* wc_le32) {
    using F = typename C::Fr;
    PoolBlock b_zp, b_all, b_part;
    if (!b_zp.alloc(ctx, (p->nq ? p->nq : 1) * 32) || !b_all.alloc(ctx, p->ndest * 32) || !b_part.alloc(ctx, (p->nchunks ? p->nchunks : 1) * 32)) return BP_ERR_DEVICE;
    void *zp = b_zp.p, *all = b_all.p;
    auto cleanup = [&]() {};
    hipStream_t s = ctx->stream;
    if (p->nq) hipLaunchKernelGGL(k_fr_powers_mont<C>, dim3(blocks_for(p->nq)), dim3(kBlock), 0, s, fr_mont_words<F>(fr_from_le<F>(z_le32)), p->nq, (ScalarWords*)zp);
    hipLaunchKernelGGL(k_r1cs_flatten<C>, dim3(blocks_for(p->ndest)), dim3(kBlock), 0, s, (const uint32_t*)p->seg, (const uint32_t*)p->tq,
                       (const ScalarWords*)p->coeff, (const ScalarWords*)zp, (uint32_t)(3 * p->n), (uint32_t)p->ndest, kFlattenLightMax, (ScalarWords*)all);
    if (p->nheavy) {
        hipLaunchKernelGGL(k_r1cs_flatten_heavy<C>, dim3((unsigned)(p->nchunks < 2048 ? p->nchunks : 2048)), dim3(kBlock), 0, s, (const uint32_t*)p->chunk,
                           (uint32_t)p->nchunks, (const uint32_t*)p->tq, (const ScalarWords*)p->coeff, (const ScalarWords*)zp, (ScalarWords*)b_part.p);
        hipLaunchKernelGGL(k_r1cs_flatten_heavy_final<C>, dim3(blocks_for(p->nheavy)), dim3(kBlock), 0, s, (const uint32_t*)p->heavy, (const uint32_t*)p->hfirst,
                           (uint32_t)p->nheavy, (const ScalarWords*)b_part.p, (uint32_t)(3 * p->n), (ScalarWords*)all);
    }
    if (hipGetLastError() != hipSuccess) { cleanup(); return BP_ERR_DEVICE; }
    const size_t lens[4] = {p->n, p->n, p->n, p->m}, offs[4] = {0, p->n, 2 * p->n, 3 * p->n};
    int rc = BP_OK;
    for (int k = 0; k < 4 && rc == BP_OK; k++) {
        rc = bp_frvec_alloc(ctx, lens[k], &outv[k]);
        if (rc == BP_OK && lens[k] &&
            hipMemcpyAsync(outv[k]->d, (const uint8_t*)all + offs[k] * 32, lens[k] * 32, hipMemcpyDeviceToDevice, s) != hipSuccess)
            rc = BP_ERR_DEVICE;
    }
    if (rc == BP_OK && wc_le32 && hipMemcpyAsync(wc_le32, (const uint8_t*)all + (p->ndest - 1) * 32, 32, hipMemcpyDeviceToHost, s) != hipSuccess) rc = BP_ERR_DEVICE;
    if (hipStreamSynchronize(s) != hipSuccess && rc == BP_OK) rc = BP_ERR_DEVICE;
    cleanup();
    if (rc) for (int k = 0; k < 4; k++) { bp_frvec_free(outv[k]); outv[k] = nullptr; }
    return rc;
}

template <class C>
static int commit_pairs_impl(bp_ctx* ctx, const uint8_t* g_le, const uint8_t* h_le, const bp_frvec* k1, const bp_frvec* k2, bp_g1vec* out) {
    // the two fixed points go through the same upload path as any other points (canonical bytes -> resident Montgomery rows)
    bp_g1vec* gh = nullptr;
    std::vector<uint8_t> both(4 * (size_t)(4 * C::Fp::NW));
    memcpy(both.data(), g_le, 2 * (size_t)(4 * C::Fp::NW));
    memcpy(both.data() + 2 * (size_t)(4 * C::Fp::NW), h_le, 2 * (size_t)(4 * C::Fp::NW));
    int rc = bp_g1vec_upload(ctx, both.data(), 2, BP_FMT_LE, &gh);
    if (rc) return rc;
    AffPacked<C> hostgh[2];
    if (hipMemcpyAsync(hostgh, gh->d, sizeof hostgh, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        bp_g1vec_free(gh);
        return BP_ERR_DEVICE;
    }
    bp_g1vec_free(gh);
    hipLaunchKernelGGL(k_commit_pairs<C>, dim3(blocks_for(k1->n)), dim3(kBlock), 0, ctx->stream, hostgh[0], hostgh[1], (const ScalarWords*)k1->d,
                       (const ScalarWords*)k2->d, k1->n, (AffPacked<C>*)out->d);
    HIPCHK(hipGetLastError());
    return BP_OK;
}

// ---- IPP::create_ipp with the generators sharded by index range over several contexts (second sentence) ----------
// Shard s holds G[k0_s .. k0_s + n_s), H[...], their factors and -- because the generators are never folded (bp_ipp.cuh) -- only the
// matching slices of the coefficient vectors c_G, c_H; the short vectors a, b are replicated and folded on every shard.  A round is:
// every shard computes c_L / c_R (redundantly: two inner products of the current a, b), its slice of the L / R scalars and the
// device stage of its paired MSM (asynchronously, one common window width); the host then folds the N record sets of L and of R
// (the "all-gather of the L, R partials" of the survey, through pinned host memory as in bp_msm_g1_multi), runs the transcript and
// hands u, u^-1 back to every shard's fold kernel.  Same L, R, a, b as the single-device prover, bit for bit.
namespace {
struct IppShard {
    bp_ctx* ctx = nullptr;
    size_t k0 = 0, nloc = 0;
    PoolBlock a, b, cG, cH, sL, sR, pall, cLR, partial, raw;
    bp_g1table* table = nullptr;
    ~IppShard() { if (table) bp_internal_table_free(table); }
};

template <class C>
int ipp_create_multi(bp_ctx* const* ctxs, size_t N, Transcript& t, const uint8_t* Q_le, const bp_frvec* const* Gf, const bp_frvec* const* Hf,
                     const bp_g1vec* const* G, const bp_g1vec* const* H, const uint8_t* a_le, const uint8_t* b_le, size_t n, uint8_t* L_out,
                     uint8_t* R_out, size_t* lg_n_out, uint8_t* a_out, uint8_t* b_out) {
    using F = typename C::Fr;
    using I = Ipp<C>;
    constexpr size_t kPt = sizeof(AffPacked<C>);
    constexpr int kFb = 4 * C::Fp::NW;
    std::vector<IppShard> sh(N);
    size_t k0 = 0, nmax = 0;
    bool tables = true;
    for (size_t s = 0; s < N; s++) {
        sh[s].ctx = ctxs[s]; sh[s].k0 = k0; sh[s].nloc = G[s]->n;
        k0 += G[s]->n;
        if (G[s]->n > nmax) nmax = G[s]->n;
        tables = tables && G[s]->table && H[s]->table && G[s]->table->c == G[0]->table->c && H[s]->table->c == G[0]->table->c;
    }
    int rc;
    // canonical scalars only (as bp_frvec_upload checks): validate a, b once, on the first shard
    {
        bp_frvec *ta = nullptr, *tb = nullptr;
        rc = bp_frvec_upload(ctxs[0], a_le, n, &ta);
        if (!rc) rc = bp_frvec_upload(ctxs[0], b_le, n, &tb);
        bp_frvec_free(ta); bp_frvec_free(tb);
        if (rc) return rc;
    }
    const uint8_t zero_pt[2 * kFb] = {0};
    for (size_t s = 0; s < N; s++) {
        IppShard& x = sh[s];
        bp_ctx* ctx = x.ctx;
        if ((rc = bp_internal_set_device(ctx))) return rc;
        const size_t nl = x.nloc, m = 2 * nl + 1;
        if (!x.a.alloc(ctx, n * 32) || !x.b.alloc(ctx, n * 32) || !x.cG.alloc(ctx, nl * 32) || !x.cH.alloc(ctx, nl * 32) || !x.sL.alloc(ctx, m * 32) ||
            !x.sR.alloc(ctx, m * 32) || !x.pall.alloc(ctx, m * kPt) || !x.cLR.alloc(ctx, 64) || !x.partial.alloc(ctx, (kInnerBlocks + 1) * 32) || !x.raw.alloc(ctx, kPt))
            return BP_ERR_DEVICE;
        hipStream_t st = ctx->stream;
        if ((rc = ctx->flags.reserve(ctx, 64))) return rc;
        uint32_t host_flag = 0;
        HIPCHK(hipMemsetAsync(ctx->flags.p, 0, 4, st));
        HIPCHK(hipMemcpyAsync(x.a.p, a_le, n * 32, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x.b.p, b_le, n * 32, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x.cG.p, Gf[s]->d, nl * 32, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(x.cH.p, Hf[s]->d, nl * 32, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(x.pall.p, G[s]->d, nl * kPt, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync((uint8_t*)x.pall.p + nl * kPt, H[s]->d, nl * kPt, hipMemcpyDeviceToDevice, st));
        // last term: Q on the first shard (validated like any point from outside), the identity elsewhere
        HIPCHK(hipMemcpyAsync(x.raw.p, s == 0 ? Q_le : zero_pt, 2 * kFb, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_points_to_resident<C>, dim3(1), dim3(kBlock), 0, st, (const uint32_t*)x.raw.p, (size_t)1, (AffPacked<C>*)x.pall.p + 2 * nl, (uint32_t*)ctx->flags.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&host_flag, ctx->flags.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));               // a_le / b_le / Q_le are borrowed host buffers
        if (host_flag) return BP_ERR_ARG;
        if (tables && (rc = bp_internal_table_concat(ctx, G[s], 0, H[s], 0, nl, s == 0 ? Q_le : zero_pt, &x.table))) return rc;
        if (tables && !x.table) tables = false;
    }
    if (!tables) for (auto& x : sh) if (x.table) { bp_internal_table_free(x.table); x.table = nullptr; }
    const int c = tables ? 0 : bp_internal_pair_width(ctxs[0], 2 * nmax + 1, nmax + 1);
    if (!tables && c <= 0) return BP_ERR_ARG;

    I::ipp_domain_sep(t, n);
    std::vector<uint16_t> rpos(kMaxRecords);
    std::vector<XyzzPacked<C>> recL, recR;
    size_t nj = n, k = 0;
    while (nj != 1) {
        const size_t h = nj / 2;
        int nrec = 0;
        for (size_t s = 0; s < N; s++) {                       // queue every shard's round; nothing waits here
            IppShard& x = sh[s];
            bp_ctx* ctx = x.ctx;
            if ((rc = bp_internal_set_device(ctx))) return rc;
            auto* a = (ScalarWords*)x.a.p; auto* b = (ScalarWords*)x.b.p;
            unsigned g = blocks_for(h);
            if (g > kInnerBlocks / 2) g = kInnerBlocks / 2;
            if (g == 0) g = 1;
            hipLaunchKernelGGL(k_fr_inner2<C>, dim3(g, 2), dim3(kBlock), 0, ctx->stream, a, b + h, a + h, b, h, (ScalarWords*)x.partial.p);
            hipLaunchKernelGGL(k_fr_inner2_final<C>, dim3(2), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)x.partial.p, g, (ScalarWords*)x.cLR.p);
            hipLaunchKernelGGL(k_ipp_round_scalars<C>, dim3(blocks_for(x.nloc)), dim3(kBlock), 0, ctx->stream, a, b, (const ScalarWords*)x.cG.p,
                               (const ScalarWords*)x.cH.p, (const ScalarWords*)x.cLR.p, x.nloc, x.k0, nj, (ScalarWords*)x.sL.p, (ScalarWords*)x.sR.p, s == 0 ? 1 : 0);
            HIPCHK(hipGetLastError());
            int nr = 0;
            if ((rc = bp_internal_msm2_begin(ctx, x.pall.p, x.sL.p, x.sR.p, 2 * x.nloc + 1, c, x.nloc + 1, x.table, &nr, rpos.data()))) return rc;
            if (s && nr != nrec) return BP_ERR_DEVICE;            // one geometry for all shards
            nrec = nr;
        }
        const int R1 = nrec / 2;
        recL.resize(N * (size_t)R1);
        recR.resize(N * (size_t)R1);
        for (size_t s = 0; s < N; s++) {
            if ((rc = bp_internal_set_device(sh[s].ctx))) return rc;
            HIPCHK(hipStreamSynchronize(sh[s].ctx->stream));
            const XyzzPacked<C>* r = (const XyzzPacked<C>*)sh[s].ctx->host_pinned;
            memcpy(&recL[s * (size_t)R1], r, (size_t)R1 * sizeof(XyzzPacked<C>));
            memcpy(&recR[s * (size_t)R1], r + R1, (size_t)R1 * sizeof(XyzzPacked<C>));
        }
        uint8_t* L = L_out + k * 2 * kFb;
        uint8_t* R = R_out + k * 2 * kFb;
        const void* recs[2] = {recL.data(), recR.data()};
        const uint16_t* poss[2] = {rpos.data(), rpos.data() + R1};
        uint8_t* outs[2] = {L, R};
        if ((rc = bp_internal_fold_sets(ctxs[0], 2, recs, N, R1, poss, outs))) return rc;
        I::commit_point(t, "L", L);
        I::commit_point(t, "R", R);
        Fe<F> u = I::challenge_scalar(t, "u");
        Fe<F> ui = fr_inv_fast<F>(u);
        for (size_t s = 0; s < N; s++) {
            IppShard& x = sh[s];
            if ((rc = bp_internal_set_device(x.ctx))) return rc;
            const size_t span = x.nloc > h ? x.nloc : h;
            hipLaunchKernelGGL(k_ipp_fold_scalars<C>, dim3(blocks_for(span)), dim3(kBlock), 0, x.ctx->stream, (ScalarWords*)x.a.p, (ScalarWords*)x.b.p,
                               (ScalarWords*)x.cG.p, (ScalarWords*)x.cH.p, fr_mont_words<F>(u), fr_mont_words<F>(ui), x.nloc, x.k0, nj);
            HIPCHK(hipGetLastError());
        }
        nj = h;
        k++;
    }
    if (lg_n_out) *lg_n_out = k;
    if ((rc = bp_internal_set_device(ctxs[0]))) return rc;
    HIPCHK(hipMemcpyAsync(a_out, sh[0].a.p, 32, hipMemcpyDeviceToHost, ctxs[0]->stream));
    HIPCHK(hipMemcpyAsync(b_out, sh[0].b.p, 32, hipMemcpyDeviceToHost, ctxs[0]->stream));
    for (size_t s = 0; s < N; s++) {                           // the shards' blocks return to their pools: nothing may be in flight
        if ((rc = bp_internal_set_device(ctxs[s]))) return rc;
        HIPCHK(hipStreamSynchronize(ctxs[s]->stream));
    }
    return BP_OK;
}
}  // namespace


// Compaction table of a vector (bp_g1vec_precompute): affine digit multiples m 2^(64 k) P_i (m = 1 .. 8, k < 4) from the rows
// w = 64 k / c of its window-multiples table.  *out stays NULL when c does not divide 64 (the rows do not exist).
template <class C>
static int ctable_build_impl(bp_ctx* ctx, const bp_g1table* wt, bp_g1table** out) {
    constexpr size_t kPt = sizeof(AffPacked<C>);
    constexpr int K = 4;
    const size_t n = wt->n, rows = (size_t)1 << (kSmallDigitBits - 1);
    const int step = 64 / wt->c;
    if ((uint64_t)rows * K * n >= ((uint64_t)1 << 31)) return BP_OK;
    PoolBlock gathered, tmp;
    if (!gathered.alloc(ctx, K * n * kPt) || !tmp.alloc(ctx, rows * K * n * sizeof(XyzzPacked<C>))) return BP_ERR_DEVICE;
    bp_g1table* t = new (std::nothrow) bp_g1table();
    if (!t) return BP_ERR_DEVICE;
    t->pool = ctx->pool; t->device = ctx->device; t->n = n; t->c = kSmallDigitBits; t->W = (int)rows; t->digits = true; t->affine = true; t->K = K;
    t->d = ctx->pool->get(rows * K * n * kPt, &t->cap);
    if (!t->d) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
    hipStream_t s = ctx->stream;
    for (int k = 0; k < K; k++) {
        if ((size_t)k * step >= (size_t)wt->W) { bp_internal_table_free(t); return BP_OK; }      // (cannot happen for c | 64: W >= 256 / c)
        if (hipMemcpyAsync((uint8_t*)gathered.p + (size_t)k * n * kPt, (const uint8_t*)wt->d + (size_t)k * step * n * kPt, n * kPt, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            bp_internal_table_free(t); return BP_ERR_DEVICE;
        }
    }
    hipLaunchKernelGGL(k_digit_table_build<C>, dim3((unsigned)((K * n + 63) / 64)), dim3(64), 0, s, (const AffPacked<C>*)gathered.p, (uint32_t)(K * n), (XyzzPacked<C>*)tmp.p, (uint32_t)rows);
    int rc = hipGetLastError() == hipSuccess ? Ipp<C>::batch_to_affine(ctx, (const XyzzPacked<C>*)tmp.p, rows * K * n, (AffPacked<C>*)t->d, s, nullptr) : BP_ERR_DEVICE;
    if (rc) { bp_internal_table_free(t); return rc; }
    *out = t;
    return BP_OK;
}
int bp_internal_ctable_build(bp_ctx* ctx, const bp_g1table* wt, bp_g1table** out) {
    *out = nullptr;
    if (!wt || wt->digits || wt->c <= 0 || 64 % wt->c) return BP_OK;
    try {
        return ctx->curve == BP_CURVE_BLS12_381 ? ctable_build_impl<Bls381>(ctx, wt, out) : ctable_build_impl<Bn254>(ctx, wt, out);
    } catch (...) { return BP_ERR_DEVICE; }
}

extern "C" {

// ---- transcript ---------------------------------------------------------------------------------------------
int bp_transcript_new(const uint8_t* label, size_t label_len, bp_transcript** out) {
    return bp_guard([&]() -> int {
    if (!out || (!label && label_len)) return BP_ERR_ARG;
    *out = new (std::nothrow) bp_transcript(label, label_len);
    return *out ? BP_OK : BP_ERR_DEVICE;
    });
}
int bp_transcript_free(bp_transcript* t) { delete t; return BP_OK; }
int bp_transcript_append_message(bp_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    return bp_guard([&]() -> int {
    if (!t) return BP_ERR_ARG;
    t->t.append_message(label, label_len, msg, msg_len);
    return BP_OK;
    });
}
int bp_transcript_append_u64(bp_transcript* t, const uint8_t* label, size_t label_len, uint64_t x) {
    return bp_guard([&]() -> int {
    if (!t) return BP_ERR_ARG;
    t->t.append_u64(label, label_len, x);
    return BP_OK;
    });
}
int bp_transcript_challenge_bytes(bp_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len) {
    return bp_guard([&]() -> int {
    if (!t || (!out && out_len)) return BP_ERR_ARG;
    t->t.challenge_bytes(label, label_len, out, out_len);
    return BP_OK;
    });
}
int bp_transcript_commit_point(bp_transcript* t, int curve_id, const char* label, const uint8_t* point_le) {
    return bp_guard([&]() -> int {
    if (!t || !curve_ok(curve_id) || !label || !point_le) return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) Ipp<Bls381>::commit_point(t->t, label, point_le); else Ipp<Bn254>::commit_point(t->t, label, point_le);
    return BP_OK;
    });
}
int bp_transcript_commit_points(bp_transcript* t, int curve_id, const char* label, const uint8_t* points_le, size_t n) {
    return bp_guard([&]() -> int {
    if (!t || !curve_ok(curve_id) || !label || (!points_le && n)) return BP_ERR_ARG;
    const size_t pb = curve_id == BP_CURVE_BLS12_381 ? 2 * 4 * Bls381::Fp::NW : 2 * 4 * Bn254::Fp::NW;
    for (size_t i = 0; i < n; i++) {
        if (curve_id == BP_CURVE_BLS12_381) Ipp<Bls381>::commit_point(t->t, label, points_le + i * pb); else Ipp<Bn254>::commit_point(t->t, label, points_le + i * pb);
    }
    return BP_OK;
    });
}
int bp_transcript_commit_scalar(bp_transcript* t, int curve_id, const char* label, const uint8_t* scalar_le32) {
    return bp_guard([&]() -> int {
    if (!t || !curve_ok(curve_id) || !label || !scalar_le32) return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) Ipp<Bls381>::commit_scalar(t->t, label, scalar_le32); else Ipp<Bn254>::commit_scalar(t->t, label, scalar_le32);
    return BP_OK;
    });
}
int bp_transcript_challenge_scalar(bp_transcript* t, int curve_id, const char* label, uint8_t* out_le32) {
    return bp_guard([&]() -> int {
    if (!t || !curve_ok(curve_id) || !label || !out_le32) return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) fr_to_le<Bls381Fr>(Ipp<Bls381>::challenge_scalar(t->t, label), out_le32);
    else fr_to_le<Bn254Fr>(Ipp<Bn254>::challenge_scalar(t->t, label), out_le32);
    return BP_OK;
    });
}

// ---- host Fr helpers ----------------------------------------------------------------------------------------
int bp_fr_inverse(int curve_id, const uint8_t* in_le32, uint8_t* out_le32) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || !in_le32 || !out_le32) return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) fr_to_le<Bls381Fr>(fr_inv_fast<Bls381Fr>(fr_from_le<Bls381Fr>(in_le32)), out_le32);
    else fr_to_le<Bn254Fr>(fr_inv_fast<Bn254Fr>(fr_from_le<Bn254Fr>(in_le32)), out_le32);
    return BP_OK;
    });
}

// FieldElement::random() (src/r1cs/verifier.rs:392 and the provers' blindings): n uniform non-zero scalars from the OS
// (getrandom), by rejection of fr_bits-bit draws.
int bp_fr_random(int curve_id, uint8_t* out_le32, size_t n) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || (!out_le32 && n)) return BP_ERR_ARG;
    const int bits = curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS;
    for (size_t i = 0; i < n; i++) {
        uint8_t* o = out_le32 + 32 * i;
        for (int tries = 0;; tries++) {
            if (tries > 1000) return BP_ERR_DEVICE;
            size_t got = 0;
            while (got < 32) {
                ssize_t k = getrandom(o + got, 32 - got, 0);
                if (k < 0) { if (errno == EINTR) continue; return BP_ERR_DEVICE; }
                got += (size_t)k;
            }
            if (bits < 256) o[31] &= (uint8_t)((1u << (bits - 248)) - 1);
            if (bp_fr_is_canonical_nonzero(curve_id, o)) break;
        }
    }
    return BP_OK;
    });
}

int bp_fr_is_canonical_nonzero(int curve_id, const uint8_t* x_le32) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || !x_le32) return 0;
    uint32_t w[8];
    memcpy(w, x_le32, 32);
    uint32_t any = 0;
    for (int i = 0; i < 8; i++) any |= w[i];
    if (!any) return 0;
    return curve_id == BP_CURVE_BLS12_381 ? (int)words_lt_mod<Bls381Fr>(w) : (int)words_lt_mod<Bn254Fr>(w);
    });
}

// ---- FieldElementVector kernels -----------------------------------------------------------------------------
int bp_fr_inner_product(bp_ctx* ctx, const bp_frvec* a, size_t aoff, const bp_frvec* b, size_t boff, size_t n, uint8_t* out_le32) {
    return bp_guard([&]() -> int {
    if (!ctx || !a || !b || !out_le32) return BP_ERR_ARG;
    if (aoff > a->n || n > a->n - aoff || boff > b->n || n > b->n - boff) return BP_ERR_LENGTH;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    if ((rc = ctx->scratch.reserve(ctx, (kInnerBlocks + 2) * 32))) return rc;
    auto* part = (ScalarWords*)ctx->scratch.p;
    if (ctx->curve == BP_CURVE_BLS12_381) rc = Ipp<Bls381>::inner(ctx, (const ScalarWords*)a->d + aoff, (const ScalarWords*)b->d + boff, n, part + 1, part);
    else rc = Ipp<Bn254>::inner(ctx, (const ScalarWords*)a->d + aoff, (const ScalarWords*)b->d + boff, n, part + 1, part);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out_le32, part, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BP_OK;
    });
}

int bp_fr_hadamard(bp_ctx* ctx, const bp_frvec* a, const bp_frvec* b, bp_frvec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !a || !b || !out) return BP_ERR_ARG;
    if (a->n != b->n) return BP_ERR_LENGTH;
    int rc = alloc_frvec(ctx, a->n, out); if (rc) return rc;
    if (a->n == 0) return BP_OK;
    if (ctx->curve == BP_CURVE_BLS12_381)
        hipLaunchKernelGGL(k_fr_hadamard<Bls381>, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)a->d, (const ScalarWords*)b->d, a->n, (ScalarWords*)(*out)->d);
    else
        hipLaunchKernelGGL(k_fr_hadamard<Bn254>, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)a->d, (const ScalarWords*)b->d, a->n, (ScalarWords*)(*out)->d);
    HIPCHK(hipGetLastError());
    return BP_OK;
    });
}

int bp_fr_scaled_by(bp_ctx* ctx, const bp_frvec* a, const uint8_t* s_le32, bp_frvec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !a || !s_le32 || !out) return BP_ERR_ARG;
    int rc = alloc_frvec(ctx, a->n, out); if (rc) return rc;
    if (a->n == 0) return BP_OK;
    if (ctx->curve == BP_CURVE_BLS12_381)
        hipLaunchKernelGGL(k_fr_scale<Bls381>, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)a->d, fr_mont_words<Bls381Fr>(fr_from_le<Bls381Fr>(s_le32)), a->n, (ScalarWords*)(*out)->d);
    else
        hipLaunchKernelGGL(k_fr_scale<Bn254>, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)a->d, fr_mont_words<Bn254Fr>(fr_from_le<Bn254Fr>(s_le32)), a->n, (ScalarWords*)(*out)->d);
    HIPCHK(hipGetLastError());
    return BP_OK;
    });
}

// out[i] = the two halves of a[i] under the curve's endomorphism, as the prover's kernels use them (bp_compact.cuh: glv_decompose):
// 16 bytes s1 then 16 bytes s2, little-endian, a[i] = s1 + s2 LAMBDA mod r.  BLS12-381: both halves plain numbers < 2^128; BN254: s2 mod
// 2^128, negative when >= 2^66 (bp_curve.cuh).  Same length and byte size as a scalar vector, so it travels as one.
int bp_fr_glv_split(bp_ctx* ctx, const bp_frvec* a, bp_frvec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !a || !out) return BP_ERR_ARG;
    int rc = alloc_frvec(ctx, a->n, out); if (rc) return rc;
    if (a->n == 0) return BP_OK;
    if (ctx->curve == BP_CURVE_BLS12_381)
        hipLaunchKernelGGL(k_glv_decompose<Bls381>, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)a->d, (const ScalarWords*)nullptr, a->n, (ScalarWords*)(*out)->d, (ScalarWords*)nullptr);
    else
        hipLaunchKernelGGL(k_glv_decompose<Bn254>, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)a->d, (const ScalarWords*)nullptr, a->n, (ScalarWords*)(*out)->d, (ScalarWords*)nullptr);
    HIPCHK(hipGetLastError());
    return BP_OK;
    });
}

int bp_fr_vandermonde(bp_ctx* ctx, const uint8_t* e_le32, size_t n, bp_frvec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !e_le32 || !out) return BP_ERR_ARG;
    int rc = alloc_frvec(ctx, n, out); if (rc) return rc;
    if (n == 0) return BP_OK;
    if (ctx->curve == BP_CURVE_BLS12_381)
        hipLaunchKernelGGL(k_fr_vandermonde<Bls381>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, fr_mont_words<Bls381Fr>(fr_from_le<Bls381Fr>(e_le32)), n, (ScalarWords*)(*out)->d);
    else
        hipLaunchKernelGGL(k_fr_vandermonde<Bn254>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, fr_mont_words<Bn254Fr>(fr_from_le<Bn254Fr>(e_le32)), n, (ScalarWords*)(*out)->d);
    HIPCHK(hipGetLastError());
    return BP_OK;
    });
}

// ---- vector polynomials (src/utils/vector_poly.rs) ----------------------------------------------------------
static int same_len(const bp_frvec* const* v, int k, size_t* n) {
    for (int i = 0; i < k; i++) if (!v[i]) return BP_ERR_ARG;
    *n = v[0]->n;
    for (int i = 1; i < k; i++) if (v[i]->n != *n) return BP_ERR_LENGTH;
    return BP_OK;
}

int bp_vecpoly3_special_inner_product(bp_ctx* ctx, const bp_frvec* const lhs[4], const bp_frvec* const rhs[4], uint8_t* out_t1_to_t6) {
    return bp_guard([&]() -> int {
    if (!ctx || !lhs || !rhs || !out_t1_to_t6) return BP_ERR_ARG;
    size_t n, n2;
    int rc;
    if ((rc = same_len(lhs, 4, &n)) || (rc = same_len(rhs, 4, &n2))) return rc;
    if (n != n2) return BP_ERR_LENGTH;
    if ((rc = bp_internal_set_device(ctx))) return rc;
    if ((rc = ctx->scratch.reserve(ctx, (6 * kInnerBlocks + 8) * 32))) return rc;
    auto* out = (ScalarWords*)ctx->scratch.p;
    auto* part = out + 8;
    unsigned g = blocks_for(n);
    if (g > kInnerBlocks) g = kInnerBlocks;
    if (g == 0) g = 1;
    auto L = [&](int i) { return (const ScalarWords*)lhs[i]->d; };
    auto R = [&](int i) { return (const ScalarWords*)rhs[i]->d; };
    if (ctx->curve == BP_CURVE_BLS12_381) {
        hipLaunchKernelGGL(k_vecpoly3_special<Bls381>, dim3(g), dim3(kBlock), 0, ctx->stream, L(1), L(2), L(3), R(0), R(1), R(3), n, part);
        hipLaunchKernelGGL(k_fr_multi_final<Bls381>, dim3(1), dim3(kBlock), 0, ctx->stream, part, g, 6u, out);
    } else {
        hipLaunchKernelGGL(k_vecpoly3_special<Bn254>, dim3(g), dim3(kBlock), 0, ctx->stream, L(1), L(2), L(3), R(0), R(1), R(3), n, part);
        hipLaunchKernelGGL(k_fr_multi_final<Bn254>, dim3(1), dim3(kBlock), 0, ctx->stream, part, g, 6u, out);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_t1_to_t6, out, 6 * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BP_OK;
    });
}

int bp_vecpoly1_inner_product(bp_ctx* ctx, const bp_frvec* const l[2], const bp_frvec* const r[2], uint8_t* out_t0_t1_t2) {
    return bp_guard([&]() -> int {
    if (!ctx || !l || !r || !out_t0_t1_t2) return BP_ERR_ARG;
    size_t n, n2;
    int rc;
    if ((rc = same_len(l, 2, &n)) || (rc = same_len(r, 2, &n2))) return rc;
    if (n != n2) return BP_ERR_LENGTH;
    if ((rc = bp_internal_set_device(ctx))) return rc;
    if ((rc = ctx->scratch.reserve(ctx, (3 * kInnerBlocks + 8) * 32))) return rc;
    auto* out = (ScalarWords*)ctx->scratch.p;
    auto* part = out + 8;
    unsigned g = blocks_for(n);
    if (g > kInnerBlocks) g = kInnerBlocks;
    if (g == 0) g = 1;
    if (ctx->curve == BP_CURVE_BLS12_381) {
        hipLaunchKernelGGL(k_vecpoly1_inner<Bls381>, dim3(g), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)l[0]->d, (const ScalarWords*)l[1]->d,
                           (const ScalarWords*)r[0]->d, (const ScalarWords*)r[1]->d, n, part);
        hipLaunchKernelGGL(k_fr_multi_final<Bls381>, dim3(1), dim3(kBlock), 0, ctx->stream, part, g, 3u, out);
    } else {
        hipLaunchKernelGGL(k_vecpoly1_inner<Bn254>, dim3(g), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)l[0]->d, (const ScalarWords*)l[1]->d,
                           (const ScalarWords*)r[0]->d, (const ScalarWords*)r[1]->d, n, part);
        hipLaunchKernelGGL(k_fr_multi_final<Bn254>, dim3(1), dim3(kBlock), 0, ctx->stream, part, g, 3u, out);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_t0_t1_t2, out, 3 * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BP_OK;
    });
}

int bp_vecpoly_eval(bp_ctx* ctx, const bp_frvec* const* p, int degree, const uint8_t* x_le32, bp_frvec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !p || !x_le32 || !out || (degree != 1 && degree != 3)) return BP_ERR_ARG;
    size_t n;
    int rc;
    if ((rc = same_len(p, degree + 1, &n))) return rc;
    if ((rc = alloc_frvec(ctx, n, out))) return rc;
    if (n == 0) return BP_OK;
    const ScalarWords* q[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i <= degree; i++) q[i] = (const ScalarWords*)p[i]->d;
    if (ctx->curve == BP_CURVE_BLS12_381)
        hipLaunchKernelGGL(k_vecpoly_eval<Bls381>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, q[0], q[1], q[2], q[3], degree,
                           fr_mont_words<Bls381Fr>(fr_from_le<Bls381Fr>(x_le32)), n, (ScalarWords*)(*out)->d);
    else
        hipLaunchKernelGGL(k_vecpoly_eval<Bn254>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, q[0], q[1], q[2], q[3], degree,
                           fr_mont_words<Bn254Fr>(fr_from_le<Bn254Fr>(x_le32)), n, (ScalarWords*)(*out)->d);
    HIPCHK(hipGetLastError());
    return BP_OK;
    });
}

// ---- R1CS vector pipeline (src/r1cs/prover.rs:458-563, src/r1cs/verifier.rs:342-390) ------------------------

int bp_r1cs_prover_polys(bp_ctx* ctx, const bp_frvec* const in[8], const uint8_t* y_le32, bp_frvec* out[6]) {
    return bp_guard([&]() -> int {
    if (!ctx || !in || !y_le32 || !out) return BP_ERR_ARG;
    size_t n;
    int rc;
    if ((rc = same_len(in, 8, &n))) return rc;
    if ((rc = bp_internal_set_device(ctx))) return rc;
    for (int k = 0; k < 6; k++) out[k] = nullptr;
    for (int k = 0; k < 6; k++) if ((rc = alloc_frvec(ctx, n, &out[k]))) { for (int j = 0; j < k; j++) bp_frvec_free(out[j]); return rc; }
    if (n == 0) return BP_OK;
    rc = ctx->curve == BP_CURVE_BLS12_381 ? r1cs_prover_polys_impl<Bls381>(ctx, in, y_le32, n, out) : r1cs_prover_polys_impl<Bn254>(ctx, in, y_le32, n, out);
    if (rc) for (int k = 0; k < 6; k++) { bp_frvec_free(out[k]); out[k] = nullptr; }
    return rc;
    });
}


int bp_r1cs_ipp_inputs(bp_ctx* ctx, const bp_frvec* l_eval, const bp_frvec* r_eval, const uint8_t* y_le32, const uint8_t* u_le32, size_t n1,
                       size_t padded_n, bp_frvec* out[4]) {
    return bp_guard([&]() -> int {
    if (!ctx || !l_eval || !r_eval || !y_le32 || !u_le32 || !out) return BP_ERR_ARG;
    if (l_eval->n != r_eval->n) return BP_ERR_LENGTH;
    if (padded_n < l_eval->n || n1 > l_eval->n || padded_n == 0 || (padded_n & (padded_n - 1))) return BP_ERR_ARG;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    for (int k = 0; k < 4; k++) out[k] = nullptr;
    for (int k = 0; k < 4; k++) if ((rc = alloc_frvec(ctx, padded_n, &out[k]))) { for (int j = 0; j < k; j++) bp_frvec_free(out[j]); return rc; }
    rc = ctx->curve == BP_CURVE_BLS12_381 ? r1cs_ipp_inputs_impl<Bls381>(ctx, l_eval, r_eval, y_le32, u_le32, n1, padded_n, out)
                                          : r1cs_ipp_inputs_impl<Bn254>(ctx, l_eval, r_eval, y_le32, u_le32, n1, padded_n, out);
    if (rc) for (int k = 0; k < 4; k++) { bp_frvec_free(out[k]); out[k] = nullptr; }
    return rc;
    });
}


int bp_r1cs_verifier_scalars(bp_ctx* ctx, bp_transcript* t, const uint8_t* L_le, const uint8_t* R_le, size_t lg_n, size_t padded_n, size_t n1,
                             const bp_frvec* wL, const bp_frvec* wR, const bp_frvec* wO, const uint8_t* y_inv_le32, const uint8_t* x_le32,
                             const uint8_t* u_le32, const uint8_t* a_le32, const uint8_t* b_le32, uint8_t* u_sq_out, uint8_t* u_inv_sq_out,
                             bp_frvec** g_scalars, bp_frvec** h_scalars) {
    return bp_guard([&]() -> int {
    if (!ctx || !t || !wL || !wR || !wO || !y_inv_le32 || !x_le32 || !u_le32 || !a_le32 || !b_le32 || !u_sq_out || !u_inv_sq_out || !g_scalars ||
        !h_scalars || (lg_n && (!L_le || !R_le)))
        return BP_ERR_ARG;
    if (wL->n != wR->n || wL->n != wO->n) return BP_ERR_LENGTH;
    if (lg_n >= 32 || padded_n != ((size_t)1 << lg_n)) return BP_ERR_VERIFY;
    if (wL->n > padded_n || n1 > wL->n) return BP_ERR_ARG;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    *g_scalars = *h_scalars = nullptr;
    if ((rc = alloc_frvec(ctx, padded_n, g_scalars))) return rc;
    if ((rc = alloc_frvec(ctx, padded_n, h_scalars))) { bp_frvec_free(*g_scalars); *g_scalars = nullptr; return rc; }
    rc = ctx->curve == BP_CURVE_BLS12_381
             ? r1cs_verifier_scalars_impl<Bls381>(ctx, t->t, L_le, R_le, lg_n, padded_n, n1, wL, wR, wO, y_inv_le32, x_le32, u_le32, a_le32, b_le32,
                                                  u_sq_out, u_inv_sq_out, *g_scalars, *h_scalars)
             : r1cs_verifier_scalars_impl<Bn254>(ctx, t->t, L_le, R_le, lg_n, padded_n, n1, wL, wR, wO, y_inv_le32, x_le32, u_le32, a_le32, b_le32,
                                                 u_sq_out, u_inv_sq_out, *g_scalars, *h_scalars);
    if (rc) { bp_frvec_free(*g_scalars); bp_frvec_free(*h_scalars); *g_scalars = *h_scalars = nullptr; }
    return rc;
    });
}

// ---- IPP device-resident state ------------------------------------------------------------------------------
int bp_ipp_state_free(bp_ipp_state* st) {
    return bp_guard([&]() -> int {
    if (!st) return BP_OK;
    if (st->side_pending) (void)hipEventSynchronize(st->ev_side);      // blocks in use on the sibling stream must not return to the pool yet
    if (st->ev_side) (void)hipEventDestroy(st->ev_side);
    if (st->table) bp_internal_table_free(st->table);
    if (st->blocks) {
        for (auto& b : *st->blocks) st->pool->put(b.first, b.second);
        delete st->blocks;
    }
    delete st;
    return BP_OK;
    });
}

int bp_ipp_state_create(bp_ctx* ctx, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* Q_le, const bp_frvec* Gf, const bp_frvec* Hf,
                        const bp_frvec* a, const bp_frvec* b, bp_ipp_state** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !G || !H || !Q_le || !Gf || !Hf || !a || !b || !out) return BP_ERR_ARG;
    *out = nullptr;
    size_t n = G->n;
    if (n == 0 || (n & (n - 1))) return BP_ERR_ARG;                                                  // assert!(n.is_power_of_two())  ipp.rs:48
    if (H->n != n || a->n != n || b->n != n || Gf->n != n || Hf->n != n) return BP_ERR_ARG;          // assert_eq! lengths            ipp.rs:51-55
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    size_t pt = 2 * (size_t)fp_bytes_of(ctx->curve);
    bp_ipp_state* st = new (std::nothrow) bp_ipp_state();
    if (!st) return BP_ERR_DEVICE;
    memset(st, 0, sizeof *st);
    st->ctx = ctx; st->n0 = st->n = n; st->first = true; st->device = ctx->device;
    st->fold_generators = ctx->ipp_fold_generators;
    st->pool = ctx->pool;
    st->blocks = new (std::nothrow) std::vector<std::pair<void*, size_t>>();
    if (!st->blocks) { delete st; return BP_ERR_DEVICE; }
    hipStream_t s = ctx->stream;
    if ((rc = ctx->flags.reserve(ctx, 64))) { bp_ipp_state_free(st); return rc; }
    uint32_t* flag = (uint32_t*)ctx->flags.p;
    uint32_t host_flag = 0;
    bool ok = st->take(&st->a, n * 32) && st->take(&st->b, n * 32) && st->take(&st->cLR, 64) && st->take(&st->partial, (kInnerBlocks + 1) * 32) &&
              st->take(&st->Q, pt) && st->take(&st->pts_tmp, 2 * (n + 1) * pt) && hipMemsetAsync(flag, 0, 4, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(st->a, a->d, n * 32, hipMemcpyDeviceToDevice, s) == hipSuccess &&                       // clones, ipp.rs:57-60
         hipMemcpyAsync(st->b, b->d, n * 32, hipMemcpyDeviceToDevice, s) == hipSuccess;
    if (ok) {
        // Q: host bytes -> resident form (pts_tmp doubles as the raw staging area)
        ok = hipMemcpyAsync(st->pts_tmp, Q_le, pt, hipMemcpyHostToDevice, s) == hipSuccess;
        if (ok) {
            if (ctx->curve == BP_CURVE_BLS12_381)
                hipLaunchKernelGGL(k_points_to_resident<Bls381>, dim3(1), dim3(kBlock), 0, s, (const uint32_t*)st->pts_tmp, (size_t)1, (AffPacked<Bls381>*)st->Q, flag);
            else
                hipLaunchKernelGGL(k_points_to_resident<Bn254>, dim3(1), dim3(kBlock), 0, s, (const uint32_t*)st->pts_tmp, (size_t)1, (AffPacked<Bn254>*)st->Q, flag);
            ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&host_flag, flag, 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipStreamSynchronize(s) == hipSuccess;
            if (ok && host_flag) { bp_ipp_state_free(st); return BP_ERR_ARG; }      // Q is not a point of the curve
        }
    }
    if (ok && st->fold_generators) {
        // reference-shaped mode: working copies of G, H are folded in place every round (k_ipp_fold)
        ok = st->take(&st->G, n * pt) && st->take(&st->H, n * pt) && st->take(&st->gf, n * 32) && st->take(&st->hf, n * 32) &&
             st->take(&st->sc_tmp, 2 * (n + 1) * 32) &&
             hipMemcpyAsync(st->G, G->d, n * pt, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipMemcpyAsync(st->H, H->d, n * pt, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipMemcpyAsync(st->gf, Gf->d, n * 32, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipMemcpyAsync(st->hf, Hf->d, n * 32, hipMemcpyDeviceToDevice, s) == hipSuccess;
    } else if (ok) {
        // default: [G | H | Q] resident and never modified; coefficients start as the factors
        size_t m = 2 * n + 1;
        ok = st->take(&st->Pall, m * pt) && st->take(&st->cG, n * 32) && st->take(&st->cH, n * 32) && st->take(&st->sL, m * 32) && st->take(&st->sR, m * 32) &&
             hipMemcpyAsync(st->Pall, G->d, n * pt, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipMemcpyAsync((uint8_t*)st->Pall + n * pt, H->d, n * pt, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipMemcpyAsync((uint8_t*)st->Pall + 2 * n * pt, st->Q, pt, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipMemcpyAsync(st->cG, Gf->d, n * 32, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipMemcpyAsync(st->cH, Hf->d, n * 32, hipMemcpyDeviceToDevice, s) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) { bp_ipp_state_free(st); return BP_ERR_DEVICE; }
    const bool digits = !st->fold_generators && n >= 16 && 2 * n + 1 <= kSmallDigitMax && st->ctx->tuning.small_msm && st->ctx->c_override <= 0;
    {   // generator compaction: automatic for proofs of >= 8192 generators (at a live length of 4096); BP_TUNE_COMPACT_AT moves or disables it
        const size_t knob = ctx->tuning.compact_at;
        const size_t at = knob == 0 ? (n >= 8192 ? 4096 : 0) : knob == 1 ? 0 : knob;
        st->compact_at = !st->fold_generators && ctx->tuning.small_msm && ctx->c_override <= 0 && at >= 16 && n > at && 2 * at + 1 <= kSmallDigitMax ? at : 0;
    }
    if (st->compact_at) {
        const bp_g1table *cg = G->ctable ? G->ctable : G->cview, *ch = H->ctable ? H->ctable : H->cview;
        const size_t og = G->ctable ? 0 : G->tview_off, oh = H->ctable ? 0 : H->tview_off;
        if (cg && ch && cg->K == 4 && ch->K == 4 && og + n <= cg->n && oh + n <= ch->n && cg->device == ctx->device && ch->device == ctx->device) {
            st->ctG = cg; st->ctH = ch; st->ctG_off = og; st->ctH_off = oh;
        } else {
            st->glv = ctx->tuning.glv && (ctx->curve == BP_CURVE_BLS12_381 ? Bls381::HAS_GLV : Bn254::HAS_GLV);
            rc = ctx->curve == BP_CURVE_BLS12_381 ? Ipp<Bls381>::build_original_multiples(st) : Ipp<Bn254>::build_original_multiples(st);
            if (rc) { bp_ipp_state_free(st); return rc; }
        }
    }
    if (!st->fold_generators && !digits && 2 * n + 1 > kSmallMsmMax) {     // smaller rounds run as ONE launch (k_small_msm): nothing for a table to merge
        rc = bp_internal_table_concat(ctx, G, 0, H, 0, n, Q_le, &st->table);
        if (rc) { bp_ipp_state_free(st); return rc; }
    } else if (digits) {
        // ... but every round is a small MSM over the SAME 2n + 1 points: their digit multiples, once (one ~0.1 ms launch for lg n rounds
        // that each lose the per-lane doubling chain; same-box A/B: n = 64 2.24-2.51 -> 2.04-2.07 ms per proof, n = 128 2.94-3.26 ->
        // 2.38-2.42, n = 16 1.35-1.60 -> 1.28-1.30; at n = 4 the launch costs more than two rounds save: 0.66 -> 0.72)
        // ... from ~4 terms per lane on as AFFINE rows (one batch inversion, ~60 us with its host round trip): the lanes' serial chains are
        // then mixed additions (8M + 2S instead of 12M + 2S per term) -- and, with the scalars split by the curve's endomorphism
        // (BP_TUNE_GLV), 16 multiples for 26 windows of 5 bits per half instead of 8 for 64 windows of 4 bits: k_small_msm_glv
        // (from 64 generators on; below 1024 over packed XYZZ rows -- with affine rows the batch inversion's host round trip cost what the
        // shorter rounds save: n = 64 1.43-1.53 ms either way, BN254 n = 512 1.74-1.93 without against 1.93-1.97 with.  XYZZ rows, same box,
        // three runs each: BLS12-381 n = 64 1.53-1.75 -> 1.44-1.58 ms, 256 2.23-2.54 -> 2.03-2.19, 512 2.72-3.12 -> 2.39-2.69; BN254 n = 64
        // 1.08-1.25 -> 1.00-1.09, 512 1.80-1.95 -> 1.62-1.72; n = 16 equal)
        if (n >= 64 && ctx->tuning.glv && (ctx->curve == BP_CURVE_BLS12_381 ? Bls381::HAS_GLV : Bn254::HAS_GLV)) {
            rc = ctx->curve == BP_CURVE_BLS12_381 ? Ipp<Bls381>::glv_round_table(st, n >= 1024) : Ipp<Bn254>::glv_round_table(st, n >= 1024);
        } else {
            rc = bp_internal_digit_table_build(ctx, st->Pall, 2 * n + 1, &st->table);
            if (!rc && n >= 1024) rc = ctx->curve == BP_CURVE_BLS12_381 ? Ipp<Bls381>::digit_table_to_affine(ctx, st->table) : Ipp<Bn254>::digit_table_to_affine(ctx, st->table);
        }
        if (rc) { bp_ipp_state_free(st); return rc; }
    }
    *out = st;
    return BP_OK;
    });
}

size_t bp_ipp_state_len(const bp_ipp_state* st) { return st ? st->n : 0; }

int bp_ctx_set_ipp_fold_generators(bp_ctx* ctx, int on) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_ERR_ARG;
    ctx->ipp_fold_generators = on != 0;
    return BP_OK;
    });
}

int bp_ipp_round(bp_ipp_state* st, uint8_t* L_le, uint8_t* R_le) {
    return bp_guard([&]() -> int {
    if (!st || !L_le || !R_le || st->n < 2) return BP_ERR_ARG;
    int rc = bp_internal_set_device(st->ctx); if (rc) return rc;
    IPP_DISPATCH(st->ctx->curve, I::round(st, L_le, R_le));
    });
}

int bp_ipp_fold(bp_ipp_state* st, const uint8_t* u_le32, const uint8_t* u_inv_le32) {
    return bp_guard([&]() -> int {
    if (!st || !u_le32 || !u_inv_le32 || st->n < 2) return BP_ERR_ARG;
    int rc = bp_internal_set_device(st->ctx); if (rc) return rc;
    IPP_DISPATCH(st->ctx->curve, I::fold(st, u_le32, u_inv_le32));
    });
}

int bp_ipp_state_finish(bp_ipp_state* st, uint8_t* a_le32, uint8_t* b_le32) {
    return bp_guard([&]() -> int {
    if (!st || !a_le32 || !b_le32 || st->n != 1) return BP_ERR_ARG;
    int rc = bp_internal_set_device(st->ctx); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(a_le32, st->a, 32, hipMemcpyDeviceToHost, st->ctx->stream));
    HIPCHK(hipMemcpyAsync(b_le32, st->b, 32, hipMemcpyDeviceToHost, st->ctx->stream));
    HIPCHK(hipStreamSynchronize(st->ctx->stream));
    return BP_OK;
    });
}

// ---- IPP::create_ipp / verify_ipp with the library's host transcript ----------------------------------------
int bp_ipp_create(bp_ctx* ctx, bp_transcript* t, const uint8_t* Q_le, const bp_frvec* G_factors, const bp_frvec* H_factors, const bp_g1vec* G,
                  const bp_g1vec* H, const bp_frvec* a, const bp_frvec* b, uint8_t* L_out, uint8_t* R_out, size_t* lg_n_out, uint8_t* a_out_le32,
                  uint8_t* b_out_le32) {
    return bp_guard([&]() -> int {
    if (!t || !a_out_le32 || !b_out_le32) return BP_ERR_ARG;
    bp_ipp_state* st = nullptr;
    int rc = bp_ipp_state_create(ctx, G, H, Q_le, G_factors, H_factors, a, b, &st);
    if (rc) return rc;
    if (st->n > 1 && (!L_out || !R_out)) { bp_ipp_state_free(st); return BP_ERR_ARG; }
    if (ctx->curve == BP_CURVE_BLS12_381) rc = Ipp<Bls381>::create(st, t->t, L_out, R_out, lg_n_out, a_out_le32, b_out_le32);
    else rc = Ipp<Bn254>::create(st, t->t, L_out, R_out, lg_n_out, a_out_le32, b_out_le32);
    bp_ipp_state_free(st);
    return rc;
    });
}

int bp_ipp_create_multi(bp_ctx* const* ctxs, size_t n_shards, bp_transcript* t, const uint8_t* Q_le, const bp_frvec* const* G_factors,
                        const bp_frvec* const* H_factors, const bp_g1vec* const* G, const bp_g1vec* const* H, const uint8_t* a_le32,
                        const uint8_t* b_le32, size_t n, uint8_t* L_out, uint8_t* R_out, size_t* lg_n_out, uint8_t* a_out_le32, uint8_t* b_out_le32) {
    return bp_guard([&]() -> int {
    if (!ctxs || n_shards == 0 || n_shards > 64 || !t || !Q_le || !G_factors || !H_factors || !G || !H || !a_le32 || !b_le32 || !a_out_le32 || !b_out_le32)
        return BP_ERR_ARG;
    if (n == 0 || (n & (n - 1))) return BP_ERR_ARG;                                             // assert!(n.is_power_of_two())  ipp.rs:48
    if (n > 1 && (!L_out || !R_out)) return BP_ERR_ARG;
    size_t total = 0;
    for (size_t s = 0; s < n_shards; s++) {
        if (!ctxs[s] || !G[s] || !H[s] || !G_factors[s] || !H_factors[s] || ctxs[s]->curve != ctxs[0]->curve) return BP_ERR_ARG;
        for (size_t j = 0; j < s; j++) if (ctxs[j] == ctxs[s]) return BP_ERR_ARG;              // one shard per context
        if (G[s]->n == 0 || H[s]->n != G[s]->n || G_factors[s]->n != G[s]->n || H_factors[s]->n != G[s]->n) return BP_ERR_ARG;   // ipp.rs:51-55
        total += G[s]->n;
    }
    if (total != n) return BP_ERR_ARG;
    try {
        if (ctxs[0]->curve == BP_CURVE_BLS12_381)
            return ipp_create_multi<Bls381>(ctxs, n_shards, t->t, Q_le, G_factors, H_factors, G, H, a_le32, b_le32, n, L_out, R_out, lg_n_out, a_out_le32, b_out_le32);
        return ipp_create_multi<Bn254>(ctxs, n_shards, t->t, Q_le, G_factors, H_factors, G, H, a_le32, b_le32, n, L_out, R_out, lg_n_out, a_out_le32, b_out_le32);
    } catch (...) {
        for (size_t s = 0; s < n_shards; s++) { (void)bp_internal_set_device(ctxs[s]); (void)hipStreamSynchronize(ctxs[s]->stream); }
        return BP_ERR_DEVICE;
    }
    });
}

int bp_ipp_verify(bp_ctx* ctx, bp_transcript* t, size_t n, const bp_frvec* G_factors, const bp_frvec* H_factors, const uint8_t* P_le,
                  const uint8_t* Q_le, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* a_le32, const uint8_t* b_le32, const uint8_t* L_le,
                  const uint8_t* R_le, size_t lg_n) {
    return bp_guard([&]() -> int {
    if (!ctx || !t || !G_factors || !H_factors || !P_le || !Q_le || !G || !H || !a_le32 || !b_le32 || (lg_n && (!L_le || !R_le))) return BP_ERR_ARG;
    if (lg_n >= 32 || n != ((size_t)1 << lg_n)) return BP_ERR_VERIFY;           // verification_scalars, ipp.rs:269-276
    if (G->n < n || H->n < n || G_factors->n < n || H_factors->n < n) return BP_ERR_LENGTH;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    IPP_DISPATCH(ctx->curve, I::verify(ctx, t->t, n, G_factors, H_factors, P_le, Q_le, G, H, a_le32, b_le32, L_le, R_le, lg_n));
    });
}

int bp_ipp_verify_batch(bp_ctx* ctx, size_t n, size_t lg_n, const bp_frvec* G_factors, const bp_frvec* H_factors, const bp_g1vec* G,
                        const bp_g1vec* H, const bp_ipp_proof_ref* proofs, size_t m, const uint8_t* weights_le32) {
    return bp_guard([&]() -> int {
    if (!ctx || !G_factors || !H_factors || !G || !H || (m && !proofs)) return BP_ERR_ARG;
    // weights: the caller's (tests) or, with NULL, fresh ones from the OS; a zero weight would drop its proof from the check
    std::vector<uint8_t> wbuf;
    if (m && !weights_le32) {
        wbuf.resize(m * 32);
        int rcw = bp_fr_random(ctx->curve, wbuf.data(), m);
        if (rcw) return rcw;
        weights_le32 = wbuf.data();
    } else {
        for (size_t p = 0; p < m; p++) if (!bp_fr_is_canonical_nonzero(ctx->curve, weights_le32 + 32 * p)) return BP_ERR_ARG;
    }
    for (size_t p = 0; p < m; p++) {
        const bp_ipp_proof_ref& r = proofs[p];
        if (!r.transcript || !r.P_le || !r.Q_le || !r.a_le32 || !r.b_le32 || (lg_n && (!r.L_le || !r.R_le))) return BP_ERR_ARG;
    }
    if (lg_n >= 32 || n != ((size_t)1 << lg_n)) return BP_ERR_VERIFY;
    if (G->n < n || H->n < n || G_factors->n < n || H_factors->n < n) return BP_ERR_LENGTH;
    if (m == 0) return BP_OK;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    IPP_DISPATCH(ctx->curve, I::verify_batch(ctx, n, lg_n, G_factors, H_factors, G, H, proofs, m, weights_le32));
    });
}

int bp_g1vec_commit_pairs(bp_ctx* ctx, const uint8_t* g_le, const uint8_t* h_le, const bp_frvec* k1, const bp_frvec* k2, bp_g1vec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !g_le || !h_le || !k1 || !k2 || !out) return BP_ERR_ARG;
    *out = nullptr;
    if (k1->n != k2->n) return BP_ERR_LENGTH;
    int rc = bp_g1vec_alloc(ctx, k1->n, out);
    if (rc || k1->n == 0) return rc;
    if (ctx->curve == BP_CURVE_BLS12_381) rc = commit_pairs_impl<Bls381>(ctx, g_le, h_le, k1, k2, *out);
    else rc = commit_pairs_impl<Bn254>(ctx, g_le, h_le, k1, k2, *out);
    if (rc) { bp_g1vec_free(*out); *out = nullptr; }
    return rc;
    });
}

int bp_r1cs_plan_create(bp_ctx* ctx, size_t n_terms, const uint32_t* term_constraint, const uint8_t* term_kind, const uint32_t* term_index,
                        const uint8_t* coeff_le32, size_t n_constraints, size_t n, size_t m, bp_r1cs_plan** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out || (n_terms && (!term_constraint || !term_kind || !term_index || !coeff_le32))) return BP_ERR_ARG;
    *out = nullptr;
    if (n_terms >= ((size_t)1 << 32) || 3 * n + m + 1 >= ((size_t)1 << 32) || n_constraints >= ((size_t)1 << 32)) return BP_ERR_ARG;
    const size_t ndest = 3 * n + m + 1;
    std::vector<uint32_t> dest(n_terms), seg(ndest + 1, 0);
    for (size_t t = 0; t < n_terms; t++) {
        const uint8_t k = term_kind[t];
        const uint32_t i = term_index[t];
        if (term_constraint[t] >= n_constraints) return BP_ERR_ARG;
        if (k <= BP_VAR_MUL_OUTPUT) { if (i >= n) return BP_ERR_ARG; dest[t] = (uint32_t)(k * n + i); }
        else if (k == BP_VAR_COMMITTED) { if (i >= m) return BP_ERR_ARG; dest[t] = (uint32_t)(3 * n + i); }
        else if (k == BP_VAR_ONE) dest[t] = (uint32_t)(3 * n + m);
        else return BP_ERR_ARG;
        seg[dest[t] + 1]++;
    }
    for (size_t d = 0; d < ndest; d++) seg[d + 1] += seg[d];
    std::vector<uint32_t> cur(seg.begin(), seg.end() - 1), tq(n_terms ? n_terms : 1), heavy;
    std::vector<uint8_t> coeff((n_terms ? n_terms : 1) * 32);
    for (size_t t = 0; t < n_terms; t++) {                 // counting sort by destination (order inside one is irrelevant)
        const uint32_t pos = cur[dest[t]]++;
        tq[pos] = term_constraint[t];
        memcpy(&coeff[(size_t)pos * 32], coeff_le32 + 32 * t, 32);
    }
    std::vector<uint32_t> chunk, hfirst;
    for (size_t d = 0; d < ndest; d++) {
        if (seg[d + 1] - seg[d] <= kFlattenLightMax) continue;
        heavy.push_back((uint32_t)d);
        hfirst.push_back((uint32_t)(chunk.size() / 2));
        for (uint32_t lo = seg[d]; lo < seg[d + 1]; lo += kFlattenChunk) {
            chunk.push_back(lo);
            chunk.push_back(seg[d + 1] - lo > kFlattenChunk ? lo + kFlattenChunk : seg[d + 1]);
        }
    }
    hfirst.push_back((uint32_t)(chunk.size() / 2));
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    bp_r1cs_plan* p = new (std::nothrow) bp_r1cs_plan{ctx->device, ctx->curve, n, m, n_constraints, n_terms, ndest, heavy.size(), nullptr, nullptr, nullptr, nullptr};
    if (!p) return BP_ERR_DEVICE;
    p->nchunks = chunk.size() / 2;
    auto fail = [&]() { bp_r1cs_plan_free(p); return BP_ERR_DEVICE; };
    if (hipMalloc(&p->seg, (ndest + 1) * 4) != hipSuccess || hipMalloc(&p->tq, tq.size() * 4) != hipSuccess ||
        hipMalloc(&p->coeff, coeff.size()) != hipSuccess || hipMalloc(&p->heavy, (heavy.size() ? heavy.size() : 1) * 4) != hipSuccess ||
        hipMalloc(&p->chunk, (chunk.size() ? chunk.size() : 1) * 4) != hipSuccess || hipMalloc(&p->hfirst, hfirst.size() * 4) != hipSuccess)
        return fail();
    hipStream_t s = ctx->stream;
    if (hipMemcpyAsync(p->seg, seg.data(), (ndest + 1) * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(p->tq, tq.data(), tq.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(p->coeff, coeff.data(), coeff.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
        (heavy.size() && hipMemcpyAsync(p->heavy, heavy.data(), heavy.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) ||
        (chunk.size() && hipMemcpyAsync(p->chunk, chunk.data(), chunk.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) ||
        hipMemcpyAsync(p->hfirst, hfirst.data(), hfirst.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return fail();
    *out = p;
    return BP_OK;
    });
}

int bp_r1cs_plan_free(bp_r1cs_plan* p) {
    return bp_guard([&]() -> int {
    if (!p) return BP_OK;
    (void)hipSetDevice(p->device);
    for (void* b : {p->seg, p->tq, p->coeff, p->heavy, p->chunk, p->hfirst}) if (b) (void)hipFree(b);
    delete p;
    return BP_OK;
    });
}

int bp_r1cs_flattened_constraints(bp_ctx* ctx, const bp_r1cs_plan* plan, const uint8_t* z_le32, bp_frvec* out[4], uint8_t* wc_le32) {
    return bp_guard([&]() -> int {
    if (!ctx || !plan || !z_le32 || !out) return BP_ERR_ARG;
    for (int k = 0; k < 4; k++) out[k] = nullptr;
    if (plan->curve != ctx->curve || plan->device != ctx->device) return BP_ERR_ARG;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    if (ctx->curve == BP_CURVE_BLS12_381) return flattened_constraints_impl<Bls381>(ctx, plan, z_le32, out, wc_le32);
    return flattened_constraints_impl<Bn254>(ctx, plan, z_le32, out, wc_le32);
    });
}

// IPP::verification_scalars (ipp.rs:262-315): (u_j^2, u_j^-2, s) as canonical LE scalars; host arithmetic.
int bp_ipp_verification_scalars(int curve_id, bp_transcript* t, const uint8_t* L_le, const uint8_t* R_le, size_t lg_n, size_t n, uint8_t* u_sq,
                                uint8_t* u_inv_sq, uint8_t* s) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || !t || !u_sq || !u_inv_sq || !s || (lg_n && (!L_le || !R_le))) return BP_ERR_ARG;
    auto run = [&](auto tag) -> int {
        using C = decltype(tag);
        using F = typename C::Fr;
        std::vector<Fe<F>> ch, ch_inv;
        int rc = Ipp<C>::verification_scalars(t->t, L_le, R_le, lg_n, n, ch, ch_inv);
        if (rc) return rc;
        std::vector<Fe<F>> usq(lg_n);
        Fe<F> prod_inv = fe_one<F>();
        for (size_t j = 0; j < lg_n; j++) {
            usq[j] = fe_sqr(ch[j]);
            fr_to_le<F>(usq[j], u_sq + 32 * j);
            fr_to_le<F>(fe_sqr(ch_inv[j]), u_inv_sq + 32 * j);
            prod_inv = fe_mul(prod_inv, ch_inv[j]);
        }
        std::vector<Fe<F>> sv(n);
        sv[0] = prod_inv;                                                               // :304
        for (size_t i = 1; i < n; i++) {                                                // :305-312
            int lg_i = 63 - __builtin_clzll((unsigned long long)i);
            sv[i] = fe_mul(sv[i - ((size_t)1 << lg_i)], usq[(lg_n - 1) - lg_i]);
        }
        for (size_t i = 0; i < n; i++) fr_to_le<F>(sv[i], s + 32 * i);
        return BP_OK;
    };
    return curve_id == BP_CURVE_BLS12_381 ? run(Bls381{}) : run(Bn254{});
    });
}

}  // extern "C"
