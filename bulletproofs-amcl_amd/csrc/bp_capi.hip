// bp_capi.hip -- implementation of the C ABI in include/bpmsm.h (libbpmsm.so).
// Host orchestration of the gfx950 kernels in bp_kernels.cuh; no CPU fallback for any compute entry point.
#include <new>
#include <vector>

#include "bp_internal.hpp"
#include "bp_host_tail.hpp"

using namespace bp;

// ------------------------------------------------------------------------------------------------ geometry
struct MsmGeom {
    int c;           // target window width
    WinTab tab;      // W windows of nearly equal width covering fr_bits + 1 bits
    uint32_t m;      // buckets per reduce thread (a power of two; windows with fewer buckets use their bucket count)
    bool small;      // n <= kSmallMsmMax: the single-launch path, one record per window
    int small_blocks; // ... per block of the window: 2 above 512 terms, else 1
    bool merged;     // MSM over a window-multiples table: tab describes the digit windows, tabv the merged buckets (one "window" per scalar set)
    WinTab tabv;     // merged only: the view the kernels after the coarse scatter run with (k_fine_place .. k_window_sums)
    // tail records handed to the host: record r carries weight 2^rpos[r] (bp_host_tail.hpp folds any such list)
    int nrec;
    uint16_t rpos[kMaxRecords];
    // window SUBSET (bp_msm_g1_windows_subset: shards that split the windows as well as the index range): tab then holds windows
    // [sub_first, sub_first + sub_count) of the W_full windows of the scalar -- same recoding, same bit offsets -- renumbered from 0
    int sub_first, sub_count, W_full;
    uint8_t cw_first_full, cw_last_full, n_wide_full;
};

static int ilog2(uint32_t v) { int l = 0; while ((1u << (l + 1)) <= v) l++; return l; }

// Reduce geometry on the table the reduce kernels run with (t = g.tab, or g.tabv in merged mode): buckets per reduce thread, the
// compact block grid, and the tail records with their bit positions.
static int geom_reduce(MsmGeom& g, WinTab& t, const bp_tuning* tn) {
    const int W = t.W;
    // Buckets per reduce thread: the smallest power of two whose blocks fit one per CU.  A 257th block makes some SIMD run two of
    // these dependent chains back to back (measured: c = 14 paired, 304 blocks 1.26 ms, 152 blocks 0.86 ms; scripts/time_pair.py).
    auto blocks_for = [&](uint32_t m) {
        uint32_t blocks = 0;
        for (int w = 0; w < W; w++) {
            uint32_t B = t.boff[w + 1] - t.boff[w], mw = m < B ? m : B;
            blocks += (B / mw + kBlock - 1) / kBlock;
        }
        return blocks;
    };
    // ... and among those the one with the shortest dependent chain of k_bucket_reduce for the widest window:
    //   2m (running sums) + log2(threads per block) (block tree) + passes * log2(blocks per window) (window level, last block;
    //   its LDS holds kReduceSlots / blocks partial kinds per pass)
    // m = 1 is not always best: a merged (table) MSM at c = 16 has 128 blocks per scalar set and would fold its 10 partial kinds
    // in three passes of 7 steps; m = 4 gives 8 + 8 + 5.
    auto chain = [&](uint32_t m) {
        uint32_t worst = 0;
        for (int w = 0; w < W; w++) {
            const uint32_t B = t.boff[w + 1] - t.boff[w], mw = m < B ? m : B, T = B / mw, nblk = (T + kBlock - 1) / kBlock;
            const uint32_t lgT = (uint32_t)ilog2(T < (uint32_t)kBlock ? T : (uint32_t)kBlock), lgB = (uint32_t)ilog2(nblk), kinds = 2 + lgT;
            uint32_t per = (uint32_t)kReduceSlots / nblk;
            if (per > kinds) per = kinds;
            const uint32_t passes = nblk > 1 ? (kinds + per - 1) / per : 0;
            const uint32_t steps = 2 * mw + lgT + passes * lgB;
            if (steps > worst) worst = steps;
        }
        return worst;
    };
    uint32_t m = 1;
    while (m < (1u << 15) && blocks_for(m) > 256) m <<= 1;
    for (uint32_t m2 = m << 1; m2 <= 64; m2 <<= 1) if (chain(m2) < chain(m)) m = m2;
    if (tn && tn->reduce_m) m = tn->reduce_m;
    g.m = m;
    uint32_t rb = 0, nrec = 0;
    for (int w = 0; w < W; w++) {
        const uint32_t B = t.boff[w + 1] - t.boff[w], mw = m < B ? m : B;
        const uint32_t T = B / mw, nblk = (T + kBlock - 1) / kBlock;
        t.lgm[w] = (uint8_t)ilog2(mw);
        t.rboff[w] = (uint16_t)rb;
        rb += nblk;
        t.roff[w] = (uint16_t)nrec;
        if (g.small) {                                         // one record per window and block: the sum of the block's terms
            for (int b = 0; b < g.small_blocks; b++) if (nrec + b < (uint32_t)kMaxRecords) g.rpos[nrec + b] = t.off[w];
            nrec += (uint32_t)g.small_blocks;
        } else {                                               // tri, then one plane per bit of the reduce-thread index
            const int planes = ilog2(T);
            for (int k = 0; k <= planes; k++)
                if (nrec + k < (uint32_t)kMaxRecords) g.rpos[nrec + k] = (uint16_t)(k == 0 ? t.off[w] : t.off[w] + t.lgm[w] + (k - 1));
            nrec += 1 + planes;
        }
    }
    t.rboff[W] = (uint16_t)rb;
    t.roff[W] = (uint16_t)nrec;
    g.nrec = (int)nrec;
    if (nrec > (uint32_t)kMaxRecords || rb > 65535u) return BP_ERR_ARG;
    return BP_OK;
}

// tn: the context's validated tuning knobs (nullptr: defaults)
static int msm_geom(MsmGeom& g, int fr_bits, size_t n, int c_override, int nsets = 1, size_t nnz = 0, const bp_tuning* tn = nullptr, size_t small_max = kSmallMsmMax) {
    int c = c_override;
    if (c <= 0) {
        int lg = 0;
        while (((size_t)1 << (lg + 1)) <= n) lg++;
        c = lg - 1;   // measured (scripts/time_msm.py sweeps): short per-bucket chains beat fewer buckets up to c = 16
        if (c > 16) c = 16;
        // ... unless that leaves > 2^14 nearly empty buckets (the bucket reduce costs ~2 ns per bucket whatever they
        // hold): shrink c until the mean occupancy reaches 8.  nnz = non-zero scalars per set when the caller knows
        // it (the IPP rounds: half of the generators carry a zero), else n.  (scripts/sweep_ipp_c.py)
        const size_t live = nnz ? nnz : n;
        while (c > 8 && nsets > 1) {   // measured for the paired (IPP round) shape only; single-set sweeps favour lg n - 1 throughout
            uint64_t W1 = (uint64_t)((fr_bits + 1 + c - 1) / c);
            uint64_t buckets = (uint64_t)nsets * W1 << (c - 1), entries = (uint64_t)nsets * W1 * live;
            if (buckets <= (1u << 14) || entries >= 8 * buckets) break;   // floor 2^14 (2^17 until round 2: n = 2^12, 2^13 ran 4-7 % slower at c = 12)
            c--;
        }
    }
    const bool small_ok = !tn || tn->small_msm;
    // The single-launch path emits one record per window, the bucket pipeline 1 + log2(reduce threads): callers that must agree on
    // a record layout across shards fix the window width (bp_ctx_set_window_bits / bp_msm_g1_multi), and a fixed width always means
    // the pipeline's layout -- a 5-point shard beside a 2^20-point one then folds with it.
    g.small = n <= small_max && small_ok && c_override <= 0;       // small_max: kSmallDigitMax when the caller holds the points' digit multiples
    g.small_blocks = g.small && n > 512 ? 2 : 1;                    // more than two terms per lane: two blocks per window, a record each
    // k_small_msm (n <= kSmallMsmMax): each lane multiplies by its digit, so narrow windows shorten the chain; below c = 4 the
    // extra windows cost more on the host (one addition per window in the tail) than they save on the device
    if (c_override <= 0 && g.small && c > kSmallDigitBits) c = kSmallDigitBits;
    if (c < 2) c = 2;
    if (c > 16) c = 16;
    g.c = c;
    int cover = fr_bits + 1;
    int W1 = (cover + c - 1) / c;                 // windows per scalar set
    int base = cover / W1, extra = cover % W1;    // `extra` windows of base+1 bits, the rest base bits (all <= c)
    WinTab& t = g.tab;
    memset(&t, 0, sizeof t);
    const int W = W1 * nsets;
    t.W = W;
    t.nsets = nsets;
    uint32_t bias[8] = {0};
    int off = 0;
    uint32_t nb = 0, rows = 0;
    for (int w = 0; w < W; w++) {
        const int w1 = w % W1;
        if (w1 == 0) off = 0;
        int cw = base + (w1 < extra ? 1 : 0);
        t.cw[w] = (uint8_t)cw;
        t.off[w] = (uint16_t)off;
        t.boff[w] = nb;
        int fb = cw - 1 < 8 ? cw - 1 : 8;
        t.fbits[w] = (uint8_t)fb;
        t.hoff[w] = (uint16_t)rows;
        rows += 1u << (cw - 1 - fb);
        uint32_t B = 1u << (cw - 1);
        nb += B;
        if (w >= W1) { off += cw; continue; }     // the bias is per scalar: accumulate it over the first set only
        // bias += (2^(cw-1) - 1) << off
        uint64_t half1 = (uint64_t)B - 1;
        int word = off >> 5, sh = off & 31;
        unsigned __int128 add = (unsigned __int128)half1 << sh;
        uint64_t carry = 0;
        for (int k = word; k < 8; k++) {
            uint64_t v = (uint64_t)bias[k] + (uint64_t)(add & 0xffffffffu) + carry;
            bias[k] = (uint32_t)v;
            carry = v >> 32;
            add >>= 32;
            if (!add && !carry) break;
        }
        off += cw;
    }
    t.boff[W] = nb;
    t.hoff[W] = (uint16_t)rows;
    t.nbuckets = nb;
    memcpy(t.bias.w, bias, sizeof bias);
    // (Round 2 also carried an opt-in pipeline of window groups on several streams here.  It was measured slower at every group
    // count -- 4.44 -> 4.94 / 5.69 / 9.6 ms for 2 / 4 / 8 groups at n = 2^20, profiles/r02_window_groups.txt -- and was removed in
    // round 3; DESIGN.md section 5 keeps the analysis.)
    g.merged = false;
    g.sub_first = 0; g.sub_count = W; g.W_full = W;
    g.cw_first_full = t.cw[0]; g.cw_last_full = t.cw[W - 1]; g.n_wide_full = 0;
    for (int w = 0; w < W; w++) if (t.cw[w] == t.cw[0]) g.n_wide_full++;
    return geom_reduce(g, t, tn);
}

// Restrict a (single-set, bucket-pipeline) geometry to the windows [w0, w0 + wn): the kernels run over a table of wn windows whose
// bit offsets are the originals (k_digits_bin shifts the biased scalar down to the first one), the records carry the original bit
// positions, so record blocks of different window groups simply add up in the host fold.
static int geom_subset(MsmGeom& g, int w0, int wn, const bp_tuning* tn) {
    WinTab& t = g.tab;
    if (g.small || g.merged || t.nsets != 1 || w0 < 0 || wn <= 0 || w0 + wn > t.W) return BP_ERR_ARG;
    if (w0 == 0 && wn == t.W) return BP_OK;
    uint8_t cw[kMaxWindows], fb[kMaxWindows];
    uint16_t off[kMaxWindows];
    for (int w = 0; w < wn; w++) { cw[w] = t.cw[w0 + w]; fb[w] = t.fbits[w0 + w]; off[w] = t.off[w0 + w]; }
    uint32_t nb = 0, rows = 0;
    for (int w = 0; w < wn; w++) {
        t.cw[w] = cw[w]; t.fbits[w] = fb[w]; t.off[w] = off[w];
        t.boff[w] = nb; t.hoff[w] = (uint16_t)rows;
        nb += 1u << (cw[w] - 1);
        rows += 1u << (cw[w] - 1 - fb[w]);
    }
    t.boff[wn] = nb; t.hoff[wn] = (uint16_t)rows; t.nbuckets = nb; t.W = wn;
    g.sub_first = w0; g.sub_count = wn;
    return geom_reduce(g, t, tn);
}

// Geometry of an MSM over a window-multiples table (bp_g1vec_precompute) with window width c: W1 windows per scalar set at bit
// offsets c w (the last one narrower), all of a set's windows sharing its 2^(c-1) buckets.
static int msm_geom_table(MsmGeom& g, int fr_bits, int c, int W1, int nsets, const bp_tuning* tn) {
    const int cover = fr_bits + 1;
    if (c < 2 || c > 16 || W1 != (cover + c - 1) / c || W1 * nsets > kMaxWindows) return BP_ERR_ARG;
    g.c = c;
    g.small = false;
    g.small_blocks = 1;
    g.merged = true;
    WinTab& t = g.tab;
    memset(&t, 0, sizeof t);
    const int W = W1 * nsets;
    const int fb = c - 1 < 8 ? c - 1 : 8;
    const uint32_t B = 1u << (c - 1), bins = 1u << (c - 1 - fb);
    t.W = W; t.nsets = nsets; t.merged = 1; t.W1 = (uint16_t)W1;
    uint32_t bias[8] = {0};
    for (int w = 0; w < W; w++) {
        const int w1 = w % W1, set = w / W1, off = c * w1;
        const int cw = cover - off < c ? cover - off : c;
        t.cw[w] = (uint8_t)cw;
        t.off[w] = (uint16_t)off;
        t.boff[w] = (uint32_t)set * B;
        t.fbits[w] = (uint8_t)fb;
        t.hoff[w] = (uint16_t)(set * bins);
        if (set) continue;
        unsigned __int128 add = (unsigned __int128)(((uint64_t)1 << (cw - 1)) - 1) << (off & 31);
        uint64_t carry = 0;
        for (int k = off >> 5; k < 8; k++) {
            uint64_t v = (uint64_t)bias[k] + (uint64_t)(add & 0xffffffffu) + carry;
            bias[k] = (uint32_t)v;
            carry = v >> 32;
            add >>= 32;
            if (!add && !carry) break;
        }
    }
    t.boff[W] = (uint32_t)nsets * B;
    t.hoff[W] = (uint16_t)(nsets * bins);
    t.nbuckets = (uint32_t)nsets * B;
    memcpy(t.bias.w, bias, sizeof bias);
    // the view: one window of c bits per scalar set
    WinTab& v = g.tabv;
    memset(&v, 0, sizeof v);
    v.W = nsets; v.nsets = nsets; v.W1 = 1;
    for (int s2 = 0; s2 <= nsets; s2++) { v.boff[s2] = (uint32_t)s2 * B; v.hoff[s2] = (uint16_t)(s2 * bins); }
    for (int s2 = 0; s2 < nsets; s2++) { v.cw[s2] = (uint8_t)c; v.off[s2] = 0; v.fbits[s2] = (uint8_t)fb; }
    v.nbuckets = t.nbuckets;
    return geom_reduce(g, v, tn);
}

// ------------------------------------------------------------------------------------------------ per-curve code
template <class C>
struct Impl {
    using Fp = typename C::Fp;
    static constexpr size_t kPointBytes = sizeof(AffPacked<C>);
    static constexpr size_t kXyzzBytes = sizeof(XyzzPacked<C>);

    static int ensure_events(bp_ctx* ctx) {
        if (ctx->ev_ready) return BP_OK;
        for (auto& e : ctx->ev) HIPCHK(hipEventCreate(&e));
        ctx->ev_ready = true;
        return BP_OK;
    }

    // Device stage: tail records of  sum_i s_i P_i  into ctx->window_sum (g.nrec records).
    static int msm_windows(bp_ctx* ctx, const AffPacked<C>* pts, const ScalarWords* sc, size_t n, MsmGeom& g, const ScalarWords* sc2 = nullptr,
                           size_t nnz = 0, const bp_g1table* tb = nullptr) {
        // tb: window-multiples table of `pts` (same n): the merged-window pipeline over its rows
        const bp_g1table* dm = nullptr;                     // digit multiples for the single-launch small MSM (bp_g1table::digits)
        if (tb && tb->digits) { dm = tb; tb = nullptr; }
        if (dm && dm->glv) dm = nullptr;                    // a GLV-split table has its own kernel (bp_capi_ipp.hip); here it is just "no table"
        int rc = tb ? msm_geom_table(g, C::Fr::BITS, tb->c, tb->W, sc2 ? 2 : 1, &ctx->tuning)
                    : msm_geom(g, C::Fr::BITS, n, ctx->c_override, sc2 ? 2 : 1, nnz, &ctx->tuning, ctx->win_count ? 0 : dm && dm->n == n ? kSmallDigitMax : kSmallMsmMax);
        if (rc) return rc;
        if (ctx->win_count) {                               // a window group of a sharded MSM (bp_msm_g1_windows_subset): always the pipeline's layout
            if (tb || dm || sc2 || g.small) return BP_ERR_ARG;
            if ((rc = geom_subset(g, ctx->win_first, ctx->win_count, &ctx->tuning))) return rc;
        }
        if (tb) { if (tb->n != n || (uint64_t)tb->W * n >= ((uint64_t)1 << 31)) return BP_ERR_ARG; pts = (const AffPacked<C>*)tb->d; }
        const WinTab& tab = g.tab;                          // digit windows: k_digits_bin, k_coarse_scatter
        const WinTab& tabR = g.merged ? g.tabv : g.tab;     // buckets as the later kernels see them
        const int W = tab.W, WR = tabR.W;
        if (bp_trace_on()) fprintf(stderr, "[bpmsm trace] msm n=%zu c=%d W=%d merged=%d nbuckets=%u m=%u reduce_blocks=%u records=%d\n", n, g.c, W, (int)g.merged, tab.nbuckets, g.m, (unsigned)tabR.rboff[WR], g.nrec);
        if (n >= ((size_t)1 << 31)) return BP_ERR_ARG;                      // the sign lives in bit 31 of an index
        if ((uint64_t)W * n >= ((uint64_t)1 << 32)) return BP_ERR_ARG;      // 32-bit slot offsets
        hipStream_t st = ctx->stream;
        const bool tm = ctx->timing;
        if (tm && (rc = ensure_events(ctx))) return rc;
        if ((rc = ctx->window_sum.reserve(ctx, (size_t)g.nrec * kXyzzBytes))) return rc;
        if (g.small) {   // one launch: block per window, lane per term (k_small_msm)
            if (tm) for (int e = 0; e < 6; e++) HIPCHK(hipEventRecord(ctx->ev[e], st));
            // the multiples only when they are this vector's and cover the digits of this geometry
            const void* mult = dm && dm->n == n && g.c <= dm->c ? dm->d : nullptr;
            const dim3 grid(W, g.small_blocks);
            auto* ws = (XyzzPacked<C>*)ctx->window_sum.p;
            // an inner-product round (ctx->ipp_live set by bp_capi_ipp.hip around this call): only the terms that can be non-zero
            IppSparse sp = {0, 0, 0};
            if (ctx->ipp_live >= 2 && sc2 && n == 2 * (size_t)ctx->ipp_n0 + 1) sp = IppSparse{ctx->ipp_n0, ctx->ipp_live, ctx->ipp_live / 2};
            if (!mult) hipLaunchKernelGGL((k_small_msm<C, 0>), grid, dim3(kBlock), 0, st, pts, sc, sc2, (uint32_t)n, tab, ws, mult, sp);
            else if (dm->affine) hipLaunchKernelGGL((k_small_msm<C, 2>), grid, dim3(kBlock), 0, st, pts, sc, sc2, (uint32_t)n, tab, ws, mult, sp);
            else hipLaunchKernelGGL((k_small_msm<C, 1>), grid, dim3(kBlock), 0, st, pts, sc, sc2, (uint32_t)n, tab, ws, mult, sp);
            BP_TRACE_SYNC(ctx, "k_small_msm<C>");
            if (tm) HIPCHK(hipEventRecord(ctx->ev[6], st));
            HIPCHK(hipGetLastError());
            return BP_OK;
        }
        const size_t nb = tab.nbuckets;
        // Task length (task_len, bp_kernels.cuh): chosen on the device from the entries and non-empty buckets the sort finds.  The host
        // computes the same rule for the worst case -- every digit non-zero, every bucket used -- to SIZE the task arrays: the device
        // may go down to an eighth of it (structured scalars: few entries, few buckets), never below.
        // (target: 1.5x the 131 072 resident lanes of k_accumulate.  It was 2x while the length came from the host's n W; with the true
        // entry count an inner-product round -- half of each scalar set is zero -- fell to L = 8, four task sums per bucket for the
        // reduce to add: accumulate -22 us, reduce +50 us per round on one box.  1.5x restores L = 16 there; uniform MSMs of 2^14 ..
        // 2^20 measured the same at 1x, 1.5x and 2x.)
        const uint64_t kTaskTarget = ctx->tuning.task_target ? ctx->tuning.task_target : 3 * 65536;
        const uint64_t entries = (uint64_t)W * (nnz ? nnz : n);
        uint32_t L = 8;
        // (Round 3 tried "plentiful" from a quarter of the target on, so that the 2^16 merged buckets of a table MSM stay one task each
        // and the reduce chain loses the recombination: reduce 0.48 -> 0.34 ms, but the accumulate then lasts as long as its LONGEST
        // bucket -- one task per lane, nothing to balance -- 0.31 -> 0.51 ms.  Kept as it was.)
        if (nb >= kTaskTarget) { while ((uint64_t)L * nb < 2 * entries && L < (1u << 20)) L <<= 1; if (L < 128) L = 128; }
        else { while ((uint64_t)L * kTaskTarget < entries && L < (1u << 20)) L <<= 1; }
        const uint32_t lmin = L / 8 > 8 ? L / 8 : 8;
        L = lmin;                                                           // what the arrays are sized for
        const size_t slots = (size_t)W * n;
        size_t max_split = slots / L + 1;                                 // tasks beyond one per bucket
        if (max_split > slots) max_split = slots;
        const size_t max_tasks = nb + max_split;
        const size_t max_heavy = (nb < max_split ? nb : max_split) + 1;
        const size_t max_chunks = max_heavy + max_tasks / kBlock + 1;
        const size_t scan_blocks = (nb + kScanPerBlock - 1) / kScanPerBlock;
        if ((rc = ctx->count.reserve(ctx, nb * 4))) return rc;
        if ((rc = ctx->cursor.reserve(ctx, nb * 4))) return rc;
        if ((rc = ctx->ntasks.reserve(ctx, nb * 4))) return rc;
        if ((rc = ctx->task_off.reserve(ctx, nb * 4))) return rc;
        if ((rc = ctx->idx.reserve(ctx, (size_t)W * n * 4))) return rc;
        if ((rc = ctx->code.reserve(ctx, (size_t)W * n * 2))) return rc;
        if ((rc = ctx->order.reserve(ctx, max_tasks * 4))) return rc;
        if ((rc = ctx->t_start.reserve(ctx, max_tasks * 4))) return rc;
        if ((rc = ctx->t_len.reserve(ctx, max_tasks * 4))) return rc;
        if ((rc = ctx->tsum.reserve(ctx, max_tasks * kXyzzBytes))) return rc;
        if ((rc = ctx->heavy.reserve(ctx, max_heavy * 4))) return rc;
        if ((rc = ctx->heavy_chunks.reserve(ctx, max_chunks * sizeof(uint2)))) return rc;
        constexpr size_t kMetaWords = ((kTaskBins + 3 + 15) / 16) * 16 + 2 * kMaxWindows + 16;   // + per window: "blocks done" (k_bucket_reduce), non-empty buckets; + the slice count of huge bins
        if ((rc = ctx->meta.reserve(ctx, kMetaWords * 4))) return rc;
        if ((rc = ctx->partial.reserve(ctx, (size_t)tabR.rboff[WR] * kPartPerBlock * kXyzzBytes))) return rc;
        uint32_t* count = (uint32_t*)ctx->count.p;       // bucket starts
        uint32_t* cursor = (uint32_t*)ctx->cursor.p;     // bucket ends
        uint32_t* ntasks = (uint32_t*)ctx->ntasks.p;
        uint32_t* task_off = (uint32_t*)ctx->task_off.p;
        uint32_t* idx = (uint32_t*)ctx->idx.p;
        uint16_t* code = (uint16_t*)ctx->code.p;
        auto* partial = (XyzzPacked<C>*)ctx->partial.p;
        auto* wsum = (XyzzPacked<C>*)ctx->window_sum.p;

        // Tile = scalars per block of the binning passes (2048 .. 16384 measured within 1 % of each other at 2^18 .. 2^22).
        // Below ~2^19 scalars a 2048-scalar tile leaves the per-scalar passes with a few dozen blocks for 256 CUs (n = 2^17: 65 blocks,
        // k_digits_bin 71 us); the tile shrinks (never below one scalar per lane) until there are ~512 of them.
        uint32_t tile = kTile;
        while (tile > (uint32_t)kBlock && (n + tile - 1) / tile < 512) tile >>= 1;
        if (ctx->tuning.tile) tile = ctx->tuning.tile;              // validated by bp_ctx_set_tuning: a multiple of kBlock
        const uint32_t ntiles = (uint32_t)((n + tile - 1) / tile);
        const uint32_t ncols = g.merged ? ntiles * (uint32_t)tab.W1 : ntiles;     // columns of the tile histogram: (window of the set, tile) when merged
        const uint32_t rows = tab.hoff[W];
        const size_t nhist = (size_t)rows * ncols;
        const size_t hist_blocks = (nhist + kScanPerBlock - 1) / kScanPerBlock;
        if ((rc = ctx->tile_hist.reserve(ctx, nhist * 4))) return rc;
        if ((rc = ctx->tmp_idx.reserve(ctx, (size_t)W * n * 8))) return rc;            // (point index, digit code) records of the coarse pass
        if ((rc = ctx->block_sums.reserve(ctx, (hist_blocks + 16 + scan_blocks + 16) * 4))) return rc;
        uint32_t* hsum = (uint32_t*)ctx->block_sums.p;                 // scan of the tile histogram; hsum[hist_blocks] = grand total
        uint32_t* bsum = hsum + hist_blocks + 16;                      // scan of the task counts
        uint32_t* tile_hist = (uint32_t*)ctx->tile_hist.p;
        uint2* tmp_rec = (uint2*)ctx->tmp_idx.p;
        uint32_t* bins = (uint32_t*)ctx->meta.p;                       // [kTaskBins] bin counts -> bin cursors
        uint32_t* total_tasks = bins + kTaskBins;
        uint32_t* nheavy = bins + kTaskBins + 1;
        uint32_t* nchunks = bins + kTaskBins + 2;
        uint32_t* win_done = bins + ((kTaskBins + 3 + 15) / 16) * 16;
        uint32_t* nonempty = win_done + kMaxWindows;                    // per window: buckets that hold anything (k_fine_place -> task_len)
        uint32_t* nslices = nonempty + kMaxWindows;
        // Huge coarse bins (structured scalars: bit vectors, a few distinct values, small values) get a block per 8192 records instead of
        // one block for the bin -- from 2^20 records on (2^16 scalars x 16 windows): below, a bin cannot hold more than a block streams in
        // ~0.1 ms, and the two extra launches (~5 us when there is nothing to do) would be felt.
        const size_t nrecs = (size_t)W * n;
        const size_t max_slices = nrecs >= ((size_t)1 << 20) ? nrecs / kHugeSlice + nrecs / kHugeMin + 2 : 0;
        if (max_slices && (rc = ctx->huge.reserve(ctx, max_slices * (sizeof(HugeSlice) + kBlock * 4)))) return rc;
        HugeSlice* slices = max_slices ? (HugeSlice*)ctx->huge.p : nullptr;
        uint32_t* slice_hist = max_slices ? (uint32_t*)((uint8_t*)ctx->huge.p + max_slices * sizeof(HugeSlice)) : nullptr;
        uint32_t* order = (uint32_t*)ctx->order.p;
        uint32_t* t_start = (uint32_t*)ctx->t_start.p;
        uint32_t* t_len = (uint32_t*)ctx->t_len.p;
        auto* tsum = (XyzzPacked<C>*)ctx->tsum.p;
        uint32_t* heavy = (uint32_t*)ctx->heavy.p;
        uint2* chunks = (uint2*)ctx->heavy_chunks.p;

        // scalar negation (k_digits_bin): one bit per scalar and set, one flag word per tile and set
        const size_t nw64 = (n + 63) / 64;
        if ((rc = ctx->negbits.reserve(ctx, (size_t)tab.nsets * (nw64 * 8 + (size_t)ntiles * 4)))) return rc;
        uint64_t* negbits = (uint64_t*)ctx->negbits.p;
        uint32_t* tile_neg = (uint32_t*)(negbits + (size_t)tab.nsets * nw64);
        ScalarWords rmod, rneg;
        for (int k = 0; k < 8; k++) rmod.w[k] = rneg.w[k] = C::Fr::Words::MODW[k];
        for (int k = 4; k < 8; k++) { const bool more = rneg.w[k] == 0; rneg.w[k]--; if (!more) break; }             // r - 2^128
        if (tm) HIPCHK(hipEventRecord(ctx->ev[0], st));
        HIPCHK(hipMemsetAsync(ctx->meta.p, 0, kMetaWords * 4, st));
        hipLaunchKernelGGL(k_digits_bin, dim3(ntiles), dim3(kBlock), 0, st, sc, sc2, n, tab, ntiles, tile, code, tile_hist, rmod, rneg, negbits, nw64, tile_neg);
        BP_TRACE_SYNC(ctx, "k_digits_bin");
        if (tm) HIPCHK(hipEventRecord(ctx->ev[1], st));
        hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)hist_blocks), dim3(kBlock), 0, st, tile_hist, nhist, hsum);
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)hist_blocks), dim3(kBlock), 0, st, tile_hist, nhist, hsum, tile_hist, (uint32_t*)nullptr);
        BP_TRACE_SYNC(ctx, "scan tile_hist");
        if (tm) HIPCHK(hipEventRecord(ctx->ev[2], st));
        hipLaunchKernelGGL(k_coarse_scatter, dim3(ntiles, W), dim3(kBlock), 0, st, code, n, tab, ntiles, tile_hist, tmp_rec, 0, tile, negbits, nw64, tile_neg);
        BP_TRACE_SYNC(ctx, "k_coarse_scatter");
        hipLaunchKernelGGL(k_fine_place, dim3(128, WR), dim3(kBlock), 0, st, tmp_rec, tabR, ncols, tile_hist, hsum + hist_blocks, count, cursor, idx, 0, nonempty, slices, nslices);
        if (max_slices) {
            const unsigned sg = (unsigned)(max_slices < 1024 ? max_slices : 1024);
            hipLaunchKernelGGL(k_fine_huge_count, dim3(sg), dim3(kBlock), 0, st, tmp_rec, tabR, slices, nslices, slice_hist);
            hipLaunchKernelGGL(k_fine_huge_place, dim3(sg), dim3(kBlock), 0, st, tmp_rec, tabR, slices, nslices, slice_hist, count, cursor, idx, nonempty);
        }
        BP_TRACE_SYNC(ctx, "k_fine_place");
        if (tm) HIPCHK(hipEventRecord(ctx->ev[3], st));
        // count[] = bucket starts, cursor[] = bucket ends
        const unsigned bgrid = (unsigned)((nb + kBlock * kTaskPer - 1) / (kBlock * kTaskPer));   // kTaskPer buckets per lane
        hipLaunchKernelGGL(k_task_count, dim3(bgrid), dim3(kBlock), 0, st, count, cursor, (uint32_t)nb, hsum + hist_blocks, nonempty, (uint32_t)WR, (uint32_t)kTaskTarget, lmin, ntasks, bins);
        hipLaunchKernelGGL(k_task_bins_scan, dim3(1), dim3(kBlock), 0, st, bins, total_tasks);
        hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)scan_blocks), dim3(kBlock), 0, st, ntasks, nb, bsum);
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)scan_blocks), dim3(kBlock), 0, st, ntasks, nb, bsum, task_off, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_task_emit, dim3(bgrid), dim3(kBlock), 0, st, count, cursor, (uint32_t)nb, hsum + hist_blocks, nonempty, (uint32_t)WR, (uint32_t)kTaskTarget, lmin, task_off, bins, order, t_start, t_len,
                           heavy, nheavy, chunks, nchunks);
        BP_TRACE_SYNC(ctx, "k_task_emit");
        if (tm) HIPCHK(hipEventRecord(ctx->ev[4], st));
        // (WPS = 3 -- 168 VGPRs and 156 B of scratch with the round-3 multiplier -- measured again in round 3: 2.24-2.34 against 2.24-2.26 ms)
        hipLaunchKernelGGL((k_accumulate<C, 2>), dim3((unsigned)((max_tasks + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, pts, idx, order, t_start, t_len, total_tasks, tsum);
        BP_TRACE_SYNC(ctx, "k_accumulate<C>");
        if (tm) HIPCHK(hipEventRecord(ctx->ev[5], st));
        hipLaunchKernelGGL(k_combine_chunks<C>, dim3((unsigned)(max_chunks < 256 ? max_chunks : 256)), dim3(kBlock), 0, st, chunks, nchunks, task_off, ntasks, tsum);
        hipLaunchKernelGGL(k_combine_heavy<C>, dim3((unsigned)(max_heavy < 256 ? max_heavy : 256)), dim3(kBlock), 0, st, heavy, nheavy, task_off, ntasks, tsum);
        BP_TRACE_SYNC(ctx, "k_combine_heavy<C>");
        hipLaunchKernelGGL(k_bucket_reduce<C>, dim3(tabR.rboff[WR]), dim3(kBlock), 0, st, tsum, task_off, ntasks, tabR, partial, win_done, wsum);
        BP_TRACE_SYNC(ctx, "k_bucket_reduce<C>");
        if (tm) HIPCHK(hipEventRecord(ctx->ev[6], st));
        HIPCHK(hipGetLastError());
        return BP_OK;
    }

    // last_ms: [0] whole device pipeline, [1] digits + histograms, [2] scan, [3] scatter, [4] task lists, [5] accumulate,
    // [6] end of the accumulate -> end of the pipeline (combine, bucket reduce, window sums)
    static void collect_timing(bp_ctx* ctx) {
        ctx->last_ms_n = 0;
        if (!ctx->timing || !ctx->ev_ready) return;
        if (hipEventSynchronize(ctx->ev[6]) != hipSuccess) return;
        float t;
        if (hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[6]) == hipSuccess) ctx->last_ms[0] = t;
        for (int i = 0; i < 6; i++)
            if (hipEventElapsedTime(&t, ctx->ev[i], ctx->ev[i + 1]) == hipSuccess) ctx->last_ms[1 + i] = t;
        ctx->last_ms_n = 7;
    }

    static const host::Tail<C>& tail() { static const host::Tail<C> t; return t; }

    // nfolds independent folds (1, or the 2 of a paired MSM), each dealt to `chains` Horner walks that run on the context's helper
    // threads beside the calling thread (bp_host_tail.hpp).  Few records: one chain per fold (the walk is then 255 doublings and a
    // handful of additions; threads would only add their wake-up latency).
    static void fold_parallel(bp_ctx* ctx, int nfolds, const XyzzPacked<C>* const* rec, size_t sets, int nrec, const uint16_t* const* pos, uint8_t* const* out_le) {
        using Jac = typename host::Tail<C>::Jac;
        const host::Tail<C>& tl = tail();
        // 4 chains for one set of bit-plane records; 8 / 16 when the sets of several shards are folded together (N x 208 additions:
        // each chain still walks all ~255 doublings, so more chains only thin out the additions)
        const size_t pairs = (size_t)nrec * sets;
        int chains = ctx && ctx->tail_chains > 0 ? ctx->tail_chains : (pairs >= 1200 ? 16 : pairs >= 400 ? 8 : pairs >= 48 ? 4 : 1);
        if (chains > host::Tail<C>::kMaxChains) chains = host::Tail<C>::kMaxChains;
        if (!ctx) chains = 1;
        Jac parts[2 * host::Tail<C>::kMaxChains];
        const int njobs = nfolds * chains;
        std::function<void(int)> job = [&](int j) { tl.fold_chain(rec[j / chains], sets, nrec, pos[j / chains], j % chains, chains, &parts[j]); };
        if (njobs == 1) job(0);
        else ctx->tail_pool.run(njobs, job, njobs - 1);
        if (nfolds == 2) {
            std::function<void(int)> fin = [&](int f) { tl.finish(parts + f * chains, chains, out_le[f]); };
            ctx->tail_pool.run(2, fin, 1);
        } else tl.finish(parts, chains, out_le[0]);
    }
    static void fold1(bp_ctx* ctx, const XyzzPacked<C>* rec, size_t sets, int nrec, const uint16_t* pos, uint8_t* out_le) {
        fold_parallel(ctx, 1, &rec, sets, nrec, &pos, &out_le);
    }

    static void aff_to_le(const Aff<C>& a, uint8_t* out) {
        uint32_t w[Fp::NW];
        fe_pack_words<Fp>(w, fe_from_mont<Fp>(a.x));
        memcpy(out, w, 4 * Fp::NW);
        fe_pack_words<Fp>(w, fe_from_mont<Fp>(a.y));
        memcpy(out + 4 * Fp::NW, w, 4 * Fp::NW);
    }

    static int msm(bp_ctx* ctx, const void* pts, size_t poff, const void* sc, size_t soff, size_t n, uint8_t* out_le, const bp_g1table* tb = nullptr) {
        if (n == 0) { memset(out_le, 0, 2 * 4 * Fp::NW); ctx->last_ms_n = 0; return BP_OK; }
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts + poff, (const ScalarWords*)sc + soff, n, g, nullptr, 0, tb);
        if (rc) return rc;
        if ((rc = host_pinned_reserve(ctx, (size_t)g.nrec * kXyzzBytes))) return rc;
        if (ctx->device_tail) {   // all-device variant: one lane folds the records (see k_tail_fold)
            const size_t pos_bytes = ((size_t)g.nrec * 2 + 15) & ~(size_t)15;
            if ((rc = ctx->scratch.reserve(ctx, pos_bytes + 2 * 4 * Fp::NW))) return rc;
            memcpy(ctx->host_pinned, g.rpos, (size_t)g.nrec * 2);          // host_pinned holds at least nrec records: room for nrec positions
            HIPCHK(hipMemcpyAsync(ctx->scratch.p, ctx->host_pinned, (size_t)g.nrec * 2, hipMemcpyHostToDevice, ctx->stream));
            uint32_t* dev_out = (uint32_t*)((uint8_t*)ctx->scratch.p + pos_bytes);
            hipLaunchKernelGGL(k_tail_fold<C>, dim3(1), dim3(64), 0, ctx->stream, (const XyzzPacked<C>*)ctx->window_sum.p, (const uint16_t*)ctx->scratch.p, g.nrec, dev_out);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(out_le, dev_out, 2 * 4 * Fp::NW, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            collect_timing(ctx);
            return BP_OK;
        }
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, ctx->window_sum.p, (size_t)g.nrec * kXyzzBytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        collect_timing(ctx);
        fold1(ctx, (const XyzzPacked<C>*)ctx->host_pinned, 1, g.nrec, g.rpos, out_le);
        return BP_OK;
    }

    // Asynchronous form of msm(): begin() queues the device pipeline and the D2H copy of the window sums on the
    // context's stream and returns; end() waits for them and runs the host tail.  One MSM in flight per context.
    static int msm_begin(bp_ctx* ctx, const void* pts, size_t poff, const void* sc, size_t soff, size_t n, const bp_g1table* tb = nullptr) {
        ctx->pending = false;
        ctx->pending_n = n;
        if (n == 0) { ctx->pending = true; return BP_OK; }
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts + poff, (const ScalarWords*)sc + soff, n, g, nullptr, 0, tb);
        if (rc) return rc;
        if ((rc = host_pinned_reserve(ctx, (size_t)g.nrec * kXyzzBytes))) return rc;
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, ctx->window_sum.p, (size_t)g.nrec * kXyzzBytes, hipMemcpyDeviceToHost, ctx->stream));
        ctx->pending_nrec = g.nrec;                     // _end folds with the geometry that produced the records
        memcpy(ctx->pending_rpos, g.rpos, sizeof ctx->pending_rpos);
        ctx->pending = true;
        return BP_OK;
    }
    static int msm_end(bp_ctx* ctx, uint8_t* out_le) {
        if (!ctx->pending) return BP_ERR_ARG;
        ctx->pending = false;
        if (ctx->pending_n == 0) { memset(out_le, 0, 2 * 4 * Fp::NW); return BP_OK; }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        collect_timing(ctx);
        fold1(ctx, (const XyzzPacked<C>*)ctx->host_pinned, 1, ctx->pending_nrec, ctx->pending_rpos, out_le);
        return BP_OK;
    }

    // Two scalar sets over the same points in ONE pipeline pass (2W windows): out1 = <sc1, pts>, out2 = <sc2, pts>.
    static int msm2(bp_ctx* ctx, const void* pts, const void* sc1, const void* sc2, size_t n, uint8_t* out1_le, uint8_t* out2_le, size_t nnz = 0, const bp_g1table* tb = nullptr) {
        if (n == 0) { memset(out1_le, 0, 2 * 4 * Fp::NW); memset(out2_le, 0, 2 * 4 * Fp::NW); return BP_OK; }
        MsmGeom g;
        bp_prof().lap(0);
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts, (const ScalarWords*)sc1, n, g, (const ScalarWords*)sc2, nnz, tb);
        if (rc) return rc;
        const int R1 = g.nrec / 2;                      // records of one scalar set
        if ((rc = host_pinned_reserve(ctx, (size_t)g.nrec * kXyzzBytes))) return rc;
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, ctx->window_sum.p, (size_t)g.nrec * kXyzzBytes, hipMemcpyDeviceToHost, ctx->stream));
        bp_prof().lap(1);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        bp_prof().lap(2);
        collect_timing(ctx);
        // the two tails are independent: their Horner chains run side by side on the context's helper threads
        const XyzzPacked<C>* rec = (const XyzzPacked<C>*)ctx->host_pinned;
        const XyzzPacked<C>* recs[2] = {rec, rec + R1};
        const uint16_t* poss[2] = {g.rpos, g.rpos + R1};
        uint8_t* outs[2] = {out1_le, out2_le};
        fold_parallel(ctx, 2, recs, 1, R1, poss, outs);
        bp_prof().lap(3);
        return BP_OK;
    }

    // Record block of the two-stage (sharded) form: W window records followed by ONE header record that names the geometry
    // which produced them, so that bp_msm_g1_finish can refuse sets that do not fit together (ranks whose shard sizes straddle
    // a power of two pick different window widths unless the caller fixes c with bp_ctx_set_window_bits).
    struct RecHeader { uint32_t magic, c, W, fr_bits, cw_first, cw_last, n_wide, nrec, m, small, w0, wn; };   // widths: n_wide windows of cw_first bits, then cw_last; w0, wn: the window group
    static_assert(sizeof(RecHeader) <= sizeof(XyzzPacked<C>), "header must fit one record");
    static constexpr uint32_t kRecMagic = 0x33575042u;   // "BPW3" (round 4: + the window group; blocks of rounds 2 / 3 -- "BPW1" / "BPW2" -- are refused)
    static void fill_header(RecHeader& h, const MsmGeom& g) {
        memset(&h, 0, sizeof h);
        h.magic = kRecMagic; h.c = (uint32_t)g.c; h.W = (uint32_t)g.W_full; h.fr_bits = (uint32_t)C::Fr::BITS;
        h.cw_first = g.cw_first_full; h.cw_last = g.cw_last_full; h.n_wide = g.n_wide_full;
        h.nrec = (uint32_t)g.nrec; h.m = g.m; h.small = g.small ? (uint32_t)g.small_blocks : 0u;
        h.w0 = (uint32_t)g.sub_first; h.wn = (uint32_t)g.sub_count;
    }

    // block_records (window groups only): the header goes to the LAST record of a block of that many (groups differ in record count)
    static int msm_windows_to(bp_ctx* ctx, const void* pts, size_t poff, const void* sc, size_t soff, size_t n, void* device_out, size_t block_records = 0) {
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts + poff, (const ScalarWords*)sc + soff, n, g);
        if (rc) return rc;
        const int W = g.nrec;
        if (block_records && block_records < (size_t)W + 1) return BP_ERR_ARG;
        const size_t hdr_at = block_records ? block_records - 1 : (size_t)W;
        HIPCHK(hipMemcpyAsync(device_out, ctx->window_sum.p, (size_t)W * kXyzzBytes, hipMemcpyDeviceToDevice, ctx->stream));
        if (hdr_at > (size_t)W) HIPCHK(hipMemsetAsync((uint8_t*)device_out + (size_t)W * kXyzzBytes, 0, (hdr_at - W) * kXyzzBytes, ctx->stream));
        if ((rc = host_pinned_reserve(ctx, kXyzzBytes))) return rc;
        memset(ctx->host_pinned, 0, kXyzzBytes);
        fill_header(*(RecHeader*)ctx->host_pinned, g);
        HIPCHK(hipMemcpyAsync((uint8_t*)device_out + hdr_at * kXyzzBytes, ctx->host_pinned, kXyzzBytes, hipMemcpyHostToDevice, ctx->stream));
        // The records are about to be read by another stream (RCCL's) or another device: complete them before returning.
        HIPCHK(hipStreamSynchronize(ctx->stream));
        collect_timing(ctx);
        return BP_OK;
    }

    // host_rec: sets x (W + 1) records in host memory (header last in each set); validates the headers and folds
    static int finish_host(int c_override, const XyzzPacked<C>* host_rec, size_t sets, size_t n_per_set, uint8_t* out_le, const bp_tuning* tn = nullptr, bp_ctx* ctx = nullptr) {
        MsmGeom g;
        if (msm_geom(g, C::Fr::BITS, n_per_set, c_override, 1, 0, tn)) return BP_ERR_ARG;
        const int W = g.nrec;
        RecHeader want;
        fill_header(want, g);
        std::vector<XyzzPacked<C>> packed;
        try { packed.resize(sets * (size_t)W); } catch (...) { return BP_ERR_DEVICE; }     // no exception crosses the C ABI
        for (size_t s = 0; s < sets; s++) {
            const XyzzPacked<C>* set = host_rec + s * (size_t)(W + 1);
            if (memcmp(&set[W], &want, sizeof want) != 0) return BP_ERR_ARG;     // geometry of this set differs from the caller's
            memcpy(&packed[s * (size_t)W], set, (size_t)W * kXyzzBytes);
        }
        fold1(ctx, packed.data(), sets, W, g.rpos, out_le);
        return BP_OK;
    }

    // Blocks of `stride` records each, the LAST one the header, records [0, header.nrec) valid (the rest padding): shards of a 2-D split
    // (index range x window group) have different record counts per block.  Every header must name the caller's full geometry; its
    // window group gives the block's bit positions.  One fold over all records.
    static int finish_blocks_host(int c_override, const XyzzPacked<C>* host_rec, size_t nblocks, size_t stride, size_t n_per_set, uint8_t* out_le,
                                  const bp_tuning* tn = nullptr, bp_ctx* ctx = nullptr) {
        if (stride < 2) return BP_ERR_ARG;
        MsmGeom full;
        if (msm_geom(full, C::Fr::BITS, n_per_set, c_override, 1, 0, tn, 0)) return BP_ERR_ARG;
        std::vector<XyzzPacked<C>> packed;
        std::vector<uint16_t> pos;
        uint32_t cover[kMaxWindows] = {0};             // how many blocks hold window w: equal for all w, or the groups do not tile the windows
        for (size_t b = 0; b < nblocks; b++) {
            const XyzzPacked<C>* blk = host_rec + b * stride;
            RecHeader h;
            memcpy(&h, &blk[stride - 1], sizeof h);
            MsmGeom g = full;
            if (h.w0 >= (uint32_t)full.tab.W || h.wn == 0 || h.wn > (uint32_t)full.tab.W - h.w0) return BP_ERR_ARG;
            for (uint32_t w = h.w0; w < h.w0 + h.wn; w++) cover[w]++;
            if (geom_subset(g, (int)h.w0, (int)h.wn, tn)) return BP_ERR_ARG;
            RecHeader want;
            fill_header(want, g);
            if (memcmp(&h, &want, sizeof want) != 0 || (size_t)g.nrec > stride - 1) return BP_ERR_ARG;
            packed.insert(packed.end(), blk, blk + g.nrec);
            pos.insert(pos.end(), g.rpos, g.rpos + g.nrec);
        }
        for (int w = 1; w < full.tab.W; w++) if (cover[w] != cover[0]) return BP_ERR_ARG;
        if (packed.size() > (size_t)kMaxRecords) return BP_ERR_ARG;
        fold1(ctx, packed.data(), 1, (int)packed.size(), pos.data(), out_le);
        return BP_OK;
    }

    // bp_msm_g1_multi, device half of one shard: queue the pipeline with the shared window width c and the D2H copy of the W
    // window sums into this context's pinned buffer; no synchronisation.
    static int multi_begin(bp_ctx* ctx, const bp_g1vec* pts, const bp_frvec* sc, int c, const bp_tuning* tn, int* W_out) {
        const int saved = ctx->c_override;
        const bp_tuning saved_tn = ctx->tuning;
        ctx->c_override = c;
        ctx->tuning = *tn;                       // one geometry for every shard: the first context's knobs
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts->d, (const ScalarWords*)sc->d, pts->n, g);
        ctx->c_override = saved;
        ctx->tuning = saved_tn;
        if (rc) return rc;
        if ((rc = host_pinned_reserve(ctx, (size_t)g.nrec * kXyzzBytes))) return rc;
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, ctx->window_sum.p, (size_t)g.nrec * kXyzzBytes, hipMemcpyDeviceToHost, ctx->stream));
        *W_out = g.nrec;
        return BP_OK;
    }

    // one affine point (canonical LE) -> a packed XYZZ record in the device's Montgomery radix (host arithmetic)
    static void record_from_affine(const uint8_t* le, XyzzPacked<C>* out) {
        uint32_t xw[Fp::NW], yw[Fp::NW];
        memcpy(xw, le, 4 * Fp::NW);
        memcpy(yw, le + 4 * Fp::NW, 4 * Fp::NW);
        Aff<C> a;
        a.x = fe_to_mont<Fp>(fe_unpack_words<Fp>(xw));
        a.y = fe_to_mont<Fp>(fe_unpack_words<Fp>(yw));
        *out = xyzz_pack(xyzz_from_aff(a));
    }

    static int msm_finish(bp_ctx* ctx, const void* device_records, size_t sets, size_t n_per_set, uint8_t* out_le) {
        MsmGeom g;
        if (msm_geom(g, C::Fr::BITS, n_per_set, ctx->c_override, 1, 0, &ctx->tuning)) return BP_ERR_ARG;
        size_t bytes = sets * (size_t)(g.nrec + 1) * kXyzzBytes;
        int rc;
        if ((rc = host_pinned_reserve(ctx, bytes))) return rc;
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, device_records, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        collect_timing(ctx);
        return finish_host(ctx->c_override, (const XyzzPacked<C>*)ctx->host_pinned, sets, n_per_set, out_le, &ctx->tuning, ctx);
    }

    // validate = true: BP_ERR_ARG if a coordinate is >= p or a point is off the curve (see k_points_to_resident)
    static int upload_points(bp_ctx* ctx, const uint8_t* le, size_t n, void* d_out, bool validate) {
        size_t bytes = n * 2 * 4 * Fp::NW;
        int rc;
        if ((rc = ctx->scratch.reserve(ctx, bytes ? bytes : 16))) return rc;
        if ((rc = ctx->flags.reserve(ctx, 64))) return rc;
        uint32_t* flag = validate ? (uint32_t*)ctx->flags.p : nullptr;
        uint32_t host_flag = 0;
        if (flag) HIPCHK(hipMemsetAsync(flag, 0, 4, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->scratch.p, le, bytes, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_points_to_resident<C>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                           (const uint32_t*)ctx->scratch.p, n, (AffPacked<C>*)d_out, flag);
        HIPCHK(hipGetLastError());
        if (flag) HIPCHK(hipMemcpyAsync(&host_flag, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));   // `le` is a borrowed host buffer
        return host_flag ? BP_ERR_ARG : BP_OK;
    }

    static int check_scalars(bp_ctx* ctx, const void* d_sc, size_t n) {
        int rc;
        if ((rc = ctx->flags.reserve(ctx, 64))) return rc;
        uint32_t* flag = (uint32_t*)ctx->flags.p;
        uint32_t host_flag = 0;
        HIPCHK(hipMemsetAsync(flag, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_check_scalars<C>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const ScalarWords*)d_sc, n, flag);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&host_flag, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return host_flag ? BP_ERR_ARG : BP_OK;
    }

    static int download_points(bp_ctx* ctx, const void* d_in, size_t offset, size_t n, uint8_t* le) {
        size_t bytes = n * 2 * 4 * Fp::NW;
        int rc;
        if ((rc = ctx->scratch.reserve(ctx, bytes ? bytes : 16))) return rc;
        hipLaunchKernelGGL(k_points_from_resident<C>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                           (const AffPacked<C>*)d_in + offset, n, (uint32_t*)ctx->scratch.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(le, ctx->scratch.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return BP_OK;
    }

    // table of d * 2^(4j) * G via the generic scalar-mul kernel (960 elements, once per context)
    static int ensure_fixed_base_table(bp_ctx* ctx) {
        if (ctx->fixed_base_ready) return BP_OK;
        const size_t m = (size_t)kFixedBaseWindows * 15;
        int rc;
        if ((rc = ctx->fixed_base_table.reserve(ctx, m * kPointBytes))) return rc;
        if ((rc = ctx->scratch.reserve(ctx, m * 32))) return rc;
        std::vector<ScalarWords> ks(m);
        for (int j = 0; j < kFixedBaseWindows; j++)
            for (int d = 1; d <= 15; d++) {
                ScalarWords s;
                memset(&s, 0, sizeof s);
                s.w[(4 * j) >> 5] = (uint32_t)d << ((4 * j) & 31);      // d * 2^(4j): a nibble never straddles a word
                ks[(size_t)j * 15 + d - 1] = s;
            }
        HIPCHK(hipMemcpyAsync(ctx->scratch.p, ks.data(), m * 32, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_scalar_mul<C>, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const AffPacked<C>*)nullptr,
                           (const ScalarWords*)ctx->scratch.p, m, (AffPacked<C>*)ctx->fixed_base_table.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));   // ks is a host temporary
        ctx->fixed_base_ready = true;
        return BP_OK;
    }

    static int fixed_base(bp_ctx* ctx, const void* k, size_t n, void* out) {
        int rc = ensure_fixed_base_table(ctx);
        if (rc) return rc;
        hipLaunchKernelGGL(k_fixed_base<C>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                           (const AffPacked<C>*)ctx->fixed_base_table.p, (const ScalarWords*)k, n, (AffPacked<C>*)out);
        HIPCHK(hipGetLastError());
        return BP_OK;
    }

    static int scalar_mul(bp_ctx* ctx, const void* base, const void* k, size_t n, void* out) {
        hipLaunchKernelGGL(k_scalar_mul<C>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                           (const AffPacked<C>*)base, (const ScalarWords*)k, n, (AffPacked<C>*)out);
        HIPCHK(hipGetLastError());
        return BP_OK;
    }
};

#define DISPATCH(ctx_, expr)                                              \
    do {                                                                  \
        if ((ctx_)->curve == BP_CURVE_BLS12_381) { using I = Impl<Bls381>; return expr; } \
        else { using I = Impl<Bn254>; return expr; }                      \
    } while (0)

static int set_device(const bp_ctx* ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    return BP_OK;
}
int bp_internal_set_device(const bp_ctx* ctx) { return set_device(ctx); }

int bp_internal_msm(bp_ctx* ctx, const void* points, const void* scalars, size_t n, uint8_t* out_le, const bp_g1table* tb) {
    DISPATCH(ctx, I::msm(ctx, points, 0, scalars, 0, n, out_le, tb));
}

int bp_internal_msm2(bp_ctx* ctx, const void* points, const void* scalars1, const void* scalars2, size_t n, uint8_t* out1_le, uint8_t* out2_le,
                     size_t nnz, const bp_g1table* tb) {
    DISPATCH(ctx, I::msm2(ctx, points, scalars1, scalars2, n, out1_le, out2_le, nnz, tb));
}

// ---- building blocks of the sharded inner-product argument (bp_capi_ipp.hip: bp_ipp_create_multi) ----
// The window width a paired MSM of n terms (nnz non-zero per set) picks on its own: the shards then all run with it FIXED.
int bp_internal_pair_width(bp_ctx* ctx, size_t n, size_t nnz) {
    MsmGeom g;
    const int bits = ctx->curve == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS;
    if (msm_geom(g, bits, n, ctx->c_override, 2, nnz, &ctx->tuning)) return 0;
    return g.c;
}
// Device stage of a paired MSM with the window width fixed to c (or the table's width): the 2 x (nrec / 2) tail records are
// copied to ctx->host_pinned asynchronously; the caller synchronises ctx->stream before reading them.
template <class C>
static int msm2_begin_impl(bp_ctx* ctx, const void* pts, const void* sc1, const void* sc2, size_t n, int c, size_t nnz, const bp_g1table* tb, int* nrec_out, uint16_t* rpos_out) {
    const int saved = ctx->c_override;
    ctx->c_override = c;
    MsmGeom g;
    int rc = Impl<C>::msm_windows(ctx, (const AffPacked<C>*)pts, (const ScalarWords*)sc1, n, g, (const ScalarWords*)sc2, nnz, tb);
    ctx->c_override = saved;
    if (rc) return rc;
    if ((rc = host_pinned_reserve(ctx, (size_t)g.nrec * Impl<C>::kXyzzBytes))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->host_pinned, ctx->window_sum.p, (size_t)g.nrec * Impl<C>::kXyzzBytes, hipMemcpyDeviceToHost, ctx->stream));
    *nrec_out = g.nrec;
    memcpy(rpos_out, g.rpos, (size_t)g.nrec * sizeof(uint16_t));
    return BP_OK;
}
int bp_internal_msm2_begin(bp_ctx* ctx, const void* pts, const void* sc1, const void* sc2, size_t n, int c, size_t nnz, const bp_g1table* tb, int* nrec_out, uint16_t* rpos_out) {
    if (ctx->curve == BP_CURVE_BLS12_381) return msm2_begin_impl<Bls381>(ctx, pts, sc1, sc2, n, c, nnz, tb, nrec_out, rpos_out);
    return msm2_begin_impl<Bn254>(ctx, pts, sc1, sc2, n, c, nnz, tb, nrec_out, rpos_out);
}
// out_le[f] = sum over the `sets` record sets of fold f (f < nfolds <= 2); rec[f] holds sets x nrec records, set-major
int bp_internal_fold_sets(bp_ctx* ctx, int nfolds, const void* const* rec, size_t sets, int nrec, const uint16_t* const* pos, uint8_t* const* out_le) {
    if (ctx->curve == BP_CURVE_BLS12_381) Impl<Bls381>::fold_parallel(ctx, nfolds, (const XyzzPacked<Bls381>* const*)rec, sets, nrec, pos, out_le);
    else Impl<Bn254>::fold_parallel(ctx, nfolds, (const XyzzPacked<Bn254>* const*)rec, sets, nrec, pos, out_le);
    return BP_OK;
}

static bool bp_fr_is_canonical_or_zero(int curve, const uint8_t* x_le32) {
    uint32_t w[8];
    memcpy(w, x_le32, 32);
    return curve == BP_CURVE_BLS12_381 ? words_lt_mod<Bls381Fr>(w) : words_lt_mod<Bn254Fr>(w);
}
// out[j] = k1[j] g + k2[j] h for j < count on the HOST (a handful of two-term commitments; bp_host_tail.hpp: Tail::mul2), the count
// jobs side by side on the context's helper threads.  Inputs must be valid (the callers pass generators and canonical scalars).
int bp_internal_host_mul2(bp_ctx* ctx, const uint8_t* g_le, const uint8_t* h_le, const uint8_t* k1_le32, const uint8_t* k2_le32, int count, uint8_t* const* out_le) {
    if (count <= 0) return BP_OK;
    const bool ok = ctx->curve == BP_CURVE_BLS12_381 ? Impl<Bls381>::tail().valid_affine(g_le) && Impl<Bls381>::tail().valid_affine(h_le)
                                                     : Impl<Bn254>::tail().valid_affine(g_le) && Impl<Bn254>::tail().valid_affine(h_le);
    if (!ok) return BP_ERR_ARG;                              // as bp_g1vec_upload would say of the same bytes
    for (int j = 0; j < count; j++)
        if (!bp_fr_is_canonical_or_zero(ctx->curve, k1_le32 + 32 * j) || !bp_fr_is_canonical_or_zero(ctx->curve, k2_le32 + 32 * j)) return BP_ERR_ARG;
    std::function<void(int)> job = [&](int j) {
        if (ctx->curve == BP_CURVE_BLS12_381) Impl<Bls381>::tail().mul2(g_le, h_le, k1_le32 + 32 * j, k2_le32 + 32 * j, out_le[j]);
        else Impl<Bn254>::tail().mul2(g_le, h_le, k1_le32 + 32 * j, k2_le32 + 32 * j, out_le[j]);
    };
    if (count == 1) job(0);
    else ctx->tail_pool.run(count, job, count - 1);
    return BP_OK;
}

static uint64_t next_table_id() {
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

// Window-multiples table of n resident points (bp_g1vec_precompute): allocated from the context's pool, built on its stream.
int bp_internal_table_build(bp_ctx* ctx, const void* points, size_t n, int c, bp_g1table** out) {
    *out = nullptr;
    const int fr_bits = ctx->curve == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS;
    if (c < 2 || c > 16 || n == 0) return BP_ERR_ARG;
    const int W1 = (fr_bits + 1 + c - 1) / c;
    if ((uint64_t)W1 * n >= ((uint64_t)1 << 31)) return BP_ERR_ARG;           // a table row index shares its word with the sign bit
    const size_t pt = 2 * (size_t)fp_bytes_of(ctx->curve), xz = bp_msm_record_bytes(ctx->curve);
    bp_g1table* t = new (std::nothrow) bp_g1table();
    if (!t) return BP_ERR_DEVICE;
    t->pool = ctx->pool; t->device = ctx->device; t->n = n; t->c = c; t->W = W1; t->id = next_table_id();
    t->d = ctx->pool->get((size_t)W1 * n * pt, &t->cap);
    PoolBlock tmp, pre;                                                        // parked XYZZ values and running products (returned to the pool in stream order)
    if (!t->d || !tmp.alloc(ctx, (size_t)(W1 - 1) * n * xz + 16) || !pre.alloc(ctx, (size_t)(W1 - 1) * n * (pt / 2) + 16)) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (ctx->curve == BP_CURVE_BLS12_381)
        hipLaunchKernelGGL(k_table_build<Bls381>, grid, dim3(kBlock), 0, ctx->stream, (const AffPacked<Bls381>*)points, n, c, W1, (XyzzPacked<Bls381>*)tmp.p,
                           (FePacked<Bls381Fp>*)pre.p, (AffPacked<Bls381>*)t->d);
    else
        hipLaunchKernelGGL(k_table_build<Bn254>, grid, dim3(kBlock), 0, ctx->stream, (const AffPacked<Bn254>*)points, n, c, W1, (XyzzPacked<Bn254>*)tmp.p,
                           (FePacked<Bn254Fp>*)pre.p, (AffPacked<Bn254>*)t->d);
    if (hipGetLastError() != hipSuccess) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
    *out = t;
    return BP_OK;
}

// Digit multiples m P_i (m = 1 .. 8) of n <= kSmallDigitMax resident points for k_small_msm: one launch, ~100 us, n x 8 x 192 B.
int bp_internal_digit_table_build(bp_ctx* ctx, const void* points, size_t n, bp_g1table** out) {
    *out = nullptr;
    if (n == 0 || n > kSmallDigitMax) return BP_ERR_ARG;
    const size_t xz = bp_msm_record_bytes(ctx->curve);
    const int rows = 1 << (kSmallDigitBits - 1);
    bp_g1table* t = new (std::nothrow) bp_g1table();
    if (!t) return BP_ERR_DEVICE;
    t->pool = ctx->pool; t->device = ctx->device; t->n = n; t->c = kSmallDigitBits; t->W = rows; t->digits = true;
    t->d = ctx->pool->get((size_t)rows * n * xz, &t->cap);
    if (!t->d) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
    const dim3 grid((unsigned)((n + 63) / 64));
    const bool par = (size_t)rows * n <= kDigitTableParMax;                       // small: a lane per (point, multiple), depth 3 + 2 instead of 8
    const dim3 pgrid((unsigned)(((size_t)rows * n + kBlock - 1) / kBlock));
    if (ctx->curve == BP_CURVE_BLS12_381) {
        if (par) hipLaunchKernelGGL(k_digit_table_build_par<Bls381>, pgrid, dim3(kBlock), 0, ctx->stream, (const AffPacked<Bls381>*)points, (uint32_t)n, (XyzzPacked<Bls381>*)t->d, (uint32_t)rows);
        else hipLaunchKernelGGL(k_digit_table_build<Bls381>, grid, dim3(64), 0, ctx->stream, (const AffPacked<Bls381>*)points, (uint32_t)n, (XyzzPacked<Bls381>*)t->d, (uint32_t)rows);
    } else {
        if (par) hipLaunchKernelGGL(k_digit_table_build_par<Bn254>, pgrid, dim3(kBlock), 0, ctx->stream, (const AffPacked<Bn254>*)points, (uint32_t)n, (XyzzPacked<Bn254>*)t->d, (uint32_t)rows);
        else hipLaunchKernelGGL(k_digit_table_build<Bn254>, grid, dim3(64), 0, ctx->stream, (const AffPacked<Bn254>*)points, (uint32_t)n, (XyzzPacked<Bn254>*)t->d, (uint32_t)rows);
    }
    if (hipGetLastError() != hipSuccess) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
    *out = t;
    return BP_OK;
}

// Table of the concatenation [G[offG .. offG+n) | H[offH .. offH+n) | extra] from the tables of G and H (device copies, row by row)
// and the window multiples of the one extra point (host arithmetic, ~0.1 ms) -- the vectors the prover's MSMs run over
// ([G | H | Q] of an inner-product argument, [G | H | B_blinding] of the R1CS commitments).  *out stays NULL (BP_OK) when G or H has
// no table, or their widths differ: the caller then runs the plain pipeline.
int bp_internal_table_concat(bp_ctx* ctx, const bp_g1vec* G, size_t offG, const bp_g1vec* H, size_t offH, size_t n, const uint8_t* extra_le, bp_g1table** out) {
    *out = nullptr;
    if (offG > G->n || n > G->n - offG || offH > H->n || n > H->n - offH) return BP_ERR_LENGTH;
    const bp_g1table *tg = G->table ? G->table : G->tview, *th = H->table ? H->table : H->tview;
    if (!G->table) offG += G->tview_off;
    if (!H->table) offH += H->tview_off;
    if (!tg || !th || tg->c != th->c || tg->W != th->W || offG + n > tg->n || offH + n > th->n) return BP_OK;
    const size_t m = 2 * n + (extra_le ? 1 : 0);         // extra_le == NULL: [G | H] alone (the verifiers' table, bp_internal_gh_table)
    const int W1 = tg->W, c = tg->c;
    if ((uint64_t)W1 * m >= ((uint64_t)1 << 31)) return BP_OK;
    const size_t pt = 2 * (size_t)fp_bytes_of(ctx->curve);
    bp_g1table* t = new (std::nothrow) bp_g1table();
    if (!t) return BP_ERR_DEVICE;
    t->pool = ctx->pool; t->device = ctx->device; t->n = m; t->c = c; t->W = W1; t->id = next_table_id();
    t->d = ctx->pool->get((size_t)W1 * m * pt, &t->cap);
    PoolBlock raw, conv;
    if (!t->d || !raw.alloc(ctx, (size_t)W1 * pt) || !conv.alloc(ctx, (size_t)W1 * pt)) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
    int rc = host_pinned_reserve(ctx, (size_t)W1 * pt);
    if (rc) { bp_internal_table_free(t); return rc; }
    hipStream_t s = ctx->stream;
    // the pinned staging buffer may still be the source / target of an earlier copy on this stream
    if (hipStreamSynchronize(s) != hipSuccess) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
    bool ok = true;
    if (extra_le) {
        if (ctx->curve == BP_CURVE_BLS12_381) Impl<Bls381>::tail().window_multiples(extra_le, c, W1, (uint8_t*)ctx->host_pinned);
        else Impl<Bn254>::tail().window_multiples(extra_le, c, W1, (uint8_t*)ctx->host_pinned);
        ok = hipMemcpyAsync(raw.p, ctx->host_pinned, (size_t)W1 * pt, hipMemcpyHostToDevice, s) == hipSuccess;
    }
    if (ok && extra_le) {
        if (ctx->curve == BP_CURVE_BLS12_381)
            hipLaunchKernelGGL(k_points_to_resident<Bls381>, dim3((unsigned)((W1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, (const uint32_t*)raw.p, (size_t)W1, (AffPacked<Bls381>*)conv.p, (uint32_t*)nullptr);
        else
            hipLaunchKernelGGL(k_points_to_resident<Bn254>, dim3((unsigned)((W1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, (const uint32_t*)raw.p, (size_t)W1, (AffPacked<Bn254>*)conv.p, (uint32_t*)nullptr);
        ok = hipGetLastError() == hipSuccess;
    }
    uint8_t* dst = (uint8_t*)t->d;
    if (ok && n) {
        ok = hipMemcpy2DAsync(dst, m * pt, (const uint8_t*)tg->d + offG * pt, tg->n * pt, n * pt, (size_t)W1, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipMemcpy2DAsync(dst + n * pt, m * pt, (const uint8_t*)th->d + offH * pt, th->n * pt, n * pt, (size_t)W1, hipMemcpyDeviceToDevice, s) == hipSuccess;
    }
    if (extra_le) ok = ok && hipMemcpy2DAsync(dst + 2 * n * pt, m * pt, conv.p, pt, pt, (size_t)W1, hipMemcpyDeviceToDevice, s) == hipSuccess;
    // the multiples live in the pinned buffer until the H2D copy has run: wait (this path runs once per proof, not per round)
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) { bp_internal_table_free(t); return BP_ERR_DEVICE; }
    *out = t;
    return BP_OK;
}

// The verifiers' table of [G[0 .. n) | H[0 .. n)]: the rows of the two vectors' own tables side by side -- 2 W n row copies
// (201 MB at n = 2^16, c = 16), made on the first verification against these generators and kept on the context.  The key is the pair of
// table ids (unique per build, so a vector that was freed or precomputed again can never be mistaken) and the row ranges.
static int bp_internal_gh_table(bp_ctx* ctx, const bp_g1vec* G, const bp_g1vec* H, size_t n, const bp_g1table** out) {
    *out = nullptr;
    const bp_g1table *tg = G->table ? G->table : G->tview, *th = H->table ? H->table : H->tview;
    if (!tg || !th || tg->device != ctx->device || th->device != ctx->device || tg->c != th->c || tg->W != th->W) return BP_OK;   // (whatever is kept stays)
    const size_t offG = G->table ? 0 : G->tview_off, offH = H->table ? 0 : H->tview_off;
    if (ctx->gh_table && ctx->gh_id[0] == tg->id && ctx->gh_id[1] == th->id && ctx->gh_off[0] == offG && ctx->gh_off[1] == offH && ctx->gh_n == n) {
        *out = ctx->gh_table;
        return BP_OK;
    }
    if (ctx->gh_table) {                    // other generators: the old table's rows may still be read by an MSM queued on a sibling stream
        for (bp_ctx* h : ctx->helper) if (h && hipStreamSynchronize(h->stream) != hipSuccess) return BP_ERR_DEVICE;
        bp_internal_table_free(ctx->gh_table);
        ctx->gh_table = nullptr;
    }
    bp_g1table* t = nullptr;
    int rc = bp_internal_table_concat(ctx, G, 0, H, 0, n, nullptr, &t);
    if (rc || !t) return rc;
    ctx->gh_table = t;
    ctx->gh_id[0] = tg->id; ctx->gh_id[1] = th->id; ctx->gh_off[0] = offG; ctx->gh_off[1] = offH; ctx->gh_n = n;
    *out = t;
    return BP_OK;
}

template <class C>
static int msm_extras_gh_impl(bp_ctx* ctx, const void* xpts, const void* xsc, size_t nx, const void* gh_sc, const bp_g1table* T, size_t n, uint8_t* out_le) {
    using I = Impl<C>;
    bp_ctx* side = nx ? bp_internal_helper(ctx, 0) : nullptr;
    if (nx && !side) return BP_ERR_DEVICE;
    int rc = BP_OK;
    if (side) {                                  // the other terms first: their (latency-bound) launches fill the gaps of the big pipeline
        if ((rc = bp_internal_fork(ctx, side))) return rc;
        rc = I::msm_begin(side, xpts, 0, xsc, 0, nx);
    }
    if (rc == BP_OK) rc = I::msm_begin(ctx, T->d, 0, gh_sc, 0, 2 * n, T);
    bool ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (side) ok = (hipStreamSynchronize(side->stream) == hipSuccess) && ok;     // both drained whatever happened: the callers' temporaries are in use there
    ctx->pending = false;
    if (side) side->pending = false;
    if (rc) return rc;
    if (!ok) return BP_ERR_DEVICE;
    const int ra = ctx->pending_nrec, rb = side ? side->pending_nrec : 0;
    if (ra + rb > kMaxRecords) return BP_ERR_DEVICE;
    std::vector<XyzzPacked<C>> rec((size_t)(ra + rb));
    std::vector<uint16_t> pos((size_t)(ra + rb));
    memcpy(rec.data(), ctx->host_pinned, (size_t)ra * I::kXyzzBytes);
    memcpy(pos.data(), ctx->pending_rpos, (size_t)ra * sizeof(uint16_t));
    if (rb) {
        memcpy(rec.data() + ra, side->host_pinned, (size_t)rb * I::kXyzzBytes);
        memcpy(pos.data() + ra, side->pending_rpos, (size_t)rb * sizeof(uint16_t));
    }
    I::fold1(ctx, rec.data(), 1, ra + rb, pos.data(), out_le);
    return BP_OK;
}

static int gh_table_for(bp_ctx* ctx, const bp_g1vec* G, const bp_g1vec* H, size_t n, const bp_g1table** T) {
    *T = nullptr;
    const uint32_t lim = ctx->tuning.verify_tables;          // 0 (default): never -- measured slower than the plain MSM on MI355X
    if (lim < 2 || n < lim || n > G->n || n > H->n || ctx->device_tail || ctx->win_count || ctx->pending) return BP_OK;    // (pending: the caller's own bp_msm_g1_begin owns the record buffers)
    return bp_internal_gh_table(ctx, G, H, n, T);
}
int bp_internal_gh_ready(bp_ctx* ctx, const bp_g1vec* G, const bp_g1vec* H, size_t n, bool* yes) {
    const bp_g1table* T = nullptr;
    int rc = gh_table_for(ctx, G, H, n, &T);
    *yes = rc == BP_OK && T != nullptr;
    return rc;
}
int bp_internal_msm_extras_gh(bp_ctx* ctx, const void* xpts, const void* xsc, size_t nx, const void* gh_sc, const bp_g1vec* G, const bp_g1vec* H, size_t n,
                              uint8_t* out_le, bool* done) {
    *done = false;
    const bp_g1table* T = nullptr;
    int rc = gh_table_for(ctx, G, H, n, &T);
    if (rc || !T) return rc;
    *done = true;
    if (ctx->curve == BP_CURVE_BLS12_381) return msm_extras_gh_impl<Bls381>(ctx, xpts, xsc, nx, gh_sc, T, n, out_le);
    return msm_extras_gh_impl<Bn254>(ctx, xpts, xsc, nx, gh_sc, T, n, out_le);
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* bp_version(void) { return "bpmsm 0.1 (gfx950; unsaturated 30-bit limbs; Pippenger/XYZZ)"; }

int bp_device_count(void) {
    return bp_guard([&]() -> int {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
    });
}

int bp_curve_params(int curve_id, bp_curve_info* out) {
    return bp_guard([&]() -> int {
    if (!out || !curve_ok(curve_id)) return BP_ERR_ARG;
    memset(out, 0, sizeof *out);
    out->curve_id = curve_id;
    out->fr_bytes = 32;
    if (curve_id == BP_CURVE_BLS12_381) {
        out->fp_bytes = 48; out->modbytes = Bls381::MODBYTES; out->fr_bits = Bls381Fr::BITS;
        memcpy(out->p_le, Bls381FpW::MODW, 48); memcpy(out->r_le, Bls381FrW::MODW, 32);
        memcpy(out->gen_le, Bls381::GX, 48); memcpy(out->gen_le + 48, Bls381::GY, 48);
    } else {
        out->fp_bytes = 32; out->modbytes = Bn254::MODBYTES; out->fr_bits = Bn254Fr::BITS;
        memcpy(out->p_le, Bn254FpW::MODW, 32); memcpy(out->r_le, Bn254FrW::MODW, 32);
        memcpy(out->gen_le, Bn254::GX, 32); memcpy(out->gen_le + 32, Bn254::GY, 32);
    }
    return BP_OK;
    });
}

int bp_ctx_create(int curve_id, int device_ordinal, bp_ctx** out) {
    return bp_guard([&]() -> int {
    if (!out || !curve_ok(curve_id)) return BP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BP_ERR_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= ndev) return BP_ERR_ARG;
    HIPCHK(hipSetDevice(device_ordinal));
    bp_ctx* ctx = new (std::nothrow) bp_ctx();
    if (!ctx) return BP_ERR_DEVICE;
    ctx->curve = curve_id;
    ctx->device = device_ordinal;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return BP_ERR_DEVICE; }
    ctx->stream = ctx->own_stream;
    ctx->pool = new (std::nothrow) DevPool();
    if (!ctx->pool) { (void)hipStreamDestroy(ctx->own_stream); delete ctx; return BP_ERR_DEVICE; }
    ctx->pool->device = device_ordinal;
    *out = ctx;
    return BP_OK;
    });
}

int bp_ctx_destroy(bp_ctx* ctx) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_OK;
    for (auto& h : ctx->helper) { if (h) bp_ctx_destroy(h); h = nullptr; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    ctx->fixed_base_table.release();
    if (ctx->gh_table) { bp_internal_table_free(ctx->gh_table); ctx->gh_table = nullptr; }
    for (DevBuf* b : {&ctx->count, &ctx->cursor, &ctx->block_sums, &ctx->idx, &ctx->code, &ctx->tile_hist, &ctx->tmp_idx, &ctx->ntasks, &ctx->task_off, &ctx->order, &ctx->t_start,
                      &ctx->t_len, &ctx->tsum, &ctx->heavy, &ctx->heavy_chunks, &ctx->meta, &ctx->partial, &ctx->window_sum, &ctx->scratch, &ctx->flags, &ctx->huge, &ctx->negbits}) b->release();
    if (ctx->pool) { ctx->pool->trim(); ctx->pool->release(); }     // cached blocks go back to the driver now; live handles keep the (empty) pool alive
    if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->ev_ready) for (auto& e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return BP_OK;
    });
}

// Sibling context k of `ctx` (same curve, device and options; its own stream and workspace), created on first use and destroyed
// with `ctx`.  bp_internal_fork makes the sibling's stream wait for everything queued on ctx->stream so far, so vectors built on
// the parent can be consumed there.  Used for independent MSMs in flight from one host thread (bp_capi_r1cs.hip).
bp_ctx* bp_internal_helper(bp_ctx* ctx, int k) {
    if (!ctx || k < 0 || k > 1) return nullptr;
    if (!ctx->helper[k]) {
        bp_ctx* h = nullptr;
        if (bp_ctx_create(ctx->curve, ctx->device, &h) != BP_OK) return nullptr;
        ctx->helper[k] = h;
    }
    bp_ctx* h = ctx->helper[k];
    h->c_override = ctx->c_override;
    h->device_tail = ctx->device_tail;
    h->tuning = ctx->tuning;
    return h;
}
int bp_internal_fork(bp_ctx* ctx, bp_ctx* sibling) {
    return bp_guard([&]() -> int {
    if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(hipStreamWaitEvent(sibling->stream, ctx->ev_fork, 0));
    return BP_OK;
    });
}

int bp_ctx_set_stream(bp_ctx* ctx, void* hip_stream) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_ERR_ARG;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next != ctx->stream) {   // pool blocks are recycled in stream order: drain the old stream before work moves to another
        int rc = set_device(ctx); if (rc) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ctx->stream = next;
    return BP_OK;
    });
}

int bp_ctx_synchronize(bp_ctx* ctx) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BP_OK;
    });
}

int bp_ctx_set_window_bits(bp_ctx* ctx, int c) {
    return bp_guard([&]() -> int {
    if (!ctx || c < 0 || c > 16 || c == 1) return BP_ERR_ARG;
    ctx->c_override = c;
    return BP_OK;
    });
}

int bp_ctx_set_tuning(bp_ctx* ctx, int knob, long value) {
    return bp_guard([&]() -> int {
    if (!ctx || value < 0) return BP_ERR_ARG;
    switch (knob) {
    case BP_TUNE_TILE:          // k_digits_bin / k_coarse_scatter give every lane tile / 256 scalars: anything else would skip scalars
        if (value != 0 && (value < kBlock || value > 16384 || value % kBlock)) return BP_ERR_ARG;
        ctx->tuning.tile = (uint32_t)value;
        return BP_OK;
    case BP_TUNE_REDUCE_M:      // the bit-plane records need a power of two
        if (value != 0 && (value > 16384 || (value & (value - 1)))) return BP_ERR_ARG;
        ctx->tuning.reduce_m = (uint32_t)value;
        return BP_OK;
    case BP_TUNE_TASK_TARGET:
        if (value != 0 && (value < 1024 || value > (1L << 28))) return BP_ERR_ARG;
        ctx->tuning.task_target = (uint64_t)value;
        return BP_OK;
    case BP_TUNE_TAIL_CHAINS:   // independent Horner walks of the host tail (helper threads): 1 .. 16
        if (value > host::Tail<Bls381>::kMaxChains) return BP_ERR_ARG;
        ctx->tail_chains = (int)value;
        return BP_OK;
    case BP_TUNE_SMALL_MSM:
        if (value > 1) return BP_ERR_ARG;
        ctx->tuning.small_msm = value != 0;
        return BP_OK;
    case BP_TUNE_GLV:           // 0 automatic (on where the curve has the split: BLS12-381), 1 off
        if (value > 1) return BP_ERR_ARG;
        ctx->tuning.glv = value == 0;
        return BP_OK;
    case BP_TUNE_VERIFY_TABLES: // 0 never (default), else the smallest n (>= 2) whose verification uses the tables
        if (value == 1 || value > (1L << 30)) return BP_ERR_ARG;
        ctx->tuning.verify_tables = (uint32_t)value;
        return BP_OK;
    case BP_TUNE_COMPACT_AT:    // 0 automatic, 1 never, else the live length (a power of two whose rounds are single launches) to compact at
        if (value > 1 && (value < 16 || (value & (value - 1)) || 2 * (size_t)value + 1 > kSmallDigitMax)) return BP_ERR_ARG;
        ctx->tuning.compact_at = (uint32_t)value;
        return BP_OK;
    default:
        return BP_ERR_ARG;
    }
    });
}

int bp_ctx_set_device_tail(bp_ctx* ctx, int on) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_ERR_ARG;
    ctx->device_tail = on != 0;
    return BP_OK;
    });
}

int bp_ctx_enable_timing(bp_ctx* ctx, int on) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_ERR_ARG;
    ctx->timing = on != 0;
    return BP_OK;
    });
}

int bp_msm_last_timing(bp_ctx* ctx, float* ms, int cap) {
    return bp_guard([&]() -> int {
    if (!ctx || !ms) return 0;
    int k = ctx->last_ms_n < cap ? ctx->last_ms_n : cap;
    for (int i = 0; i < k; i++) ms[i] = ctx->last_ms[i];
    return k;
    });
}

// ---- G1Vector ----
static size_t point_bytes(const bp_ctx* ctx) { return 2 * (size_t)fp_bytes_of(ctx->curve); }

int bp_g1vec_alloc(bp_ctx* ctx, size_t n, bp_g1vec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out) return BP_ERR_ARG;
    *out = nullptr;
    int rc = set_device(ctx); if (rc) return rc;
    size_t bytes = (n ? n : 1) * point_bytes(ctx), cap = 0;
    void* d = ctx->pool->get(bytes, &cap);
    if (!d) return BP_ERR_DEVICE;
    if (hipMemsetAsync(d, 0, bytes, ctx->stream) != hipSuccess) { ctx->pool->put(d, cap); return BP_ERR_DEVICE; }
    *out = new bp_g1vec{ctx, d, n, true, ctx->device, ctx->pool, cap};
    return BP_OK;
    });
}

int bp_g1vec_upload(bp_ctx* ctx, const uint8_t* points, size_t n, int fmt, bp_g1vec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out || (!points && n) || (fmt != BP_FMT_LE && fmt != BP_FMT_AMCL)) return BP_ERR_ARG;
    int rc = bp_g1vec_alloc(ctx, n, out);
    if (rc) return rc;
    if (n == 0) return BP_OK;
    std::vector<uint8_t> le;
    const uint8_t* src = points;
    int fb = fp_bytes_of(ctx->curve);
    if (fmt == BP_FMT_AMCL) {
        // 04 || X || Y big-endian (MODBYTES == fp_bytes for both curves); identity = 04 || 0 || 1
        le.assign(n * 2 * fb, 0);
        for (size_t i = 0; i < n; i++) {
            const uint8_t* p = points + i * (2 * fb + 1);
            if (p[0] != 0x04) { bp_g1vec_free(*out); *out = nullptr; return BP_ERR_ARG; }
            bool xz = true, y1 = p[2 * fb] == 1;
            for (int k = 0; k < fb; k++) { if (p[1 + k]) xz = false; if (k < fb - 1 && p[1 + fb + k]) y1 = false; }
            if (xz && y1) continue;   // identity -> all-zero
            for (int k = 0; k < fb; k++) { le[i * 2 * fb + k] = p[fb - k]; le[i * 2 * fb + fb + k] = p[2 * fb - k]; }
        }
        src = le.data();
    }
    if (ctx->curve == BP_CURVE_BLS12_381) rc = Impl<Bls381>::upload_points(ctx, src, n, (*out)->d, true);
    else rc = Impl<Bn254>::upload_points(ctx, src, n, (*out)->d, true);
    if (rc) { bp_g1vec_free(*out); *out = nullptr; }
    return rc;
    });
}

int bp_g1vec_download(bp_ctx* ctx, const bp_g1vec* v, size_t offset, size_t n, int fmt, uint8_t* out) {
    return bp_guard([&]() -> int {
    if (!ctx || !v || (!out && n) || (fmt != BP_FMT_LE && fmt != BP_FMT_AMCL)) return BP_ERR_ARG;
    if (offset > v->n || n > v->n - offset) return BP_ERR_LENGTH;
    if (n == 0) return BP_OK;
    int rc = set_device(ctx); if (rc) return rc;
    int fb = fp_bytes_of(ctx->curve);
    std::vector<uint8_t> le;
    uint8_t* dst = out;
    if (fmt == BP_FMT_AMCL) { le.resize(n * 2 * fb); dst = le.data(); }
    if (ctx->curve == BP_CURVE_BLS12_381) rc = Impl<Bls381>::download_points(ctx, v->d, offset, n, dst);
    else rc = Impl<Bn254>::download_points(ctx, v->d, offset, n, dst);
    if (rc) return rc;
    if (fmt == BP_FMT_AMCL) {
        for (size_t i = 0; i < n; i++) {
            uint8_t* p = out + i * (2 * fb + 1);
            const uint8_t* q = le.data() + i * 2 * fb;
            memset(p, 0, 2 * fb + 1);
            p[0] = 0x04;
            bool z = true;
            for (int k = 0; k < 2 * fb; k++) if (q[k]) { z = false; break; }
            if (z) { p[2 * fb] = 1; continue; }
            for (int k = 0; k < fb; k++) { p[fb - k] = q[k]; p[2 * fb - k] = q[fb + k]; }
        }
    }
    return BP_OK;
    });
}

void bp_internal_table_free(bp_g1table* t) {
    if (!t) return;
    if (t->d) t->pool->put(t->d, t->cap);
    delete t;
}

// An owned block whose raw pointer was handed out (bp_*_device_ptr: torch / RCCL streams, views on other contexts) may still be
// read by streams this library does not know: wait for the device, as the hipFree of the pre-pool code did, before the block can
// be handed out again.  Everything else is ordered on the owner's stream and goes back without synchronisation.
static void recycle_block(DevPool* pool, int device, void* d, size_t cap, bool exported) {
    if (exported) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (hipSetDevice(device) == hipSuccess) (void)hipDeviceSynchronize();
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    }
    pool->put(d, cap);
}

int bp_g1vec_free(bp_g1vec* v) {
    return bp_guard([&]() -> int {
    if (!v) return BP_OK;
    if (v->table) bp_internal_table_free(v->table);
    if (v->ctable) bp_internal_table_free(v->ctable);
    if (v->owned && v->d) recycle_block(v->pool, v->device, v->d, v->cap, v->exported);
    delete v;
    return BP_OK;
    });
}

int bp_g1vec_precompute(bp_ctx* ctx, bp_g1vec* v, int window_bits) {
    return bp_guard([&]() -> int {
    if (!ctx || !v || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    if (v->n == 0) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    int c = window_bits;
    if (c == 0) {                                   // as the plain pipeline picks it, but never below 8 (a table row per 8 bits at most)
        int lg = 0;
        while (((size_t)1 << (lg + 1)) <= v->n) lg++;
        c = lg - 1 < 8 ? 8 : lg - 1 > 16 ? 16 : lg - 1;
    }
    if (ctx->device != v->device) return BP_ERR_ARG;            // the table lives in ctx's pool and is built on ctx's stream: same device as the vector
    if (v->table) { bp_internal_table_free(v->table); v->table = nullptr; }
    if (v->ctable) { bp_internal_table_free(v->ctable); v->ctable = nullptr; }
    rc = bp_internal_table_build(ctx, v->d, v->n, c, &v->table);
    if (rc == BP_OK) rc = bp_internal_ctable_build(ctx, v->table, &v->ctable);
    // A one-time cost: complete the build before returning, so that MSMs issued on OTHER contexts (shards, sibling streams) can
    // never read rows that k_table_build has not written yet.
    if (rc == BP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = BP_ERR_DEVICE;
    if (rc != BP_OK) {
        if (v->table) { bp_internal_table_free(v->table); v->table = nullptr; }
        if (v->ctable) { bp_internal_table_free(v->ctable); v->ctable = nullptr; }
    }
    return rc;
    });
}

int bp_g1vec_drop_table(bp_g1vec* v) {
    return bp_guard([&]() -> int {
    if (!v) return BP_ERR_ARG;
    if (v->table) { bp_internal_table_free(v->table); v->table = nullptr; }
    if (v->ctable) { bp_internal_table_free(v->ctable); v->ctable = nullptr; }
    return BP_OK;
    });
}

int bp_ctx_verify_table_info(const bp_ctx* ctx, size_t* n, size_t* bytes) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_ERR_ARG;
    const bp_g1table* t = ctx->gh_table;
    if (n) *n = t ? ctx->gh_n : 0;
    if (bytes) *bytes = t ? (size_t)t->W * t->n * 2 * (size_t)fp_bytes_of(ctx->curve) : 0;
    return BP_OK;
    });
}

int bp_ctx_drop_verify_table(bp_ctx* ctx) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_ERR_ARG;
    if (!ctx->gh_table) return BP_OK;
    int rc = set_device(ctx); if (rc) return rc;
    for (bp_ctx* h : ctx->helper) if (h) HIPCHK(hipStreamSynchronize(h->stream));
    bp_internal_table_free(ctx->gh_table);          // back to the pool in the order of ctx->stream, where its MSMs ran
    ctx->gh_table = nullptr;
    return BP_OK;
    });
}

int bp_g1vec_table_info(const bp_g1vec* v, int* window_bits, int* windows, size_t* bytes) {
    return bp_guard([&]() -> int {
    if (!v) return BP_ERR_ARG;
    const bp_g1table* t = v->table;
    if (window_bits) *window_bits = t ? t->c : 0;
    if (windows) *windows = t ? t->W : 0;
    if (bytes) *bytes = (t ? (size_t)t->W * t->n : 0) * 2 * (size_t)fp_bytes_of(v->ctx->curve) +
                        (v->ctable ? (size_t)v->ctable->W * v->ctable->K * v->ctable->n * 2 * (size_t)fp_bytes_of(v->ctx->curve) : 0);   // + the compaction table
    return BP_OK;
    });
}

size_t bp_g1vec_len(const bp_g1vec* v) { return v ? v->n : 0; }
void* bp_g1vec_device_ptr(bp_g1vec* v) { if (!v) return nullptr; v->exported = true; return v->d; }

int bp_g1vec_wrap_device(bp_ctx* ctx, void* device_ptr, size_t n, bp_g1vec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out || (!device_ptr && n) || ((uintptr_t)device_ptr & 15)) return BP_ERR_ARG;
    *out = new bp_g1vec{ctx, device_ptr, n, false, ctx->device, nullptr, 0};
    return BP_OK;
    });
}

int bp_g1vec_scalar_mul(bp_ctx* ctx, const bp_g1vec* p, const bp_frvec* k, bp_g1vec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !k || !out) return BP_ERR_ARG;
    if (p && p->n != k->n) return BP_ERR_LENGTH;
    int rc = bp_g1vec_alloc(ctx, k->n, out);
    if (rc) return rc;
    if (k->n == 0) return BP_OK;
    if (!p) {   // fixed base: table of generator multiples
        if (ctx->curve == BP_CURVE_BLS12_381) rc = Impl<Bls381>::fixed_base(ctx, k->d, k->n, (*out)->d);
        else rc = Impl<Bn254>::fixed_base(ctx, k->d, k->n, (*out)->d);
    } else if (ctx->curve == BP_CURVE_BLS12_381) rc = Impl<Bls381>::scalar_mul(ctx, p->d, k->d, k->n, (*out)->d);
    else rc = Impl<Bn254>::scalar_mul(ctx, p->d, k->d, k->n, (*out)->d);
    if (rc) { bp_g1vec_free(*out); *out = nullptr; }
    return rc;
    });
}

int bp_g1vec_fixed_base_mul(bp_ctx* ctx, const bp_frvec* k, bp_g1vec** out) { return bp_g1vec_scalar_mul(ctx, nullptr, k, out); }

// ---- FieldElementVector ----
int bp_frvec_alloc(bp_ctx* ctx, size_t n, bp_frvec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out) return BP_ERR_ARG;
    *out = nullptr;
    int rc = set_device(ctx); if (rc) return rc;
    size_t bytes = (n ? n : 1) * 32, cap = 0;
    void* d = ctx->pool->get(bytes, &cap);
    if (!d) return BP_ERR_DEVICE;
    if (hipMemsetAsync(d, 0, bytes, ctx->stream) != hipSuccess) { ctx->pool->put(d, cap); return BP_ERR_DEVICE; }
    *out = new bp_frvec{ctx, d, n, true, ctx->device, ctx->pool, cap};
    return BP_OK;
    });
}

int bp_frvec_upload(bp_ctx* ctx, const uint8_t* scalars_le32, size_t n, bp_frvec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out || (!scalars_le32 && n)) return BP_ERR_ARG;
    int rc = bp_frvec_alloc(ctx, n, out);
    if (rc) return rc;
    if (n == 0) return BP_OK;
    if (hipMemcpyAsync((*out)->d, scalars_le32, n * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        bp_frvec_free(*out); *out = nullptr; return BP_ERR_DEVICE;
    }
    // canonical scalars only (< r): the window recoding silently wraps otherwise.  The check also completes the copy of the
    // borrowed host buffer (it ends with a stream synchronisation).
    rc = ctx->curve == BP_CURVE_BLS12_381 ? Impl<Bls381>::check_scalars(ctx, (*out)->d, n) : Impl<Bn254>::check_scalars(ctx, (*out)->d, n);
    if (rc) { bp_frvec_free(*out); *out = nullptr; }
    return rc;
    });
}

// `bytes` of LIBRARY-made host data -> device memory on the context's stream without waiting for it: through the page-locked ring (a
// pageable source makes hipMemcpyAsync stage the bytes synchronously, ~10 us per call; the verifiers issue four or five such copies per
// proof).  Larger than a quarter of the ring: a plain copy and a synchronisation (the source may be a temporary of the caller).
int bp_internal_stage_h2d(bp_ctx* ctx, const void* src, size_t bytes, void* dst) {
    constexpr size_t kStage = (size_t)1 << 20;
    if (bytes == 0) return BP_OK;
    if (!ctx->stage && hipHostMalloc(&ctx->stage, kStage, hipHostMallocDefault) == hipSuccess) { ctx->stage_cap = kStage; ctx->stage_cur = 0; }
    if (!ctx->stage || bytes > kStage / 4) {
        (void)hipGetLastError();
        if (!ctx->stage) ctx->stage = nullptr;
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return BP_OK;
    }
    if (ctx->stage_cur + bytes > ctx->stage_cap) {          // wrap: everything staged so far must have left the ring
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->stage_cur = 0;
    }
    uint8_t* slot = (uint8_t*)ctx->stage + ctx->stage_cur;
    ctx->stage_cur += (bytes + 63) & ~(size_t)63;
    memcpy(slot, src, bytes);
    HIPCHK(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return BP_OK;
}

int bp_internal_frvec_upload_trusted(bp_ctx* ctx, const uint8_t* le32, size_t n, bp_frvec** out) {
    constexpr size_t kStage = (size_t)1 << 20;
    const size_t bytes = n * 32;
    if (bytes == 0 || bytes > kStage / 4) return bp_frvec_upload(ctx, le32, n, out);
    if (!ctx->stage) {
        if (hipHostMalloc(&ctx->stage, kStage, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->stage = nullptr; return bp_frvec_upload(ctx, le32, n, out); }
        ctx->stage_cap = kStage; ctx->stage_cur = 0;
    }
    if (ctx->stage_cur + bytes > ctx->stage_cap) {          // wrap: everything staged so far must have left the ring
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->stage_cur = 0;
    }
    int rc = bp_frvec_alloc(ctx, n, out);
    if (rc) return rc;
    uint8_t* slot = (uint8_t*)ctx->stage + ctx->stage_cur;
    ctx->stage_cur += (bytes + 63) & ~(size_t)63;
    memcpy(slot, le32, bytes);
    if (hipMemcpyAsync((*out)->d, slot, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { bp_frvec_free(*out); *out = nullptr; return BP_ERR_DEVICE; }
    return BP_OK;
}

int bp_frvec_download(bp_ctx* ctx, const bp_frvec* v, size_t offset, size_t n, uint8_t* out_le32) {
    return bp_guard([&]() -> int {
    if (!ctx || !v || (!out_le32 && n)) return BP_ERR_ARG;
    if (offset > v->n || n > v->n - offset) return BP_ERR_LENGTH;
    if (n == 0) return BP_OK;
    int rc = set_device(ctx); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out_le32, (const uint8_t*)v->d + offset * 32, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BP_OK;
    });
}

int bp_frvec_copy(bp_ctx* ctx, bp_frvec* dst, size_t dst_off, const bp_frvec* src, size_t src_off, size_t n) {
    return bp_guard([&]() -> int {
    if (!ctx || !dst || !src) return BP_ERR_ARG;
    if (dst_off > dst->n || n > dst->n - dst_off || src_off > src->n || n > src->n - src_off) return BP_ERR_LENGTH;
    if (n == 0) return BP_OK;
    int rc = set_device(ctx); if (rc) return rc;
    HIPCHK(hipMemcpyAsync((uint8_t*)dst->d + dst_off * 32, (const uint8_t*)src->d + src_off * 32, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    return BP_OK;
    });
}

int bp_frvec_free(bp_frvec* v) {
    return bp_guard([&]() -> int {
    if (!v) return BP_OK;
    if (v->owned && v->d) recycle_block(v->pool, v->device, v->d, v->cap, v->exported);
    delete v;
    return BP_OK;
    });
}

size_t bp_frvec_len(const bp_frvec* v) { return v ? v->n : 0; }
void* bp_frvec_device_ptr(bp_frvec* v) { if (!v) return nullptr; v->exported = true; return v->d; }

int bp_frvec_wrap_device(bp_ctx* ctx, void* device_ptr, size_t n, bp_frvec** out) {
    return bp_guard([&]() -> int {
    if (!ctx || !out || (!device_ptr && n) || ((uintptr_t)device_ptr & 15)) return BP_ERR_ARG;
    *out = new bp_frvec{ctx, device_ptr, n, false, ctx->device, nullptr, 0};
    return BP_OK;
    });
}

// ---- MSM ----
int bp_msm_g1_range(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n, uint8_t* out_le) {
    return bp_guard([&]() -> int {
    if (!ctx || !points || !scalars || !out_le) return BP_ERR_ARG;
    if (poff > points->n || n > points->n - poff || soff > scalars->n || n > scalars->n - soff) return BP_ERR_LENGTH;
    int rc = set_device(ctx); if (rc) return rc;
    const bp_g1table* tb = (points->table && poff == 0 && n == points->n && points->table->n == n) ? points->table : nullptr;
    DISPATCH(ctx, I::msm(ctx, points->d, poff, scalars->d, soff, n, out_le, tb));
    });
}

int bp_msm_g1(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars, uint8_t* out_le) {
    return bp_guard([&]() -> int {
    if (!ctx || !points || !scalars || !out_le) return BP_ERR_ARG;
    if (points->n != scalars->n) return BP_ERR_LENGTH;
    return bp_msm_g1_range(ctx, points, 0, scalars, 0, points->n, out_le);
    });
}

int bp_msm_g1_begin(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars) {
    return bp_guard([&]() -> int {
    if (!ctx || !points || !scalars) return BP_ERR_ARG;
    if (points->n != scalars->n) return BP_ERR_LENGTH;
    int rc = set_device(ctx); if (rc) return rc;
    const bp_g1table* tb = points->table && points->table->n == points->n ? points->table : nullptr;
    DISPATCH(ctx, I::msm_begin(ctx, points->d, 0, scalars->d, 0, points->n, tb));
    });
}

int bp_msm_g1_end(bp_ctx* ctx, uint8_t* out_le) {
    return bp_guard([&]() -> int {
    if (!ctx || !out_le) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    DISPATCH(ctx, I::msm_end(ctx, out_le));
    });
}

int bp_msm_g1_pair(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars1, const bp_frvec* scalars2, uint8_t* out1_le, uint8_t* out2_le) {
    return bp_guard([&]() -> int {
    if (!ctx || !points || !scalars1 || !scalars2 || !out1_le || !out2_le) return BP_ERR_ARG;
    if (points->n != scalars1->n || points->n != scalars2->n) return BP_ERR_LENGTH;
    int rc = set_device(ctx); if (rc) return rc;
    return bp_internal_msm2(ctx, points->d, scalars1->d, scalars2->d, points->n, out1_le, out2_le, 0, points->table && points->table->n == points->n ? points->table : nullptr);
    });
}

size_t bp_msm_window_records(bp_ctx* ctx, size_t n) {
    if (!ctx) return 0;
    int bits = ctx->curve == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS;
    MsmGeom g;
    if (msm_geom(g, bits, n, ctx->c_override, 1, 0, &ctx->tuning)) return 0;
    return (size_t)g.nrec + 1;   // tail records + the geometry header (see RecHeader)
}

size_t bp_msm_record_bytes(int curve_id) { return curve_id == BP_CURVE_BLS12_381 ? sizeof(XyzzPacked<Bls381>) : sizeof(XyzzPacked<Bn254>); }

int bp_msm_g1_windows(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n, void* device_out) {
    return bp_guard([&]() -> int {
    if (!ctx || !points || !scalars || !device_out || n == 0) return BP_ERR_ARG;
    if (poff > points->n || n > points->n - poff || soff > scalars->n || n > scalars->n - soff) return BP_ERR_LENGTH;
    // the context-free helpers a receiving host uses (bp_msm_record_positions / _geometry / _record_header / _finish_host) know the
    // DEFAULT record layout only: a context tuned to another one must not hand out record blocks
    if (ctx->tuning.reduce_m != 0 || !ctx->tuning.small_msm) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    DISPATCH(ctx, I::msm_windows_to(ctx, points->d, poff, scalars->d, soff, n, device_out));
    });
}

// ---- 2-D sharding: index range x window group ----------------------------------------------------------------
size_t bp_msm_window_records_subset(bp_ctx* ctx, size_t n, int w_first, int w_count) {
    if (!ctx || n == 0) return 0;
    MsmGeom g;
    const int bits = ctx->curve == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS;
    if (msm_geom(g, bits, n, ctx->c_override, 1, 0, &ctx->tuning, 0)) return 0;
    if (geom_subset(g, w_first, w_count, &ctx->tuning)) return 0;
    return (size_t)g.nrec + 1;
}

int bp_msm_g1_windows_subset(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n, int w_first, int w_count,
                             size_t block_records, void* device_out) {
    return bp_guard([&]() -> int {
    if (!ctx || !points || !scalars || !device_out || n == 0 || w_first < 0 || w_count <= 0 || block_records < 2) return BP_ERR_ARG;
    if (poff > points->n || n > points->n - poff || soff > scalars->n || n > scalars->n - soff) return BP_ERR_LENGTH;
    int rc = set_device(ctx); if (rc) return rc;
    ctx->win_first = w_first; ctx->win_count = w_count;
    rc = ctx->curve == BP_CURVE_BLS12_381 ? Impl<Bls381>::msm_windows_to(ctx, points->d, poff, scalars->d, soff, n, device_out, block_records)
                                          : Impl<Bn254>::msm_windows_to(ctx, points->d, poff, scalars->d, soff, n, device_out, block_records);
    ctx->win_first = 0; ctx->win_count = 0;
    return rc;
    });
}

int bp_msm_g1_finish_blocks(bp_ctx* ctx, const void* device_records, size_t n_blocks, size_t block_records, size_t n_per_set, uint8_t* out_le) {
    return bp_guard([&]() -> int {
    if (!ctx || !device_records || !out_le || n_blocks == 0 || block_records < 2 || n_per_set == 0) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    const size_t bytes = n_blocks * block_records * bp_msm_record_bytes(ctx->curve);
    if ((rc = host_pinned_reserve(ctx, bytes))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->host_pinned, device_records, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->curve == BP_CURVE_BLS12_381)
        return Impl<Bls381>::finish_blocks_host(ctx->c_override, (const XyzzPacked<Bls381>*)ctx->host_pinned, n_blocks, block_records, n_per_set, out_le, &ctx->tuning, ctx);
    return Impl<Bn254>::finish_blocks_host(ctx->c_override, (const XyzzPacked<Bn254>*)ctx->host_pinned, n_blocks, block_records, n_per_set, out_le, &ctx->tuning, ctx);
    });
}

int bp_msm_g1_finish_blocks_host(int curve_id, const void* host_records, size_t n_blocks, size_t block_records, size_t n_per_set, int window_bits, uint8_t* out_le) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || !host_records || !out_le || n_blocks == 0 || block_records < 2 || n_per_set == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1)
        return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) return Impl<Bls381>::finish_blocks_host(window_bits, (const XyzzPacked<Bls381>*)host_records, n_blocks, block_records, n_per_set, out_le);
    return Impl<Bn254>::finish_blocks_host(window_bits, (const XyzzPacked<Bn254>*)host_records, n_blocks, block_records, n_per_set, out_le);
    });
}

// context-free forms for a host that only receives blocks (default tuning): bit positions and header of one window group's block
int bp_msm_record_positions_subset(int curve_id, size_t n, int window_bits, int w_first, int w_count, int* nrec_out, uint16_t* pos_out) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || n == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    MsmGeom g;
    if (msm_geom(g, curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n, window_bits, 1, 0, nullptr, 0)) return BP_ERR_ARG;
    if (geom_subset(g, w_first, w_count, nullptr)) return BP_ERR_ARG;
    if (nrec_out) *nrec_out = g.nrec;
    if (pos_out) for (int r = 0; r < g.nrec; r++) pos_out[r] = g.rpos[r];
    return BP_OK;
    });
}
int bp_msm_record_header_subset(int curve_id, size_t n, int window_bits, int w_first, int w_count, void* record_out) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || !record_out || n == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    MsmGeom g;
    if (msm_geom(g, curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n, window_bits, 1, 0, nullptr, 0)) return BP_ERR_ARG;
    if (geom_subset(g, w_first, w_count, nullptr)) return BP_ERR_ARG;
    memset(record_out, 0, bp_msm_record_bytes(curve_id));
    if (curve_id == BP_CURVE_BLS12_381) Impl<Bls381>::fill_header(*(Impl<Bls381>::RecHeader*)record_out, g);
    else Impl<Bn254>::fill_header(*(Impl<Bn254>::RecHeader*)record_out, g);
    return BP_OK;
    });
}

int bp_msm_g1_finish(bp_ctx* ctx, const void* device_records, size_t sets, size_t n_per_set, uint8_t* out_le) {
    return bp_guard([&]() -> int {
    if (!ctx || !device_records || !out_le || sets == 0 || n_per_set == 0) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    DISPATCH(ctx, I::msm_finish(ctx, device_records, sets, n_per_set, out_le));
    });
}

int bp_msm_g1_finish_host(int curve_id, const void* host_records, size_t sets, size_t n_per_set, int window_bits, uint8_t* out_le) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || !host_records || !out_le || sets == 0 || n_per_set == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) return Impl<Bls381>::finish_host(window_bits, (const XyzzPacked<Bls381>*)host_records, sets, n_per_set, out_le);
    return Impl<Bn254>::finish_host(window_bits, (const XyzzPacked<Bn254>*)host_records, sets, n_per_set, out_le);
    });
}

int bp_msm_geometry(int curve_id, size_t n, int window_bits, int* c_out, int* W_out, uint8_t* cw_out, uint16_t* off_out, uint8_t* bias_le32) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || n == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    MsmGeom g;
    if (msm_geom(g, curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n, window_bits)) return BP_ERR_ARG;
    if (c_out) *c_out = g.c;
    if (W_out) *W_out = g.tab.W;
    for (int w = 0; w < g.tab.W; w++) { if (cw_out) cw_out[w] = g.tab.cw[w]; if (off_out) off_out[w] = g.tab.off[w]; }
    if (bias_le32) memcpy(bias_le32, g.tab.bias.w, 32);
    return BP_OK;
    });
}

int bp_msm_record_positions(int curve_id, size_t n, int window_bits, int* nrec_out, uint16_t* pos_out) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || n == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    MsmGeom g;
    if (msm_geom(g, curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n, window_bits)) return BP_ERR_ARG;
    if (nrec_out) *nrec_out = g.nrec;
    if (pos_out) for (int r = 0; r < g.nrec; r++) pos_out[r] = g.rpos[r];
    return BP_OK;
    });
}

int bp_msm_record_from_affine(int curve_id, const uint8_t* point_le, void* record_out) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || !point_le || !record_out) return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) Impl<Bls381>::record_from_affine(point_le, (XyzzPacked<Bls381>*)record_out);
    else Impl<Bn254>::record_from_affine(point_le, (XyzzPacked<Bn254>*)record_out);
    return BP_OK;
    });
}

int bp_msm_record_header(int curve_id, size_t n, int window_bits, void* record_out) {
    return bp_guard([&]() -> int {
    if (!curve_ok(curve_id) || !record_out || n == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    MsmGeom g;
    if (msm_geom(g, curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n, window_bits)) return BP_ERR_ARG;
    memset(record_out, 0, bp_msm_record_bytes(curve_id));
    if (curve_id == BP_CURVE_BLS12_381) Impl<Bls381>::fill_header(*(Impl<Bls381>::RecHeader*)record_out, g);
    else Impl<Bn254>::fill_header(*(Impl<Bn254>::RecHeader*)record_out, g);
    return BP_OK;
    });
}

// One host thread, several devices (or several contexts on one device), no RCCL: shard i = (points[i], scalars[i]) lives with
// ctxs[i].  Every shard's device stage is queued from its context's helper thread (launch overhead in parallel), all with ONE
// window width so that the records fit together; each device copies its W window sums (W x 192 B) to pinned host memory and
// the calling thread folds the N record sets.  The "reduce" of north_star is this gather: N x 3 KiB, latency-bound.
int bp_msm_g1_multi(bp_ctx* const* ctxs, const bp_g1vec* const* points, const bp_frvec* const* scalars, size_t n_shards, uint8_t* out_le) {
    return bp_guard([&]() -> int {
    if (!ctxs || !points || !scalars || !out_le || n_shards == 0 || n_shards > 64) return BP_ERR_ARG;
    size_t n_max = 0;
    for (size_t i = 0; i < n_shards; i++) {
        if (!ctxs[i] || !points[i] || !scalars[i] || ctxs[i]->curve != ctxs[0]->curve) return BP_ERR_ARG;
        if (points[i]->n != scalars[i]->n) return BP_ERR_LENGTH;
        for (size_t j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) return BP_ERR_ARG;      // one shard in flight per context
        if (points[i]->n > n_max) n_max = points[i]->n;
    }
    const int curve = ctxs[0]->curve;
    const size_t pbytes = 2 * (size_t)fp_bytes_of(curve);
    if (n_max == 0) { memset(out_le, 0, pbytes); return BP_OK; }
    MsmGeom g;
    const int fr_bits = curve == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS;
    if (msm_geom(g, fr_bits, n_max, ctxs[0]->c_override, 1, 0, &ctxs[0]->tuning)) return BP_ERR_ARG;
    if (msm_geom(g, fr_bits, n_max, g.c, 1, 0, &ctxs[0]->tuning)) return BP_ERR_ARG;     // the shards run with this width FIXED: the record layout of a fixed width
    const int c = g.c, W = g.nrec;
    const bp_tuning tn0 = ctxs[0]->tuning;
    try {                                        // std::vector / std::function may throw: nothing crosses the C ABI
    std::vector<int> rcs(n_shards, BP_OK), Ws(n_shards, W);
    std::vector<char> queued(n_shards, 0), live(n_shards, 0);
    for (size_t i = 0; i < n_shards; i++) {
        if (points[i]->n == 0) continue;
        live[i] = 1;
        auto job = [&, i]() {
            bp_ctx* cx = ctxs[i];
            int rc = set_device(cx);
            if (!rc) rc = curve == BP_CURVE_BLS12_381 ? Impl<Bls381>::multi_begin(cx, points[i], scalars[i], c, &tn0, &Ws[i])
                                                      : Impl<Bn254>::multi_begin(cx, points[i], scalars[i], c, &tn0, &Ws[i]);
            rcs[i] = rc;
        };
        if (n_shards > 1 && ctxs[i]->worker.submit(job)) queued[i] = 1;
        else job();
    }
    for (size_t i = 0; i < n_shards; i++) if (queued[i]) ctxs[i]->worker.wait();
    int rc = BP_OK;
    for (size_t i = 0; i < n_shards; i++) {
        if (!live[i]) continue;
        if (!rcs[i] && Ws[i] != W) rcs[i] = BP_ERR_DEVICE;
        if (set_device(ctxs[i]) != BP_OK || hipStreamSynchronize(ctxs[i]->stream) != hipSuccess) rcs[i] = rcs[i] ? rcs[i] : BP_ERR_DEVICE;
        if (rcs[i] && !rc) rc = rcs[i];
    }
    if (rc) return rc;
    const size_t rec = bp_msm_record_bytes(curve);
    std::vector<uint8_t> all;
    size_t sets = 0;
    for (size_t i = 0; i < n_shards; i++) {
        if (!live[i]) continue;
        all.insert(all.end(), (const uint8_t*)ctxs[i]->host_pinned, (const uint8_t*)ctxs[i]->host_pinned + (size_t)W * rec);
        sets++;
    }
    if (curve == BP_CURVE_BLS12_381) Impl<Bls381>::fold1(ctxs[0], (const XyzzPacked<Bls381>*)all.data(), sets, W, g.rpos, out_le);
    else Impl<Bn254>::fold1(ctxs[0], (const XyzzPacked<Bn254>*)all.data(), sets, W, g.rpos, out_le);
    return BP_OK;
    } catch (...) {
        for (size_t i = 0; i < n_shards; i++) if (ctxs[i]) { (void)set_device(ctxs[i]); (void)hipStreamSynchronize(ctxs[i]->stream); }
        return BP_ERR_DEVICE;
    }
    });
}

int bp_ctx_trim(bp_ctx* ctx) {
    return bp_guard([&]() -> int {
    if (!ctx) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->pool->trim();
    return BP_OK;
    });
}

}  // extern "C"
