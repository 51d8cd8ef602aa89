// bp_capi.hip -- implementation of the C ABI in include/bpmsm.h (libbpmsm.so).
// Host orchestration of the gfx950 kernels in bp_kernels.cuh; no CPU fallback for any compute entry point.
#include <new>
#include <vector>

#include "bp_internal.hpp"
#include "bp_host_tail.hpp"

using namespace bp;

// ------------------------------------------------------------------------------------------------ geometry
constexpr int kMaxGroups = 8;
struct MsmGeom {
    int c;           // target window width
    WinTab tab;      // W windows of nearly equal width covering fr_bits + 1 bits
    uint32_t m;      // buckets per reduce thread (windows outside the last group)
    int ngroups;     // window groups processed as a software pipeline on two streams
    int gw[kMaxGroups + 1];   // group k = windows [gw[k], gw[k+1])
    // tail records handed to the host: kRecPerWin per window; record r carries weight 2^rpos[r] (bp_host_tail.hpp folds any such list)
    int nrec;
    uint16_t rpos[kRecPerWin * kMaxWindows];
};

static void msm_geom(MsmGeom& g, int fr_bits, size_t n, int c_override, int nsets = 1, size_t nnz = 0) {
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
    // k_small_msm (n <= kSmallMsmMax): each lane multiplies by its digit, so narrow windows shorten the chain; below c = 4 the
    // extra windows cost more on the host (one addition per window in the tail) than they save on the device
    if (c_override <= 0 && n <= kSmallMsmMax && c > 4) c = 4;
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
    // Window groups (msm_windows; opt-in with BP_GROUPS=k, default 1): the W windows processed as k contiguous groups, the tail of
    // one group (bucket reduce: ~2m + 30 dependent point operations at one wave per SIMD) and the memory-bound sort of the next
    // running beside the ALU-bound accumulate of another.  Built and measured in round 2 as VERDICT r1 #3 asked -- it LOSES:
    // 4.44 ms (1 group) -> 4.94 (2) -> 5.69 (4) -> 9.6 ms (8) at n = 2^20.  A group's accumulate has only W/k * 2^15 tasks for the
    // 131072 resident lanes, so the longest-first balancing has nothing to balance with (4 windows: one task per lane, the kernel
    // lasts as long as the longest of 131072 Poisson(32) buckets, ~1.6x the mean), and the reduce chain of the last, exposed group
    // is as long as the chain for all windows (its length is set by log2 of the buckets per window, not by their number).
    static const int g_env = getenv("BP_GROUPS") ? atoi(getenv("BP_GROUPS")) : 0;
    int G = 1;                      // measured (profiles/r02_window_groups.txt): more groups are SLOWER on this part, see below
    if (g_env > 0) G = g_env;
    if (G > W) G = W;
    if (G > kMaxGroups) G = kMaxGroups;
    g.ngroups = G;
    for (int k = 0; k <= G; k++) g.gw[k] = (int)((long)W * k / G);
    // Buckets per reduce thread, per group: the smallest m whose ACTIVE blocks fit one per CU -- all windows together for the
    // groups whose reduce is hidden (work-efficient: few long chains), the last group alone for the one that is exposed (shorter
    // chain).  A 257th block makes some SIMD run two such chains back to back (measured: c = 14 paired, 304 blocks 1.26 ms,
    // 152 blocks 0.86 ms; scripts/time_pair.py).
    static const uint32_t m_env = getenv("BP_REDUCE_M") ? (uint32_t)atoi(getenv("BP_REDUCE_M")) : 0;
    auto pick_m = [&](int w0, int w1) {
        uint32_t m = 1;
        for (; m < 16; m++) {
            uint32_t blocks = 0;
            for (int w = w0; w < w1; w++) { uint32_t B = t.boff[w + 1] - t.boff[w]; blocks += ((B + m - 1) / m + kBlock - 1) / kBlock; }
            if (blocks <= 256) break;
        }
        return m_env ? m_env : m;
    };
    const uint32_t m_all = pick_m(0, W), m_last = G > 1 ? pick_m(g.gw[G - 1], W) : m_all;
    g.m = m_all;
    uint32_t rb = 0;
    for (int w = 0; w < W; w++) {
        const uint32_t m = w >= g.gw[G - 1] ? m_last : m_all;
        t.mw[w] = (uint8_t)m;
        t.rboff[w] = (uint16_t)rb;
        uint32_t B = t.boff[w + 1] - t.boff[w];
        rb += ((B + m - 1) / m + kBlock - 1) / kBlock;
    }
    t.rboff[W] = (uint16_t)rb;
    for (int w = 0; w < W; w++) g.rpos[w] = t.off[w];
    g.nrec = kRecPerWin * W;
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
        for (auto& e : ctx->ev_acc) HIPCHK(hipEventCreate(&e));
        for (auto& e : ctx->ev_sync) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : ctx->ev_tail) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_ready = true;
        return BP_OK;
    }

    // Device stage: window sums of  sum_i s_i P_i  into ctx->window_sum (W records).
    static int msm_windows(bp_ctx* ctx, const AffPacked<C>* pts, const ScalarWords* sc, size_t n, MsmGeom& g, const ScalarWords* sc2 = nullptr,
                           size_t nnz = 0) {
        msm_geom(g, C::Fr::BITS, n, ctx->c_override, sc2 ? 2 : 1, nnz);
        const WinTab& tab = g.tab;
        const int W = tab.W;
        if (getenv("BP_TRACE")) fprintf(stderr, "[bpmsm trace] msm n=%zu c=%d W=%d nbuckets=%u m=%u reduce_blocks=%u\n", n, g.c, W, tab.nbuckets, g.m, (unsigned)tab.rboff[W]);
        if (n >= ((size_t)1 << 31)) return BP_ERR_ARG;                      // the sign lives in bit 31 of an index
        if ((uint64_t)W * n >= ((uint64_t)1 << 32)) return BP_ERR_ARG;      // 32-bit slot offsets
        hipStream_t st = ctx->stream;
        static const bool small_path = getenv("BP_SMALL_MSM") ? atoi(getenv("BP_SMALL_MSM")) != 0 : true;
        if (n <= kSmallMsmMax && small_path) {   // one launch: block per window, lane per term (k_small_msm)
            int rc0;
            if ((rc0 = ctx->window_sum.reserve((size_t)g.nrec * kXyzzBytes))) return rc0;
            const bool tm0 = ctx->timing;
            ctx->last_groups = 0;
            if (tm0) { if ((rc0 = ensure_events(ctx))) return rc0; for (int e = 0; e < 6; e++) HIPCHK(hipEventRecord(ctx->ev[e], st)); }
            hipLaunchKernelGGL(k_small_msm<C>, dim3(W), dim3(kBlock), 0, st, pts, sc, sc2, (uint32_t)n, tab, (XyzzPacked<C>*)ctx->window_sum.p);
            BP_TRACE_SYNC(ctx, "k_small_msm<C>");
            if (tm0) HIPCHK(hipEventRecord(ctx->ev[6], st));
            HIPCHK(hipGetLastError());
            return BP_OK;
        }
        const size_t nb = tab.nbuckets;
        const int G = g.ngroups;
        // task length (see bp_kernels.cuh): >= 2x the mean bucket size when buckets are plentiful, else small enough
        // for ~kTaskTarget tasks (a few times the 131072 resident lanes of k_accumulate)
        static const uint64_t kTaskTarget = getenv("BP_TASK_TARGET") ? (uint64_t)atoll(getenv("BP_TASK_TARGET")) : 2 * 131072;
        const uint64_t entries = (uint64_t)W * (nnz ? nnz : n);
        uint32_t L = 8;
        if (nb >= kTaskTarget) { while ((uint64_t)L * nb < 2 * entries && L < (1u << 20)) L <<= 1; if (L < 128) L = 128; }
        else { while ((uint64_t)L * kTaskTarget < entries && L < (1u << 20)) L <<= 1; }
        uint32_t lshift = 0;
        while ((128u << lshift) < L) lshift++;
        // per-group capacities (group k owns buckets [boff[gw[k]], boff[gw[k+1]]) and its own task / heavy lists)
        struct Grp { int w0, w1; size_t b0, nb, max_tasks, max_heavy, max_chunks, scan_blocks, task_base, heavy_base, chunk_base, bsum_base; };
        Grp gr[kMaxGroups];
        size_t tot_tasks = 0, tot_heavy = 0, tot_chunks = 0, tot_bsum = 0;
        for (int k = 0; k < G; k++) {
            Grp& q = gr[k];
            q.w0 = g.gw[k]; q.w1 = g.gw[k + 1];
            q.b0 = tab.boff[q.w0]; q.nb = tab.boff[q.w1] - q.b0;
            const size_t slots = (size_t)(q.w1 - q.w0) * n;
            size_t max_split = slots / L + 1;                                 // tasks beyond one per bucket
            if (max_split > slots) max_split = slots;
            q.max_tasks = q.nb + max_split;
            q.max_heavy = (q.nb < max_split ? q.nb : max_split) + 1;
            q.max_chunks = q.max_heavy + q.max_tasks / kBlock + 1;
            q.scan_blocks = (q.nb + kScanPerBlock - 1) / kScanPerBlock;
            q.task_base = tot_tasks; q.heavy_base = tot_heavy; q.chunk_base = tot_chunks; q.bsum_base = tot_bsum;
            tot_tasks += q.max_tasks; tot_heavy += q.max_heavy; tot_chunks += q.max_chunks; tot_bsum += q.scan_blocks + 16;
        }
        int rc;
        if ((rc = ctx->count.reserve(nb * 4))) return rc;
        if ((rc = ctx->cursor.reserve(nb * 4))) return rc;
        if ((rc = ctx->ntasks.reserve(nb * 4))) return rc;
        if ((rc = ctx->task_off.reserve(nb * 4))) return rc;
        if ((rc = ctx->idx.reserve((size_t)W * n * 4))) return rc;
        if ((rc = ctx->code.reserve((size_t)W * n * 2))) return rc;
        if ((rc = ctx->order.reserve(tot_tasks * 4))) return rc;
        if ((rc = ctx->t_start.reserve(tot_tasks * 4))) return rc;
        if ((rc = ctx->t_len.reserve(tot_tasks * 4))) return rc;
        if ((rc = ctx->tsum.reserve(tot_tasks * kXyzzBytes))) return rc;
        if ((rc = ctx->heavy.reserve(tot_heavy * 4))) return rc;
        if ((rc = ctx->heavy_chunks.reserve(tot_chunks * sizeof(uint2)))) return rc;
        constexpr size_t kMetaWords = ((kTaskBins + 3 + 15) / 16) * 16;      // per group, 64-byte aligned
        if ((rc = ctx->meta.reserve(kMaxGroups * kMetaWords * 4))) return rc;
        if ((rc = ctx->partial.reserve((size_t)tab.rboff[W] * kXyzzBytes))) return rc;
        if ((rc = ctx->window_sum.reserve((size_t)g.nrec * kXyzzBytes))) return rc;
        uint32_t* count = (uint32_t*)ctx->count.p;       // bucket starts
        uint32_t* cursor = (uint32_t*)ctx->cursor.p;     // bucket ends
        uint32_t* ntasks = (uint32_t*)ctx->ntasks.p;
        uint32_t* task_off = (uint32_t*)ctx->task_off.p; // task ids are local to the bucket's group
        uint32_t* idx = (uint32_t*)ctx->idx.p;
        uint16_t* code = (uint16_t*)ctx->code.p;
        auto* partial = (XyzzPacked<C>*)ctx->partial.p;
        auto* wsum = (XyzzPacked<C>*)ctx->window_sum.p;

        // Tile = scalars per block of the binning passes (BP_TILE: 2048 .. 16384 measured within 1 % of each other at 2^18 .. 2^22).
        static const uint32_t tile_env = getenv("BP_TILE") ? (uint32_t)atoi(getenv("BP_TILE")) : 0;
        // Below ~2^19 scalars a 2048-scalar tile leaves the per-scalar passes with a few dozen blocks for 256 CUs (n = 2^17: 65 blocks,
        // k_digits_bin 71 us); the tile shrinks (never below one scalar per lane) until there are ~512 of them.
        uint32_t tile = kTile;
        while (tile > (uint32_t)kBlock && (n + tile - 1) / tile < 512) tile >>= 1;
        if (tile_env) tile = tile_env;
        const uint32_t ntiles = (uint32_t)((n + tile - 1) / tile);
        const uint32_t rows = tab.hoff[W];
        const size_t nhist = (size_t)rows * ntiles;
        const size_t hist_blocks = (nhist + kScanPerBlock - 1) / kScanPerBlock;
        if ((rc = ctx->tile_hist.reserve(nhist * 4))) return rc;
        if ((rc = ctx->tmp_idx.reserve((size_t)W * n * 8))) return rc;            // (point index, digit code) records of the coarse pass
        if ((rc = ctx->block_sums.reserve((hist_blocks + 16 + tot_bsum) * 4))) return rc;
        uint32_t* hsum = (uint32_t*)ctx->block_sums.p;                 // scan of the tile histogram; hsum[hist_blocks] = grand total
        uint32_t* gsum = hsum + hist_blocks + 16;                      // per-group task scans
        uint32_t* tile_hist = (uint32_t*)ctx->tile_hist.p;
        uint2* tmp_rec = (uint2*)ctx->tmp_idx.p;

        const bool tm = ctx->timing;
        if ((rc = ensure_events(ctx))) return rc;
        if (G > 1) {
            if (!ctx->aux_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
            for (int k = 0; k < G; k++) if (!ctx->tail_stream[k]) HIPCHK(hipStreamCreateWithFlags(&ctx->tail_stream[k], hipStreamNonBlocking));
        }
        if (tm) HIPCHK(hipEventRecord(ctx->ev[0], st));
        HIPCHK(hipMemsetAsync(ctx->meta.p, 0, (size_t)G * kMetaWords * 4, st));
        hipLaunchKernelGGL(k_digits_bin, dim3(ntiles), dim3(kBlock), 0, st, sc, sc2, n, tab, ntiles, tile, code, tile_hist);
        BP_TRACE_SYNC(ctx, "k_digits_bin");
        if (tm) HIPCHK(hipEventRecord(ctx->ev[1], st));
        hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)hist_blocks), dim3(kBlock), 0, st, tile_hist, nhist, hsum);
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)hist_blocks), dim3(kBlock), 0, st, tile_hist, nhist, hsum, tile_hist, (uint32_t*)nullptr);
        BP_TRACE_SYNC(ctx, "scan tile_hist");
        if (tm) HIPCHK(hipEventRecord(ctx->ev[2], st));
        if (G > 1) {   // fork: the auxiliary stream continues from here
            HIPCHK(hipEventRecord(ctx->ev_sync[0], st));
            HIPCHK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_sync[0], 0));
        }
        static const int wps = getenv("BP_ACC_WPS") ? atoi(getenv("BP_ACC_WPS")) : 2;
        // Schedule: group k sorts on front stream k & 1 and accumulates there as soon as group k - 1 has finished accumulating
        // (the accumulates run back to back, they are what bounds the MSM); its tail -- combine, bucket reduce, window sums:
        // ~2m + 30 dependent point operations at one wave per SIMD -- goes to a stream of its own and runs beside the next
        // groups' accumulates, as does the memory-bound sort of the next group.  Only the last group's tail is exposed.
        for (int k = 0; k < G; k++) {
            const Grp& q = gr[k];
            hipStream_t sk = (k & 1) ? ctx->aux_stream : st;
            const int Wg = q.w1 - q.w0;
            uint32_t* bins = (uint32_t*)ctx->meta.p + (size_t)k * kMetaWords;      // [kTaskBins] bin counts -> bin cursors
            uint32_t* total_tasks = bins + kTaskBins;
            uint32_t* nheavy = bins + kTaskBins + 1;
            uint32_t* nchunks = bins + kTaskBins + 2;
            uint32_t* order = (uint32_t*)ctx->order.p + q.task_base;
            uint32_t* t_start = (uint32_t*)ctx->t_start.p + q.task_base;
            uint32_t* t_len = (uint32_t*)ctx->t_len.p + q.task_base;
            auto* tsum = (XyzzPacked<C>*)ctx->tsum.p + q.task_base;
            uint32_t* heavy = (uint32_t*)ctx->heavy.p + q.heavy_base;
            uint2* chunks = (uint2*)ctx->heavy_chunks.p + q.chunk_base;
            uint32_t* bsum = gsum + q.bsum_base;
            hipLaunchKernelGGL(k_coarse_scatter, dim3(ntiles, Wg), dim3(kBlock), 0, sk, code, n, tab, ntiles, tile_hist, tmp_rec, q.w0, tile);
            BP_TRACE_SYNC(ctx, "k_coarse_scatter");
            hipLaunchKernelGGL(k_fine_place, dim3(128, Wg), dim3(kBlock), 0, sk, tmp_rec, tab, ntiles, tile_hist, hsum + hist_blocks, count, cursor, idx, q.w0);
            BP_TRACE_SYNC(ctx, "k_fine_place");
            if (tm && k == 0) HIPCHK(hipEventRecord(ctx->ev[3], sk));
            // count[] = bucket starts, cursor[] = bucket ends.  Tasks of this group (bucket ids local to the group from here on):
            const unsigned bgrid = (unsigned)((q.nb + kBlock * kTaskPer - 1) / (kBlock * kTaskPer));   // kTaskPer buckets per lane
            hipLaunchKernelGGL(k_task_count, dim3(bgrid), dim3(kBlock), 0, sk, count + q.b0, cursor + q.b0, (uint32_t)q.nb, L, lshift, ntasks + q.b0, bins);
            hipLaunchKernelGGL(k_task_bins_scan, dim3(1), dim3(kBlock), 0, sk, bins, total_tasks);
            hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)q.scan_blocks), dim3(kBlock), 0, sk, ntasks + q.b0, q.nb, bsum);
            hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)q.scan_blocks), dim3(kBlock), 0, sk, ntasks + q.b0, q.nb, bsum, task_off + q.b0, (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_task_emit, dim3(bgrid), dim3(kBlock), 0, sk, count + q.b0, cursor + q.b0, (uint32_t)q.nb, L, lshift, task_off + q.b0, bins, order, t_start, t_len,
                               heavy, nheavy, chunks, nchunks);
            BP_TRACE_SYNC(ctx, "k_task_emit");
            if (tm && k == 0) HIPCHK(hipEventRecord(ctx->ev[4], sk));
            if (k > 0) HIPCHK(hipStreamWaitEvent(sk, ctx->ev_acc[2 * (k - 1) + 1], 0));
            if (tm) HIPCHK(hipEventRecord(ctx->ev_acc[2 * k], sk));
            {
                dim3 agrid((unsigned)((q.max_tasks + kBlock - 1) / kBlock));
                if (wps == 2) hipLaunchKernelGGL((k_accumulate<C, 2>), agrid, dim3(kBlock), 0, sk, pts, idx, order, t_start, t_len, total_tasks, tsum);
                else if (wps == 4) hipLaunchKernelGGL((k_accumulate<C, 4>), agrid, dim3(kBlock), 0, sk, pts, idx, order, t_start, t_len, total_tasks, tsum);
                else hipLaunchKernelGGL((k_accumulate<C, 3>), agrid, dim3(kBlock), 0, sk, pts, idx, order, t_start, t_len, total_tasks, tsum);
            }
            if (G > 1 || tm) HIPCHK(hipEventRecord(ctx->ev_acc[2 * k + 1], sk));
            if (tm && k == G - 1) HIPCHK(hipEventRecord(ctx->ev[5], sk));
            hipStream_t tk = sk;
            if (G > 1) { tk = ctx->tail_stream[k]; HIPCHK(hipStreamWaitEvent(tk, ctx->ev_acc[2 * k + 1], 0)); }
            hipLaunchKernelGGL(k_combine_chunks<C>, dim3((unsigned)(q.max_chunks < 256 ? q.max_chunks : 256)), dim3(kBlock), 0, tk, chunks, nchunks, task_off + q.b0, ntasks + q.b0, tsum);
            hipLaunchKernelGGL(k_combine_heavy<C>, dim3((unsigned)(q.max_heavy < 256 ? q.max_heavy : 256)), dim3(kBlock), 0, tk, heavy, nheavy, task_off + q.b0, ntasks + q.b0, tsum);
            BP_TRACE_SYNC(ctx, "k_combine_heavy<C>");
            hipLaunchKernelGGL(k_bucket_reduce<C>, dim3(tab.rboff[q.w1] - tab.rboff[q.w0]), dim3(kBlock), 0, tk, tsum, task_off, ntasks, tab, (uint32_t)tab.rboff[q.w0], partial);
            BP_TRACE_SYNC(ctx, "k_bucket_reduce<C>");
            hipLaunchKernelGGL(k_window_sums<C>, dim3(Wg), dim3(kBlock), 0, tk, partial, tab, wsum, q.w0);
            BP_TRACE_SYNC(ctx, "k_window_sums<C>");
            if (G > 1) HIPCHK(hipEventRecord(ctx->ev_tail[k], tk));
        }
        if (G > 1) {   // join: everything the caller queues on the context's stream next sees all groups (and both front streams)
            HIPCHK(hipEventRecord(ctx->ev_sync[1], ctx->aux_stream));
            HIPCHK(hipStreamWaitEvent(st, ctx->ev_sync[1], 0));
            for (int k = 0; k < G; k++) HIPCHK(hipStreamWaitEvent(st, ctx->ev_tail[k], 0));
        }
        if (tm) HIPCHK(hipEventRecord(ctx->ev[6], st));
        ctx->last_groups = G;
        HIPCHK(hipGetLastError());
        return BP_OK;
    }

    // last_ms: [0] whole device pipeline, [1] digits + histograms, [2] scan, [3] scatter and [4] task lists of the first window
    // group, [5] accumulate = SUM over the window groups' launches (each measured on its own stream; with more than one group
    // other kernels run beside them), [6] end of the last accumulate -> end of the pipeline (the exposed tail)
    static void collect_timing(bp_ctx* ctx) {
        ctx->last_ms_n = 0;
        if (!ctx->timing || !ctx->ev_ready) return;
        if (hipEventSynchronize(ctx->ev[6]) != hipSuccess) return;
        float t;
        if (hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[6]) == hipSuccess) ctx->last_ms[0] = t;
        for (int i = 0; i < 6; i++)
            if (hipEventElapsedTime(&t, ctx->ev[i], ctx->ev[i + 1]) == hipSuccess) ctx->last_ms[1 + i] = t;
        if (ctx->last_groups > 0) {
            float acc = 0;
            for (int k = 0; k < ctx->last_groups; k++)
                if (hipEventElapsedTime(&t, ctx->ev_acc[2 * k], ctx->ev_acc[2 * k + 1]) == hipSuccess) acc += t;
            ctx->last_ms[5] = acc;
        }
        ctx->last_ms_n = 7;
    }

    static const host::Tail<C>& tail() { static const host::Tail<C> t; return t; }

    static void aff_to_le(const Aff<C>& a, uint8_t* out) {
        uint32_t w[Fp::NW];
        fe_pack_words<Fp>(w, fe_from_mont<Fp>(a.x));
        memcpy(out, w, 4 * Fp::NW);
        fe_pack_words<Fp>(w, fe_from_mont<Fp>(a.y));
        memcpy(out + 4 * Fp::NW, w, 4 * Fp::NW);
    }

    static int msm(bp_ctx* ctx, const void* pts, size_t poff, const void* sc, size_t soff, size_t n, uint8_t* out_le) {
        if (n == 0) { memset(out_le, 0, 2 * 4 * Fp::NW); ctx->last_ms_n = 0; return BP_OK; }
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts + poff, (const ScalarWords*)sc + soff, n, g);
        if (rc) return rc;
        if ((rc = host_pinned_reserve(ctx, (size_t)g.nrec * kXyzzBytes))) return rc;
        if (ctx->device_tail) {   // all-device variant: one lane folds the windows (see k_tail_fold)
            if ((rc = ctx->scratch.reserve(2 * 4 * Fp::NW))) return rc;
            hipLaunchKernelGGL(k_tail_fold<C>, dim3(1), dim3(64), 0, ctx->stream, (const XyzzPacked<C>*)ctx->window_sum.p, g.tab, 0, g.tab.W,
                               (uint32_t*)ctx->scratch.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(out_le, ctx->scratch.p, 2 * 4 * Fp::NW, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            collect_timing(ctx);
            return BP_OK;
        }
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, ctx->window_sum.p, (size_t)g.nrec * kXyzzBytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        collect_timing(ctx);
        tail().fold((const XyzzPacked<C>*)ctx->host_pinned, 1, g.nrec, g.rpos, out_le);
        return BP_OK;
    }

    // Asynchronous form of msm(): begin() queues the device pipeline and the D2H copy of the window sums on the
    // context's stream and returns; end() waits for them and runs the host tail.  One MSM in flight per context.
    static int msm_begin(bp_ctx* ctx, const void* pts, size_t poff, const void* sc, size_t soff, size_t n) {
        ctx->pending = false;
        ctx->pending_n = n;
        if (n == 0) { ctx->pending = true; return BP_OK; }
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts + poff, (const ScalarWords*)sc + soff, n, g);
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
        tail().fold((const XyzzPacked<C>*)ctx->host_pinned, 1, ctx->pending_nrec, ctx->pending_rpos, out_le);
        return BP_OK;
    }

    // Two scalar sets over the same points in ONE pipeline pass (2W windows): out1 = <sc1, pts>, out2 = <sc2, pts>.
    static int msm2(bp_ctx* ctx, const void* pts, const void* sc1, const void* sc2, size_t n, uint8_t* out1_le, uint8_t* out2_le, size_t nnz = 0) {
        if (n == 0) { memset(out1_le, 0, 2 * 4 * Fp::NW); memset(out2_le, 0, 2 * 4 * Fp::NW); return BP_OK; }
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts, (const ScalarWords*)sc1, n, g, (const ScalarWords*)sc2, nnz);
        if (rc) return rc;
        const int R1 = g.nrec / 2;                      // records of one scalar set
        if ((rc = host_pinned_reserve(ctx, (size_t)g.nrec * kXyzzBytes))) return rc;
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, ctx->window_sum.p, (size_t)g.nrec * kXyzzBytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        collect_timing(ctx);
        // the two serial tails (~0.12 ms each: 255 dependent doublings) are independent: fold the second on the context's helper thread
        const host::Tail<C>& tl = tail();
        const XyzzPacked<C>* rec = (const XyzzPacked<C>*)ctx->host_pinned;
        const bool helped = ctx->worker.submit([&]() { tl.fold(rec + R1, 1, R1, g.rpos + R1, out2_le); });
        tl.fold(rec, 1, R1, g.rpos, out1_le);
        if (helped) ctx->worker.wait();
        else tl.fold(rec + R1, 1, R1, g.rpos + R1, out2_le);   // no thread available: fold both here
        return BP_OK;
    }

    // Record block of the two-stage (sharded) form: W window records followed by ONE header record that names the geometry
    // which produced them, so that bp_msm_g1_finish can refuse sets that do not fit together (ranks whose shard sizes straddle
    // a power of two pick different window widths unless the caller fixes c with bp_ctx_set_window_bits).
    struct RecHeader { uint32_t magic, c, W, fr_bits, cw_first, cw_last, n_wide, rec_per_win; };   // widths: n_wide windows of cw_first bits, then cw_last
    static_assert(sizeof(RecHeader) <= sizeof(XyzzPacked<C>), "header must fit one record");
    static constexpr uint32_t kRecMagic = 0x31575042u;   // "BPW1"
    static void fill_header(RecHeader& h, const MsmGeom& g) {
        memset(&h, 0, sizeof h);
        h.magic = kRecMagic; h.c = (uint32_t)g.c; h.W = (uint32_t)g.tab.W; h.fr_bits = (uint32_t)C::Fr::BITS;
        h.cw_first = g.tab.cw[0]; h.cw_last = g.tab.cw[g.tab.W - 1];
        for (int w = 0; w < g.tab.W; w++) if (g.tab.cw[w] == g.tab.cw[0]) h.n_wide++;
        h.rec_per_win = kRecPerWin;
    }

    static int msm_windows_to(bp_ctx* ctx, const void* pts, size_t poff, const void* sc, size_t soff, size_t n, void* device_out) {
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts + poff, (const ScalarWords*)sc + soff, n, g);
        if (rc) return rc;
        const int W = g.nrec;
        HIPCHK(hipMemcpyAsync(device_out, ctx->window_sum.p, (size_t)W * kXyzzBytes, hipMemcpyDeviceToDevice, ctx->stream));
        if ((rc = host_pinned_reserve(ctx, kXyzzBytes))) return rc;
        memset(ctx->host_pinned, 0, kXyzzBytes);
        fill_header(*(RecHeader*)ctx->host_pinned, g);
        HIPCHK(hipMemcpyAsync((uint8_t*)device_out + (size_t)W * kXyzzBytes, ctx->host_pinned, kXyzzBytes, hipMemcpyHostToDevice, ctx->stream));
        // The records are about to be read by another stream (RCCL's) or another device: complete them before returning.
        HIPCHK(hipStreamSynchronize(ctx->stream));
        collect_timing(ctx);
        return BP_OK;
    }

    // host_rec: sets x (W + 1) records in host memory (header last in each set); validates the headers and folds
    static int finish_host(int c_override, const XyzzPacked<C>* host_rec, size_t sets, size_t n_per_set, uint8_t* out_le) {
        MsmGeom g;
        msm_geom(g, C::Fr::BITS, n_per_set, c_override);
        const int W = g.nrec;
        RecHeader want;
        fill_header(want, g);
        std::vector<XyzzPacked<C>> packed(sets * (size_t)W);
        for (size_t s = 0; s < sets; s++) {
            const XyzzPacked<C>* set = host_rec + s * (size_t)(W + 1);
            if (memcmp(&set[W], &want, sizeof want) != 0) return BP_ERR_ARG;     // geometry of this set differs from the caller's
            memcpy(&packed[s * (size_t)W], set, (size_t)W * kXyzzBytes);
        }
        tail().fold(packed.data(), sets, W, g.rpos, out_le);
        return BP_OK;
    }

    // bp_msm_g1_multi, device half of one shard: queue the pipeline with the shared window width c and the D2H copy of the W
    // window sums into this context's pinned buffer; no synchronisation.
    static int multi_begin(bp_ctx* ctx, const bp_g1vec* pts, const bp_frvec* sc, int c, int* W_out) {
        const int saved = ctx->c_override;
        ctx->c_override = c;
        MsmGeom g;
        int rc = msm_windows(ctx, (const AffPacked<C>*)pts->d, (const ScalarWords*)sc->d, pts->n, g);
        ctx->c_override = saved;
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
        msm_geom(g, C::Fr::BITS, n_per_set, ctx->c_override);
        size_t bytes = sets * (size_t)(g.nrec + 1) * kXyzzBytes;
        int rc;
        if ((rc = host_pinned_reserve(ctx, bytes))) return rc;
        HIPCHK(hipMemcpyAsync(ctx->host_pinned, device_records, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        collect_timing(ctx);
        return finish_host(ctx->c_override, (const XyzzPacked<C>*)ctx->host_pinned, sets, n_per_set, out_le);
    }

    // validate = true: BP_ERR_ARG if a coordinate is >= p or a point is off the curve (see k_points_to_resident)
    static int upload_points(bp_ctx* ctx, const uint8_t* le, size_t n, void* d_out, bool validate) {
        size_t bytes = n * 2 * 4 * Fp::NW;
        int rc;
        if ((rc = ctx->scratch.reserve(bytes ? bytes : 16))) return rc;
        if ((rc = ctx->flags.reserve(64))) return rc;
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
        if ((rc = ctx->flags.reserve(64))) return rc;
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
        if ((rc = ctx->scratch.reserve(bytes ? bytes : 16))) return rc;
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
        if ((rc = ctx->fixed_base_table.reserve(m * kPointBytes))) return rc;
        if ((rc = ctx->scratch.reserve(m * 32))) return rc;
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

int bp_internal_msm(bp_ctx* ctx, const void* points, const void* scalars, size_t n, uint8_t* out_le) {
    DISPATCH(ctx, I::msm(ctx, points, 0, scalars, 0, n, out_le));
}

int bp_internal_msm2(bp_ctx* ctx, const void* points, const void* scalars1, const void* scalars2, size_t n, uint8_t* out1_le, uint8_t* out2_le,
                     size_t nnz) {
    DISPATCH(ctx, I::msm2(ctx, points, scalars1, scalars2, n, out1_le, out2_le, nnz));
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* bp_version(void) { return "bpmsm 0.1 (gfx950; unsaturated 30-bit limbs; Pippenger/XYZZ)"; }

int bp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bp_curve_params(int curve_id, bp_curve_info* out) {
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
}

int bp_ctx_create(int curve_id, int device_ordinal, bp_ctx** out) {
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
}

int bp_ctx_destroy(bp_ctx* ctx) {
    if (!ctx) return BP_OK;
    for (auto& h : ctx->helper) { if (h) bp_ctx_destroy(h); h = nullptr; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    ctx->fixed_base_table.release();
    for (DevBuf* b : {&ctx->count, &ctx->cursor, &ctx->block_sums, &ctx->idx, &ctx->code, &ctx->tile_hist, &ctx->tmp_idx, &ctx->ntasks, &ctx->task_off, &ctx->order, &ctx->t_start,
                      &ctx->t_len, &ctx->tsum, &ctx->heavy, &ctx->heavy_chunks, &ctx->meta, &ctx->partial, &ctx->window_sum, &ctx->scratch, &ctx->flags}) b->release();
    if (ctx->pool) ctx->pool->release();     // cached blocks go back to the driver; live handles keep the pool itself alive
    if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
    if (ctx->ev_ready) {
        for (auto& e : ctx->ev) (void)hipEventDestroy(e);
        for (auto& e : ctx->ev_acc) (void)hipEventDestroy(e);
        for (auto& e : ctx->ev_sync) (void)hipEventDestroy(e);
        for (auto& e : ctx->ev_tail) (void)hipEventDestroy(e);
    }
    for (auto& t : ctx->tail_stream) if (t) { (void)hipStreamSynchronize(t); (void)hipStreamDestroy(t); }
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return BP_OK;
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
    return h;
}
int bp_internal_fork(bp_ctx* ctx, bp_ctx* sibling) {
    if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(hipStreamWaitEvent(sibling->stream, ctx->ev_fork, 0));
    return BP_OK;
}

int bp_ctx_set_stream(bp_ctx* ctx, void* hip_stream) {
    if (!ctx) return BP_ERR_ARG;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next != ctx->stream) {   // pool blocks are recycled in stream order: drain the old stream before work moves to another
        int rc = set_device(ctx); if (rc) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ctx->stream = next;
    return BP_OK;
}

int bp_ctx_synchronize(bp_ctx* ctx) {
    if (!ctx) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BP_OK;
}

int bp_ctx_set_window_bits(bp_ctx* ctx, int c) {
    if (!ctx || c < 0 || c > 16 || c == 1) return BP_ERR_ARG;
    ctx->c_override = c;
    return BP_OK;
}

int bp_ctx_set_device_tail(bp_ctx* ctx, int on) {
    if (!ctx) return BP_ERR_ARG;
    ctx->device_tail = on != 0;
    return BP_OK;
}

int bp_ctx_enable_timing(bp_ctx* ctx, int on) {
    if (!ctx) return BP_ERR_ARG;
    ctx->timing = on != 0;
    return BP_OK;
}

int bp_msm_last_timing(bp_ctx* ctx, float* ms, int cap) {
    if (!ctx || !ms) return 0;
    int k = ctx->last_ms_n < cap ? ctx->last_ms_n : cap;
    for (int i = 0; i < k; i++) ms[i] = ctx->last_ms[i];
    return k;
}

// ---- G1Vector ----
static size_t point_bytes(const bp_ctx* ctx) { return 2 * (size_t)fp_bytes_of(ctx->curve); }

int bp_g1vec_alloc(bp_ctx* ctx, size_t n, bp_g1vec** out) {
    if (!ctx || !out) return BP_ERR_ARG;
    *out = nullptr;
    int rc = set_device(ctx); if (rc) return rc;
    size_t bytes = (n ? n : 1) * point_bytes(ctx), cap = 0;
    void* d = ctx->pool->get(bytes, &cap);
    if (!d) return BP_ERR_DEVICE;
    if (hipMemsetAsync(d, 0, bytes, ctx->stream) != hipSuccess) { ctx->pool->put(d, cap); return BP_ERR_DEVICE; }
    *out = new bp_g1vec{ctx, d, n, true, ctx->device, ctx->pool, cap};
    return BP_OK;
}

int bp_g1vec_upload(bp_ctx* ctx, const uint8_t* points, size_t n, int fmt, bp_g1vec** out) {
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
}

int bp_g1vec_download(bp_ctx* ctx, const bp_g1vec* v, size_t offset, size_t n, int fmt, uint8_t* out) {
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
}

int bp_g1vec_free(bp_g1vec* v) {
    if (!v) return BP_OK;
    if (v->owned && v->d) v->pool->put(v->d, v->cap);     // back to the context's pool (no device synchronisation)
    delete v;
    return BP_OK;
}

size_t bp_g1vec_len(const bp_g1vec* v) { return v ? v->n : 0; }
void* bp_g1vec_device_ptr(bp_g1vec* v) { return v ? v->d : nullptr; }

int bp_g1vec_wrap_device(bp_ctx* ctx, void* device_ptr, size_t n, bp_g1vec** out) {
    if (!ctx || !out || (!device_ptr && n) || ((uintptr_t)device_ptr & 15)) return BP_ERR_ARG;
    *out = new bp_g1vec{ctx, device_ptr, n, false, ctx->device, nullptr, 0};
    return BP_OK;
}

int bp_g1vec_scalar_mul(bp_ctx* ctx, const bp_g1vec* p, const bp_frvec* k, bp_g1vec** out) {
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
}

int bp_g1vec_fixed_base_mul(bp_ctx* ctx, const bp_frvec* k, bp_g1vec** out) { return bp_g1vec_scalar_mul(ctx, nullptr, k, out); }

// ---- FieldElementVector ----
int bp_frvec_alloc(bp_ctx* ctx, size_t n, bp_frvec** out) {
    if (!ctx || !out) return BP_ERR_ARG;
    *out = nullptr;
    int rc = set_device(ctx); if (rc) return rc;
    size_t bytes = (n ? n : 1) * 32, cap = 0;
    void* d = ctx->pool->get(bytes, &cap);
    if (!d) return BP_ERR_DEVICE;
    if (hipMemsetAsync(d, 0, bytes, ctx->stream) != hipSuccess) { ctx->pool->put(d, cap); return BP_ERR_DEVICE; }
    *out = new bp_frvec{ctx, d, n, true, ctx->device, ctx->pool, cap};
    return BP_OK;
}

int bp_frvec_upload(bp_ctx* ctx, const uint8_t* scalars_le32, size_t n, bp_frvec** out) {
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
}

int bp_frvec_download(bp_ctx* ctx, const bp_frvec* v, size_t offset, size_t n, uint8_t* out_le32) {
    if (!ctx || !v || (!out_le32 && n)) return BP_ERR_ARG;
    if (offset > v->n || n > v->n - offset) return BP_ERR_LENGTH;
    if (n == 0) return BP_OK;
    int rc = set_device(ctx); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out_le32, (const uint8_t*)v->d + offset * 32, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BP_OK;
}

int bp_frvec_copy(bp_ctx* ctx, bp_frvec* dst, size_t dst_off, const bp_frvec* src, size_t src_off, size_t n) {
    if (!ctx || !dst || !src) return BP_ERR_ARG;
    if (dst_off > dst->n || n > dst->n - dst_off || src_off > src->n || n > src->n - src_off) return BP_ERR_LENGTH;
    if (n == 0) return BP_OK;
    int rc = set_device(ctx); if (rc) return rc;
    HIPCHK(hipMemcpyAsync((uint8_t*)dst->d + dst_off * 32, (const uint8_t*)src->d + src_off * 32, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    return BP_OK;
}

int bp_frvec_free(bp_frvec* v) {
    if (!v) return BP_OK;
    if (v->owned && v->d) v->pool->put(v->d, v->cap);
    delete v;
    return BP_OK;
}

size_t bp_frvec_len(const bp_frvec* v) { return v ? v->n : 0; }
void* bp_frvec_device_ptr(bp_frvec* v) { return v ? v->d : nullptr; }

int bp_frvec_wrap_device(bp_ctx* ctx, void* device_ptr, size_t n, bp_frvec** out) {
    if (!ctx || !out || (!device_ptr && n) || ((uintptr_t)device_ptr & 15)) return BP_ERR_ARG;
    *out = new bp_frvec{ctx, device_ptr, n, false, ctx->device, nullptr, 0};
    return BP_OK;
}

// ---- MSM ----
int bp_msm_g1_range(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n, uint8_t* out_le) {
    if (!ctx || !points || !scalars || !out_le) return BP_ERR_ARG;
    if (poff > points->n || n > points->n - poff || soff > scalars->n || n > scalars->n - soff) return BP_ERR_LENGTH;
    int rc = set_device(ctx); if (rc) return rc;
    DISPATCH(ctx, I::msm(ctx, points->d, poff, scalars->d, soff, n, out_le));
}

int bp_msm_g1(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars, uint8_t* out_le) {
    if (!ctx || !points || !scalars || !out_le) return BP_ERR_ARG;
    if (points->n != scalars->n) return BP_ERR_LENGTH;
    return bp_msm_g1_range(ctx, points, 0, scalars, 0, points->n, out_le);
}

int bp_msm_g1_begin(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars) {
    if (!ctx || !points || !scalars) return BP_ERR_ARG;
    if (points->n != scalars->n) return BP_ERR_LENGTH;
    int rc = set_device(ctx); if (rc) return rc;
    DISPATCH(ctx, I::msm_begin(ctx, points->d, 0, scalars->d, 0, points->n));
}

int bp_msm_g1_end(bp_ctx* ctx, uint8_t* out_le) {
    if (!ctx || !out_le) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    DISPATCH(ctx, I::msm_end(ctx, out_le));
}

int bp_msm_g1_pair(bp_ctx* ctx, const bp_g1vec* points, const bp_frvec* scalars1, const bp_frvec* scalars2, uint8_t* out1_le, uint8_t* out2_le) {
    if (!ctx || !points || !scalars1 || !scalars2 || !out1_le || !out2_le) return BP_ERR_ARG;
    if (points->n != scalars1->n || points->n != scalars2->n) return BP_ERR_LENGTH;
    int rc = set_device(ctx); if (rc) return rc;
    return bp_internal_msm2(ctx, points->d, scalars1->d, scalars2->d, points->n, out1_le, out2_le, 0);
}

size_t bp_msm_window_records(bp_ctx* ctx, size_t n) {
    if (!ctx) return 0;
    int bits = ctx->curve == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS;
    MsmGeom g;
    msm_geom(g, bits, n, ctx->c_override);
    return (size_t)g.nrec + 1;   // kRecPerWin records per window + the geometry header (see RecHeader)
}

size_t bp_msm_record_bytes(int curve_id) { return curve_id == BP_CURVE_BLS12_381 ? sizeof(XyzzPacked<Bls381>) : sizeof(XyzzPacked<Bn254>); }

int bp_msm_g1_windows(bp_ctx* ctx, const bp_g1vec* points, size_t poff, const bp_frvec* scalars, size_t soff, size_t n, void* device_out) {
    if (!ctx || !points || !scalars || !device_out || n == 0) return BP_ERR_ARG;
    if (poff > points->n || n > points->n - poff || soff > scalars->n || n > scalars->n - soff) return BP_ERR_LENGTH;
    int rc = set_device(ctx); if (rc) return rc;
    DISPATCH(ctx, I::msm_windows_to(ctx, points->d, poff, scalars->d, soff, n, device_out));
}

int bp_msm_g1_finish(bp_ctx* ctx, const void* device_records, size_t sets, size_t n_per_set, uint8_t* out_le) {
    if (!ctx || !device_records || !out_le || sets == 0 || n_per_set == 0) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    DISPATCH(ctx, I::msm_finish(ctx, device_records, sets, n_per_set, out_le));
}

int bp_msm_g1_finish_host(int curve_id, const void* host_records, size_t sets, size_t n_per_set, int window_bits, uint8_t* out_le) {
    if (!curve_ok(curve_id) || !host_records || !out_le || sets == 0 || n_per_set == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) return Impl<Bls381>::finish_host(window_bits, (const XyzzPacked<Bls381>*)host_records, sets, n_per_set, out_le);
    return Impl<Bn254>::finish_host(window_bits, (const XyzzPacked<Bn254>*)host_records, sets, n_per_set, out_le);
}

int bp_msm_geometry(int curve_id, size_t n, int window_bits, int* c_out, int* W_out, uint8_t* cw_out, uint16_t* off_out, uint8_t* bias_le32) {
    if (!curve_ok(curve_id) || n == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    MsmGeom g;
    msm_geom(g, curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n, window_bits);
    if (c_out) *c_out = g.c;
    if (W_out) *W_out = g.tab.W;
    for (int w = 0; w < g.tab.W; w++) { if (cw_out) cw_out[w] = g.tab.cw[w]; if (off_out) off_out[w] = g.tab.off[w]; }
    if (bias_le32) memcpy(bias_le32, g.tab.bias.w, 32);
    return BP_OK;
}

int bp_msm_record_positions(int curve_id, size_t n, int window_bits, int* nrec_out, uint16_t* pos_out) {
    if (!curve_ok(curve_id) || n == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    MsmGeom g;
    msm_geom(g, curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n, window_bits);
    if (nrec_out) *nrec_out = g.nrec;
    if (pos_out) for (int r = 0; r < g.nrec; r++) pos_out[r] = g.rpos[r];
    return BP_OK;
}

int bp_msm_record_from_affine(int curve_id, const uint8_t* point_le, void* record_out) {
    if (!curve_ok(curve_id) || !point_le || !record_out) return BP_ERR_ARG;
    if (curve_id == BP_CURVE_BLS12_381) Impl<Bls381>::record_from_affine(point_le, (XyzzPacked<Bls381>*)record_out);
    else Impl<Bn254>::record_from_affine(point_le, (XyzzPacked<Bn254>*)record_out);
    return BP_OK;
}

int bp_msm_record_header(int curve_id, size_t n, int window_bits, void* record_out) {
    if (!curve_ok(curve_id) || !record_out || n == 0 || window_bits < 0 || window_bits > 16 || window_bits == 1) return BP_ERR_ARG;
    MsmGeom g;
    msm_geom(g, curve_id == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n, window_bits);
    memset(record_out, 0, bp_msm_record_bytes(curve_id));
    if (curve_id == BP_CURVE_BLS12_381) Impl<Bls381>::fill_header(*(Impl<Bls381>::RecHeader*)record_out, g);
    else Impl<Bn254>::fill_header(*(Impl<Bn254>::RecHeader*)record_out, g);
    return BP_OK;
}

// One host thread, several devices (or several contexts on one device), no RCCL: shard i = (points[i], scalars[i]) lives with
// ctxs[i].  Every shard's device stage is queued from its context's helper thread (launch overhead in parallel), all with ONE
// window width so that the records fit together; each device copies its W window sums (W x 192 B) to pinned host memory and
// the calling thread folds the N record sets.  The "reduce" of north_star is this gather: N x 3 KiB, latency-bound.
int bp_msm_g1_multi(bp_ctx* const* ctxs, const bp_g1vec* const* points, const bp_frvec* const* scalars, size_t n_shards, uint8_t* out_le) {
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
    msm_geom(g, curve == BP_CURVE_BLS12_381 ? Bls381Fr::BITS : Bn254Fr::BITS, n_max, ctxs[0]->c_override);
    const int c = g.c, W = g.nrec;
    std::vector<int> rcs(n_shards, BP_OK), Ws(n_shards, W);
    std::vector<char> queued(n_shards, 0), live(n_shards, 0);
    for (size_t i = 0; i < n_shards; i++) {
        if (points[i]->n == 0) continue;
        live[i] = 1;
        auto job = [&, i]() {
            bp_ctx* cx = ctxs[i];
            int rc = set_device(cx);
            if (!rc) rc = curve == BP_CURVE_BLS12_381 ? Impl<Bls381>::multi_begin(cx, points[i], scalars[i], c, &Ws[i])
                                                      : Impl<Bn254>::multi_begin(cx, points[i], scalars[i], c, &Ws[i]);
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
    if (curve == BP_CURVE_BLS12_381) Impl<Bls381>::tail().fold((const XyzzPacked<Bls381>*)all.data(), sets, W, g.rpos, out_le);
    else Impl<Bn254>::tail().fold((const XyzzPacked<Bn254>*)all.data(), sets, W, g.rpos, out_le);
    return BP_OK;
}

int bp_ctx_trim(bp_ctx* ctx) {
    if (!ctx) return BP_ERR_ARG;
    int rc = set_device(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->pool->trim();
    return BP_OK;
}

}  // extern "C"
