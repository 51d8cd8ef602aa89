// bp_kernels.cuh -- gfx950 kernels of the Pippenger bucket MSM and the vector helpers around it.
//
// Replaces (device side of): G1Vector::multi_scalar_mul_var_time / inner_product_var_time_with_ref_vecs /
// inner_product_const_time of amcl_wrapper as called at /root/reference src/ipp.rs:91,104,158,170,251-253 and
// src/r1cs/prover.rs:358,423, src/r1cs/verifier.rs:451.  The reference's CPU path is
// believed to be Strauss/wNAF; the bucket method is this build's choice -- parity is on the
// resulting group element.
//
// Pipeline for  sum_i s_i P_i  with W signed windows of (nearly) equal width c covering bits+1 bits, window w
// owning 2^(c_w - 1) buckets (bucket j <-> |digit| = j + 1):
//   k_digits_bin     block per tile: k' = k + bias, 16-bit digit code per window (stored transposed, code[w][i]) and
//                    LDS histograms over coarse bins of the bucket id
//   k_scan_*         exclusive scan of the (bin, tile) histogram
//   k_coarse_scatter / k_fine_place   two-level placement of point indices (+ sign bit) into idx[], grouped by
//                    (window, bucket), and bucket start/end offsets (order inside a bucket is arbitrary: the sum is
//                    commutative and the result is compared in canonical affine form)
//   k_task_*         cut buckets into tasks of <= kTaskLen points, order tasks longest first
//   k_accumulate     lane per task: XYZZ accumulator in VGPRs, gathers its points (96-B rows, 16-B vector loads) and
//                    mixed-adds them
//   k_combine_chunks / k_combine_heavy   buckets cut into more than kLightMax tasks: block per 256-task chunk, then block
//                    per bucket over its chunk sums (lighter multi-task buckets are summed inside k_bucket_reduce)
//   k_bucket_reduce  compact grid of <= 256 blocks; thread per m consecutive buckets: running sum / sum of running sums,
//                    weighted by the segment's base value, then an LDS tree per block            -> partial[]
//                    (the last block of a window to finish folds the window's blocks: tail records -> window_sum[])
//   k_small_msm      n <= kSmallMsmMax terms: the whole device stage in one launch (block per window, lane per term)
// The final  sum_w 2^(off_w) window_sum[w]  (a strictly serial chain of ~bits doublings) is folded on the host
// (bp_capi.hip): one lane of a GPU would take ~2 ms for it, the host ~0.1 ms, and the result is needed on the
// host anyway (it goes into the Fiat-Shamir transcript).
#pragma once
#include <hip/hip_runtime.h>
#include "bp_curve.cuh"

namespace bp {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------- scalar slicing
// Scalars in HBM: 8 canonical (non-Montgomery) little-endian 32-bit words, 32 B, loaded as two 16-B vectors.
struct alignas(16) ScalarWords {
    uint32_t w[8];
};

// Window table: W windows of (nearly equal) widths cw[w] covering fr_bits + 1 bits, window w starting at bit
// off[w]; window w owns buckets boff[w] .. boff[w+1]-1 (2^(cw-1) of them, bucket j <-> |digit| = j + 1).
// Equal widths matter: a narrow top window would put n / 2^(few bits) points into each of its buckets.
constexpr int kMaxWindows = 256;   // up to two scalar sets of 128 windows each
struct WinTab {
    int W;            // windows of ALL scalar sets (set s owns windows [s * W / nsets, (s + 1) * W / nsets))
    int nsets;        // scalar sets sharing the same points (1 or 2)
    uint32_t nbuckets;
    uint8_t cw[kMaxWindows];
    uint16_t off[kMaxWindows];
    uint32_t boff[kMaxWindows + 1];
    uint8_t fbits[kMaxWindows];        // fine bits of a bucket id: min(8, cw - 1); the rest are the coarse bin
    uint16_t hoff[kMaxWindows + 1];    // first coarse-bin row of window w in the tile histogram (hoff[W] = rows)
    uint16_t rboff[kMaxWindows + 1];   // first block of window w in the (compact, 1-D) grid of k_bucket_reduce
    uint8_t lgm[kMaxWindows];          // log2 of the buckets per reduce thread in window w (a power of two, <= the window's buckets)
    // Merged mode (MSM over a vector with a window-multiples table, bp_g1vec_precompute): every window of a scalar set drops its
    // entries into the SAME 2^(c-1) buckets, the entry of (window w, point i) naming table row (w % W1) * n + i = 2^(c w) P_i.
    // boff[w] / hoff[w] are then per SET (equal for all its windows) and a (window, tile) pair is its own column of the tile
    // histogram: column (w % W1) * ntiles + tile of ntiles * W1.
    uint8_t merged;
    uint16_t W1;                       // windows per scalar set
    uint16_t roff[kMaxWindows + 1];    // first tail record of window w (bucket pipeline: 1 + log2(threads of the window) records)
    ScalarWords bias;   // H = sum_w (2^(cw-1) - 1) 2^off[w]
};
// Tail records handed to the host (record r carries weight 2^rpos[r], bp_capi.hip).  The bucket pipeline emits, per window,
// the plain weighted sum of the per-thread segments plus one "bit-plane" record per bit of the reduce-thread index (see
// k_bucket_reduce); the single-launch small MSM emits one record per window.
constexpr int kMaxRecPerWin = 16;      // 1 + log2(2^15 buckets / 1 per thread)
constexpr int kMaxRecords = 4096;      // cap on W * records per window (checked in msm_geom)
constexpr int kPartPerBlock = 10;      // k_bucket_reduce output per block: TRI, RUN, planes of the 8 thread-index bits

constexpr int kTile = 2048;            // scalars per block in the binning passes (8 per lane) for large MSMs; smaller ones use smaller tiles
                                       // (a multiple of kBlock) so that the passes still have a few hundred blocks (bp_capi.hip)
constexpr int kDigitBatch = 4;         // scalars in flight per lane in k_digits_bin
constexpr int kFineBatch = 8;          // records in flight per lane in k_fine_place
constexpr int kMaxBinRows = 4096;      // sum over windows of coarse bins (c = 16: 16 x 128 per scalar set)

// Signed-digit recoding without a serial carry: with k' = k + H, the raw cw-bit window w of k' equals
// digit_w + (2^(cw-1) - 1), digit_w in [-(2^(cw-1) - 1), 2^(cw-1)], sum_w digit_w 2^off[w] = k.  (Adding
// half-1 to a window overflows it exactly when the classic recoding would emit a carry.)  The raw window --
// at most 16 bits -- is the "digit code" stored per (window, scalar).
__device__ __forceinline__ void add256(uint64_t (&q)[4], const ScalarWords& a, const ScalarWords& b) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t x = a.w[2 * i] | ((uint64_t)a.w[2 * i + 1] << 32), y = b.w[2 * i] | ((uint64_t)b.w[2 * i + 1] << 32);
        uint64_t s = x + y, s2 = s + c;
        c = (uint64_t)(s < x) + (uint64_t)(s2 < s);
        q[i] = s2;
    }
}

// q >>= k, 0 < k < 256 (k wave-uniform: the branches are scalar)
__device__ __forceinline__ void shr256(uint64_t (&q)[4], int k) {
    const int word = k >> 6, sh = k & 63;
    if (word == 1) { q[0] = q[1]; q[1] = q[2]; q[2] = q[3]; q[3] = 0; }
    else if (word == 2) { q[0] = q[2]; q[1] = q[3]; q[2] = 0; q[3] = 0; }
    else if (word == 3) { q[0] = q[3]; q[1] = 0; q[2] = 0; q[3] = 0; }
    if (sh) {
        q[0] = (q[0] >> sh) | (q[1] << (64 - sh));
        q[1] = (q[1] >> sh) | (q[2] << (64 - sh));
        q[2] = (q[2] >> sh) | (q[3] << (64 - sh));
        q[3] >>= sh;
    }
}

// a > b, and a = m - a, on canonical 8-word values (the scalar negation of k_digits_bin)
__device__ __forceinline__ bool gt256(const ScalarWords& a, const ScalarWords& b) {
    bool gt = false, eq = true;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        gt = gt || (eq && a.w[i] > b.w[i]);
        eq = eq && a.w[i] == b.w[i];
    }
    return gt;
}
__device__ __forceinline__ void rsub256(ScalarWords& a, const ScalarWords& m) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)m.w[i] - a.w[i] - borrow;
        a.w[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t mine, uint32_t* lds, uint32_t& block_total);

// Sorting point indices by (window, bucket) -- a two-level binning sort, all atomics in LDS:
//   k_digits_bin   block per tile of kTile scalars: digit codes (transposed store code[w][i]) and, per window, an LDS
//                  histogram over COARSE bins (the top bits of the bucket id) -> tile_hist[(row(w,bin)) * ntiles + tile]
//   scan           exclusive scan of tile_hist: element offsets, window-major / bin-major / tile-minor
//   k_coarse_scatter  grid (tiles, W): every (tile, bin) run is written contiguously (>= 64-byte runs on average)
//   k_fine_place   grid (coarse bins, W): one block owns a coarse bin (~8192 elements, 256 buckets): LDS histogram of
//                  the fine bits -> bucket start/end, then placement into idx[]; the bin's output region is ~32 KB, so
//                  the 4-byte stores combine in L2 instead of costing a 64-byte write-back each.
// (The first version -- global-atomic histogram + global random scatter -- took 0.63 + 1.48 ms at n = 2^20 with 1.0 GB
// of WRITE_SIZE for 64 MB of useful output: profiles/r01_bench_n1_pmc_hbm.json.)
// Scalar negation (round 4): a scalar k > r - 2^128 ("a small negative number": rneg = r - 2^128) is recoded as r - k and its point enters
// its buckets NEGATED (negbits: one bit per scalar and set, applied to the sign of the entries by k_coarse_scatter; tile_neg: whether a
// tile has any, so that tiles without pay one flag load).  The a_R = a_L - 1 of a bit decomposition (/root/reference
// src/r1cs/gadgets/helper_constraints/positive_no.rs:18-24) is 0 or r - 1: one non-zero digit instead of one in every window.
// (Negating every k > r / 2 was measured first: uniform scalars then leave half of the TOP window's buckets empty and fill the others twice
// as high -- reduce 0.42 -> 0.57 ms at 2^20.  A uniform scalar meets the 2^-127 rule never.)
static __global__ void __launch_bounds__(kBlock) k_digits_bin(const ScalarWords* __restrict__ scalars, const ScalarWords* __restrict__ scalars2, size_t n,
                                                             WinTab tab, uint32_t ntiles, uint32_t tile, uint16_t* __restrict__ code, uint32_t* __restrict__ tile_hist,
                                                             ScalarWords rmod, ScalarWords rneg, uint64_t* __restrict__ negbits, size_t nw64, uint32_t* __restrict__ tile_neg) {
    __shared__ uint32_t lh[kMaxBinRows];
    __shared__ uint32_t s_anyneg;
    if (threadIdx.x == 0) s_anyneg = 0;                   // (the barrier after the histogram's zeroing orders it)
    const bool mg = tab.merged != 0;
    const uint32_t mbins = 1u << (tab.cw[0] - 1 - tab.fbits[0]);          // merged: coarse bins of a set (same for every window)
    const uint32_t rows = mg ? (uint32_t)tab.W * mbins : tab.hoff[tab.W];  // LDS rows: one histogram per WINDOW in both modes
    for (uint32_t k = threadIdx.x; k < rows; k += kBlock) lh[k] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * tile;
    const int wps = tab.W / tab.nsets;   // windows per scalar set (same geometry for every set)
    const uint32_t per = tile / kBlock;   // scalars per lane (tile is a multiple of kBlock)
    uint32_t anyneg = 0;                  // bit `set`: this lane saw a negated scalar of that set
#pragma unroll 1
    for (uint32_t e0 = 0; e0 < per; e0 += kDigitBatch) {
      // kDigitBatch scalars per lane are loaded before the first is recoded (the loop was paced by one 32-byte load per iteration)
      for (int set = 0; set < tab.nsets; set++) {
        ScalarWords sw[kDigitBatch];
#pragma unroll
        for (int u = 0; u < kDigitBatch; u++) {
            size_t i = base + (size_t)(e0 + u) * kBlock + threadIdx.x;
            if (e0 + u < per && i < n) sw[u] = set ? scalars2[i] : scalars[i];
        }
#pragma unroll
        for (int u = 0; u < kDigitBatch; u++) {
            size_t i = base + (size_t)(e0 + u) * kBlock + threadIdx.x;
            const bool valid = e0 + u < per && i < n;
            if (negbits) {                                   // (a wave holds 64 consecutive scalars, the first a multiple of 64)
                bool neg = false;
                if (valid) { neg = gt256(sw[u], rneg); if (neg) { rsub256(sw[u], rmod); anyneg |= 1u << set; } }
                const uint64_t m = __ballot(neg);
                if ((threadIdx.x & 63) == 0 && valid) negbits[(size_t)set * nw64 + (i >> 6)] = m;
            }
            if (valid) {
                uint64_t q[4];
                add256(q, sw[u], tab.bias);
                if (const int off0 = tab.off[set * wps]) shr256(q, off0);      // a window SUBSET (sharded callers): its first window starts above bit 0
                for (int w = set * wps; w < (set + 1) * wps; w++) {
                    int c = tab.cw[w];
                    uint32_t raw = (uint32_t)q[0] & ((1u << c) - 1);
                    q[0] = (q[0] >> c) | (q[1] << (64 - c));
                    q[1] = (q[1] >> c) | (q[2] << (64 - c));
                    q[2] = (q[2] >> c) | (q[3] << (64 - c));
                    q[3] >>= c;
                    code[(size_t)w * n + i] = (uint16_t)raw;
                    int d = (int)raw - ((1 << (c - 1)) - 1);
                    if (d != 0) atomicAdd(&lh[(mg ? (uint32_t)w * mbins : (uint32_t)tab.hoff[w]) + (((uint32_t)(d < 0 ? -d : d) - 1) >> tab.fbits[w])], 1u);
                }
            }
        }
      }
    }
    if (anyneg) atomicOr(&s_anyneg, anyneg);
    __syncthreads();
    if (tile_neg && threadIdx.x < (uint32_t)tab.nsets) tile_neg[(size_t)threadIdx.x * ntiles + blockIdx.x] = (s_anyneg >> threadIdx.x) & 1u;
    if (!mg) {
        for (uint32_t k = threadIdx.x; k < rows; k += kBlock) tile_hist[(size_t)k * ntiles + blockIdx.x] = lh[k];
    } else {
        const size_t cols = (size_t)ntiles * tab.W1;
        for (uint32_t k = threadIdx.x; k < rows; k += kBlock) {
            const uint32_t w = k / mbins, bin = k - w * mbins;
            tile_hist[(size_t)(tab.hoff[w] + bin) * cols + (size_t)(w % tab.W1) * ntiles + blockIdx.x] = lh[k];
        }
    }
}

// grid = (ntiles, W).  tile_off = scanned tile_hist.  Writes (code, point index) pairs grouped by coarse bin -- as ONE 8-byte record
// per element: the kernel is bound by the address processing of its scattered stores (64 lanes, 64 different runs), and one
// store instruction per element instead of two (2-byte code + 4-byte index) matters more than the 2 extra bytes.
static __global__ void __launch_bounds__(kBlock) k_coarse_scatter(const uint16_t* __restrict__ code, size_t n, WinTab tab, uint32_t ntiles,
                                                                 const uint32_t* __restrict__ tile_off, uint2* __restrict__ tmp_rec, int w0, uint32_t tile,
                                                                 const uint64_t* __restrict__ negbits, size_t nw64, const uint32_t* __restrict__ tile_neg) {
    __shared__ uint32_t lcur[128];
    const int w = w0 + (int)blockIdx.y;
    const int c = tab.cw[w], fb = tab.fbits[w];
    const bool mg = tab.merged != 0;
    // merged: the bins are those of the SET's 2^(c-1) buckets (its first window has the full width; the last one may be narrower)
    const uint32_t nbins = mg ? 1u << (tab.cw[(w / tab.W1) * tab.W1] - 1 - fb) : (uint32_t)(tab.hoff[w + 1] - tab.hoff[w]);
    const size_t cols = mg ? (size_t)ntiles * tab.W1 : ntiles, col = mg ? (size_t)(w % tab.W1) * ntiles + blockIdx.x : blockIdx.x;
    for (uint32_t k = threadIdx.x; k < nbins; k += kBlock) lcur[k] = tile_off[(size_t)(tab.hoff[w] + k) * cols + col];
    __syncthreads();
    size_t base = (size_t)blockIdx.x * tile;
    const uint32_t half1 = (1u << (c - 1)) - 1;
    const uint32_t per = tile / kBlock;
    const uint32_t row0 = mg ? (uint32_t)(w % tab.W1) * (uint32_t)n : 0u;      // merged: the entry names a row of the window-multiples table
    const int set = w / (tab.W / tab.nsets);                                    // this window's scalar set; its negation bits only if the tile has any
    const uint64_t* nbits = negbits && tile_neg[(size_t)set * ntiles + blockIdx.x] ? negbits + (size_t)set * nw64 : nullptr;
    for (uint32_t e0 = 0; e0 < per; e0 += kFineBatch) {      // kFineBatch codes per lane in flight
        uint32_t raw[kFineBatch];
        uint64_t nb[kFineBatch];
#pragma unroll
        for (int u = 0; u < kFineBatch; u++) {
            size_t i = base + (size_t)(e0 + u) * kBlock + threadIdx.x;
            const bool valid = e0 + u < per && i < n;
            raw[u] = valid ? code[(size_t)w * n + i] : half1;
            nb[u] = valid && nbits ? nbits[i >> 6] : 0;
        }
#pragma unroll
        for (int u = 0; u < kFineBatch; u++) {
            int d = (int)raw[u] - (int)half1;
            if ((nb[u] >> (threadIdx.x & 63)) & 1) d = -d;               // the scalar was recoded as r - k: its point enters negated
            if (d != 0) {
                uint32_t pos = atomicAdd(&lcur[((uint32_t)(d < 0 ? -d : d) - 1) >> fb], 1u);
                tmp_rec[pos] = make_uint2(row0 + (uint32_t)(base + (size_t)(e0 + u) * kBlock + threadIdx.x), (uint32_t)d);     // (row, signed digit)
            }
        }
    }
}

// (Structured scalars make ONE coarse bin huge -- bit vectors put half of all records into one bucket -- and its block then streams
// 2^19 records alone: 0.89 ms at n = 2^20.  Peeling the wave's most common key off the LDS counters (one atomic per wave for it) was
// measured in round 3: no gain for bits (the block is bound by its own load / store stream, not by the counter), 2x slower for a
// huge bin with 255 live buckets (8-bit scalars).  The fix is several blocks per huge bin, i.e. a count / place pair of kernels:
// k_fine_huge_count / k_fine_huge_place below, round 4.)
// grid = (128, W); block (bin, w) owns the elements [tile_off[row * ntiles], tile_off[(row + 1) * ntiles]) of its
// coarse bin (row = hoff[w] + bin; `total` closes the last row).  Two streaming passes over them: fine histogram,
// then placement.  Writes start[g] / end[g] for its 2^fbits buckets and idx[] (point index + sign bit).
// A coarse bin with more than kHugeMin records (round 4) is not streamed by its one block: k_fine_place cuts it into slices
// of kHugeSlice records, and the k_fine_huge_count / _place pair below runs a block per slice (per-slice histograms in HBM; the bin's
// bucket bounds from their sums, a slice's cursors from the slices before it).
struct HugeSlice { uint32_t bin, w, lo, hi, first, count, bin_lo, pad; };   // records [lo, hi) of coarse bin `bin` of window w; the bin's slices are first .. first + count - 1
constexpr uint32_t kHugeSlice = 8192;
constexpr uint32_t kHugeMin = 4 * kHugeSlice;

static __global__ void __launch_bounds__(kBlock) k_fine_place(const uint2* __restrict__ tmp_rec, WinTab tab, uint32_t ntiles, const uint32_t* __restrict__ tile_off, const uint32_t* __restrict__ total,
                                                             uint32_t* __restrict__ start, uint32_t* __restrict__ end, uint32_t* __restrict__ idx, int w0,
                                                             uint32_t* __restrict__ nonempty, HugeSlice* __restrict__ slices, uint32_t* __restrict__ nslices) {
    __shared__ uint32_t lh[kBlock], lscan[kBlock / 64];
    __shared__ uint32_t s_first;
    const int w = w0 + (int)blockIdx.y;
    const uint32_t nbins = tab.hoff[w + 1] - tab.hoff[w];
    if (blockIdx.x >= nbins) return;
    const int fb = tab.fbits[w];
    const uint32_t row = tab.hoff[w] + blockIdx.x, rows = tab.hoff[tab.W];
    const uint32_t lo = tile_off[(size_t)row * ntiles];
    const uint32_t hi = row + 1 < rows ? tile_off[(size_t)(row + 1) * ntiles] : *total;
    const uint32_t fmask = (1u << fb) - 1;
    if (slices && hi - lo > kHugeMin) {                      // (block-uniform) hand the bin to the slice kernels
        const uint32_t cnt = (hi - lo + kHugeSlice - 1) / kHugeSlice;
        if (threadIdx.x == 0) s_first = atomicAdd(nslices, cnt);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cnt; k += kBlock) {
            const uint32_t a = lo + k * kHugeSlice;
            slices[s_first + k] = HugeSlice{blockIdx.x, (uint32_t)w, a, a + kHugeSlice < hi ? a + kHugeSlice : hi, s_first, cnt, lo, 0u};
        }
        return;
    }
    lh[threadIdx.x] = 0;
    __syncthreads();
    // kFineBatch records per lane are loaded before any of them is used: the loop is otherwise a chain of (load, LDS atomic) pairs
    // paced by the load latency (32 dependent round trips per pass for an 8192-record bin)
    for (uint32_t j0 = lo + threadIdx.x; j0 < hi; j0 += kFineBatch * kBlock) {
        int dd[kFineBatch];
#pragma unroll
        for (int u = 0; u < kFineBatch; u++) { uint32_t j = j0 + u * kBlock; dd[u] = j < hi ? (int)tmp_rec[j].y : 0; }   // digit 0 = not an element
#pragma unroll
        for (int u = 0; u < kFineBatch; u++) {
            const int d = dd[u];
            if (d != 0) atomicAdd(&lh[((uint32_t)(d < 0 ? -d : d) - 1) & fmask], 1u);
        }
    }
    __syncthreads();
    uint32_t cnt = lh[threadIdx.x], tot;
    uint32_t ex = block_exclusive_scan(cnt, lscan, tot) + lo;
    if (threadIdx.x <= fmask) {
        uint32_t g = tab.boff[w] + (blockIdx.x << fb) + threadIdx.x;
        start[g] = ex;
        end[g] = ex + cnt;
    }
    {   // buckets that hold anything, for the task length (task_len below): one atomic per block, on its WINDOW's counter (one
        // counter for all blocks -- 2 432 blocks x 4 waves at n = 2^16 -- serialised in L2: +26 us on a 25 us stage)
        const int ne = __syncthreads_count(threadIdx.x <= fmask && cnt != 0);
        if (threadIdx.x == 0 && ne) atomicAdd(&nonempty[w], (uint32_t)ne);
    }
    lh[threadIdx.x] = ex;
    __syncthreads();
    for (uint32_t j0 = lo + threadIdx.x; j0 < hi; j0 += kFineBatch * kBlock) {
        uint2 rec[kFineBatch];
#pragma unroll
        for (int u = 0; u < kFineBatch; u++) { uint32_t j = j0 + u * kBlock; rec[u] = j < hi ? tmp_rec[j] : make_uint2(0u, 0u); }
#pragma unroll
        for (int u = 0; u < kFineBatch; u++) {
            const int d = (int)rec[u].y;
            if (d != 0) {
                uint32_t pos = atomicAdd(&lh[((uint32_t)(d < 0 ? -d : d) - 1) & fmask], 1u);
                idx[pos] = rec[u].x | (d < 0 ? 0x80000000u : 0u);
            }
        }
    }
}

// grid-stride over the slices k_fine_place listed: hist[s * kBlock + key] = records of slice s with fine key `key`
static __global__ void __launch_bounds__(kBlock) k_fine_huge_count(const uint2* __restrict__ tmp_rec, WinTab tab, const HugeSlice* __restrict__ slices,
                                                                  const uint32_t* __restrict__ nslices, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[kBlock];
    const uint32_t ns = *nslices;
    for (uint32_t s = blockIdx.x; s < ns; s += gridDim.x) {
        const HugeSlice sl = slices[s];
        const uint32_t fmask = (1u << tab.fbits[sl.w]) - 1;
        lh[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t j0 = sl.lo + threadIdx.x; j0 < sl.hi; j0 += kFineBatch * kBlock) {
            int dd[kFineBatch];
#pragma unroll
            for (int u = 0; u < kFineBatch; u++) { uint32_t j = j0 + u * kBlock; dd[u] = j < sl.hi ? (int)tmp_rec[j].y : 0; }
#pragma unroll
            for (int u = 0; u < kFineBatch; u++) {
                const int d = dd[u];
                if (d != 0) atomicAdd(&lh[((uint32_t)(d < 0 ? -d : d) - 1) & fmask], 1u);
            }
        }
        __syncthreads();
        hist[(size_t)s * kBlock + threadIdx.x] = lh[threadIdx.x];
        __syncthreads();
    }
}

// ... and the placement: bucket bounds of the bin (written by its first slice) and this slice's cursors from the per-slice histograms
static __global__ void __launch_bounds__(kBlock) k_fine_huge_place(const uint2* __restrict__ tmp_rec, WinTab tab, const HugeSlice* __restrict__ slices,
                                                                  const uint32_t* __restrict__ nslices, const uint32_t* __restrict__ hist,
                                                                  uint32_t* __restrict__ start, uint32_t* __restrict__ end, uint32_t* __restrict__ idx,
                                                                  uint32_t* __restrict__ nonempty) {
    __shared__ uint32_t lh[kBlock], lscan[kBlock / 64];
    const uint32_t ns = *nslices;
    for (uint32_t s = blockIdx.x; s < ns; s += gridDim.x) {
        const HugeSlice sl = slices[s];
        const int fb = tab.fbits[sl.w];
        const uint32_t fmask = (1u << fb) - 1;
        uint32_t tot = 0, before = 0;
        for (uint32_t k = 0; k < sl.count; k++) {
            const uint32_t v = hist[(size_t)(sl.first + k) * kBlock + threadIdx.x];
            tot += v;
            if (sl.first + k < s) before += v;
        }
        uint32_t all;
        const uint32_t ex = block_exclusive_scan(tot, lscan, all) + sl.bin_lo;
        if (s == sl.first) {                                 // (block-uniform)
            if (threadIdx.x <= fmask) {
                const uint32_t g = tab.boff[sl.w] + (sl.bin << fb) + threadIdx.x;
                start[g] = ex;
                end[g] = ex + tot;
            }
            const int ne = __syncthreads_count(threadIdx.x <= fmask && tot != 0);
            if (threadIdx.x == 0 && ne) atomicAdd(&nonempty[sl.w], (uint32_t)ne);
        }
        lh[threadIdx.x] = ex + before;
        __syncthreads();
        for (uint32_t j0 = sl.lo + threadIdx.x; j0 < sl.hi; j0 += kFineBatch * kBlock) {
            uint2 rec[kFineBatch];
#pragma unroll
            for (int u = 0; u < kFineBatch; u++) { uint32_t j = j0 + u * kBlock; rec[u] = j < sl.hi ? tmp_rec[j] : make_uint2(0u, 0u); }
#pragma unroll
            for (int u = 0; u < kFineBatch; u++) {
                const int d = (int)rec[u].y;
                if (d != 0) {
                    uint32_t pos = atomicAdd(&lh[((uint32_t)(d < 0 ? -d : d) - 1) & fmask], 1u);
                    idx[pos] = rec[u].x | (d < 0 ? 0x80000000u : 0u);
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- exclusive scan
constexpr int kScanPerThread = 8;
constexpr int kScanPerBlock = kBlock * kScanPerThread;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t mine, uint32_t* lds /* >= 4 words */, uint32_t& block_total) {
    // inclusive scan inside the wave by shuffles, then across the block's 4 waves through LDS
    uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kBlock / 64; k++) { uint32_t t = lds[k]; if ((uint32_t)k < wave) base += t; tot += t; }
    __syncthreads();
    block_total = tot;
    return base + inc - mine;
}

static __global__ void __launch_bounds__(kBlock) k_scan_block_sums(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds[kBlock / 64];
    size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * kScanPerThread;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) if (base + j < n) s += in[base + j];
    uint32_t tot;
    block_exclusive_scan(s, lds, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of the block sums, in place; block_sums[nblocks] receives the grand total
// The second level of the scan is folded into this kernel: block b sums the b block totals before it (a few hundred to a few
// thousand words) instead of reading a pre-scanned array -- one launch fewer per scan, and these launches are ~5 us each on a path
// that is a few hundred microseconds long for the MSMs of an IPP round.  The last block leaves the grand total in block_sums[nblocks].
static __global__ void __launch_bounds__(kBlock) k_scan_apply(const uint32_t* in, size_t n, uint32_t* __restrict__ block_sums,
                                                        uint32_t* out, uint32_t* __restrict__ out_copy) {
    __shared__ uint32_t lds[kBlock / 64];
    uint32_t before = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kBlock) before += block_sums[b];
    uint32_t prefix;
    block_exclusive_scan(before, lds, prefix);                    // prefix = sum of the totals of blocks 0 .. blockIdx.x - 1
    size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread], s = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(s, lds, tot) + prefix;
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) {
        if (base + j < n) { out[base + j] = ex; if (out_copy) out_copy[base + j] = ex; }
        ex += v[j];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) block_sums[gridDim.x] = prefix + tot;
}

// ---------------------------------------------------------------------------------------------- tasks
// A bucket of `size` points is cut into ceil(size / kTaskLen) tasks of at most kTaskLen points, so that no lane
// ever walks more than kTaskLen points (structured scalars -- bit vectors, repeated values -- put thousands of
// points into one bucket).  Tasks are then ordered by length, longest first (counting sort), so that the 64
// lanes of a wave run the same number of additions: with Poisson(32) bucket sizes an unsorted wave waits for
// its longest lane, ~1.45x the mean.
// The task length L is a power of two chosen per call (task_len): when there are plenty of buckets (>> resident lanes) it is
// >= 2x the mean bucket size, so that with uniformly random scalars every bucket is one task; when buckets are few it
// is smaller, so that there are several times more tasks than lanes (otherwise the last, partly filled round of
// waves costs up to half of the kernel).  Lengths are binned into <= 129 classes, longest first.
// Round 3: L is chosen ON THE DEVICE (task_len), from what the sort found -- E entries in NE non-empty buckets -- instead of on the
// host from n W entries in all buckets: scalars with structure (bit vectors, small values, repeated values: the a_L / a_R and value
// commitments of a range proof) fill a handful of buckets of a single window, the host's "every bucket is one task of <= 128" then
// meant 128 dependent additions on a few thousand lanes (accumulate 1.05 ms for 2^17 useful additions at n = 2^18, bits).  The
// host still sizes the task arrays: lmin (its own estimate / 8) bounds L from below.
struct TaskLen { uint32_t L, lshift; };
__device__ __forceinline__ TaskLen task_len(uint32_t entries, const uint32_t* __restrict__ nonempty_w, uint32_t nwin, uint32_t target, uint32_t lmin) {
    __shared__ uint32_t s_ne;                 // non-empty buckets: the sum of the per-window counters of k_fine_place (nwin <= kBlock)
    if (threadIdx.x == 0) s_ne = 0;
    __syncthreads();
    if (threadIdx.x < nwin) { const uint32_t v = nonempty_w[threadIdx.x]; if (v) atomicAdd(&s_ne, v); }
    __syncthreads();
    const uint64_t E = entries, NE = s_ne ? s_ne : 1;
    uint64_t L = 8;
    if (NE >= target) {                       // plenty of buckets: every one of them a single task (twice the mean, at least 128)
        while (L * NE < 2 * E && L < (1u << 20)) L <<= 1;
        if (L < 128) L = 128;
    } else {                                  // few buckets: ~target tasks ...
        while (L * target < E && L < (1u << 20)) L <<= 1;
        // ... unless that cuts EVERY bucket into more than kLightMax tasks (few, fat buckets: a narrow window-multiples table has 2^13
        // buckets of ~300 points at c = 14, n = 2^17) and sends them all through k_combine_chunks, a block per bucket (measured: 5.2 ms
        // per paired MSM against 0.9 ms at c = 16): then the length that leaves a typical bucket <= 6 tasks -- summed by the reduce's
        // own lanes -- as long as a quarter of the resident lanes still get a task.
        uint64_t L2 = L;
        while (L2 * NE * 6 < E && L2 < (1u << 20)) L2 <<= 1;
        if (L2 > L && E / L2 >= 32768) L = L2;
    }
    if (L < lmin) L = lmin;
    TaskLen t;
    t.L = (uint32_t)L;
    t.lshift = 0;
    while ((128u << t.lshift) < t.L) t.lshift++;
    return t;
}
constexpr uint32_t kTaskBins = 129;
constexpr uint32_t kLightMax = 8;   // buckets of 2..kLightMax tasks are summed inside k_bucket_reduce, heavier ones by k_combine_chunks / _heavy
__device__ __forceinline__ uint32_t task_bin(uint32_t len, uint32_t L, uint32_t lshift) { return (L - len) >> lshift; }   // lshift = max(0, log2(L) - 7)

// thread per bucket: ntasks[g], and a histogram of task lengths
// kTaskPer buckets per lane: every block ends with one global atomic per occupied length bin, all blocks on the same ~129 words; with
// one bucket per lane (2048 blocks at 2^19 buckets) that serialised traffic was most of the kernel's 30 us.
constexpr int kTaskPer = 4;
static __global__ void __launch_bounds__(kBlock) k_task_count(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, uint32_t nbuckets,
                                                        const uint32_t* __restrict__ entries, const uint32_t* __restrict__ nonempty, uint32_t nwin, uint32_t target, uint32_t lmin,
                                                        uint32_t* __restrict__ ntasks, uint32_t* __restrict__ bin_count) {
    __shared__ uint32_t lh[kTaskBins];
    const TaskLen tl = task_len(*entries, nonempty, nwin, target, lmin);
    const uint32_t L = tl.L, lshift = tl.lshift;
    for (uint32_t b = threadIdx.x; b < kTaskBins; b += kBlock) lh[b] = 0;
    __syncthreads();
    uint32_t sz[kTaskPer];
#pragma unroll
    for (int u = 0; u < kTaskPer; u++) {
        uint32_t g = (blockIdx.x * kTaskPer + u) * kBlock + threadIdx.x;
        sz[u] = g < nbuckets ? end[g] - start[g] : 0;
    }
#pragma unroll
    for (int u = 0; u < kTaskPer; u++) {
        uint32_t g = (blockIdx.x * kTaskPer + u) * kBlock + threadIdx.x;
        if (g < nbuckets) {
            uint32_t full = sz[u] / L, rem = sz[u] % L;
            ntasks[g] = full + (rem ? 1 : 0);
            if (full) atomicAdd(&lh[0], full);
            if (rem) atomicAdd(&lh[task_bin(rem, L, lshift)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kTaskBins; b += kBlock) if (lh[b]) atomicAdd(&bin_count[b], lh[b]);
}
static __global__ void __launch_bounds__(kBlock) k_task_bins_scan(uint32_t* __restrict__ bin, uint32_t* __restrict__ total) {
    __shared__ uint32_t lds[kBlock / 64];
    static_assert(kTaskBins <= kBlock, "one thread per bin");
    uint32_t v = threadIdx.x < kTaskBins ? bin[threadIdx.x] : 0, tot;
    uint32_t ex = block_exclusive_scan(v, lds, tot);
    if (threadIdx.x < kTaskBins) bin[threadIdx.x] = ex;
    if (threadIdx.x == 0) *total = tot;
}

// thread per bucket: emit its tasks.  task id = task_off[g] + k; order[] lists task ids longest first (positions are
// reserved per block through LDS histograms: one global atomic per (block, length class)).  t_start/t_len describe the
// slot range of a task.  Buckets with more than kLightMax tasks have their kBlock-task chunks appended to chunks[] (and the bucket
// itself to heavy[] when it has more than one chunk).
static __global__ void __launch_bounds__(kBlock) k_task_emit(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, uint32_t nbuckets,
                                                       const uint32_t* __restrict__ entries, const uint32_t* __restrict__ nonempty, uint32_t nwin, uint32_t target, uint32_t lmin,
                                                       const uint32_t* __restrict__ task_off, uint32_t* __restrict__ bin_cursor,
                                                       uint32_t* __restrict__ order, uint32_t* __restrict__ t_start, uint32_t* __restrict__ t_len,
                                                       uint32_t* __restrict__ heavy, uint32_t* __restrict__ nheavy, uint2* __restrict__ chunks,
                                                       uint32_t* __restrict__ nchunks) {
    __shared__ uint32_t lh[kTaskBins], lbase[kTaskBins];
    __shared__ uint32_t big_n, big[kBlock * kTaskPer][4];        // buckets of more than kSerialEmit full tasks: (first task id, first slot, full tasks, first position in order[])
    const TaskLen tl = task_len(*entries, nonempty, nwin, target, lmin);      // the same inputs as k_task_count: the same length
    const uint32_t L = tl.L, lshift = tl.lshift;
    for (uint32_t b = threadIdx.x; b < kTaskBins; b += kBlock) lh[b] = 0;
    if (threadIdx.x == 0) big_n = 0;
    __syncthreads();
    uint32_t size[kTaskPer], s0[kTaskPer], toff[kTaskPer], rank_full[kTaskPer], rank_rem[kTaskPer];
#pragma unroll
    for (int u = 0; u < kTaskPer; u++) {
        uint32_t g = (blockIdx.x * kTaskPer + u) * kBlock + threadIdx.x;
        size[u] = 0; s0[u] = 0; toff[u] = 0; rank_full[u] = 0; rank_rem[u] = 0;
        if (g < nbuckets) { s0[u] = start[g]; size[u] = end[g] - s0[u]; toff[u] = task_off[g]; }
    }
#pragma unroll
    for (int u = 0; u < kTaskPer; u++) {
        const uint32_t nfull = size[u] / L, rem = size[u] % L;   // nfull tasks of exactly L points (bin 0), ranked inside the block, plus one shorter
        if (nfull) rank_full[u] = atomicAdd(&lh[0], nfull);
        if (rem) rank_rem[u] = atomicAdd(&lh[task_bin(rem, L, lshift)], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kTaskBins; b += kBlock) lbase[b] = lh[b] ? atomicAdd(&bin_cursor[b], lh[b]) : 0;
    __syncthreads();
    // A lane writes the tasks of its own bucket while they are few; a bucket of many tasks (one bucket holds HALF of all points when
    // the scalars are bits: 32 768 tasks at n = 2^20, 1.1 ms for the one lane that owned it) is handed to the whole block.
    constexpr uint32_t kSerialEmit = 16;
#pragma unroll
    for (int u = 0; u < kTaskPer; u++) {
        if (size[u] == 0) continue;
        const uint32_t g = (blockIdx.x * kTaskPer + u) * kBlock + threadIdx.x;
        const uint32_t nfull = size[u] / L, rem = size[u] % L;
        if (nfull <= kSerialEmit) {
            for (uint32_t k = 0; k < nfull; k++) {
                order[lbase[0] + rank_full[u] + k] = toff[u] + k;
                t_start[toff[u] + k] = s0[u] + k * L;
                t_len[toff[u] + k] = L;
            }
        } else {
            const uint32_t slot = atomicAdd(&big_n, 1u);
            big[slot][0] = toff[u]; big[slot][1] = s0[u]; big[slot][2] = nfull; big[slot][3] = lbase[0] + rank_full[u];
        }
        if (rem) {
            order[lbase[task_bin(rem, L, lshift)] + rank_rem[u]] = toff[u] + nfull;
            t_start[toff[u] + nfull] = s0[u] + nfull * L;
            t_len[toff[u] + nfull] = rem;
        }
        const uint32_t nt = nfull + (rem ? 1u : 0u);
        if (nt > kLightMax) {
            const uint32_t nch = (nt + kBlock - 1) / kBlock;          // chunks of kBlock task sums for k_combine_chunks
            if (nch > 1) heavy[atomicAdd(nheavy, 1u)] = g;            // k_combine_heavy: only buckets of several chunks
            const uint32_t base = atomicAdd(nchunks, nch);
            for (uint32_t j = 0; j < nch; j++) chunks[base + j] = make_uint2(g, j);
        }
    }
    __syncthreads();
    const uint32_t nbig = big_n;
    for (uint32_t b = 0; b < nbig; b++) {
        const uint32_t t0 = big[b][0], p0 = big[b][1], nfull = big[b][2], o0 = big[b][3];
        for (uint32_t k = threadIdx.x; k < nfull; k += kBlock) {
            order[o0 + k] = t0 + k;
            t_start[t0 + k] = p0 + k * L;
            t_len[t0 + k] = L;
        }
    }
}

// ---------------------------------------------------------------------------------------------- bucket accumulate
// One lane per task.  The accumulator (4 x 13 limbs) lives in VGPRs for the whole run; points are gathered as
// 96-byte rows with 16-byte vector loads.  tsum[task] receives the task's sum.  WPS = waves per SIMD the register
// allocator is asked to fit (2: 191 VGPRs, no scratch -- profiles/r04_kernel_resources.txt; 3: 168 VGPRs + 156 B scratch and no faster; for BLS12-381).
template <class C, int WPS>
__global__ void __launch_bounds__(kBlock, WPS) k_accumulate(const AffPacked<C>* __restrict__ pts, const uint32_t* __restrict__ idx,
                                                       const uint32_t* __restrict__ order, const uint32_t* __restrict__ t_start,
                                                       const uint32_t* __restrict__ t_len, const uint32_t* __restrict__ total_tasks,
                                                       XyzzPacked<C>* __restrict__ tsum) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *total_tasks) return;
    uint32_t tid = order[j];
    uint32_t s = t_start[tid], e = s + t_len[tid];
    using Fp = typename C::Fp;
    // lazy-reduction accumulator (bounded domain, bp_curve.cuh): no conditional subtraction inside the loop.
    // Two nested loops: the outer one takes ONE point through the general addition (empty accumulator, identity point, doubling,
    // cancellation), the inner one is the single-path addition and runs until the task ends or a point needs the general form.
    // With uniformly random input the outer body runs once per task (its first point).
    XyzzLazy<C> acc = xyzz_lazy_inf<C>();
    while (s < e) {
        {
            uint32_t code = idx[s++];
            Aff<C> p = aff_unpack(pts[code & 0x7fffffffu]);
            if (code >> 31) p.y = fe_neg(p.y);
            xyzz_lazy_add_aff(acc, p);
        }
        if (acc.inf) continue;
        // (Software prefetch of the next row -- plain C++ and, because the compiler's register reuse forced an early s_waitcnt, as
        // hand-issued inline-asm loads whose only wait sits at the top of the next iteration -- was measured again in round 3: 2.24-2.30
        // against 2.24-2.25 ms.  The second resident wave already covers the gather; the 8.6 % "waiting on memory" of the SQ
        // counters is time the other wave spends issuing.)
        while (s < e) {
            const uint32_t code = idx[s];
            const Aff<C> p = aff_unpack(pts[code & 0x7fffffffu]);
            if (aff_is_inf(p)) { s++; continue; }
            FeB<Fp, 2> qy = feb_widen<2>(feb_from_strict<Fp>(p.y));
            if (code >> 31) qy = feb_neg_canonical<Fp>(p.y);
            if (!xyzz_lazy_add_aff_fast(acc, feb_from_strict<Fp>(p.x), qy)) break;     // same x: the outer loop handles this point
            s++;
        }
    }
    tsum[tid] = xyzz_lazy_pack(acc);     // bounded, not canonical: every consumer works in the same lazy domain
}

// ---------------------------------------------------------------------------------------------- bucket reduce
// LDS tree over the block's kBlock partial sums (packed, 4*NW words each); result valid in thread 0.
// `active` = number of leading threads that can hold a non-identity value (the tree starts at the first power of two
// that covers them: small MSMs have a handful of segments per window, not 256).
// All bucket-sum + bucket-sum arithmetic from here on is in the bounded ("lazy") domain of bp_curve.cuh: no conditional
// subtraction (v_cndmask costs 22 cycles per wave instruction on gfx950), 17.2 instead of 20.7 us per dependent addition at
// one wave per SIMD (microbench/point_latency.hip); records stay in that domain in memory and the host reduces on entry.
template <class C>
__device__ __forceinline__ XyzzLazy<C> block_tree_sum(XyzzLazy<C> mine, XyzzPacked<C>* lds, int active = kBlock) {
    lds[threadIdx.x] = xyzz_lazy_pack(mine);
    __syncthreads();
    int s0 = kBlock / 2;
    while (s0 >= active && s0 > 0) s0 >>= 1;   // largest stride with a live partner
#pragma unroll 1
    for (int s = s0; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            mine = xyzz_lazy_add(mine, xyzz_lazy_unpack(lds[threadIdx.x + s]));
            lds[threadIdx.x] = xyzz_lazy_pack(mine);
        }
        __syncthreads();
    }
    return mine;
}

// The same tree with every addition shared by the four lanes of a quad (xyzz_lazy_add_quad, bp_curve.cuh): a level of s additions
// keeps 4 s lanes busy for ~1 900 wave instructions instead of s lanes for ~5 300 (levels of more than 64 additions run in rounds of
// 64).  Result valid in thread 0.
template <class C>
__device__ __forceinline__ XyzzLazy<C> block_tree_sum_quad(XyzzLazy<C> mine, XyzzPacked<C>* lds, int active = kBlock) {
    lds[threadIdx.x] = xyzz_lazy_pack(mine);
    __syncthreads();
    int s0 = kBlock / 2;
    while (s0 >= active && s0 > 0) s0 >>= 1;   // largest stride with a live partner
    const int quad = (int)threadIdx.x >> 2, q = (int)threadIdx.x & 3;
#pragma unroll 1
    for (int s = s0; s > 0; s >>= 1) {
#pragma unroll 1
        for (int i = quad; i < s; i += kBlock / 4) xyzz_lazy_add_quad<C>(lds, i, i + s, q);
        __syncthreads();
    }
    return xyzz_lazy_unpack(lds[0]);
}

// Heavy buckets (more than kLightMax tasks) are folded in two stages so that one bucket holding most of the points -- 0/1
// scalars, the a_L / a_R commitments of a range proof, put half of all points into ONE bucket: 4096 task sums at n = 2^16 --
// is a tree over many blocks instead of one long chain (one wave: 64 + 6 dependent additions, 1.23 ms; chunked: 8 + 1 + 4):
//   k_combine_chunks  group of lanes per (bucket, chunk of kBlock task sums), grid-stride over chunks[]: -> first record of the chunk
//   k_combine_heavy   block per bucket of SEVERAL chunks, grid-stride over heavy[]: sum of its chunk sums -> tsum[task_off[g]]
// List lengths are only known on the device, so both grids are fixed-size and stride.
// k_combine_chunks gives a chunk to a GROUP of G lanes, G chosen per launch so that the chunks fit one round of the grid where they
// can: a block (G = 256, one task sum per lane, 8 tree levels) while there are at most as many chunks as blocks -- the one giant
// bucket of bit scalars -- down to 4 lanes when there are thousands of them (256 distinct scalar values: 4096 buckets of 64 task sums,
// 1.9 ms block-per-chunk, 16 rounds of a 6-level tree on 64 of 256 lanes; a window-multiples table at c = 14: 16 384 buckets of
// 10).  Lane l of the group first adds the task sums l, l + G, ... of its chunk (a serial chain), then the group's tree runs over
// the G partial sums, four lanes per addition (xyzz_lazy_add_quad).
template <class C>
__global__ void __launch_bounds__(kBlock) k_combine_chunks(const uint2* __restrict__ chunks, const uint32_t* __restrict__ nchunks,
                                                           const uint32_t* __restrict__ task_off, const uint32_t* __restrict__ ntasks,
                                                           XyzzPacked<C>* __restrict__ tsum) {
    __shared__ XyzzPacked<C> lds[kBlock];
    __shared__ uint32_t s_most, s_live[kBlock / 4];
    const uint32_t count = *nchunks;
    uint32_t G = kBlock;
    while (G > 4 && (uint64_t)count > (uint64_t)gridDim.x * (kBlock / G)) G >>= 1;
    const uint32_t gpb = kBlock / G, lane = threadIdx.x & (G - 1), grp = threadIdx.x / G;
    for (uint32_t c0 = blockIdx.x * gpb; c0 < count; c0 += gridDim.x * gpb) {         // block-uniform bounds
        const uint32_t c = c0 + grp;
        uint32_t first = 0, cnt = 0;
        if (c < count) {
            const uint2 gc = chunks[c];
            const uint32_t left = ntasks[gc.x] - gc.y * kBlock;
            first = task_off[gc.x] + gc.y * kBlock;
            cnt = left < (uint32_t)kBlock ? left : (uint32_t)kBlock;
        }
        if (threadIdx.x == 0) s_most = 0;
        __syncthreads();
        if (lane == 0) { s_live[grp] = cnt < G ? cnt : G; if (cnt) atomicMax(&s_most, cnt); }      // lanes of the group that will hold a partial sum
        __syncthreads();
        const uint32_t nser = (s_most + G - 1) / G;                 // serial steps of the longest chunk of this round (>= 1)
        // serial phase: lane l adds the task sums l, l + G, ... of its chunk
        XyzzLazy<C> mine = lane < cnt ? xyzz_lazy_unpack(tsum[first + lane]) : xyzz_lazy_inf<C>();
#pragma unroll 1
        for (uint32_t t = 1; t < nser; t++) {
            const uint32_t k = lane + t * G;
            if (k < cnt) mine = xyzz_lazy_add(mine, xyzz_lazy_unpack(tsum[first + k]));
        }
        // tree phase over the group's partial sums, every addition on four lanes (xyzz_lazy_add_quad): item (g, i) is
        // slot[g G + i] += slot[g G + i + st], present when lane i + st of group g holds a sum
        lds[threadIdx.x] = xyzz_lazy_pack(mine);
        __syncthreads();
#pragma unroll 1
        for (uint32_t st = G >> 1; st >= 1; st >>= 1) {
#pragma unroll 1
            for (uint32_t it = threadIdx.x >> 2; it < gpb * st; it += kBlock / 4) {
                const uint32_t g = it / st, i = it - g * st;
                if (i + st < s_live[g]) xyzz_lazy_add_quad<C>(lds, (int)(g * G + i), (int)(g * G + i + st), (int)(threadIdx.x & 3));
            }
            __syncthreads();
        }
        if (lane == 0 && cnt) tsum[first] = lds[threadIdx.x];
        __syncthreads();                                            // s_most, s_live and lds are reused by the next round
    }
}

template <class C>
__global__ void __launch_bounds__(kBlock) k_combine_heavy(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ nheavy,
                                                          const uint32_t* __restrict__ task_off, const uint32_t* __restrict__ ntasks,
                                                          XyzzPacked<C>* __restrict__ tsum) {
    __shared__ XyzzPacked<C> lds[kBlock];
    const uint32_t count = *nheavy;
    for (uint32_t h = blockIdx.x; h < count; h += gridDim.x) {
        const uint32_t g = heavy[h];
        const uint32_t t0 = task_off[g], nch = (ntasks[g] + kBlock - 1) / kBlock;
        if (nch < 2) continue;   // a single chunk: k_combine_chunks already left the sum in tsum[t0] (uniform per block)
        XyzzLazy<C> mine = xyzz_lazy_inf<C>();
        for (uint32_t k = threadIdx.x; k < nch; k += kBlock) mine = xyzz_lazy_add(mine, xyzz_lazy_unpack(tsum[t0 + k * kBlock]));
        __syncthreads();   // all chunk sums read before tsum[t0] is overwritten
        mine = block_tree_sum_quad<C>(mine, lds, nch < (uint32_t)kBlock ? (int)nch : kBlock);
        if (threadIdx.x == 0) tsum[t0] = xyzz_lazy_pack(mine);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- bit-plane trees
// LDS holds `nplain + 1` arrays of `live` (a power of two <= pitch) packed lazy points: array 0 (the "plane source" X) at
// lds[0 .. live), plain array a >= 1 at lds[a * pitch ..).  After the call
//     lds[0]         = sum of X                          lds[a * pitch] = sum of plain array a
//     lds[1 << k]    = sum of the X[i] with bit k of i set   (k < log2 live)      -- the "bit planes" of X (planes == true)
// in log2(live) steps of ONE dependent addition each: at stride s the upper half of X is not consumed, it stays where it is and
// becomes the array of plane log2(s), which is then summed like any other array by the following steps.  All arrays alive at a
// step have the same length 2s, so the step is "slot[b + i] += slot[b + i + s]" over a list of bases b; item q of the step is
// (base index q / s, i = q % s) and the threads share the items (one per thread while they fit: (nplain + 1 + born) * s <= 256).
// Code size matters in the kernels around the accumulate loop: an inlined addition is ~50 KB of straight-line code and its first
// pass is paced by instruction fetch (~100-300 us cold against ~17 us warm), so every kernel here keeps its additions at as few
// call sites as it can -- this tree has one, and k_bucket_reduce calls it from ONE place for both of its levels.  (A real function
// shared by all kernels was tried in round 3: the call moves ~200 VGPRs through scratch and quadrupled the reduce, 0.49 -> 2.2 ms.)
template <class C>
__device__ __forceinline__ void plane_tree(XyzzPacked<C>* lds, int live, int pitch, int nplain, bool planes) {
    int born = 0;
#pragma unroll 1
    for (int s = live >> 1; s >= 1; s >>= 1, born += planes ? 1 : 0) {
        const int nitems = (nplain + 1 + born) * s;
        if (nitems <= kBlock / 2) {
            // few items: four lanes per addition (xyzz_lazy_add_quad: ~1 900 instead of ~5 300 wave instructions per round of 64 items)
#pragma unroll 1
            for (int it = (int)threadIdx.x >> 2; it < nitems; it += kBlock / 4) {
                const int u = it / s, i = it - u * s;
                const int base = u == 0 ? 0 : u <= born ? (live >> u) : (u - born) * pitch;
                xyzz_lazy_add_quad<C>(lds, base + i, base + i + s, (int)threadIdx.x & 3);
            }
        } else {
#pragma unroll 1
            for (int q = (int)threadIdx.x; q < nitems; q += kBlock) {
                const int u = q / s, i = q - u * s;
                // u = 0: X itself; 1 .. born: the plane born u steps ago sits at stride live >> u; then the plain arrays
                const int base = u == 0 ? 0 : u <= born ? (live >> u) : (u - born) * pitch;
                XyzzLazy<C> a = xyzz_lazy_unpack(lds[base + i]);
                a = xyzz_lazy_add(a, xyzz_lazy_unpack(lds[base + i + s]));
                lds[base + i] = xyzz_lazy_pack(a);
            }
        }
        __syncthreads();
    }
}

constexpr int kReduceSlots = 3 * kBlock;     // LDS slots of k_bucket_reduce (144 KB for BLS12-381, of the CU's 160 KB: one block per CU, as the grid intends)

// grid = tab.rboff[W] blocks, window w owning blocks rboff[w] .. rboff[w+1]-1 (exactly the blocks that have buckets: with a
// 2-D grid padded to the widest window, the blocks that exit at once skewed the dispatch and some CUs ran two of these long
// dependent chains back to back -- 1.00 ms instead of 0.54 ms at n = 2^16, c = 14; profiles/r01_reduce_grid_*.txt).
// Thread t of window w owns the m = 2^lgm consecutive buckets t*m .. t*m + m - 1 (bucket j <-> weight j + 1) and computes
//     run_t = sum_i bucket[t*m + i]          tri_t = sum_i (i + 1) bucket[t*m + i]          (2m dependent additions)
// so that the window sum is   sum_t tri_t  +  m * sum_t t * run_t.
// Rounds 1-2 multiplied run_t by t*m on the spot (double-and-add: ~22 more dependent point operations per thread, half of the
// kernel).  Now the weight t is never applied on the device:  sum_t t run_t = sum_k 2^k A_k  with A_k = sum of the run_t whose t
// has bit k set, and the A_k come out of the SAME tree that sums the block (plane_tree above) at no extra depth.  Each A_k leaves
// as its own tail record with bit position off_w + lgm + k; the host's Horner walk over bit positions (bp_host_tail.hpp) passes
// every position anyway, so a plane costs it one addition (~0.5 us) where the device paid 16 us per dependent step.
// Two levels, one kernel: every block leaves [sum tri | sum run | planes of its 8 thread-index bits] in partial[], and the LAST
// block of a window to finish (a counter per window) folds the window's blocks: plain sums of tri and of the thread-bit planes,
// and the planes of the blocks' run totals over the BLOCK index (the upper bits of the thread index).  A separate second kernel
// for that cost 146 us per MSM for two additions (cold code, see plane_tree); here the tree's code is already warm.
// Records of window w (lazy XYZZ, packed), at window_sum[roff[w] ..]:  tri | thread-bit planes | block-bit planes.
// Bucket g's sum is tsum[task_off[g]] (identity when it has no task).
// (Round 2 tried plain "digit sums" of the bucket values instead -- lost: profiles/r02_digit_sum_experiment.txt.)
template <class C>
__global__ void __launch_bounds__(kBlock) k_bucket_reduce(const XyzzPacked<C>* __restrict__ tsum, const uint32_t* __restrict__ task_off,
                                                          const uint32_t* __restrict__ ntasks, WinTab tab,
                                                          XyzzPacked<C>* __restrict__ partial, uint32_t* __restrict__ done,
                                                          XyzzPacked<C>* __restrict__ window_sum) {
    __shared__ XyzzPacked<C> lds[kReduceSlots];
    __shared__ uint32_t s_last;
    const uint32_t bid = blockIdx.x;
    uint32_t w = 0;
    while (w + 1 < (uint32_t)tab.W && tab.rboff[w + 1] <= bid) w++;   // uniform scan, W <= 256
    const uint32_t first = tab.rboff[w], nblk = tab.rboff[w + 1] - first;
    const uint32_t bx = bid - first, lgm = tab.lgm[w], m = 1u << lgm;
    const uint32_t B = tab.boff[w + 1] - tab.boff[w];
    const uint32_t T = B >> lgm;                                       // threads of the window: a power of two
    const uint32_t t = bx * kBlock + threadIdx.x;
    const int live = T < (uint32_t)kBlock ? (int)T : kBlock;
    int lgT = 0, lgB = 0;
    while ((1 << lgT) < live) lgT++;
    while ((1u << lgB) < nblk) lgB++;
    // ---- per-thread running sums: ONE addition site; an "operation" is  run += (a task sum)  or  tri += run
    XyzzLazy<C> run = xyzz_lazy_inf<C>(), tri = xyzz_lazy_inf<C>();
    if (t < T) {
        const uint32_t lo = t * m;
        uint32_t j = lo + m;                 // buckets are walked downwards: j - 1 is the current one
        uint32_t k = 0, lim = 0, t0 = 0;     // task sums of the current bucket still to add: k .. lim - 1 at tsum[t0 + k]
        bool fresh = true;                   // the current bucket has not been looked at yet
#pragma unroll 1
        while (j > lo) {
            if (fresh) {
                const uint32_t g = tab.boff[w] + j - 1;
                const uint32_t nt = ntasks[g];
                // a bucket cut into 2..kLightMax tasks is summed here (a separate lane-per-bucket kernel for it cost 0.1-0.2 ms
                // at n = 2^16); a heavier bucket was already folded into its first record by k_combine_chunks / _heavy
                lim = nt == 0 ? 0 : nt <= kLightMax ? nt : 1;
                t0 = nt ? task_off[g] : 0;
                k = 0;
                fresh = false;
            }
            const bool take = k < lim;       // run += tsum[t0 + k]   else   tri += run, next bucket
            XyzzLazy<C> a = take ? run : tri;
            const XyzzLazy<C> b = take ? xyzz_lazy_unpack(tsum[t0 + k]) : run;
            a = xyzz_lazy_add(a, b);
            if (take) { run = a; k++; }
            else { tri = a; j--; fresh = true; }
        }
    }
    // ---- two tree levels through ONE call of plane_tree: pass 0 = this block, passes >= 1 = the window (last block only)
    if ((int)threadIdx.x < live) {
        lds[threadIdx.x] = xyzz_lazy_pack(run);
        lds[kBlock + threadIdx.x] = xyzz_lazy_pack(tri);
    }
    __syncthreads();
    XyzzPacked<C>* rec = window_sum + tab.roff[w];
    const int narr = 1 + lgT;                                   // plain arrays of the window level: tri and the thread-bit planes
    int per = (int)(kReduceSlots / nblk);                       // arrays of nblk entries that fit the LDS at once
    if (per > narr + 1) per = narr + 1;
    int next_arr = 0;                                           // window level: next partial kind (0 = run totals, 1 = tri, 2 + k = plane k) to fold
    int tree_live = live, tree_pitch = kBlock, tree_plain = 1;
    bool tree_planes = true;
#pragma unroll 1
    for (int pass = 0;; pass++) {
        plane_tree<C>(lds, tree_live, tree_pitch, tree_plain, tree_planes);
        if (pass == 0) {
            if (nblk == 1) {                                    // one block: its sums are the window's records
                if ((int)threadIdx.x <= lgT) rec[threadIdx.x] = threadIdx.x == 0 ? lds[kBlock] : lds[1u << (threadIdx.x - 1)];
                return;
            }
            if ((int)threadIdx.x < 2 + lgT) {
                const int q = (int)threadIdx.x;                 // partial kinds: 0 = sum run, 1 = sum tri, 2 + k = plane of thread bit k
                partial[(size_t)bid * kPartPerBlock + q] = q == 0 ? lds[0] : q == 1 ? lds[kBlock] : lds[1u << (q - 2)];
            }
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) s_last = atomicAdd(&done[w], 1u) == nblk - 1 ? 1u : 0u;
            __syncthreads();
            if (!s_last) return;
            __threadfence();
        } else {
            // results of the group just folded: kinds [grp0, next_arr); kind 0 sits at lds[0] with its planes, kind q at slot (q - grp0) * nblk
            const int grp0 = next_arr - tree_plain - 1;         // the group held tree_plain + 1 arrays
            if (tree_planes && (int)threadIdx.x < lgB) rec[1 + lgT + threadIdx.x] = lds[1u << threadIdx.x];     // block-bit planes of the run totals
            const int nres = next_arr - grp0;
            if ((int)threadIdx.x < nres) {
                const int q = grp0 + (int)threadIdx.x;
                if (q >= 1) rec[q - 1] = lds[(size_t)threadIdx.x * nblk];        // kind 1 (tri) -> record 0, kind 2 + k -> record 1 + k
            }
            if (next_arr > narr) return;
            __syncthreads();
        }
        // load the next group of partial kinds: entry bx2 of kind q from partial[(first + bx2) * kPartPerBlock + q]
        const int take = narr + 1 - next_arr < per ? narr + 1 - next_arr : per;
        for (int e = (int)threadIdx.x; e < take * (int)nblk; e += kBlock) {
            const int qa = e / (int)nblk, bx2 = e - qa * (int)nblk;
            lds[(size_t)qa * nblk + bx2] = partial[(size_t)(first + bx2) * kPartPerBlock + next_arr + qa];
        }
        tree_planes = next_arr == 0;                            // only the run totals (kind 0) spawn planes
        tree_plain = take - 1;
        tree_live = (int)nblk;
        tree_pitch = (int)nblk;
        next_arr += take;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- small MSM (n <= kSmallMsmMax)
constexpr size_t kSmallMsmMax = 1536;   // above this the bucket pipeline wins (512 until the tree's additions ran on four lanes each; same-box
                                        // A/B after that: 1024 terms 0.475 -> 0.365 ms single-launch, 2048 terms 0.48 = 0.47 ms; IPP verify at n = 256 / 512
                                        // -- 529 / 1043 terms -- 0.51 -> 0.43 / 0.55 -> 0.50 ms).  More than 512 terms: two blocks per window.
constexpr size_t kSmallDigitMax = 8193;  // ... unless the digit multiples of the points are at hand (k_digit_table_build): a lane then pays one
                                         // addition per term.  Above 512 terms a window's terms are dealt to TWO blocks (64 windows x 2 scalar
                                         // sets x 2 = one block per CU), each leaving its own record at the window's bit position: 2n + 1 = 8193
                                         // terms are 16 serial additions + the 8-level tree.  (16 385 terms were tried: BLS12-381 n = 8192
                                         // 7.85 -> 7.55 ms, BN254 5.85 -> 6.35 ms -- left to the pipeline.)
// The bucket pipeline is a dozen dependent launches; for the 2n + 1 <= 256 terms of a small inner-product round most of its
// time is launch gaps and the depth of the bucket reduce.  Here one block per window does the whole job in one launch: lane t
// multiplies point t by its signed digit of this window (|digit| <= 2^(cw-1): a few doublings and mixed additions), then an LDS
// tree adds the n products.  Same digit recoding and window table as the pipeline, so the host tail is unchanged.
__device__ __forceinline__ uint32_t window_bits(const uint64_t (&q)[4], int off, int cw) {
    const int word = off >> 6, sh = off & 63;
    uint64_t lo = word == 0 ? q[0] : word == 1 ? q[1] : word == 2 ? q[2] : q[3];
    uint64_t hi = word == 0 ? q[1] : word == 1 ? q[2] : word == 2 ? q[3] : 0;
    uint64_t v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    return (uint32_t)v & ((1u << cw) - 1);
}

// mult != nullptr (round 3): the digit multiples m P_t, m = 1 .. 2^(cmax-1), precomputed once for a vector that many small MSMs run
// over (the [G | H | Q] of an inner-product argument of <= 255 generators: every round is an MSM over the same points).  The lane
// then LOADS its term -- no doubling / mixed-addition bodies are executed (or fetched) at all, the kernel is its tree.
constexpr int kSmallDigitBits = 4;     // the window width of the small path (msm_geom): |digit| <= 8
template <class C>
__global__ void __launch_bounds__(kBlock) k_digit_table_build(const AffPacked<C>* __restrict__ pts, uint32_t n, XyzzPacked<C>* __restrict__ mult,
                                                              uint32_t rows = 1u << (kSmallDigitBits - 1)) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const Aff<C> p = aff_unpack(pts[t]);
    XyzzLazy<C> acc = xyzz_lazy_inf<C>();
#pragma unroll 1
    for (uint32_t m = 1; m <= rows; m++) {       // one addition body: it handles the empty accumulator and P + P
        xyzz_lazy_add_aff(acc, p);
        mult[(size_t)(m - 1) * n + t] = xyzz_lazy_pack(acc);
    }
}

// The same table with a lane per (point, multiple): m P by double-and-add from P -- at most 4 doublings + 3 additions deep for m <= 16
// where the serial form above chains `rows` additions per lane.  For SMALL vectors (rows * n lanes fit the chip once): the build sits on
// the critical path of a small proof's state creation, and there the depth is what costs, not the ~3x as many point operations.
template <class C>
__global__ void __launch_bounds__(kBlock) k_digit_table_build_par(const AffPacked<C>* __restrict__ pts, uint32_t n, XyzzPacked<C>* __restrict__ mult,
                                                                  uint32_t rows) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * rows) return;
    const uint32_t m = idx / n + 1, t = idx - (m - 1) * n;
    const Aff<C> p = aff_unpack(pts[t]);
    XyzzLazy<C> acc = xyzz_lazy_inf<C>();
    xyzz_lazy_add_aff(acc, p);                                    // (identity rows stay the identity)
#pragma unroll 1
    for (int i = 30 - __clz((int)m); i >= 0; i--) {               // bits below the leading one
        acc = xyzz_lazy_dbl(acc);
        if ((m >> i) & 1) xyzz_lazy_add_aff(acc, p);
    }
    mult[(size_t)(m - 1) * n + t] = xyzz_lazy_pack(acc);
}
constexpr uint32_t kDigitTableParMax = 32768;                    // lanes up to which the parallel form is used

// The terms of an inner-product round that CAN be non-zero.  A round's scalar set over [G (n0) | H (n0) | Q] is zero on half of the
// generators by construction (bp_ipp.cuh: L takes the G_k with k mod live >= h and the H_k with k mod live < h, R the other halves),
// so a lane that walks all 2 n0 + 1 terms spends half of its serial steps skipping zeros -- whole waves at a time in the first
// rounds, lane by lane (i.e. at the full price of an addition) once h < 64.  With live != 0 the lanes enumerate the n0 + 1
// participating terms instead: e < n0 + 1 -> term index.  live = 0: identity (a plain MSM).
struct IppSparse { uint32_t n0, live, h; };
__device__ __forceinline__ uint32_t ipp_term(const IppSparse& sp, int set, uint32_t e) {
    if (sp.live == 0) return e;
    if (e >= sp.n0) return 2 * sp.n0;                                   // Q
    const uint32_t half = e >= sp.n0 / 2 ? 1u : 0u, j = e - half * (sp.n0 / 2), blk = j / sp.h, o = j - blk * sp.h;
    const bool upper = (set == 0) == (half == 0);                       // L: G upper, H lower;  R: G lower, H upper
    return half * sp.n0 + blk * sp.live + (upper ? sp.h : 0u) + o;
}

// MODE (round 4: one body per way of obtaining a term -- with all three in one kernel the BN254 instantiation needed 512 VGPRs + 256
// AGPRs and still spilled 153 registers): 0 = the lane multiplies its point by the digit (doubling chain); 1 = it loads
// the digit's multiple from `mult` as a packed lazy XYZZ point, full addition; 2 = `mult` holds AFFINE rows (AffPacked, canonical:
// the batch conversion of bp_compact.cuh), mixed addition -- 8M + 2S instead of 12M + 2S per term of the lane's serial chain.
template <class C, int MODE>
__global__ void __launch_bounds__(kBlock) k_small_msm(const AffPacked<C>* __restrict__ pts, const ScalarWords* __restrict__ sc1,
                                                      const ScalarWords* __restrict__ sc2, uint32_t n, WinTab tab,
                                                      XyzzPacked<C>* __restrict__ window_sum, const void* __restrict__ mult, IppSparse sp) {
    __shared__ XyzzPacked<C> lds[kBlock];
    const int w = blockIdx.x, wps = tab.W / tab.nsets, set = w / wps;
    const int cw = tab.cw[w], off = tab.off[w];
    const uint32_t ne = sp.live ? sp.n0 + 1 : n;                        // terms this block's lanes walk (see IppSparse)
    XyzzLazy<C> mine = xyzz_lazy_inf<C>();
    // grid.y blocks share a window's terms (block b: terms b * 256 + lane, stride 256 * grid.y) and leave one record each
    if (MODE == 2) {
        // the lane's serial chain with k_accumulate's single-path mixed addition: the accumulator starts as the first term with a
        // non-zero digit (a scan of digits only, so that every lane enters the addition loop together), the general addition takes
        // the rare doubling / cancellation
        using Fp = typename C::Fp;
        const uint32_t stride = kBlock * gridDim.y;
        uint32_t t = blockIdx.y * kBlock + threadIdx.x;
        const uint32_t n_all = n;
        n = ne;                                                         // the loops below run over the enumeration; fetch maps it to the term
        auto fetch = [&](uint32_t e, Aff<C>& p, bool& neg) -> bool {
            const uint32_t tt = ipp_term(sp, set, e);
            uint64_t q[4];
            add256(q, (set ? sc2 : sc1)[tt], tab.bias);
            const int d = (int)window_bits(q, off, cw) - ((1 << (cw - 1)) - 1);
            if (d == 0) return false;
            p = aff_unpack(((const AffPacked<C>*)mult)[(size_t)((d < 0 ? -d : d) - 1) * n_all + tt]);
            neg = d < 0;
            return !aff_is_inf(p);
        };
        auto restart = [&]() {
            Aff<C> p; bool neg;
            while (t < n && !fetch(t, p, neg)) t += stride;
            if (t < n) {
                if (neg) p.y = fe_neg(p.y);
                mine = xyzz_lazy_from_strict(xyzz_from_aff(p));
                t += stride;
            }
        };
        restart();
        while (t < n) {
            while (t < n) {
                Aff<C> p; bool neg;
                if (!fetch(t, p, neg)) { t += stride; continue; }
                FeB<Fp, 2> qy = feb_widen<2>(feb_from_strict<Fp>(p.y));
                if (neg) qy = feb_neg_canonical<Fp>(p.y);
                if (!xyzz_lazy_add_aff_fast(mine, feb_from_strict<Fp>(p.x), qy)) break;
                t += stride;
            }
            if (t < n) {
                Aff<C> p; bool neg;
                if (fetch(t, p, neg)) {
                    if (neg) p.y = fe_neg(p.y);
                    xyzz_lazy_add_aff(mine, p);
                }
                t += stride;
                if (mine.inf) restart();
            }
        }
    } else {
#pragma unroll 1
    for (uint32_t e = blockIdx.y * kBlock + threadIdx.x; e < ne; e += kBlock * gridDim.y) {      // kSmallMsmMax / kBlock terms per lane at most (16 with mult)
        const uint32_t t = ipp_term(sp, set, e);
        uint64_t q[4];
        add256(q, (set ? sc2 : sc1)[t], tab.bias);
        int d = (int)window_bits(q, off, cw) - ((1 << (cw - 1)) - 1);
        if (d == 0) continue;
        if (MODE == 1) {
            XyzzLazy<C> acc = xyzz_lazy_unpack(((const XyzzPacked<C>*)mult)[(size_t)((d < 0 ? -d : d) - 1) * n + t]);
            if (d < 0 && !acc.inf) acc.y = feb_neg<4>(acc.y);
            mine = xyzz_lazy_add(mine, acc);
        } else {
            Aff<C> p = aff_unpack(pts[t]);
            if (d < 0) { p.y = fe_neg(p.y); d = -d; }
            XyzzLazy<C> acc = xyzz_lazy_from_strict(xyzz_from_aff(p));
#pragma unroll 1
            for (int i = 30 - __clz(d); i >= 0; i--) {            // bits below the leading one
                acc = xyzz_lazy_dbl(acc);
                if ((d >> i) & 1) xyzz_lazy_add_aff(acc, p);
            }
            mine = xyzz_lazy_add(mine, acc);
        }
    }
    }
    mine = block_tree_sum_quad<C>(mine, lds, ne < (uint32_t)kBlock ? (int)ne : kBlock);
    if (threadIdx.x == 0) window_sum[tab.roff[w] + blockIdx.y] = xyzz_lazy_pack(mine);
}

// ---------------------------------------------------------------------------------------------- device tail (optional)
// sum_w 2^(off_w) S_w on the device: a strictly serial chain of ~bits doublings, executed by ONE lane.  Kept so that the
// result can stay in HBM and to put a number on the design decision: ~2 ms here against ~0.13 ms for
// the host fold (bp_host_tail.hpp).  out_le = canonical x || y little-endian words (all-zero = identity).
template <class C>
__global__ void __launch_bounds__(64) k_tail_fold(const XyzzPacked<C>* __restrict__ rec, const uint16_t* __restrict__ pos, int nrec, uint32_t* __restrict__ out_le) {
    using Fp = typename C::Fp;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int top = 0;
    for (int r = 0; r < nrec; r++) top = pos[r] > top ? pos[r] : top;
    XyzzLazy<C> acc = xyzz_lazy_inf<C>();
#pragma unroll 1
    for (int p = top; p >= 0; p--) {          // Horner over the bit positions: one doubling per position, the records of weight 2^p added on the way
        if (p < top) acc = xyzz_lazy_dbl(acc);
#pragma unroll 1
        for (int r = 0; r < nrec; r++)
            if (pos[r] == p) acc = xyzz_lazy_add(acc, xyzz_lazy_unpack(rec[r]));
    }
    Aff<C> a = xyzz_to_aff<C>(xyzz_lazy_to_strict(acc));
    uint32_t xw[Fp::NW], yw[Fp::NW];
    fe_pack_words<Fp>(xw, fe_from_mont<Fp>(a.x));
    fe_pack_words<Fp>(yw, fe_from_mont<Fp>(a.y));
    for (int k = 0; k < Fp::NW; k++) { out_le[k] = xw[k]; out_le[Fp::NW + k] = yw[k]; }
}

// ---------------------------------------------------------------------------------------------- conversions
// canonical LE words (x || y per point) <-> resident packed Montgomery affine.
// err != nullptr: VALIDATING form used for every point that arrives from outside (amcl's G1::from_bytes checks the curve
// equation; the a = 0 formulas here are only complete ON the curve): a coordinate >= p (non-canonical encoding) or a point off
// y^2 = x^3 + b sets bit 0 of *err and is stored as the identity, so that nothing downstream computes with it.
template <class C>
__global__ void __launch_bounds__(kBlock) k_points_to_resident(const uint32_t* __restrict__ raw, size_t n, AffPacked<C>* __restrict__ out, uint32_t* __restrict__ err) {
    using Fp = typename C::Fp;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* p = raw + i * 2 * Fp::NW;
    uint32_t xw[Fp::NW], yw[Fp::NW];
    for (int k = 0; k < Fp::NW; k++) { xw[k] = p[k]; yw[k] = p[Fp::NW + k]; }
    Aff<C> a;
    a.x = fe_to_mont<Fp>(fe_unpack_words<Fp>(xw));
    a.y = fe_to_mont<Fp>(fe_unpack_words<Fp>(yw));
    if (err) {
        bool ok = words_lt_mod<Fp>(xw) && words_lt_mod<Fp>(yw) && aff_on_curve(a);
        if (!ok) { atomicOr(err, 1u); a.x = fe_zero<Fp>(); a.y = fe_zero<Fp>(); }
    }
    out[i] = aff_pack(a);
}

// scalars must be canonical (< r): the signed-digit recoding adds a bias and assumes k < 2^fr_bits
template <class C>
__global__ void __launch_bounds__(kBlock) k_check_scalars(const ScalarWords* __restrict__ k, size_t n, uint32_t* __restrict__ err) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ScalarWords s = k[i];
    if (!words_lt_mod<typename C::Fr>(s.w)) atomicOr(err, 2u);
}

template <class C>
__global__ void __launch_bounds__(kBlock) k_points_from_resident(const AffPacked<C>* __restrict__ in, size_t n, uint32_t* __restrict__ raw) {
    using Fp = typename C::Fp;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<C> a = aff_unpack(in[i]);
    uint32_t xw[Fp::NW], yw[Fp::NW];
    fe_pack_words<Fp>(xw, fe_from_mont<Fp>(a.x));
    fe_pack_words<Fp>(yw, fe_from_mont<Fp>(a.y));
    uint32_t* p = raw + i * 2 * Fp::NW;
    for (int k = 0; k < Fp::NW; k++) { p[k] = xw[k]; p[Fp::NW + k] = yw[k]; }
}

// ---------------------------------------------------------------------------------------------- window-multiples table
// rows[w * n + i] = 2^(c w) P_i  for w < W1 (row 0 = the vector itself): the table behind the merged-window MSM (WinTab::merged).
// The generators of a proof system are public parameters reused by every proof (/root/reference src/r1cs/prover.rs:347-362,
// src/ipp.rs:91,104,158,170), so the table is built once per vector (bp_g1vec_precompute).
// One lane per point: c doublings per window in the lazy domain, every window's XYZZ value and the running product of its
// ZZ * ZZZ parked in HBM (tmp / pre), ONE field inversion per lane, then the walk back (Montgomery's trick) turns each parked
// value into an affine row with 6 products.  An identity point gives identity rows.
template <class C>
__global__ void __launch_bounds__(kBlock) k_table_build(const AffPacked<C>* __restrict__ pts, size_t n, int c, int W1, XyzzPacked<C>* __restrict__ tmp,
                                                        FePacked<typename C::Fp>* __restrict__ pre, AffPacked<C>* __restrict__ rows) {
    using Fp = typename C::Fp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AffPacked<C> p0 = pts[i];
    rows[i] = p0;
    const Aff<C> p = aff_unpack(p0);
    if (aff_is_inf(p)) {
        AffPacked<C> z;
        for (int k = 0; k < Fp::NW; k++) { z.x.w[k] = 0; z.y.w[k] = 0; }
        for (int w = 1; w < W1; w++) rows[(size_t)w * n + i] = z;
        return;
    }
    XyzzLazy<C> acc = xyzz_lazy_from_strict(xyzz_from_aff(p));
    FeB<Fp, 2> prod = feb_widen<2>(feb_from_strict<Fp>(fe_one<Fp>()));
#pragma unroll 1
    for (int w = 1; w < W1; w++) {
#pragma unroll 1
        for (int k = 0; k < c; k++) acc = xyzz_lazy_dbl(acc);
        tmp[(size_t)(w - 1) * n + i] = xyzz_lazy_pack(acc);
        prod = feb_mul(prod, feb_mul(acc.zz, acc.zzz));
        Fe<Fp> pv;
        for (int k = 0; k < Fp::NL; k++) pv.v[k] = prod.v[k];
        pre[(size_t)(w - 1) * n + i] = fe_pack(pv);                       // < 2p: fits the packed words
    }
    FeB<Fp, 2> inv = feb_widen<2>(feb_from_strict<Fp>(fe_inv<Fp>(feb_to_strict(prod))));
#pragma unroll 1
    for (int w = W1 - 1; w >= 1; w--) {
        const XyzzLazy<C> q = xyzz_lazy_unpack(tmp[(size_t)(w - 1) * n + i]);
        FeB<Fp, 2> prev = feb_widen<2>(feb_from_strict<Fp>(fe_one<Fp>()));
        if (w > 1) { const Fe<Fp> pv = fe_unpack(pre[(size_t)(w - 2) * n + i]); for (int k = 0; k < Fp::NL; k++) prev.v[k] = pv.v[k]; }
        const FeB<Fp, 2> tinv = feb_mul(inv, prev);                       // 1 / (ZZ_w ZZZ_w)
        inv = feb_mul(inv, feb_mul(q.zz, q.zzz));
        Aff<C> a;
        a.x = feb_to_strict(feb_mul(q.x, feb_mul(tinv, q.zzz)));          // X / ZZ
        a.y = feb_to_strict(feb_mul(q.y, feb_mul(tinv, q.zz)));           // Y / ZZZ
        rows[(size_t)w * n + i] = aff_pack(a);
    }
}

// ---------------------------------------------------------------------------------------------- batched scalar mul
// out[i] = k[i] * base[i]  (base == nullptr: the curve generator).  One lane per element, double-and-add.
template <class C>
__global__ void __launch_bounds__(kBlock) k_scalar_mul(const AffPacked<C>* __restrict__ base, const ScalarWords* __restrict__ k, size_t n,
                                                       AffPacked<C>* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<C> p = base ? aff_unpack(base[i]) : generator<C>();
    ScalarWords s = k[i];
    Xyzz<C> r = xyzz_mul_words<C>(s.w, p);
    out[i] = aff_pack(xyzz_to_aff<C>(r));
}

// out[i] = k[i] * G from a per-context table  T[j][d - 1] = d * 2^(4 j) * G  (j < 64, d = 1..15; 960 affine points):
// 64 lazy mixed additions per element and no doubling at all.  (k_scalar_mul with base == nullptr does the same by
// 255 doublings + ~128 additions per lane.)
constexpr int kFixedBaseWindows = 64;
template <class C>
__global__ void __launch_bounds__(kBlock) k_fixed_base(const AffPacked<C>* __restrict__ table, const ScalarWords* __restrict__ k, size_t n,
                                                       AffPacked<C>* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ScalarWords s = k[i];
    uint64_t q0 = s.w[0] | ((uint64_t)s.w[1] << 32), q1 = s.w[2] | ((uint64_t)s.w[3] << 32), q2 = s.w[4] | ((uint64_t)s.w[5] << 32),
             q3 = s.w[6] | ((uint64_t)s.w[7] << 32);
    XyzzLazy<C> acc = xyzz_lazy_inf<C>();
    for (int j = 0; j < kFixedBaseWindows; j++) {
        uint32_t d = (uint32_t)q0 & 15;
        q0 = (q0 >> 4) | (q1 << 60); q1 = (q1 >> 4) | (q2 << 60); q2 = (q2 >> 4) | (q3 << 60); q3 >>= 4;
        if (d) xyzz_lazy_add_aff(acc, aff_unpack(table[j * 15 + (d - 1)]));
    }
    out[i] = aff_pack(xyzz_to_aff<C>(xyzz_lazy_to_strict(acc)));
}

}  // namespace bp
