// bp_compact.cuh -- gfx950 kernels that MATERIALISE folded generators in the middle of an inner-product proof.
//
// The default prover never folds G and H (bp_ipp.cuh: every round is a paired MSM over the original [G | H | Q] with per-generator
// coefficients c_G, c_H), which replaced the reference's per-round G1::binary_scalar_mul fold (/root/reference src/ipp.rs:119,125,
// 185,187 inside the loop :138-194) -- but it keeps paying a FULL-size MSM in every round, whatever the live length.  Once the live
// length nj is small enough for the single-launch rounds (2 nj + 1 <= kSmallDigitMax terms) the folded generators
//     G'_j = sum_{t < T} c_G[j + t nj] G[j + t nj],    H'_j likewise,    T = n0 / nj            (exactly the reference's G, H at that round)
// are computed ONCE -- 2 nj independent T-term MSMs with full-size scalars -- and the remaining rounds run over [G' | H' | Q]:
//   k_digit_table_build (bp_kernels.cuh)   D[m][i] = m P_i, m = 1 .. 8, for all 2 n0 originals         (needs no challenge: queued early)
//   k_bai_*                                batch conversion of D to AFFINE rows (Montgomery's trick over the whole array)
//   k_compact_window_sums                  W[w][o] = sum_t digit_w(c_t) P_t: lane per (output o, 4-bit window w), T mixed additions
//   k_compact_horner                       S_o = sum_w 16^w W[w][o]: a strictly serial chain of 4 doublings + 1 addition per window,
//                                          every operation on the four lanes of a quad (xyzz_lazy_dbl_quad / _add_quad)
//   k_digit_table_build_xyzz, k_bai_*      digit multiples of the S_o (affine) for k_small_msm
// With a window-multiples table the caller precomputed (rows 2^(64 k) P_i) the Horner chain is 60 doublings instead of 252: a scalar
// is then K = 4 sub-scalars of 64 bits over K table rows and nwin = 16.
#pragma once
#include "bp_kernels.cuh"

namespace bp {

// ---------------------------------------------------------------------------------------------- batch XYZZ -> affine
// out[i] = affine form of in[i] (canonical Montgomery rows; the identity -> all-zero row) with ONE field inversion for the whole
// array: z_i = ZZ_i ZZZ_i (1 for the identity), product trees over lanes / blocks, the root inverted once (on the host: ~3 us of
// binary Euclid, bp_host_tail.hpp -- or by one lane when nobody waits for the result), inverses handed back down the same trees,
//     1 / ZZ = ZZZ / z,   1 / ZZZ = ZZ / z.
// Layout: block b owns elements [b * kBaiTile, (b + 1) * kBaiTile), lane l the elements b * kBaiTile + e * kBlock + l.
constexpr int kBaiPer = 4;
constexpr int kBaiTile = kBaiPer * kBlock;
constexpr int kBaiMidPer = 16;                       // block products per lane of the single middle block: n <= 256 * 16 * 1024 elements

template <class P> struct FeLds { uint32_t v[P::NL]; };
template <class P> __device__ __forceinline__ FeB<P, 2> lds_get(const FeLds<P>& s) { FeB<P, 2> r; for (int i = 0; i < P::NL; i++) r.v[i] = s.v[i]; return r; }
template <class P> __device__ __forceinline__ void lds_put(FeLds<P>& s, const FeB<P, 2>& a) { for (int i = 0; i < P::NL; i++) s.v[i] = a.v[i]; }
template <class P> __device__ __forceinline__ FeB<P, 2> feb_one2() { FeB<P, 2> r; for (int i = 0; i < P::NL; i++) r.v[i] = P::C.one[i]; return r; }
template <class P> __device__ __forceinline__ FeB<P, 2> feb_load2(const FePacked<P>& p) {
    const Fe<P> t = fe_unpack_words<P>(p.w);
    FeB<P, 2> r;
    for (int i = 0; i < P::NL; i++) r.v[i] = t.v[i];
    return r;
}
template <class P> __device__ __forceinline__ FePacked<P> feb_store2(const FeB<P, 2>& a) {
    Fe<P> t;
    for (int i = 0; i < P::NL; i++) t.v[i] = a.v[i];
    return fe_pack(t);                               // < 2p < 2^(32 NW)
}

// heap-shaped product tree over the kBlock leaves node[kBlock + l]: node[i] = node[2i] node[2i + 1], root node[1]
template <class P> __device__ __forceinline__ void bai_tree_up(FeLds<P>* node) {
#pragma unroll 1
    for (int s = kBlock / 2; s >= 1; s >>= 1) {
        __syncthreads();
        if ((int)threadIdx.x < s) {
            const int i = s + (int)threadIdx.x;
            lds_put<P>(node[i], feb_mul(lds_get<P>(node[2 * i]), lds_get<P>(node[2 * i + 1])));
        }
    }
    __syncthreads();
}
// node[1] holds the inverse of the root product: afterwards leaf node[kBlock + l] holds the inverse of leaf l's value
template <class P> __device__ __forceinline__ void bai_tree_down(FeLds<P>* node) {
#pragma unroll 1
    for (int s = 1; s < kBlock; s <<= 1) {
        __syncthreads();
        if ((int)threadIdx.x < s) {
            const int i = s + (int)threadIdx.x;
            const FeB<P, 2> iv = lds_get<P>(node[i]), a = lds_get<P>(node[2 * i]), b = lds_get<P>(node[2 * i + 1]);
            lds_put<P>(node[2 * i], feb_mul(iv, b));
            lds_put<P>(node[2 * i + 1], feb_mul(iv, a));
        }
    }
    __syncthreads();
}

// z of element i (ZZ * ZZZ in the lazy domain, < 2p), 1 for the identity / past the end
template <class C> __device__ __forceinline__ FeB<typename C::Fp, 2> bai_z(const XyzzPacked<C>* in, size_t i, size_t n) {
    using Fp = typename C::Fp;
    if (i >= n) return feb_one2<Fp>();
    const FeB<Fp, 2> zz = feb_load2<Fp>(in[i].zz);
    uint32_t any = 0;
    for (int k = 0; k < Fp::NL; k++) any |= zz.v[k];
    if (!any) return feb_one2<Fp>();
    return feb_mul(zz, feb_load2<Fp>(in[i].zzz));
}

template <class C>
__global__ void __launch_bounds__(kBlock) k_bai_block_products(const XyzzPacked<C>* __restrict__ in, size_t n, FePacked<typename C::Fp>* __restrict__ bprod) {
    using Fp = typename C::Fp;
    __shared__ FeLds<Fp> node[2 * kBlock];
    const size_t base = (size_t)blockIdx.x * kBaiTile + threadIdx.x;
    FeB<Fp, 2> prod = bai_z<C>(in, base, n);
#pragma unroll 1
    for (int e = 1; e < kBaiPer; e++) prod = feb_mul(prod, bai_z<C>(in, base + (size_t)e * kBlock, n));
    lds_put<Fp>(node[kBlock + threadIdx.x], prod);
    bai_tree_up<Fp>(node);
    if (threadIdx.x == 0) bprod[blockIdx.x] = feb_store2<Fp>(lds_get<Fp>(node[1]));
}

// Single block.  MODE 0: root[0] = product of the nb block products (the host inverts it).  MODE 1: binv[b] = 1 / bprod[b] given
// root_inv = 1 / (their product).  MODE 2: both at once, the root inverted by lane 0 (a^(p-2): ~0.4 ms of one lane's time for the
// 381-bit field -- for batches nobody is waiting for).
template <class C, int MODE>
__global__ void __launch_bounds__(kBlock) k_bai_middle(const FePacked<typename C::Fp>* __restrict__ bprod, uint32_t nb, FePacked<typename C::Fp> root_inv,
                                                       FePacked<typename C::Fp>* __restrict__ root, FePacked<typename C::Fp>* __restrict__ binv) {
    using Fp = typename C::Fp;
    __shared__ FeLds<Fp> node[2 * kBlock];
    FeB<Fp, 2> prod = feb_one2<Fp>();
#pragma unroll 1
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) prod = feb_mul(prod, feb_load2<Fp>(bprod[b]));
    lds_put<Fp>(node[kBlock + threadIdx.x], prod);
    bai_tree_up<Fp>(node);
    if (MODE == 0) {
        if (threadIdx.x == 0) root[0] = feb_store2<Fp>(lds_get<Fp>(node[1]));
        return;
    }
    if (threadIdx.x == 0) {
        FeB<Fp, 2> ri;
        if (MODE == 2) ri = feb_widen<2>(feb_from_strict<Fp>(fe_inv<Fp>(feb_to_strict(lds_get<Fp>(node[1])))));
        else ri = feb_load2<Fp>(root_inv);
        lds_put<Fp>(node[1], ri);
    }
    bai_tree_down<Fp>(node);
    // the lane's inverse covers the product of ITS block products b = l, l + 256, ...: walk them from the last to the first
    FeB<Fp, 2> run = lds_get<Fp>(node[kBlock + threadIdx.x]);
    uint32_t cnt = 0;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) cnt++;
#pragma unroll 1
    for (uint32_t k = cnt; k-- > 0;) {
        FeB<Fp, 2> pre = feb_one2<Fp>();                       // product of the lane's values before the k-th (recomputed: cnt <= kBaiMidPer)
#pragma unroll 1
        for (uint32_t k2 = 0; k2 < k; k2++) pre = feb_mul(pre, feb_load2<Fp>(bprod[threadIdx.x + k2 * kBlock]));
        const uint32_t b = threadIdx.x + k * kBlock;
        binv[b] = feb_store2<Fp>(feb_mul(run, pre));
        run = feb_mul(run, feb_load2<Fp>(bprod[b]));
    }
}

template <class C>
__global__ void __launch_bounds__(kBlock) k_bai_finish(const XyzzPacked<C>* __restrict__ in, size_t n, const FePacked<typename C::Fp>* __restrict__ binv,
                                                       AffPacked<C>* __restrict__ out) {
    using Fp = typename C::Fp;
    __shared__ FeLds<Fp> node[2 * kBlock];
    const size_t base = (size_t)blockIdx.x * kBaiTile + threadIdx.x;
    FeB<Fp, 2> z[kBaiPer], pre[kBaiPer];                         // pre[e] = z[0] .. z[e]
#pragma unroll
    for (int e = 0; e < kBaiPer; e++) {
        z[e] = bai_z<C>(in, base + (size_t)e * kBlock, n);
        pre[e] = e == 0 ? z[0] : feb_mul(pre[e - 1], z[e]);
    }
    lds_put<Fp>(node[kBlock + threadIdx.x], pre[kBaiPer - 1]);
    bai_tree_up<Fp>(node);
    if (threadIdx.x == 0) lds_put<Fp>(node[1], feb_load2<Fp>(binv[blockIdx.x]));
    bai_tree_down<Fp>(node);
    FeB<Fp, 2> run = lds_get<Fp>(node[kBlock + threadIdx.x]);    // 1 / (z[0] .. z[3])
#pragma unroll
    for (int e = kBaiPer - 1; e >= 0; e--) {
        const FeB<Fp, 2> zi = e == 0 ? run : feb_mul(run, pre[e - 1]);      // 1 / z[e]
        if (e) run = feb_mul(run, z[e]);
        const size_t i = base + (size_t)e * kBlock;
        if (i >= n) continue;
        const XyzzLazy<C> p = xyzz_lazy_unpack(in[i]);
        Aff<C> a;
        if (p.inf) { a.x = fe_zero<Fp>(); a.y = fe_zero<Fp>(); }
        else {
            a.x = feb_to_strict(feb_mul(p.x, feb_mul(zi, p.zzz)));          // X / ZZ
            a.y = feb_to_strict(feb_mul(p.y, feb_mul(zi, p.zz)));           // Y / ZZZ
        }
        out[i] = aff_pack(a);
    }
}

// ---------------------------------------------------------------------------------------------- digit multiples of XYZZ points
// mult[(m - 1) * n + t] = m S_t, m = 1 .. 8, from packed lazy points (the compacted generators): seven dependent full additions per lane
// through ONE addition site (it handles the empty accumulator and S + S).
template <class C>
__global__ void __launch_bounds__(64) k_digit_table_build_xyzz(const XyzzPacked<C>* __restrict__ S, uint32_t n, uint32_t rows, XyzzPacked<C>* __restrict__ mult) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const XyzzLazy<C> p = xyzz_lazy_unpack(S[t]);
    XyzzLazy<C> acc = xyzz_lazy_inf<C>();
#pragma unroll 1
    for (uint32_t m = 1; m <= rows; m++) {
        acc = xyzz_lazy_add(acc, p);
        mult[(size_t)(m - 1) * n + t] = xyzz_lazy_pack(acc);
    }
}

// ---------------------------------------------------------------------------------------------- window sums of the compaction
// Lane (o, w): output o < 2 nj (o < nj: G'_o over c_G, else H'_(o - nj) over c_H), 4-bit window w < nwin of every sub-scalar:
//     W[w][o] = sum_{t < T} sum_{k < K} digit_(nwin k + w)(c[j + t nj]) * Row_k(P_(j + t nj))
// with the signed digits of c + bias (bias = 0x77..7: digit = nibble - 7 in [-7, 8], as the pipeline recodes), Row_k(P_i) = 2^(4 nwin k) P_i
// and the digit's multiple LOADED from the vector's table (DG for o < nj, DH above): D[(|d| - 1) * drows + k * kstride + i], affine rows.
// K = 1, nwin = 64: plain.  K = 4, nwin = 16: over the rows 2^(64 k) P_i of a precomputed table.
// The loop is k_accumulate's single-path addition; the general one only for the rare doubling / cancellation.
template <class C>
__global__ void __launch_bounds__(kBlock, 2) k_compact_window_sums(const AffPacked<C>* __restrict__ DG, const AffPacked<C>* __restrict__ DH, size_t drowsG, size_t drowsH,
                                                                   size_t kstrideG, size_t kstrideH, const ScalarWords* __restrict__ cG,
                                                                   const ScalarWords* __restrict__ cH, uint32_t n0, uint32_t nj, int lgK, int nwin, ScalarWords bias,
                                                                   uint32_t pitch, XyzzPacked<C>* __restrict__ wsum) {
    using Fp = typename C::Fp;
    const uint32_t o = blockIdx.x * kBlock + threadIdx.x, w = blockIdx.y;
    if (o >= 2 * nj) return;
    const uint32_t vec = o >= nj ? 1u : 0u, j = o - vec * nj;
    const ScalarWords* sc = vec ? cH : cG;
    const AffPacked<C>* D = vec ? DH : DG;                      // rows of this output's vector: D[(m - 1) * drows + k * kstride + i]
    const size_t drows = vec ? drowsH : drowsG, npts = vec ? kstrideH : kstrideG, pbase = j;
    const uint32_t K = 1u << lgK, terms = (n0 / nj) << lgK;
    XyzzLazy<C> acc = xyzz_lazy_inf<C>();
    uint32_t e = 0;
    // term e: t = e >> lgK, k = e & (K - 1)
    auto fetch = [&](uint32_t ee, Aff<C>& p, bool& neg) -> bool {
        const uint32_t t = ee >> lgK, k = ee & (K - 1);
        uint64_t q[4];
        add256(q, sc[j + (size_t)t * nj], bias);
        const int d = (int)window_bits(q, 4 * (int)(nwin * k + w), 4) - 7;
        if (d == 0) return false;
        const uint32_t m = (uint32_t)(d < 0 ? -d : d);
        p = aff_unpack(D[(size_t)(m - 1) * drows + (size_t)k * npts + pbase + (size_t)t * nj]);
        neg = d < 0;
        return !aff_is_inf(p);
    };
    // The accumulator starts as the first term with a non-zero digit, found by a scan of DIGITS only: were that term taken inside the
    // addition loop (k_accumulate's shape), the lanes whose first digit is zero -- one in sixteen -- would sit out the whole inner loop of
    // their wave and then run theirs alone: twice the time (measured: 2.49 ms for 8.4 M additions).
    auto restart = [&]() {
        Aff<C> p; bool neg;
        while (e < terms && !fetch(e, p, neg)) e++;
        if (e < terms) {
            if (neg) p.y = fe_neg(p.y);
            acc = xyzz_lazy_from_strict(xyzz_from_aff(p));
            e++;
        }
    };
    restart();
    while (e < terms) {
        while (e < terms) {
            Aff<C> p; bool neg;
            if (!fetch(e, p, neg)) { e++; continue; }
            FeB<Fp, 2> qy = feb_widen<2>(feb_from_strict<Fp>(p.y));
            if (neg) qy = feb_neg_canonical<Fp>(p.y);
            if (!xyzz_lazy_add_aff_fast(acc, feb_from_strict<Fp>(p.x), qy)) break;     // same x: doubling or cancellation, below
            e++;
        }
        if (e < terms) {
            Aff<C> p; bool neg;
            if (fetch(e, p, neg)) {
                if (neg) p.y = fe_neg(p.y);
                xyzz_lazy_add_aff(acc, p);
            }
            e++;
            if (acc.inf) restart();
        }
    }
    wsum[(size_t)w * pitch + o] = xyzz_lazy_pack(acc);
}

// ---------------------------------------------------------------------------------------------- Horner over the window sums
// S_o = sum_w 16^w W[w][o]: per output a strictly serial chain of doublings -- 4 (nwin - 1) of them, whatever is done -- and additions.
// A quad of lanes shares every operation through two LDS slots (accumulator, incoming window sum), and TWO quads share an output:
// the compaction has only ~2 nj outputs, a quarter of the chip's SIMDs at one wave of 16 quads each, so the upper half of the windows
// (quad 0: Horner, then 4 h more doublings) and the lower half (quad 1) run side by side and meet in one addition -- the critical path
// is 4 (nwin - 1) doublings + nwin / 2 additions instead of nwin - 1.  Blocks of one wave (8 outputs).  No s_barrier: the quads of a
// wave run different trip counts; LDS operations of one wave execute in order, the fence only keeps the compiler from moving them.
// extra (optional): one more affine point appended as S_nout (the Q of the inner-product argument), so that the digit multiples of
// [G' | H' | Q] come out of one array.
constexpr int kHornerQuads = 16;
__device__ __forceinline__ void quad_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class C>
__global__ void __launch_bounds__(4 * kHornerQuads) k_compact_horner(const XyzzPacked<C>* __restrict__ wsum, uint32_t nout, uint32_t pitch, int nwin, int cw,
                                                                     const AffPacked<C>* __restrict__ extra, XyzzPacked<C>* __restrict__ out) {
    using Fp = typename C::Fp;
    constexpr int NW = Fp::NW;
    __shared__ XyzzPacked<C> lds[2 * kHornerQuads];
    const int quad = (int)threadIdx.x >> 2, q = (int)threadIdx.x & 3, seg = quad & 1;
    const uint32_t o = blockIdx.x * (kHornerQuads / 2) + (quad >> 1);
    const uint32_t oo = o < nout ? o : nout - 1;                 // a pair of quads past the end shadows the last output and stores nothing
    const int h = nwin / 2;                                      // quad 0: windows [h, nwin), quad 1: windows [0, h)
    const int w_hi = seg ? h - 1 : nwin - 1, w_lo = seg ? 0 : h;
    uint32_t* A = (uint32_t*)&lds[2 * quad];
    uint32_t* B = (uint32_t*)&lds[2 * quad + 1];
    if (w_hi >= w_lo) {   // lane q moves field q
        const uint32_t* src = (const uint32_t*)&wsum[(size_t)w_hi * pitch + oo];
        for (int i = 0; i < NW; i++) A[q * NW + i] = src[q * NW + i];
    } else {
        for (int i = 0; i < NW; i++) A[q * NW + i] = 0;          // (nwin = 1: no lower half) the identity
    }
    quad_lds_fence();
#pragma unroll 1
    for (int w = w_hi - 1; w >= w_lo; w--) {
#pragma unroll 1
        for (int k = 0; k < cw; k++) {
            xyzz_lazy_dbl_quad<C>(lds, 2 * quad, q);
            quad_lds_fence();
        }
        const uint32_t* src = (const uint32_t*)&wsum[(size_t)w * pitch + oo];
        for (int i = 0; i < NW; i++) B[q * NW + i] = src[q * NW + i];
        quad_lds_fence();
        xyzz_lazy_add_quad<C>(lds, 2 * quad, 2 * quad + 1, q);
        quad_lds_fence();
    }
    if (seg == 0) {
#pragma unroll 1
        for (int k = 0; k < cw * h; k++) {                       // the upper half's weight 2^(cw h)
            xyzz_lazy_dbl_quad<C>(lds, 2 * quad, q);
            quad_lds_fence();
        }
        xyzz_lazy_add_quad<C>(lds, 2 * quad, 2 * quad + 2, q);   // + the lower half (its quad finished long ago: same wave, in order)
        quad_lds_fence();
        if (o < nout) {
            uint32_t* dst = (uint32_t*)&out[o];
            for (int i = 0; i < NW; i++) dst[q * NW + i] = A[q * NW + i];
        }
    }
    if (extra && blockIdx.x == 0 && threadIdx.x == 0) {
        const Aff<C> p = aff_unpack(*extra);
        out[nout] = xyzz_lazy_pack(xyzz_lazy_from_strict(xyzz_from_aff(p)));
    }
}

// ---------------------------------------------------------------------------------------------- GLV: half-length scalars (BLS12-381, BN254)
// s P = s1 P + s2 phi(P) with s1, s2 < 2^128 (bp_curve.cuh: GLV_LAMBDA): everything that is a strictly serial chain over the bits of a
// scalar halves -- the Horner chain of the compaction (252 -> 125 doublings) and the host tail of every round that follows (records
// at bit positions < 130 instead of < 256) -- at the same number of additions: twice the terms, half the windows.
// Decomposed scalars travel as ScalarWords: words 0..3 = s1, words 4..7 = s2.  Windows of the halves: kGlvWin = 26 windows of 5 bits
// (130 bits) with the signed recoding of the pipeline, digit = window(k + bias) - 15 in [-15, 16]: 16 digit multiples per point (the
// rounds after the compaction); the compaction itself: kGlvCWin = 33 windows of 4 bits over the SAME 8 multiples of the originals as
// the plain form.
constexpr int kGlvBits = 5, kGlvWin = 26, kGlvRows = 1 << (kGlvBits - 1);

__device__ __forceinline__ void mul64x64(uint64_t a, uint64_t b, uint64_t& lo, uint64_t& hi) { lo = a * b; hi = __umul64hi(a, b); }
// acc (3 words) += a * b at word offset 0 (carry into the upper words)
__device__ __forceinline__ void mac192(uint64_t (&acc)[3], uint64_t a, uint64_t b) {
    uint64_t lo, hi;
    mul64x64(a, b, lo, hi);
    acc[0] += lo; hi += acc[0] < lo;
    acc[1] += hi; acc[2] += acc[1] < hi;
}

// s (canonical, < r) -> s1 = s mod LAMBDA, s2 = floor(s / LAMBDA): Barrett with m = 2^128 + MLO = floor(2^256 / LAMBDA)
//   q' = floor((s + floor(s MLO / 2^128)) / 2^128) is q or q - 1 (exhaustively at the boundaries + 2e5 random values, Python integers)
// P (6 words) = s (4 words) * (m1 2^64 + m0)
__device__ __forceinline__ void mul256x128(const uint64_t (&s)[4], uint64_t m0, uint64_t m1, uint64_t (&P)[6]) {
    uint64_t acc[3] = {0, 0, 0};
    mac192(acc, s[0], m0);
    P[0] = acc[0]; acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0;
    mac192(acc, s[0], m1); mac192(acc, s[1], m0);
    P[1] = acc[0]; acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0;
    mac192(acc, s[1], m1); mac192(acc, s[2], m0);
    P[2] = acc[0]; acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0;
    mac192(acc, s[2], m1); mac192(acc, s[3], m0);
    P[3] = acc[0]; acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0;
    mac192(acc, s[3], m1);
    P[4] = acc[0]; P[5] = acc[1];
}
// (lo, hi) = low 128 bits of (a1 2^64 + a0) * (b1 2^64 + b0)
__device__ __forceinline__ void mul128lo(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1, uint64_t& lo, uint64_t& hi) {
    mul64x64(a0, b0, lo, hi);
    hi += a0 * b1 + a1 * b0;
}

// The lattice form (BN254, bp_curve.cuh): s1 in [0, 2^128), s2 mod 2^128 (negative when >= 2^66)
template <class C>
__device__ __forceinline__ void glv_decompose_lattice(const uint64_t (&s)[4], uint64_t (&s1)[2], uint64_t (&s2)[2]) {
    uint64_t P[6];
    mul256x128(s, C::GLV_M1[0], C::GLV_M1[1], P);
    const uint64_t e1 = (P[4] >> 61) | (P[5] << 3);                                      // >> 317
    mul256x128(s, C::GLV_M2[0], C::GLV_M2[1], P);
    const uint64_t e20 = (P[3] >> 62) | (P[4] << 2), e21 = (P[4] >> 62) | (P[5] << 2);   // >> 254
    uint64_t x0, x1, y0, y1;
    // s2 = e1 B - e2 A  (mod 2^128)
    mul128lo(e1, 0, C::GLV_B[0], C::GLV_B[1], x0, x1);
    mul128lo(e20, e21, C::GLV_A, 0, y0, y1);
    s2[0] = x0 - y0;
    s2[1] = x1 - y1 - (uint64_t)(x0 < y0);
    // s1 = s - e1 A - e2 C  (mod 2^128; the true value is in [0, 2^128))
    mul128lo(e1, 0, C::GLV_A, 0, x0, x1);
    mul128lo(e20, e21, C::GLV_C[0], C::GLV_C[1], y0, y1);
    uint64_t t0 = s[0] - x0, t1 = s[1] - x1 - (uint64_t)(s[0] < x0);
    s1[0] = t0 - y0;
    s1[1] = t1 - y1 - (uint64_t)(t0 < y0);
}

template <class C>
__device__ __forceinline__ void glv_decompose(const ScalarWords& sw, uint64_t (&s1)[2], uint64_t (&s2)[2]) {
    const uint64_t s[4] = {sw.w[0] | ((uint64_t)sw.w[1] << 32), sw.w[2] | ((uint64_t)sw.w[3] << 32), sw.w[4] | ((uint64_t)sw.w[5] << 32),
                           sw.w[6] | ((uint64_t)sw.w[7] << 32)};
    if constexpr (C::GLV_SIGNED) { glv_decompose_lattice<C>(s, s1, s2); return; }
    const uint64_t m0 = C::GLV_MLO[0], m1 = C::GLV_MLO[1], l0 = C::GLV_LAMBDA[0], l1 = C::GLV_LAMBDA[1];
    // t = (s * MLO) >> 128: columns of the 4 x 2 word product, three-word running accumulator
    uint64_t acc[3] = {0, 0, 0}, t[4];
    mac192(acc, s[0], m0);                                   // column 0
    acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0;
    mac192(acc, s[0], m1); mac192(acc, s[1], m0);            // column 1
    acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0;
    mac192(acc, s[1], m1); mac192(acc, s[2], m0);            // column 2 -> t[0]
    t[0] = acc[0]; acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0;
    mac192(acc, s[2], m1); mac192(acc, s[3], m0);            // column 3 -> t[1]
    t[1] = acc[0]; acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0;
    mac192(acc, s[3], m1);                                   // column 4 -> t[2], t[3]
    t[2] = acc[0]; t[3] = acc[1];
    // u = s + t (< 2^256); q = u >> 128
    uint64_t u[4], cy = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const uint64_t a = s[i] + t[i], b = a + cy; cy = (uint64_t)(a < s[i]) + (uint64_t)(b < a); u[i] = b; }
    uint64_t q0 = u[2], q1 = u[3];
    // rem = s - q * LAMBDA (fits three words with room: < 2 LAMBDA)
    uint64_t p[3] = {0, 0, 0};
    mac192(p, q0, l0);
    { uint64_t hi[3] = {0, 0, 0}; mac192(hi, q0, l1); mac192(hi, q1, l0); p[1] += hi[0]; p[2] += hi[1] + (p[1] < hi[0]); }
    p[2] += q1 * l1;
    uint64_t r0 = s[0] - p[0], b0 = s[0] < p[0];
    uint64_t r1 = s[1] - p[1] - b0, b1 = (s[1] < p[1]) || (s[1] == p[1] && b0);
    uint64_t r2 = s[2] - p[2] - b1;
    // one conditional correction: rem >= LAMBDA  <=>  r2 != 0 or (r1, r0) >= (l1, l0)
    const bool ge = r2 != 0 || r1 > l1 || (r1 == l1 && r0 >= l0);
    if (ge) {
        const uint64_t n0 = r0 - l0, bb = r0 < l0;
        r1 = r1 - l1 - bb; r0 = n0;
        q0 += 1; q1 += q0 == 0;
    }
    s1[0] = r0; s1[1] = r1; s2[0] = q0; s2[1] = q1;
}

// dst[i] = the two halves of a canonical scalar (the store of k_ipp_round_prep / _final<C, true>, bp_ipp.cuh)
template <class C> __device__ __forceinline__ void glv_split_store(ScalarWords* dst, size_t i, const ScalarWords& canonical) {
    uint64_t a[2], b[2];
    glv_decompose<C>(canonical, a, b);
    ScalarWords o;
    o.w[0] = (uint32_t)a[0]; o.w[1] = (uint32_t)(a[0] >> 32); o.w[2] = (uint32_t)a[1]; o.w[3] = (uint32_t)(a[1] >> 32);
    o.w[4] = (uint32_t)b[0]; o.w[5] = (uint32_t)(b[0] >> 32); o.w[6] = (uint32_t)b[1]; o.w[7] = (uint32_t)(b[1] >> 32);
    dst[i] = o;
}

// out1[i] = split(in1[i]), out2[i] = split(in2[i]) (in place allowed); in2 == nullptr: one vector
template <class C>
__global__ void __launch_bounds__(kBlock) k_glv_decompose(const ScalarWords* in1, const ScalarWords* in2, size_t n, ScalarWords* out1, ScalarWords* out2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll 1
    for (int v = 0; v < (in2 ? 2 : 1); v++) {
        uint64_t a[2], b[2];
        glv_decompose<C>((v ? in2 : in1)[i], a, b);
        ScalarWords o;
        o.w[0] = (uint32_t)a[0]; o.w[1] = (uint32_t)(a[0] >> 32); o.w[2] = (uint32_t)a[1]; o.w[3] = (uint32_t)(a[1] >> 32);
        o.w[4] = (uint32_t)b[0]; o.w[5] = (uint32_t)(b[0] >> 32); o.w[6] = (uint32_t)b[1]; o.w[7] = (uint32_t)(b[1] >> 32);
        (v ? out2 : out1)[i] = o;
    }
}

// signed BITS-bit digit w of a 128-bit half (lo, hi): window w of (half + bias), bias = sum_w (2^(BITS-1) - 1) 2^(BITS w) over NWIN windows
// (NWIN * BITS >= 130 bits, so the sum never overflows its windows); in [-(2^(BITS-1) - 1), 2^(BITS-1)].  w is wave-uniform.
template <int BITS, int NWIN>
__device__ __forceinline__ int glv_digit(uint64_t lo, uint64_t hi, int w) {
    static_assert(BITS * NWIN >= 130 && BITS * NWIN <= 192, "three words");
    constexpr uint64_t half1 = (1u << (BITS - 1)) - 1;
    // bias words at compile time
    uint64_t b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
    for (int i = 0; i < NWIN; i++) {
        const int off = BITS * i;
        if (off < 64) { b0 |= half1 << off; if (off + BITS > 64) b1 |= half1 >> (64 - off); }
        else if (off < 128) { b1 |= half1 << (off - 64); if (off + BITS > 128) b2 |= half1 >> (128 - off); }
        else b2 |= half1 << (off - 128);
    }
    uint64_t k[3];
    k[0] = lo + b0;
    const uint64_t c0 = k[0] < lo;
    const uint64_t a1 = hi + b1;
    k[1] = a1 + c0;
    k[2] = b2 + (uint64_t)(a1 < hi) + (uint64_t)(k[1] < a1);
    const int off = BITS * w, word = off >> 6, sh = off & 63;
    const uint64_t x = word == 0 ? k[0] : word == 1 ? k[1] : k[2], y = word == 0 ? k[1] : word == 1 ? k[2] : 0;
    const uint64_t v = sh ? (x >> sh) | (y << (64 - sh)) : x;
    return (int)(v & ((1u << BITS) - 1)) - (int)half1;
}
// the 128-bit half `kind` of a decomposed scalar, one 16-byte load (kind is block-uniform everywhere: no per-lane selection)
__device__ __forceinline__ void glv_half(const ScalarWords* sc, size_t i, int kind, uint64_t& lo, uint64_t& hi) {
    const uint4 h = reinterpret_cast<const uint4*>(sc + i)[kind];
    lo = h.x | ((uint64_t)h.y << 32);
    hi = h.z | ((uint64_t)h.w << 32);
}

// ... and its sign: BLS12-381's halves are plain numbers; BN254's second half (kind 1) is a number mod 2^128 that is negative when
// >= 2^66 (bp_curve.cuh) -- the lane then works with the magnitude and negates its digit.  kind is block-uniform.
template <class C> __device__ __forceinline__ bool glv_half_signed(const ScalarWords* sc, size_t i, int kind, uint64_t& lo, uint64_t& hi) {
    glv_half(sc, i, kind, lo, hi);
    if constexpr (C::GLV_SIGNED) {
        if (kind == 1 && (hi >> 2) != 0) {
            lo = 0 - lo;
            hi = ~hi + (uint64_t)(lo == 0);
            return true;
        }
    }
    return false;
}

template <class C> __device__ __forceinline__ Fe<typename C::Fp> glv_beta_mont() { return fe_to_mont<typename C::Fp>(fe_unpack_words<typename C::Fp>(C::BETA)); }
// phi of a lazy XYZZ point: (BETA X, Y, ZZ, ZZZ) -- phi is a group homomorphism, so it is applied ONCE to a finished partial sum of
// phi-side terms instead of to every term
template <class C> __device__ __forceinline__ void xyzz_lazy_phi(XyzzLazy<C>& a) {
    using Fp = typename C::Fp;
    if (a.inf) return;
    a.x = feb_widen<8>(feb_mul(a.x, feb_from_strict<Fp>(glv_beta_mont<C>())));
}

// Window sums of the compaction over HALF-LENGTH scalars: lane (o, w, kind) -- output o, window w of the kGlvCWin 4-bit windows of a
// half, kind 0: the s1 halves over P, kind 1: the s2 halves over the same rows of D, with phi applied to the finished sum --
//     part[(kind * kGlvCWin + w) * pitch + o] = [phi] sum_t digit_w(half_kind(c_t)) P_t           (T mixed additions per lane)
// D as in k_compact_window_sums (8 multiples, K = 1); cG / cH arrive DECOMPOSED (k_glv_decompose).  k_compact_merge_halves adds the
// two parts of every (o, w).
constexpr int kGlvCBits = 4, kGlvCWin = 33;
template <class C>
__global__ void __launch_bounds__(kBlock, 2) k_compact_window_sums_glv(const AffPacked<C>* __restrict__ DG, const AffPacked<C>* __restrict__ DH, size_t drows,
                                                                       const ScalarWords* __restrict__ cG, const ScalarWords* __restrict__ cH, uint32_t n0, uint32_t nj,
                                                                       uint32_t pitch, XyzzPacked<C>* __restrict__ part) {
    using Fp = typename C::Fp;
    const uint32_t o = blockIdx.x * kBlock + threadIdx.x, w = blockIdx.y, kind = blockIdx.z;
    if (o >= 2 * nj) return;
    const uint32_t vec = o >= nj ? 1u : 0u, j = o - vec * nj;
    const ScalarWords* sc = vec ? cH : cG;
    const AffPacked<C>* D = vec ? DH : DG;
    const uint32_t terms = n0 / nj;
    XyzzLazy<C> acc = xyzz_lazy_inf<C>();
    uint32_t e = 0;
    auto fetch = [&](uint32_t t, Aff<C>& p, bool& neg) -> bool {
        uint64_t lo, hi;
        const bool flip = glv_half_signed<C>(sc, j + (size_t)t * nj, (int)kind, lo, hi);
        int d = glv_digit<kGlvCBits, kGlvCWin>(lo, hi, (int)w);
        if (flip) d = -d;
        if (d == 0) return false;
        p = aff_unpack(D[(size_t)((d < 0 ? -d : d) - 1) * drows + j + (size_t)t * nj]);
        neg = d < 0;
        return !aff_is_inf(p);
    };
    auto restart = [&]() {
        Aff<C> p; bool neg;
        while (e < terms && !fetch(e, p, neg)) e++;
        if (e < terms) {
            if (neg) p.y = fe_neg(p.y);
            acc = xyzz_lazy_from_strict(xyzz_from_aff(p));
            e++;
        }
    };
    restart();
    while (e < terms) {
        while (e < terms) {
            Aff<C> p; bool neg;
            if (!fetch(e, p, neg)) { e++; continue; }
            FeB<Fp, 2> qy = feb_widen<2>(feb_from_strict<Fp>(p.y));
            if (neg) qy = feb_neg_canonical<Fp>(p.y);
            if (!xyzz_lazy_add_aff_fast(acc, feb_from_strict<Fp>(p.x), qy)) break;
            e++;
        }
        if (e < terms) {
            Aff<C> p; bool neg;
            if (fetch(e, p, neg)) {
                if (neg) p.y = fe_neg(p.y);
                xyzz_lazy_add_aff(acc, p);
            }
            e++;
            if (acc.inf) restart();
        }
    }
    if (kind) xyzz_lazy_phi<C>(acc);
    part[((size_t)kind * kGlvCWin + w) * pitch + o] = xyzz_lazy_pack(acc);
}

// wsum[w * pitch + o] = part[w * pitch + o] + part[(nwin + w) * pitch + o]     (one full addition per lane)
template <class C>
__global__ void __launch_bounds__(kBlock) k_compact_merge_halves(const XyzzPacked<C>* __restrict__ part, uint32_t nout, uint32_t pitch, int nwin, XyzzPacked<C>* __restrict__ wsum) {
    const uint32_t o = blockIdx.x * kBlock + threadIdx.x, w = blockIdx.y;
    if (o >= nout) return;
    const XyzzLazy<C> a = xyzz_lazy_unpack(part[(size_t)w * pitch + o]), b = xyzz_lazy_unpack(part[((size_t)nwin + w) * pitch + o]);
    wsum[(size_t)w * pitch + o] = xyzz_lazy_pack(xyzz_lazy_add(a, b));
}

// One round of the inner-product argument over a compacted, GLV-split generator set: block (w, set, z) with z = (split, kind) sums the
// digit-w terms of its share of the n terms for ONE half of the scalars (kind = z & 1) -- the digit's multiple is LOADED from
// mult[(m - 1) * n + t] = m P_t (affine, 16 multiples) for both kinds; phi is applied once to the block's finished sum of a kind-1 block --
// and leaves one record at bit position 5 w: window_sum[(set * kGlvWin + w) * gridDim.z + z].  sc1 / sc2: the round's two scalar sets,
// decomposed.  gridDim.z is even.
// XROWS: `mult` holds packed lazy XYZZ rows instead of affine ones (small proofs: the batch inversion that makes rows affine costs a host
// round trip, more than the mixed additions save when a lane has one or two terms) -- full additions, as k_small_msm<C, 1>.
template <class C, bool XROWS>
__global__ void __launch_bounds__(kBlock) k_small_msm_glv(const ScalarWords* __restrict__ sc1, const ScalarWords* __restrict__ sc2, uint32_t n_all,
                                                          const void* __restrict__ mult_rows, XyzzPacked<C>* __restrict__ window_sum, IppSparse sp) {
    using Fp = typename C::Fp;
    __shared__ XyzzPacked<C> lds[kBlock];
    const int w = blockIdx.x, set = blockIdx.y, kind = blockIdx.z & 1;
    const ScalarWords* sc = set ? sc2 : sc1;
    const uint32_t stride = kBlock * (gridDim.z >> 1);
    const uint32_t n = sp.live ? sp.n0 + 1 : n_all;             // the lanes walk the terms that can be non-zero (IppSparse, bp_kernels.cuh)
    uint32_t t = (blockIdx.z >> 1) * kBlock + threadIdx.x;
    XyzzLazy<C> mine = xyzz_lazy_inf<C>();
    if constexpr (XROWS) {
        const XyzzPacked<C>* mult = (const XyzzPacked<C>*)mult_rows;
#pragma unroll 1
        for (; t < n; t += stride) {
            const uint32_t tt = ipp_term(sp, set, t);
            uint64_t lo, hi;
            const bool flip = glv_half_signed<C>(sc, tt, kind, lo, hi);
            int d = glv_digit<kGlvBits, kGlvWin>(lo, hi, w);
            if (flip) d = -d;
            if (d == 0) continue;
            XyzzLazy<C> acc = xyzz_lazy_unpack(mult[(size_t)((d < 0 ? -d : d) - 1) * n_all + tt]);
            if (d < 0 && !acc.inf) acc.y = feb_neg<4>(acc.y);
            mine = xyzz_lazy_add(mine, acc);
        }
        const uint32_t per = (n + (gridDim.z >> 1) - 1) / (gridDim.z >> 1);      // lanes of this block that can hold a term
        mine = block_tree_sum_quad<C>(mine, lds, per < (uint32_t)kBlock ? (int)per : kBlock);
        if (threadIdx.x == 0) {
            if (kind) xyzz_lazy_phi<C>(mine);
            window_sum[((size_t)set * kGlvWin + w) * gridDim.z + blockIdx.z] = xyzz_lazy_pack(mine);
        }
        return;
    }
    const AffPacked<C>* mult = (const AffPacked<C>*)mult_rows;
    auto fetch = [&](uint32_t e, Aff<C>& p, bool& neg) -> bool {
        const uint32_t tt = ipp_term(sp, set, e);
        uint64_t lo, hi;
        const bool flip = glv_half_signed<C>(sc, tt, kind, lo, hi);
        int d = glv_digit<kGlvBits, kGlvWin>(lo, hi, w);
        if (flip) d = -d;
        if (d == 0) return false;
        p = aff_unpack(mult[(size_t)((d < 0 ? -d : d) - 1) * n_all + tt]);
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
    mine = block_tree_sum_quad<C>(mine, lds, kBlock);
    if (threadIdx.x == 0) {
        if (kind) xyzz_lazy_phi<C>(mine);
        window_sum[((size_t)set * kGlvWin + w) * gridDim.z + blockIdx.z] = xyzz_lazy_pack(mine);
    }
}

// v[i] = 1 (canonical) for i < n: the coefficient vectors of a freshly compacted generator set
static __global__ void __launch_bounds__(kBlock) k_fr_fill_one(ScalarWords* __restrict__ a, ScalarWords* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ScalarWords one;
    one.w[0] = 1;
    for (int k = 1; k < 8; k++) one.w[k] = 0;
    a[i] = one;
    b[i] = one;
}

}  // namespace bp
