// Latency of ONE dependent chain of point operations per lane, at 1 and 2 waves per SIMD: what the latency-bound tails of the
// MSM (bucket reduce, small-MSM trees, IPP folds) pay per step.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../bulletproofs-amcl_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "bp_curve.cuh"
using namespace bp;
using C = Bls381;
constexpr int N = 64;

namespace bp {
// Experiment (round 2): two INDEPENDENT products with their column chains interleaved instruction by instruction (the pinned order
// keeps them so), to give a lone wave instruction-level parallelism.  Measured: no difference at any occupancy (14.45 vs 14.46 us per
// 14 products at one wave per SIMD) -- a lone wave already issues its mad chain at ~4.8 cycles per instruction, the half-rate limit.
template <class P, int B1, int B2, int B3, int B4>
BP_HD void feb_mul2(const FeB<P, B1>& a1, const FeB<P, B2>& b1, const FeB<P, B3>& a2, const FeB<P, B4>& b2, FeB<P, 2>& r1, FeB<P, 2>& r2) {
    static_assert(B1 * B2 <= kMaxProd && B3 * B4 <= kMaxProd, "operands too large for a lazy Montgomery product");
    constexpr int N = P::NL;
    uint32_t t1[2 * N], t2[2 * N];
    uint64_t acc1 = 0, acc2 = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); i++) {
            acc1 += (uint64_t)a1.v[i] * b1.v[k - i]; BP_KEEP_ORDER(acc1);
            acc2 += (uint64_t)a2.v[i] * b2.v[k - i]; BP_KEEP_ORDER(acc2);
        }
        t1[k] = (uint32_t)acc1 & LMASK; acc1 >>= LB;
        t2[k] = (uint32_t)acc2 & LMASK; acc2 >>= LB;
    }
    t1[2 * N - 1] = (uint32_t)acc1;
    t2[2 * N - 1] = (uint32_t)acc2;
    uint32_t m1[N], m2[N];
    uint32_t one = 1;
    BP_OPAQUE_ONE(one);
    acc1 = 0; acc2 = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        acc1 += (uint64_t)t1[k] * one; BP_KEEP_ORDER(acc1);
        acc2 += (uint64_t)t2[k] * one; BP_KEEP_ORDER(acc2);
#pragma unroll
        for (int i = 0; i < k; i++) {
            acc1 += (uint64_t)m1[i] * P::C.mod[k - i]; BP_KEEP_ORDER(acc1);
            acc2 += (uint64_t)m2[i] * P::C.mod[k - i]; BP_KEEP_ORDER(acc2);
        }
        m1[k] = ((uint32_t)acc1 * P::C.inv) & LMASK;
        m2[k] = ((uint32_t)acc2 * P::C.inv) & LMASK;
        acc1 += (uint64_t)m1[k] * P::C.mod[0]; BP_KEEP_ORDER(acc1);
        acc2 += (uint64_t)m2[k] * P::C.mod[0]; BP_KEEP_ORDER(acc2);
        acc1 >>= LB; acc2 >>= LB;
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
        acc1 += (uint64_t)t1[k] * one; BP_KEEP_ORDER(acc1);
        acc2 += (uint64_t)t2[k] * one; BP_KEEP_ORDER(acc2);
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            acc1 += (uint64_t)m1[i] * P::C.mod[k - i]; BP_KEEP_ORDER(acc1);
            acc2 += (uint64_t)m2[i] * P::C.mod[k - i]; BP_KEEP_ORDER(acc2);
        }
        r1.v[k - N] = (uint32_t)acc1 & LMASK; acc1 >>= LB;
        r2.v[k - N] = (uint32_t)acc2 & LMASK; acc2 >>= LB;
    }
}

}  // namespace bp

template <int MODE>
__global__ void __launch_bounds__(256) k_chain(const AffPacked<C>* pts, XyzzPacked<C>* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Aff<C> p = aff_unpack(pts[i & 1023]), q = aff_unpack(pts[(i + 7) & 1023]);
    if (MODE == 0) {            // strict full addition
        Xyzz<C> a = xyzz_from_aff(p), b = xyzz_dbl_aff(q);
        for (int k = 0; k < N; k++) a = xyzz_add(a, b);
        out[i] = xyzz_pack(a);
    } else if (MODE == 1) {     // lazy full addition
        XyzzLazy<C> a = xyzz_lazy_from_strict(xyzz_from_aff(p)), b = xyzz_lazy_from_strict(xyzz_dbl_aff(q));
        for (int k = 0; k < N; k++) a = xyzz_lazy_add(a, b);
        out[i] = xyzz_lazy_pack(a);
    } else if (MODE == 2) {     // lazy doubling
        XyzzLazy<C> a = xyzz_lazy_from_strict(xyzz_from_aff(p));
        for (int k = 0; k < N; k++) a = xyzz_lazy_dbl(a);
        out[i] = xyzz_lazy_pack(a);
    } else if (MODE == 3) {     // lazy mixed addition
        XyzzLazy<C> a = xyzz_lazy_from_strict(xyzz_from_aff(p));
        for (int k = 0; k < N; k++) xyzz_lazy_add_aff(a, q);
        out[i] = xyzz_lazy_pack(a);
    } else if (MODE == 5) {     // two independent multiplication chains, interleaved (feb_mul2): 2 x 7 products per step
        FeB<C::Fp, 2> x = feb_widen<2>(feb_from_strict<C::Fp>(p.x)), y = feb_widen<2>(feb_from_strict<C::Fp>(q.y)), x2 = y, y2 = x;
        for (int k = 0; k < N * 7; k++) feb_mul2(x, y, x2, y2, x, x2);
        XyzzLazy<C> a = xyzz_lazy_inf<C>();
        a.inf = false; a.x = feb_widen<8>(x); a.zz = x2;
        out[i] = xyzz_lazy_pack(a);
    } else {                    // bare multiplications (dependent chain of 14 per step)
        FeB<C::Fp, 2> x = feb_widen<2>(feb_from_strict<C::Fp>(p.x)), y = feb_widen<2>(feb_from_strict<C::Fp>(q.y));
        for (int k = 0; k < N * 14; k++) x = feb_mul(x, y);
        XyzzLazy<C> a = xyzz_lazy_inf<C>();
        a.inf = false; a.x = feb_widen<8>(x); a.zz = y;
        out[i] = xyzz_lazy_pack(a);
    }
}

// tree-like step: partner from LDS -> add -> barrier -> result to LDS -> barrier  (the shape of the reduce kernels)
template <class M, bool BARRIERS>
__global__ void __launch_bounds__(256) k_tree(const AffPacked<C>* pts, XyzzPacked<C>* out) {
    __shared__ XyzzPacked<C> lds[256];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Aff<C> p = aff_unpack(pts[i & 1023]), q = aff_unpack(pts[(i + 7) & 1023]);
    XyzzLazy<C> a = xyzz_lazy_from_strict(xyzz_from_aff(p));
    lds[threadIdx.x] = xyzz_lazy_pack(xyzz_lazy_from_strict(xyzz_dbl_aff(q)));
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < N; k++) {
        a = xyzz_lazy_add<C, M>(a, xyzz_lazy_unpack(lds[(threadIdx.x + 1 + (k & 3)) & 255]));
        if (BARRIERS) {
            __syncthreads();
            lds[threadIdx.x] = xyzz_lazy_pack(a);
            __syncthreads();
        }
    }
    out[i] = xyzz_lazy_pack(a);
}

__global__ void k_gen(AffPacked<C>* pts) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k[8] = {i * 2654435761u + 12345u, i + 99u, 7u, 0, 0, 0, 0, 0};
    pts[i] = aff_pack(xyzz_to_aff<C>(xyzz_mul_words<C>(k, generator<C>())));
}

int main() {
    AffPacked<C>* pts; XyzzPacked<C>* out;
    hipMalloc(&pts, 1024 * sizeof *pts); hipMalloc(&out, 1024 * 256 * sizeof *out);
    hipLaunchKernelGGL(k_gen, dim3(4), dim3(256), 0, 0, pts);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[6] = {"strict add (12M+2S)", "lazy add (12M+2S)", "lazy dbl (6M+3S)", "lazy mixed add (8M+2S)", "14 dependent feb_mul", "7 x feb_mul2 (2 chains)"};
    for (int blocks : {256, 512, 1024}) {
        printf("--- %d waves/SIMD (%d blocks x 256 threads), %d dependent steps per lane\n", blocks / 256, blocks, N);
        for (int mode = 0; mode < 6; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, pts, out); break;
                    case 1: hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, pts, out); break;
                    case 2: hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(256), 0, 0, pts, out); break;
                    case 3: hipLaunchKernelGGL(k_chain<3>, dim3(blocks), dim3(256), 0, 0, pts, out); break;
                    case 4: hipLaunchKernelGGL(k_chain<4>, dim3(blocks), dim3(256), 0, 0, pts, out); break;
                    default: hipLaunchKernelGGL(k_chain<5>, dim3(blocks), dim3(256), 0, 0, pts, out); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%-26s %8.3f ms   %7.2f us per step   %6.3f us per Fp product-equivalent\n", names[mode], best, best * 1e3 / N,
                   best * 1e3 / N / (mode == 2 ? 8.3 : mode == 3 ? 9.6 : 13.6));
        }
    }
    printf("--- tree-shaped steps (LDS partner, 2 barriers), 1 wave/SIMD\n");
    for (int v = 0; v < 4; v++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            switch (v) {
                case 0: hipLaunchKernelGGL((k_tree<MulInline, false>), dim3(256), dim3(256), 0, 0, pts, out); break;
                case 1: hipLaunchKernelGGL((k_tree<MulInline, true>), dim3(256), dim3(256), 0, 0, pts, out); break;
                case 2: hipLaunchKernelGGL((k_tree<MulCall, false>), dim3(256), dim3(256), 0, 0, pts, out); break;
                default: hipLaunchKernelGGL((k_tree<MulCall, true>), dim3(256), dim3(256), 0, 0, pts, out); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        static const char* vn[4] = {"inline, LDS partner", "inline, LDS + barriers", "call, LDS partner", "call, LDS + barriers"};
        printf("%-26s %8.3f ms   %7.2f us per step\n", vn[v], best, best * 1e3 / N);
    }
    // few blocks (64) as in k_digit_weighted
    for (int v = 0; v < 2; v++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL((k_tree<MulInline, true>), dim3(64), dim3(256), 0, 0, pts, out);
            else hipLaunchKernelGGL((k_tree<MulCall, true>), dim3(64), dim3(256), 0, 0, pts, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("64 blocks, %-15s %8.3f ms   %7.2f us per step\n", v ? "call" : "inline", best, best * 1e3 / N);
    }
    return 0;
}
