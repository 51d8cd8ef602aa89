// bp_ipp.cuh -- gfx950 kernels for the per-round vector work of the inner-product argument.
//
// Replaces (device side of) /root/reference src/ipp.rs:
//   FieldElementVector::inner_product      :77-78, :145-146         -> k_fr_inner / k_fr_inner_final
//   hadamard_product with G/H factors      :81-82, :94-95           -> fused into k_ipp_pack_round (first round)
//   L / R term assembly                    :80-104, :148-170        -> k_ipp_pack_round (then the MSM pipeline)
//   scalar fold + binary_scalar_mul fold   :115-130, :181-188       -> k_ipp_fold  (the reference's dominant CPU cost)
//   verification scalars / MSM terms       :220-249, :303-312       -> k_ipp_verify_terms
// Scalars (Fr) live in HBM as canonical 8-word values.  A Montgomery product of a canonical x with a constant held
// in Montgomery form (u*R) is the canonical product x*u, so folds need no domain conversion.
#pragma once
#include "bp_kernels.cuh"

namespace bp {

template <class F>
__device__ __forceinline__ Fe<F> fr_load(const ScalarWords* v, size_t i) {
    ScalarWords s = v[i];
    return fe_unpack_words<F>(s.w);
}
template <class F>
__device__ __forceinline__ void fr_store(ScalarWords* v, size_t i, const Fe<F>& x) {
    ScalarWords s;
    fe_pack_words<F>(s.w, x);
    v[i] = s;
}

// block-level sum of one Fr value per thread; result valid in thread 0
template <class F>
__device__ __forceinline__ Fe<F> block_fr_sum(Fe<F> mine, ScalarWords* lds) {
    fe_pack_words<F>(lds[threadIdx.x].w, mine);
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            mine = fe_add(mine, fe_unpack_words<F>(lds[threadIdx.x + s].w));
            fe_pack_words<F>(lds[threadIdx.x].w, mine);
        }
        __syncthreads();
    }
    return mine;
}

// partial[blockIdx.x] = sum_i a[i] * b[i] / R over a grid-stride slice (Montgomery-scaled; fixed by the final kernel)
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_inner(const ScalarWords* __restrict__ a, const ScalarWords* __restrict__ b, size_t n,
                                                     ScalarWords* __restrict__ partial) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    Fe<F> acc = fe_zero<F>();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc = fe_add(acc, fe_mul(fr_load<F>(a, i), fr_load<F>(b, i)));
    acc = block_fr_sum<F>(acc, lds);
    if (threadIdx.x == 0) fr_store<F>(partial, blockIdx.x, acc);
}

// single block: out[0] = (sum of m partials) * R   (undo the 1/R of the Montgomery products)
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_inner_final(const ScalarWords* __restrict__ partial, uint32_t m, ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    Fe<F> acc = fe_zero<F>();
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) acc = fe_add(acc, fr_load<F>(partial, i));
    acc = block_fr_sum<F>(acc, lds);
    if (threadIdx.x == 0) fr_store<F>(out, 0, fe_to_mont<F>(acc));
}

// Two inner products of the same length in one launch pair (c_L = <a_lo, b_hi> and c_R = <a_hi, b_lo> of an IPP round,
// src/ipp.rs:77-78): blockIdx.y selects the pair; partial holds gridDim.x sums per product; out[0], out[1].
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_inner2(const ScalarWords* __restrict__ a0, const ScalarWords* __restrict__ b0, const ScalarWords* __restrict__ a1,
                                                      const ScalarWords* __restrict__ b1, size_t n, ScalarWords* __restrict__ partial) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    const ScalarWords* a = blockIdx.y ? a1 : a0;
    const ScalarWords* b = blockIdx.y ? b1 : b0;
    Fe<F> acc = fe_zero<F>();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc = fe_add(acc, fe_mul(fr_load<F>(a, i), fr_load<F>(b, i)));
    acc = block_fr_sum<F>(acc, lds);
    if (threadIdx.x == 0) fr_store<F>(partial, (size_t)blockIdx.y * gridDim.x + blockIdx.x, acc);
}
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_inner2_final(const ScalarWords* __restrict__ partial, uint32_t m, ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    Fe<F> acc = fe_zero<F>();
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) acc = fe_add(acc, fr_load<F>(partial, (size_t)blockIdx.x * m + i));
    acc = block_fr_sum<F>(acc, lds);
    if (threadIdx.x == 0) fr_store<F>(out, blockIdx.x, fe_to_mont<F>(acc));
}

// out[i] = a[i] * b[i]   (FieldElementVector::hadamard_product)
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_hadamard(const ScalarWords* __restrict__ a, const ScalarWords* __restrict__ b, size_t n,
                                                        ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store<F>(out, i, fe_to_mont<F>(fe_mul(fr_load<F>(a, i), fr_load<F>(b, i))));
}

// out[i] = a[i] * s   (FieldElementVector::scaled_by); s_mont = s in Montgomery form
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_scale(const ScalarWords* __restrict__ a, ScalarWords s_mont, size_t n, ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store<F>(out, i, fe_mul(fr_load<F>(a, i), fe_unpack_words<F>(s_mont.w)));
}

// out[i] = e^i   (FieldElementVector::new_vandermonde_vector); one lane per element, square-and-multiply on i
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_vandermonde(ScalarWords e_mont, size_t n, ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> base = fe_unpack_words<F>(e_mont.w), acc = fe_one<F>();
    for (size_t k = i; k; k >>= 1) {
        if (k & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
    }
    fr_store<F>(out, i, fe_from_mont<F>(acc));
}

// ---------------------------------------------------------------------------------------------- vector polynomials
// /root/reference src/utils/vector_poly.rs (R1CS t(x), l(x), r(x)).
// VecPoly3::special_inner_product (:79-97): lhs.0 == 0 and rhs.2 == 0 by construction; the nine inner products are
// fused into one pass over the six vectors that matter.  partial[(blockIdx.x * 6 + j)] = block sum of t_(j+1) / R.
template <class C>
__global__ void __launch_bounds__(kBlock) k_vecpoly3_special(const ScalarWords* __restrict__ l1, const ScalarWords* __restrict__ l2,
                                                             const ScalarWords* __restrict__ l3, const ScalarWords* __restrict__ r0,
                                                             const ScalarWords* __restrict__ r1, const ScalarWords* __restrict__ r3, size_t n,
                                                             ScalarWords* __restrict__ partial) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    Fe<F> t[6];
    for (int j = 0; j < 6; j++) t[j] = fe_zero<F>();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fe<F> a1 = fr_load<F>(l1, i), a2 = fr_load<F>(l2, i), a3 = fr_load<F>(l3, i);
        Fe<F> b0 = fr_load<F>(r0, i), b1 = fr_load<F>(r1, i), b3 = fr_load<F>(r3, i);
        t[0] = fe_add(t[0], fe_mul(a1, b0));
        t[1] = fe_add(t[1], fe_add(fe_mul(a1, b1), fe_mul(a2, b0)));
        t[2] = fe_add(t[2], fe_add(fe_mul(a2, b1), fe_mul(a3, b0)));
        t[3] = fe_add(t[3], fe_add(fe_mul(a1, b3), fe_mul(a3, b1)));
        t[4] = fe_add(t[4], fe_mul(a2, b3));
        t[5] = fe_add(t[5], fe_mul(a3, b3));
    }
    for (int j = 0; j < 6; j++) {
        Fe<F> s = block_fr_sum<F>(t[j], lds);
        if (threadIdx.x == 0) fr_store<F>(partial, (size_t)blockIdx.x * 6 + j, s);
        __syncthreads();
    }
}

// single block: out[j] = R * sum_b partial[b * k + j], j < k  (k <= 6)
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_multi_final(const ScalarWords* __restrict__ partial, uint32_t m, uint32_t k, ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    for (uint32_t j = 0; j < k; j++) {
        Fe<F> acc = fe_zero<F>();
        for (uint32_t i = threadIdx.x; i < m; i += kBlock) acc = fe_add(acc, fr_load<F>(partial, (size_t)i * k + j));
        acc = block_fr_sum<F>(acc, lds);
        if (threadIdx.x == 0) fr_store<F>(out, j, fe_to_mont<F>(acc));
        __syncthreads();
    }
}

// VecPoly1::inner_product (:36-53): (t0, t1, t2) = (<l0,r0>, <l0,r1> + <l1,r0>, <l1,r1>)  (the reference's Karatsuba
// form computes the same t1).
template <class C>
__global__ void __launch_bounds__(kBlock) k_vecpoly1_inner(const ScalarWords* __restrict__ l0, const ScalarWords* __restrict__ l1,
                                                           const ScalarWords* __restrict__ r0, const ScalarWords* __restrict__ r1, size_t n,
                                                           ScalarWords* __restrict__ partial) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    Fe<F> t[3];
    for (int j = 0; j < 3; j++) t[j] = fe_zero<F>();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fe<F> a0 = fr_load<F>(l0, i), a1 = fr_load<F>(l1, i), b0 = fr_load<F>(r0, i), b1 = fr_load<F>(r1, i);
        t[0] = fe_add(t[0], fe_mul(a0, b0));
        t[1] = fe_add(t[1], fe_add(fe_mul(a0, b1), fe_mul(a1, b0)));
        t[2] = fe_add(t[2], fe_mul(a1, b1));
    }
    for (int j = 0; j < 3; j++) {
        Fe<F> s = block_fr_sum<F>(t[j], lds);
        if (threadIdx.x == 0) fr_store<F>(partial, (size_t)blockIdx.x * 3 + j, s);
        __syncthreads();
    }
}

// VecPoly3::eval (:99-106) / VecPoly1::eval (:55-62): Horner per element; deg = 3 or 1.  x_mont in Montgomery form.
template <class C>
__global__ void __launch_bounds__(kBlock) k_vecpoly_eval(const ScalarWords* __restrict__ p0, const ScalarWords* __restrict__ p1,
                                                         const ScalarWords* __restrict__ p2, const ScalarWords* __restrict__ p3, int deg,
                                                         ScalarWords x_mont, size_t n, ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> x = fe_unpack_words<F>(x_mont.w);
    Fe<F> acc;
    if (deg == 3) {
        acc = fr_load<F>(p3, i);
        acc = fe_add(fr_load<F>(p2, i), fe_mul(acc, x));     // canonical * Montgomery x -> canonical
        acc = fe_add(fr_load<F>(p1, i), fe_mul(acc, x));
    } else {
        acc = fr_load<F>(p1, i);
    }
    acc = fe_add(fr_load<F>(p0, i), fe_mul(acc, x));
    fr_store<F>(out, i, acc);
}

// ---------------------------------------------------------------------------------------------- R1CS vector pipeline
// (the Fr work either side of the IPP in the R1CS prover and verifier.)
template <class F>
__device__ __forceinline__ Fe<F> fr_pow_index(const Fe<F>& base_mont, size_t e) {   // base^e, Montgomery in/out
    Fe<F> acc = fe_one<F>(), b = base_mont;
    for (size_t k = e; k; k >>= 1) {
        if (k & 1) acc = fe_mul(acc, b);
        b = fe_sqr(b);
    }
    return acc;
}

// Prover: l(X), r(X) coefficient vectors, /root/reference src/r1cs/prover.rs:465-486
//   l1 = a_L + y^-i wR     l2 = a_O     l3 = s_L        r0 = wO - y^i     r1 = y^i a_R + wL     r3 = y^i s_R
template <class C>
__global__ void __launch_bounds__(kBlock) k_r1cs_prover_polys(const ScalarWords* __restrict__ aL, const ScalarWords* __restrict__ aR,
                                                              const ScalarWords* __restrict__ aO, const ScalarWords* __restrict__ sL,
                                                              const ScalarWords* __restrict__ sR, const ScalarWords* __restrict__ wL,
                                                              const ScalarWords* __restrict__ wR, const ScalarWords* __restrict__ wO,
                                                              ScalarWords y_mont, ScalarWords yinv_mont, size_t n, ScalarWords* __restrict__ l1,
                                                              ScalarWords* __restrict__ l2, ScalarWords* __restrict__ l3, ScalarWords* __restrict__ r0,
                                                              ScalarWords* __restrict__ r1, ScalarWords* __restrict__ r3) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> yi = fr_pow_index<F>(fe_unpack_words<F>(y_mont.w), i), yni = fr_pow_index<F>(fe_unpack_words<F>(yinv_mont.w), i);   // Montgomery
    fr_store<F>(l1, i, fe_add(fr_load<F>(aL, i), fe_mul(fr_load<F>(wR, i), yni)));
    l2[i] = aO[i];
    l3[i] = sL[i];
    fr_store<F>(r0, i, fe_sub(fr_load<F>(wO, i), fe_from_mont<F>(yi)));
    fr_store<F>(r1, i, fe_add(fe_mul(fr_load<F>(aR, i), yi), fr_load<F>(wL, i)));
    fr_store<F>(r3, i, fe_mul(fr_load<F>(sR, i), yi));
}

// Prover: inputs of create_ipp, src/r1cs/prover.rs:526-563
//   l_vec = l(x) | 0...   r_vec = r(x) | -y^i (i >= n)   G_factors = 1 (i < n1) | u   H_factors = y^-i * G_factors
template <class C>
__global__ void __launch_bounds__(kBlock) k_r1cs_ipp_inputs(const ScalarWords* __restrict__ l_eval, const ScalarWords* __restrict__ r_eval,
                                                            ScalarWords y_mont, ScalarWords yinv_mont, ScalarWords u_mont, size_t n, size_t n1,
                                                            size_t padded_n, ScalarWords* __restrict__ l_vec, ScalarWords* __restrict__ r_vec,
                                                            ScalarWords* __restrict__ gf, ScalarWords* __restrict__ hf) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= padded_n) return;
    Fe<F> u_or_1 = i < n1 ? fe_one<F>() : fe_unpack_words<F>(u_mont.w);
    fr_store<F>(gf, i, fe_from_mont<F>(u_or_1));
    fr_store<F>(hf, i, fe_from_mont<F>(fe_mul(fr_pow_index<F>(fe_unpack_words<F>(yinv_mont.w), i), u_or_1)));
    if (i < n) {
        l_vec[i] = l_eval[i];
        r_vec[i] = r_eval[i];
    } else {
        fr_store<F>(l_vec, i, fe_zero<F>());
        fr_store<F>(r_vec, i, fe_from_mont<F>(fe_neg(fr_pow_index<F>(fe_unpack_words<F>(y_mont.w), i))));
    }
}

// Verifier: scalars of G and H in the single verification MSM, src/r1cs/verifier.rs:342-390
//   g_i = u_or_1 (x y^-i wR_i - a s_i)        h_i = u_or_1 (y^-i (x wL_i + wO_i - b / s_i) - 1)
// with wL/wR/wO = 0 for i >= n, s_i as in k_ipp_verify_terms (1/s_i = s_(padded_n-1-i)).
template <class C>
__global__ void __launch_bounds__(kBlock) k_r1cs_verifier_scalars(const ScalarWords* __restrict__ wL, const ScalarWords* __restrict__ wR,
                                                                  const ScalarWords* __restrict__ wO, const ScalarWords* __restrict__ ch,
                                                                  const ScalarWords* __restrict__ ch_inv, int lg_n, ScalarWords yinv_mont,
                                                                  ScalarWords x_mont, ScalarWords u_mont, ScalarWords a_mont, ScalarWords b_mont,
                                                                  size_t n, size_t n1, size_t padded_n, ScalarWords* __restrict__ g_sc,
                                                                  ScalarWords* __restrict__ h_sc) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= padded_n) return;
    Fe<F> s = fe_one<F>(), sinv = fe_one<F>();
    for (int j = 0; j < lg_n; j++) {
        bool bit = (i >> (lg_n - 1 - j)) & 1;
        Fe<F> uj = fr_load<F>(ch, j), ujinv = fr_load<F>(ch_inv, j);
        s = fe_mul(s, bit ? uj : ujinv);
        sinv = fe_mul(sinv, bit ? ujinv : uj);
    }
    Fe<F> x = fe_unpack_words<F>(x_mont.w), a = fe_unpack_words<F>(a_mont.w), b = fe_unpack_words<F>(b_mont.w);
    Fe<F> u_or_1 = i < n1 ? fe_one<F>() : fe_unpack_words<F>(u_mont.w);
    Fe<F> yni = fr_pow_index<F>(fe_unpack_words<F>(yinv_mont.w), i);
    Fe<F> zero = fe_zero<F>();
    // everything below in Montgomery form; canonical vector entries are lifted with fe_to_mont
    Fe<F> wl = i < n ? fe_to_mont<F>(fr_load<F>(wL, i)) : zero, wr = i < n ? fe_to_mont<F>(fr_load<F>(wR, i)) : zero,
          wo = i < n ? fe_to_mont<F>(fr_load<F>(wO, i)) : zero;
    Fe<F> g = fe_mul(u_or_1, fe_sub(fe_mul(x, fe_mul(yni, wr)), fe_mul(a, s)));
    Fe<F> h = fe_mul(u_or_1, fe_sub(fe_mul(yni, fe_sub(fe_add(fe_mul(x, wl), wo), fe_mul(b, sinv))), fe_one<F>()));
    fr_store<F>(g_sc, i, fe_from_mont<F>(g));
    fr_store<F>(h_sc, i, fe_from_mont<F>(h));
}

// ---------------------------------------------------------------------------------------------- flattened constraints
// Prover::flattened_constraints / Verifier::flattened_constraints (/root/reference src/r1cs/prover.rs:142-184,
// src/r1cs/verifier.rs:149-193): for every term (variable, coeff) of constraint q,  w_kind[index] += z^(q+1) coeff  (wV and
// the constant wc are subtracted).  The reference walks the constraints in order; the sums are order-independent, so the
// terms are regrouped ONCE per circuit by destination (plan: CSR over destinations d = kind * n + index | 3n + index | 3n + m)
// and each proof evaluates  out[d] = +- sum_t zp[q_t] coeff_t  with zp[q] = z^(q+1).
//   k_fr_powers_mont      zp[q] = z^(q+1), Montgomery form
//   k_r1cs_flatten        lane per destination with at most `light_max` terms
//   k_r1cs_flatten_heavy  (+ _final) destinations with more, cut into chunks of 2048 terms (the constant term collects one term per constraint that has one)
template <class C>
__global__ void __launch_bounds__(kBlock) k_fr_powers_mont(ScalarWords z_mont, size_t nq, ScalarWords* __restrict__ zp) {
    using F = typename C::Fr;
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    fr_store<F>(zp, q, fr_pow_index<F>(fe_unpack_words<F>(z_mont.w), q + 1));
}

template <class C>
__global__ void __launch_bounds__(kBlock) k_r1cs_flatten(const uint32_t* __restrict__ seg, const uint32_t* __restrict__ tq,
                                                         const ScalarWords* __restrict__ coeff, const ScalarWords* __restrict__ zp, uint32_t n3,
                                                         uint32_t ndest, uint32_t light_max, ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= ndest) return;
    const uint32_t lo = seg[d], hi = seg[d + 1];
    if (hi - lo > light_max) return;                       // k_r1cs_flatten_heavy
    Fe<F> acc = fe_zero<F>();
    for (uint32_t t = lo; t < hi; t++) acc = fe_add(acc, fe_mul(fr_load<F>(zp, tq[t]), fr_load<F>(coeff, t)));   // Montgomery x canonical
    fr_store<F>(out, d, d >= n3 ? fe_neg(acc) : acc);
}

// A heavy destination is cut into chunks of <= kFlattenChunk terms at plan creation (round 4: the constant term of BASELINE config 3
// collects 136 192 terms -- one block walked them alone in 0.32 ms, in the prover AND the verifier): block per chunk c, terms
// [chunk[2c], chunk[2c + 1]) -> partial[c]; then a lane per heavy destination adds its chunks' partial sums.
constexpr uint32_t kFlattenChunk = 2048;
template <class C>
__global__ void __launch_bounds__(kBlock) k_r1cs_flatten_heavy(const uint32_t* __restrict__ chunk, uint32_t nchunks, const uint32_t* __restrict__ tq,
                                                               const ScalarWords* __restrict__ coeff, const ScalarWords* __restrict__ zp,
                                                               ScalarWords* __restrict__ partial) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint32_t lo = chunk[2 * c], hi = chunk[2 * c + 1];
        Fe<F> acc = fe_zero<F>();
        for (uint32_t t = lo + threadIdx.x; t < hi; t += kBlock) acc = fe_add(acc, fe_mul(fr_load<F>(zp, tq[t]), fr_load<F>(coeff, t)));
        acc = block_fr_sum<F>(acc, lds);
        if (threadIdx.x == 0) fr_store<F>(partial, c, acc);
        __syncthreads();
    }
}
// hfirst[h] .. hfirst[h + 1]: the chunks of heavy destination h
template <class C>
__global__ void __launch_bounds__(kBlock) k_r1cs_flatten_heavy_final(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ hfirst, uint32_t nheavy,
                                                                     const ScalarWords* __restrict__ partial, uint32_t n3, ScalarWords* __restrict__ out) {
    using F = typename C::Fr;
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nheavy) return;
    Fe<F> acc = fe_zero<F>();
    for (uint32_t c = hfirst[h]; c < hfirst[h + 1]; c++) acc = fe_add(acc, fr_load<F>(partial, c));
    const uint32_t d = heavy[h];
    fr_store<F>(out, d, d >= n3 ? fe_neg(acc) : acc);
}

// ---------------------------------------------------------------------------------------------- IPP round
// Assemble the MSM terms of L and R for the current round of length n = 2h (src/ipp.rs:80-104 / :148-170):
//   L: points [G_R | H_L | Q], scalars [a_L (.Gf_R) | b_R (.Hf_L) | c_L]
//   R: points [G_L | H_R | Q], scalars [a_R (.Gf_L) | b_L (.Hf_R) | c_R]
// written as two blocks of 2h + 1 terms each (L at 0, R at 2h + 1).  gf/hf == nullptr after the first round.
template <class C>
__global__ void __launch_bounds__(kBlock) k_ipp_pack_round(const AffPacked<C>* __restrict__ G, const AffPacked<C>* __restrict__ H,
                                                           const ScalarWords* __restrict__ a, const ScalarWords* __restrict__ b,
                                                           const ScalarWords* __restrict__ gf, const ScalarWords* __restrict__ hf,
                                                           const AffPacked<C>* __restrict__ Q, const ScalarWords* __restrict__ cLR, size_t h,
                                                           AffPacked<C>* __restrict__ pts, ScalarWords* __restrict__ sc) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t blk = 2 * h + 1;
    if (i < h) {
        pts[i] = G[h + i];            pts[h + i] = H[i];
        pts[blk + i] = G[i];          pts[blk + h + i] = H[h + i];
        if (gf) {
            fr_store<F>(sc, i, fe_to_mont<F>(fe_mul(fr_load<F>(a, i), fr_load<F>(gf, h + i))));
            fr_store<F>(sc, h + i, fe_to_mont<F>(fe_mul(fr_load<F>(b, h + i), fr_load<F>(hf, i))));
            fr_store<F>(sc, blk + i, fe_to_mont<F>(fe_mul(fr_load<F>(a, h + i), fr_load<F>(gf, i))));
            fr_store<F>(sc, blk + h + i, fe_to_mont<F>(fe_mul(fr_load<F>(b, i), fr_load<F>(hf, h + i))));
        } else {
            sc[i] = a[i];             sc[h + i] = b[h + i];
            sc[blk + i] = a[h + i];   sc[blk + h + i] = b[i];
        }
    }
    if (i == 0) {
        pts[2 * h] = *Q;      sc[2 * h] = cLR[0];
        pts[blk + 2 * h] = *Q; sc[blk + 2 * h] = cLR[1];
    }
}

// k1*p + k2*q with one shared doubling chain (Shamir), k1/k2 canonical words.  (G1::binary_scalar_mul)
// The two scalars are held in 64-bit registers and shifted left one bit per step (static indexing only: no
// per-lane scratch arrays, no divergent trip counts -- leading zero bits just double the identity).
// ONE_SITE: the three kinds of addition (+ p, + q, + (p + q)) go through ONE full-addition call site on a selected operand instead of
// two mixed and one full: ~15 % more field work per step, a third of the code -- k_ipp_fold<Bls381>, which also folds a and b in the
// same kernel, needed 512 VGPRs + 256 AGPRs and still spilled 60 registers with the three sites (profiles/r04_kernel_resources.txt).
template <class C, bool ONE_SITE = false>
BP_HD Xyzz<C> xyzz_mul2_words(const ScalarWords& k1, const Aff<C>& p, const ScalarWords& k2, const Aff<C>& q) {
    Xyzz<C> pq = xyzz_add_aff(xyzz_from_aff(p), q);   // p + q, with p == +-q and identities handled
    Xyzz<C> acc = xyzz_inf<C>();
    uint64_t a0 = k1.w[0] | ((uint64_t)k1.w[1] << 32), a1 = k1.w[2] | ((uint64_t)k1.w[3] << 32), a2 = k1.w[4] | ((uint64_t)k1.w[5] << 32),
             a3 = k1.w[6] | ((uint64_t)k1.w[7] << 32);
    uint64_t b0 = k2.w[0] | ((uint64_t)k2.w[1] << 32), b1 = k2.w[2] | ((uint64_t)k2.w[3] << 32), b2 = k2.w[4] | ((uint64_t)k2.w[5] << 32),
             b3 = k2.w[6] | ((uint64_t)k2.w[7] << 32);
    for (int i = 0; i < 256; i++) {
        uint32_t ba = (uint32_t)(a3 >> 63), bb = (uint32_t)(b3 >> 63);
        a3 = (a3 << 1) | (a2 >> 63); a2 = (a2 << 1) | (a1 >> 63); a1 = (a1 << 1) | (a0 >> 63); a0 <<= 1;
        b3 = (b3 << 1) | (b2 >> 63); b2 = (b2 << 1) | (b1 >> 63); b1 = (b1 << 1) | (b0 >> 63); b0 <<= 1;
        acc = xyzz_dbl(acc);
        if (ONE_SITE) {
            if (ba | bb) {
                const Xyzz<C> sel = (ba & bb) ? pq : ba ? xyzz_from_aff(p) : xyzz_from_aff(q);
                acc = xyzz_add(acc, sel);
            }
        } else {
            if (ba & bb) acc = xyzz_add(acc, pq);
            else if (ba) acc = xyzz_add_aff(acc, p);
            else if (bb) acc = xyzz_add_aff(acc, q);
        }
    }
    return acc;
}

// Batched `commit_to_field_element(g, h, m, r)` = `g.binary_scalar_mul(h, m, r)` = m g + r h with FIXED g, h and one (m, r)
// pair per lane (/root/reference src/r1cs/prover.rs:123: one per committed value -- 3 072 in BASELINE config 3 -- and
// :496-500 for T_1..T_6).  out[i] = k1[i] * g + k2[i] * h.
template <class C>
__global__ void __launch_bounds__(kBlock) k_commit_pairs(AffPacked<C> g, AffPacked<C> h, const ScalarWords* __restrict__ k1,
                                                         const ScalarWords* __restrict__ k2, size_t n, AffPacked<C>* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = aff_pack(xyzz_to_aff<C>(xyzz_mul2_words<C>(k1[i], aff_unpack(g), k2[i], aff_unpack(h))));
}

// One lane per (vector, i):  lanes [0, h) fold G, lanes [h, 2h) fold H; every lane i < h also folds a and b.
//   a_L[i] = a_L[i] u + u^-1 a_R[i] ;  b_L[i] = b_L[i] u^-1 + u b_R[i]                         (src/ipp.rs:116-117,182-183)
//   G_L[i] = (u^-1 Gf_L[i]) G_L[i] + (u Gf_R[i]) G_R[i] ;  H_L[i] = (u Hf_L[i]) H_L[i] + (u^-1 Hf_R[i]) H_R[i]   (:119-129,185-187)
// u_mont / uinv_mont are in Montgomery form.  In place: only the lower halves are written.
template <class C>
__global__ void __launch_bounds__(kBlock) k_ipp_fold(AffPacked<C>* __restrict__ G, AffPacked<C>* __restrict__ H, ScalarWords* __restrict__ a,
                                                     ScalarWords* __restrict__ b, const ScalarWords* __restrict__ gf,
                                                     const ScalarWords* __restrict__ hf, ScalarWords u_mont, ScalarWords uinv_mont, size_t h) {
    using F = typename C::Fr;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * h) return;
    Fe<F> u = fe_unpack_words<F>(u_mont.w), ui = fe_unpack_words<F>(uinv_mont.w);
    bool isH = t >= h;
    size_t i = isH ? t - h : t;
    if (!isH) {
        Fe<F> aL = fr_load<F>(a, i), aR = fr_load<F>(a, h + i), bL = fr_load<F>(b, i), bR = fr_load<F>(b, h + i);
        fr_store<F>(a, i, fe_add(fe_mul(aL, u), fe_mul(ui, aR)));
        fr_store<F>(b, i, fe_add(fe_mul(bL, ui), fe_mul(u, bR)));
    }
    AffPacked<C>* V = isH ? H : G;
    const ScalarWords* fac = isH ? hf : gf;
    // scalar on the left half / right half: G: (u^-1, u), H: (u, u^-1)
    Fe<F> sL = isH ? u : ui, sR = isH ? ui : u;
    ScalarWords k1, k2;
    if (fac) {
        fe_pack_words<F>(k1.w, fe_mul(fr_load<F>(fac, i), sL));      // canonical * Montgomery constant = canonical product
        fe_pack_words<F>(k2.w, fe_mul(fr_load<F>(fac, h + i), sR));
    } else {
        fe_pack_words<F>(k1.w, fe_from_mont<F>(sL));
        fe_pack_words<F>(k2.w, fe_from_mont<F>(sR));
    }
    Aff<C> p = aff_unpack(V[i]), q = aff_unpack(V[h + i]);
    Xyzz<C> r = xyzz_mul2_words<C, true>(k1, p, k2, q);
    V[i] = aff_pack(xyzz_to_aff<C>(r));
}

// ---------------------------------------------------------------------------------------------- IPP without folding generators
// The folded generators are never needed for their own sake: after j rounds (current length nj = n0 >> j)
//   G^(j)_i = sum over original k with (k mod nj) == i of cG_k * G_k ,   cG_k = Gf_k * prod_t (u_t^-1 or u_t by the bit of k
// that picked the half at round t), and likewise H with cH_k (inverse choice).  Hence each round's
//   L = <a_L, G_R> + <b_R, H_L> + c_L Q   and   R = <a_R, G_L> + <b_L, H_R> + c_R Q      (src/ipp.rs:148-170)
// are MSMs over the ORIGINAL, resident [G | H | Q] with these scalars (zero where a generator does not take part):
//   L:  G_k, pos >= h: a[pos - h] cG_k     H_k, pos <  h: b[h + pos] cH_k     Q: c_L          (pos = k mod nj, h = nj / 2)
//   R:  G_k, pos <  h: a[h + pos] cG_k     H_k, pos >= h: b[pos - h] cH_k     Q: c_R
// and the per-round "fold" is four Fr multiplications per generator instead of a 255-step double-scalar multiplication
// (the reference's dominant cost, and on a GPU a ~10 ms serial chain per round).  Identical L, R, a, b.
// Sharded form (bp_ipp_create_multi): a shard holds the generators k0 .. k0 + nloc - 1 of the n0 originals (cG, cH, sL, sR are
// its slices, indexed locally; a, b are full replicated copies) and, if with_q, the point Q as its last term.  One device: k0 = 0,
// nloc = n0, with_q = 1.
template <class C>
__global__ void __launch_bounds__(kBlock) k_ipp_round_scalars(const ScalarWords* __restrict__ a, const ScalarWords* __restrict__ b,
                                                              const ScalarWords* __restrict__ cG, const ScalarWords* __restrict__ cH,
                                                              const ScalarWords* __restrict__ cLR, size_t nloc, size_t k0, size_t nj,
                                                              ScalarWords* __restrict__ sL, ScalarWords* __restrict__ sR, int with_q) {
    using F = typename C::Fr;
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    ScalarWords zero;
    for (int i = 0; i < 8; i++) zero.w[i] = 0;
    if (k < nloc) {
        size_t h = nj / 2, pos = (k0 + k) & (nj - 1);
        // canonical * canonical / R, then * R^2 / R  ->  canonical product
        if (pos >= h) {
            fr_store<F>(sL, k, fe_to_mont<F>(fe_mul(fr_load<F>(a, pos - h), fr_load<F>(cG, k))));
            sR[k] = zero;
            sL[nloc + k] = zero;
            fr_store<F>(sR, nloc + k, fe_to_mont<F>(fe_mul(fr_load<F>(b, pos - h), fr_load<F>(cH, k))));
        } else {
            sL[k] = zero;
            fr_store<F>(sR, k, fe_to_mont<F>(fe_mul(fr_load<F>(a, h + pos), fr_load<F>(cG, k))));
            fr_store<F>(sL, nloc + k, fe_to_mont<F>(fe_mul(fr_load<F>(b, h + pos), fr_load<F>(cH, k))));
            sR[nloc + k] = zero;
        }
    }
    if (k == 0) { sL[2 * nloc] = with_q ? cLR[0] : zero; sR[2 * nloc] = with_q ? cLR[1] : zero; }
}

// The same two jobs of a round in ONE launch (round 4: a single-launch round is ~140 us of MSM kernel behind ~30 us of these small
// kernels, each with its own launch gap): grid (max(g, blocks of nloc), 3) -- rows 0 / 1 the partial sums of c_L = <a_lo, b_hi> / c_R = <a_hi, b_lo> (as
// k_fr_inner2), row 2 the L / R scalars of the nloc generators (as k_ipp_round_scalars, grid-stride) -- without the Q terms, which need
// c_L, c_R and are written by k_ipp_round_final.  SPLIT (BLS12-381 after a compaction): every scalar is stored as its two GLV halves.
// glv_split_store is defined by bp_compact.cuh (included after this header by the one translation unit that instantiates SPLIT = true).
template <class C> __device__ __forceinline__ void glv_split_store(ScalarWords* dst, size_t i, const ScalarWords& canonical);
template <class C, bool SPLIT>
__global__ void __launch_bounds__(kBlock) k_ipp_round_prep(const ScalarWords* __restrict__ a, const ScalarWords* __restrict__ b,
                                                           const ScalarWords* __restrict__ cG, const ScalarWords* __restrict__ cH, size_t nloc, size_t nj,
                                                           uint32_t g, ScalarWords* __restrict__ partial, ScalarWords* __restrict__ sL, ScalarWords* __restrict__ sR) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    const size_t h = nj / 2, stride = (size_t)gridDim.x * blockDim.x;
    if (blockIdx.y < 2) {                                // the first g blocks of rows 0 / 1 (the grid is as wide as the scalar row needs)
        if (blockIdx.x >= g) return;
        const ScalarWords* x = blockIdx.y ? a + h : a;
        const ScalarWords* y = blockIdx.y ? b : b + h;
        Fe<F> acc = fe_zero<F>();
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < h; i += (size_t)g * blockDim.x) acc = fe_add(acc, fe_mul(fr_load<F>(x, i), fr_load<F>(y, i)));
        acc = block_fr_sum<F>(acc, lds);
        if (threadIdx.x == 0) fr_store<F>(partial, (size_t)blockIdx.y * g + blockIdx.x, acc);
        return;
    }
    ScalarWords zero;
    for (int i = 0; i < 8; i++) zero.w[i] = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < nloc; k += stride) {
        const size_t pos = k & (nj - 1);
        const bool upper = pos >= h;
        // canonical * canonical / R, then * R^2 / R  ->  canonical product
        ScalarWords g, hh;
        fe_pack_words<F>(g.w, fe_to_mont<F>(fe_mul(fr_load<F>(a, upper ? pos - h : h + pos), fr_load<F>(cG, k))));
        fe_pack_words<F>(hh.w, fe_to_mont<F>(fe_mul(fr_load<F>(b, upper ? pos - h : h + pos), fr_load<F>(cH, k))));
        // L: G upper, H lower;  R: G lower, H upper  (zero = zero in split form too)
        if (SPLIT) {
            if (upper) { glv_split_store<C>(sL, k, g); sR[k] = zero; sL[nloc + k] = zero; glv_split_store<C>(sR, nloc + k, hh); }
            else { sL[k] = zero; glv_split_store<C>(sR, k, g); glv_split_store<C>(sL, nloc + k, hh); sR[nloc + k] = zero; }
        } else {
            if (upper) { sL[k] = g; sR[k] = zero; sL[nloc + k] = zero; sR[nloc + k] = hh; }
            else { sL[k] = zero; sR[k] = g; sL[nloc + k] = hh; sR[nloc + k] = zero; }
        }
    }
}
// blocks 0 / 1: c_L / c_R = R * (sum of the m partials) -> cLR[blockIdx.x], and the scalar of Q in the round's L / R set
template <class C, bool SPLIT>
__global__ void __launch_bounds__(kBlock) k_ipp_round_final(const ScalarWords* __restrict__ partial, uint32_t m, ScalarWords* __restrict__ cLR, size_t nloc,
                                                            ScalarWords* __restrict__ sL, ScalarWords* __restrict__ sR) {
    using F = typename C::Fr;
    __shared__ ScalarWords lds[kBlock];
    Fe<F> acc = fe_zero<F>();
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) acc = fe_add(acc, fr_load<F>(partial, (size_t)blockIdx.x * m + i));
    acc = block_fr_sum<F>(acc, lds);
    if (threadIdx.x == 0) {
        ScalarWords c;
        fe_pack_words<F>(c.w, fe_to_mont<F>(acc));
        cLR[blockIdx.x] = c;
        ScalarWords* dst = blockIdx.x ? sR : sL;
        if (SPLIT) glv_split_store<C>(dst, 2 * nloc, c);
        else dst[2 * nloc] = c;
    }
}

// cG_k *= (pos < h ? u^-1 : u); cH_k *= (pos < h ? u : u^-1); a, b folded (src/ipp.rs:116-129, 182-187).
// grid covers max(nloc, nj / 2) threads: the coefficient slice and the (replicated) a, b are independent jobs.
template <class C>
__global__ void __launch_bounds__(kBlock) k_ipp_fold_scalars(ScalarWords* __restrict__ a, ScalarWords* __restrict__ b, ScalarWords* __restrict__ cG,
                                                             ScalarWords* __restrict__ cH, ScalarWords u_mont, ScalarWords uinv_mont, size_t nloc,
                                                             size_t k0, size_t nj) {
    using F = typename C::Fr;
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fe<F> u = fe_unpack_words<F>(u_mont.w), ui = fe_unpack_words<F>(uinv_mont.w);
    size_t h = nj / 2;
    if (k < nloc) {
        bool left = ((k0 + k) & (nj - 1)) < h;
        fr_store<F>(cG, k, fe_mul(fr_load<F>(cG, k), left ? ui : u));
        fr_store<F>(cH, k, fe_mul(fr_load<F>(cH, k), left ? u : ui));
    }
    if (k < h) {
        Fe<F> aL = fr_load<F>(a, k), aR = fr_load<F>(a, h + k), bL = fr_load<F>(b, k), bR = fr_load<F>(b, h + k);
        fr_store<F>(a, k, fe_add(fe_mul(aL, u), fe_mul(ui, aR)));
        fr_store<F>(b, k, fe_add(fe_mul(bL, ui), fe_mul(u, bR)));
    }
}

// ---------------------------------------------------------------------------------------------- IPP verification
// Terms of the single verification MSM (src/ipp.rs:220-249): for i < n
//   sc[1 + i]     = a * s_i * Gf_i         with  s_i = prod_j u_j^(+1 if bit (lg_n-1-j) of i else -1)     (:303-312)
//   sc[1 + n + i] = b * s_(n-1-i) * Hf_i   (s_(n-1-i) = 1 / s_i)
// points [Q | G | H | L | R]; sc[0] = a*b and the -u_j^2 / -u_j^-2 tail are written by the host side.
// ch[j] / ch_inv[j] (j < lg_n, creation order) and a_mont / b_mont are in Montgomery form.
template <class C>
__global__ void __launch_bounds__(kBlock) k_ipp_verify_terms(const AffPacked<C>* __restrict__ G, const AffPacked<C>* __restrict__ H,
                                                             const ScalarWords* __restrict__ gf, const ScalarWords* __restrict__ hf,
                                                             const ScalarWords* __restrict__ ch, const ScalarWords* __restrict__ ch_inv, int lg_n,
                                                             ScalarWords a_mont, ScalarWords b_mont, size_t n, AffPacked<C>* __restrict__ pts,
                                                             ScalarWords* __restrict__ sc, const ScalarWords* __restrict__ ends, ScalarWords* __restrict__ ends_to) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the 1 + 2 lg n scalars the host computed (a b on Q, -u_j^2 on L_j, -u_j^-2 on R_j) travel in the same upload as the challenges: the
    // first of them goes to sc[0], the others to ends_to[0 ..) (= sc[1 + 2n ..) of the plain layout)
    if (ends) for (size_t j = i; j < 1 + 2 * (size_t)lg_n; j += n) { if (j == 0) sc[0] = ends[0]; else ends_to[j - 1] = ends[j]; }
    Fe<F> s = fe_one<F>(), sinv = fe_one<F>();
    for (int j = 0; j < lg_n; j++) {
        bool bit = (i >> (lg_n - 1 - j)) & 1;
        Fe<F> uj = fr_load<F>(ch, j), ujinv = fr_load<F>(ch_inv, j);
        s = fe_mul(s, bit ? uj : ujinv);
        sinv = fe_mul(sinv, bit ? ujinv : uj);
    }
    // s, sinv are Montgomery; (a_mont * s) is Montgomery; times canonical factor -> canonical
    fr_store<F>(sc, 1 + i, fe_mul(fe_mul(fe_unpack_words<F>(a_mont.w), s), fr_load<F>(gf, i)));
    fr_store<F>(sc, 1 + n + i, fe_mul(fe_mul(fe_unpack_words<F>(b_mont.w), sinv), fr_load<F>(hf, i)));
    if (pts) {                               // NULL: the MSM runs over the generators' own tables (bp_internal_msm_extras_gh), scalars only
        pts[1 + i] = G[i];
        pts[1 + n + i] = H[i];
    }
}

// Batch verification of m proofs over the SAME generators (the random-linear-combination argument of
// src/r1cs/verifier.rs:392 applied across proofs): sum_j w_j * (check_j) == O with caller-chosen random weights w_j.
//   sc[i]     = Gf_i * sum_j (w_j a_j) s_(j,i)           (points G)
//   sc[n + i] = Hf_i * sum_j (w_j b_j) / s_(j,i)         (points H)
// ch / ch_inv: m blocks of lg_n challenges (Montgomery form); wa / wb: m products w_j a_j, w_j b_j (Montgomery form).
// The per-proof tail (Q_j, L_j, R_j, P_j with scalars w_j a_j b_j, -w_j u^2, -w_j u^-2, -w_j) is assembled by the host side.
template <class C>
__global__ void __launch_bounds__(kBlock) k_ipp_verify_terms_batch(const AffPacked<C>* __restrict__ G, const AffPacked<C>* __restrict__ H,
                                                                   const ScalarWords* __restrict__ gf, const ScalarWords* __restrict__ hf,
                                                                   const ScalarWords* __restrict__ ch, const ScalarWords* __restrict__ ch_inv,
                                                                   const ScalarWords* __restrict__ wa, const ScalarWords* __restrict__ wb, int lg_n,
                                                                   size_t m, size_t n, AffPacked<C>* __restrict__ pts, ScalarWords* __restrict__ sc) {
    using F = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> gacc = fe_zero<F>(), hacc = fe_zero<F>();
    for (size_t p = 0; p < m; p++) {
        Fe<F> s = fr_load<F>(wa, p), sinv = fr_load<F>(wb, p);
        for (int j = 0; j < lg_n; j++) {
            bool bit = (i >> (lg_n - 1 - j)) & 1;
            Fe<F> uj = fr_load<F>(ch, p * lg_n + j), ujinv = fr_load<F>(ch_inv, p * lg_n + j);
            s = fe_mul(s, bit ? uj : ujinv);
            sinv = fe_mul(sinv, bit ? ujinv : uj);
        }
        gacc = fe_add(gacc, s);
        hacc = fe_add(hacc, sinv);
    }
    // Montgomery accumulators times canonical factors -> canonical scalars
    fr_store<F>(sc, i, fe_mul(gacc, fr_load<F>(gf, i)));
    fr_store<F>(sc, n + i, fe_mul(hacc, fr_load<F>(hf, i)));
    if (pts) {                               // NULL: as in k_ipp_verify_terms
        pts[i] = G[i];
        pts[n + i] = H[i];
    }
}

}  // namespace bp
