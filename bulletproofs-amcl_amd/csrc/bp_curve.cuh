// bp_curve.cuh -- G1 arithmetic (y^2 = x^3 + b, a = 0) for BLS12-381 and AMCL-BN254, host + gfx950 device.
//
// What it replaces: amcl_wrapper::group_elem_g1::G1 as used by the reference -- `G1 + G1`, `&G1 * &Fr`,
// `binary_scalar_mul`, equality/identity (/root/reference src/ipp.rs:119,125,185,187,255;
// src/r1cs/prover.rs:358,423,429,550) -- and the point arithmetic inside
// G1Vector's multi-scalar multiplications (rows a1/a2).
//
// Coordinates: XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity <=> ZZ = 0).  Mixed addition of an
// affine point costs 8M + 2S against 7M + 4S for Jacobian, and the bucket-sum + bucket-sum addition
// 12M + 2S against 11M + 5S: XYZZ wins wherever the bucket method spends its time, at the price of 13
// more VGPRs per accumulator.  Every exceptional case of the incomplete formulas (P + P, P + (-P),
// identity operands) is handled explicitly: buckets DO see them (duplicate generators, P and -P in one
// bucket; tests/golden/msm.json holds such cases).
//
// Memory forms: AffPacked = x || y packed Montgomery words (96 B for BLS12-381, 64 B for BN254), all-zero
// = identity ((0,0) is on neither curve); XyzzPacked = 4 packed coordinates.
#pragma once
#include "bp_field.cuh"

namespace bp {

struct Bls381 {
    using Fp = Bls381Fp;
    using Fr = Bls381Fr;
    static constexpr int ID = 0;
    static constexpr int MODBYTES = 48;   // amcl MODBYTES
    static constexpr uint32_t B = 4;
    static constexpr bool COFACTOR_IS_ONE = false;
    static constexpr uint32_t COFACTOR[4] = {0x0000aaabu, 0x8c00aaabu, 0x5555e156u, 0x396c8c00u};   // (x-1)^2/3, amcl rom CURVE_COF
    static constexpr uint32_t GX[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                        0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    static constexpr uint32_t GY[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                        0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    // GLV endomorphism (round 4; bp_compact.cuh): phi(x, y) = (BETA x, y) = LAMBDA (x, y) on G1, LAMBDA = z^2 - 1 for the BLS parameter
    // z = -0xd201000000010000 (LAMBDA^2 + LAMBDA + 1 = r; checked with Python integers, and by every parity test that crosses the
    // compaction).  A scalar s < r splits as s = s1 + s2 LAMBDA by plain division: s1 < LAMBDA < 2^128, s2 <= r / LAMBDA < 2^128.
    // GLV_MLO = floor(2^256 / LAMBDA) - 2^128 (Barrett quotient, at most one correction).
    static constexpr bool HAS_GLV = true;
    static constexpr uint32_t BETA[12] = {0x0000aaacu, 0x8bfd0000u, 0x4f49fffdu, 0x409427ebu, 0x0fb85f9bu, 0x897d2965u,
                                          0x89759ad4u, 0xaa0d857du, 0x63d4de85u, 0xec024086u, 0x397fe699u, 0x1a0111eau};
    static constexpr uint64_t GLV_LAMBDA[2] = {0x00000000ffffffffull, 0xac45a4010001a402ull};
    static constexpr uint64_t GLV_MLO[2] = {0x63f6e522f6cfee30ull, 0x7c6becf1e01faaddull};
    static constexpr bool GLV_SIGNED = false;            // both halves are plain 128-bit numbers
    static constexpr uint64_t GLV_A = 0, GLV_B[2] = {0, 0}, GLV_C[2] = {0, 0}, GLV_M1[2] = {0, 0}, GLV_M2[2] = {0, 0};      // (the lattice form, BN254)
};

struct Bn254 {
    using Fp = Bn254Fp;
    using Fr = Bn254Fr;
    static constexpr int ID = 1;
    static constexpr int MODBYTES = 32;
    static constexpr uint32_t B = 2;
    static constexpr bool COFACTOR_IS_ONE = true;
    static constexpr uint32_t COFACTOR[4] = {1, 0, 0, 0};
    static constexpr uint32_t GX[8] = {0x00000012u, 0xa7000000u, 0x00000013u, 0x61210000u, 0x00000008u, 0xba344d80u, 0x40000001u, 0x25236482u};
    static constexpr uint32_t GY[8] = {0x00000001u, 0, 0, 0, 0, 0, 0, 0};
    // GLV endomorphism (round 4, second half): phi(x, y) = (BETA x, y) = LAMBDA (x, y) with LAMBDA = 36 u^4 - 1 mod r = -(36 u^3 + 18 u^2 + 6 u + 2)
    // = 0x9366c48000000005b696800000000013a700000000000016 (190 bits; u = -0x4080000000000001 the curve's parameter).  LAMBDA is far from
    // sqrt(r), so the split is by the lattice {(x, y): x + y LAMBDA = 0 mod r} with the reduced basis (-A, B), (C, A):
    //     A = -2u - 1,   B = 6u^2 + 4u + 1,   C = 6u^2 + 2u,   A^2 + B C = r.
    // With e1 = floor(s A / r) or one less (= (s M1) >> 317, M1 = floor(2^317 A / r)) and e2 = floor(s B / r) or one less (= (s M2) >> 254,
    // M2 = floor(2^254 B / r)):    s1 = s - e1 A - e2 C  in [0, 2A + 2C) < 2^128,     s2 = e1 B - e2 A  in (-2B, 2A),   s = s1 + s2 LAMBDA mod r.
    // s2 travels mod 2^128: a value >= 2^66 is a NEGATIVE number (|s2| < 2B < 2^128 - 2^66) -- glv_half_signed in bp_compact.cuh.
    // Constants and ranges checked with Python integers (3.3e5 scalars incl. the boundaries) and by every parity test over BN254 proofs.
    static constexpr bool HAS_GLV = true;
    static constexpr bool GLV_SIGNED = true;
    static constexpr uint32_t BETA[8] = {0x00000007u, 0xcd800000u, 0x00000006u, 0x49090000u, 0x00000002u, 0x49b36240u, 0x00000000u, 0x00000000u};
    static constexpr uint64_t GLV_A = 0x8100000000000001ull;
    static constexpr uint64_t GLV_B[2] = {0x0400000000000003ull, 0x6181800000000002ull};
    static constexpr uint64_t GLV_C[2] = {0x8500000000000004ull, 0x6181800000000002ull};
    static constexpr uint64_t GLV_M1[2] = {0x9721b09dbea093a4ull, 0x6f26f94d114d6920ull};
    static constexpr uint64_t GLV_M2[2] = {0x703acc7fcdafccd5ull, 0xa807eadf812805eeull};
    static constexpr uint64_t GLV_LAMBDA[2] = {1, 0};     // (the plain-division form is BLS12-381's)
    static constexpr uint64_t GLV_MLO[2] = {0, 0};
};

template <class C>
struct alignas(16) AffPacked {
    FePacked<typename C::Fp> x, y;
};

template <class C>
struct Aff {   // unpacked affine; identity <=> x = y = 0
    Fe<typename C::Fp> x, y;
};

template <class C>
struct Xyzz {
    Fe<typename C::Fp> x, y, zz, zzz;
};

template <class C>
struct alignas(16) XyzzPacked {
    FePacked<typename C::Fp> x, y, zz, zzz;
};

template <class C> BP_HD Aff<C> aff_unpack(const AffPacked<C>& p) { Aff<C> r; r.x = fe_unpack(p.x); r.y = fe_unpack(p.y); return r; }
template <class C> BP_HD AffPacked<C> aff_pack(const Aff<C>& p) { AffPacked<C> r; r.x = fe_pack(p.x); r.y = fe_pack(p.y); return r; }
template <class C> BP_HD bool aff_is_inf(const Aff<C>& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
template <class C> BP_HD Aff<C> aff_neg(const Aff<C>& p) { Aff<C> r; r.x = p.x; r.y = fe_neg(p.y); return r; }   // -(0,0) = (0,0)

template <class C> BP_HD Xyzz<C> xyzz_unpack(const XyzzPacked<C>& p) { Xyzz<C> r; r.x = fe_unpack(p.x); r.y = fe_unpack(p.y); r.zz = fe_unpack(p.zz); r.zzz = fe_unpack(p.zzz); return r; }
template <class C> BP_HD XyzzPacked<C> xyzz_pack(const Xyzz<C>& p) { XyzzPacked<C> r; r.x = fe_pack(p.x); r.y = fe_pack(p.y); r.zz = fe_pack(p.zz); r.zzz = fe_pack(p.zzz); return r; }

template <class C> BP_HD Xyzz<C> xyzz_inf() {
    using Fp = typename C::Fp;
    Xyzz<C> r; r.x = fe_zero<Fp>(); r.y = fe_zero<Fp>(); r.zz = fe_zero<Fp>(); r.zzz = fe_zero<Fp>();
    return r;
}
template <class C> BP_HD bool xyzz_is_inf(const Xyzz<C>& p) { return fe_is_zero(p.zz); }

template <class C> BP_HD Xyzz<C> xyzz_from_aff(const Aff<C>& p) {
    using Fp = typename C::Fp;
    if (aff_is_inf(p)) return xyzz_inf<C>();
    Xyzz<C> r; r.x = p.x; r.y = p.y; r.zz = fe_one<Fp>(); r.zzz = fe_one<Fp>();
    return r;
}

// 2 * (affine p), p != identity: mdbl-2008-s-1 (a = 0)
template <class C> BP_HD Xyzz<C> xyzz_dbl_aff(const Aff<C>& p) {
    using Fp = typename C::Fp;
    Fe<Fp> U = fe_dbl(p.y);
    Fe<Fp> V = fe_sqr(U);
    Fe<Fp> W = fe_mul(U, V);
    Fe<Fp> S = fe_mul(p.x, V);
    Fe<Fp> X2 = fe_sqr(p.x);
    Fe<Fp> M = fe_add(fe_dbl(X2), X2);
    Xyzz<C> r;
    r.x = fe_sub(fe_sub(fe_sqr(M), S), S);
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, p.y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2 * p: dbl-2008-s-1 (a = 0).  (y = 0 cannot occur: both groups have odd order.)
template <class C> BP_HD Xyzz<C> xyzz_dbl(const Xyzz<C>& p) {
    using Fp = typename C::Fp;
    if (xyzz_is_inf(p)) return p;
    Fe<Fp> U = fe_dbl(p.y);
    Fe<Fp> V = fe_sqr(U);
    Fe<Fp> W = fe_mul(U, V);
    Fe<Fp> S = fe_mul(p.x, V);
    Fe<Fp> X2 = fe_sqr(p.x);
    Fe<Fp> M = fe_add(fe_dbl(X2), X2);
    Xyzz<C> r;
    r.x = fe_sub(fe_sub(fe_sqr(M), S), S);
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, p.y));
    r.zz = fe_mul(V, p.zz);
    r.zzz = fe_mul(W, p.zzz);
    return r;
}

// acc + (affine q): madd-2008-s, exceptional cases handled.
template <class C> BP_HD Xyzz<C> xyzz_add_aff(const Xyzz<C>& a, const Aff<C>& q) {
    using Fp = typename C::Fp;
    if (aff_is_inf(q)) return a;
    if (xyzz_is_inf(a)) return xyzz_from_aff(q);
    Fe<Fp> U2 = fe_mul(q.x, a.zz);
    Fe<Fp> S2 = fe_mul(q.y, a.zzz);
    Fe<Fp> Pp = fe_sub(U2, a.x);
    Fe<Fp> Rr = fe_sub(S2, a.y);
    if (fe_is_zero(Pp)) {
        if (fe_is_zero(Rr)) return xyzz_dbl_aff(q);
        return xyzz_inf<C>();
    }
    Fe<Fp> PP = fe_sqr(Pp);
    Fe<Fp> PPP = fe_mul(Pp, PP);
    Fe<Fp> Q = fe_mul(a.x, PP);
    Xyzz<C> r;
    r.x = fe_sub(fe_sub(fe_sub(fe_sqr(Rr), PPP), Q), Q);
    r.y = fe_sub(fe_mul(Rr, fe_sub(Q, r.x)), fe_mul(a.y, PPP));
    r.zz = fe_mul(a.zz, PP);
    r.zzz = fe_mul(a.zzz, PPP);
    return r;
}

// a + b: add-2008-s, exceptional cases handled.
template <class C> BP_HD Xyzz<C> xyzz_add(const Xyzz<C>& a, const Xyzz<C>& b) {
    using Fp = typename C::Fp;
    if (xyzz_is_inf(b)) return a;
    if (xyzz_is_inf(a)) return b;
    Fe<Fp> U1 = fe_mul(a.x, b.zz);
    Fe<Fp> U2 = fe_mul(b.x, a.zz);
    Fe<Fp> S1 = fe_mul(a.y, b.zzz);
    Fe<Fp> S2 = fe_mul(b.y, a.zzz);
    Fe<Fp> Pp = fe_sub(U2, U1);
    Fe<Fp> Rr = fe_sub(S2, S1);
    if (fe_is_zero(Pp)) {
        if (fe_is_zero(Rr)) return xyzz_dbl(a);
        return xyzz_inf<C>();
    }
    Fe<Fp> PP = fe_sqr(Pp);
    Fe<Fp> PPP = fe_mul(Pp, PP);
    Fe<Fp> Q = fe_mul(U1, PP);
    Xyzz<C> r;
    r.x = fe_sub(fe_sub(fe_sub(fe_sqr(Rr), PPP), Q), Q);
    r.y = fe_sub(fe_mul(Rr, fe_sub(Q, r.x)), fe_mul(S1, PPP));
    r.zz = fe_mul(fe_mul(a.zz, b.zz), PP);
    r.zzz = fe_mul(fe_mul(a.zzz, b.zzz), PPP);
    return r;
}

template <class C> BP_HD Xyzz<C> xyzz_neg(const Xyzz<C>& p) { Xyzz<C> r = p; r.y = fe_neg(p.y); return r; }

// -> affine.  One field inversion (of ZZ * ZZZ).
template <class C> BP_HD_NOINLINE Aff<C> xyzz_to_aff(const Xyzz<C>& p) {
    using Fp = typename C::Fp;
    Aff<C> r;
    if (xyzz_is_inf(p)) { r.x = fe_zero<Fp>(); r.y = fe_zero<Fp>(); return r; }
    Fe<Fp> i5 = fe_inv<Fp>(fe_mul(p.zz, p.zzz));
    r.x = fe_mul(p.x, fe_mul(i5, p.zzz));   // X / ZZ
    r.y = fe_mul(p.y, fe_mul(i5, p.zz));    // Y / ZZZ
    return r;
}

// y^2 == x^3 + b ?  (identity counts as on the curve)
template <class C> BP_HD bool aff_on_curve(const Aff<C>& p) {
    using Fp = typename C::Fp;
    if (aff_is_inf(p)) return true;
    Fe<Fp> braw = fe_zero<Fp>();
    braw.v[0] = C::B;
    Fe<Fp> rhs = fe_add(fe_mul(fe_sqr(p.x), p.x), fe_to_mont<Fp>(braw));
    return fe_eq(fe_sqr(p.y), rhs);
}

template <class C> BP_HD Aff<C> generator() {
    using Fp = typename C::Fp;
    Aff<C> g;
    g.x = fe_to_mont<Fp>(fe_unpack_words<Fp>(C::GX));
    g.y = fe_to_mont<Fp>(fe_unpack_words<Fp>(C::GY));
    return g;
}

// k * p, k given as 8 canonical (non-Montgomery) 32-bit words; double-and-add, msb first.  The scalar sits in
// 64-bit registers shifted left one bit per step (static indexing; leading zeros double the identity).
template <class C> BP_HD Xyzz<C> xyzz_mul_words(const uint32_t (&k)[8], const Aff<C>& p) {
    Xyzz<C> acc = xyzz_inf<C>();
    uint64_t a0 = k[0] | ((uint64_t)k[1] << 32), a1 = k[2] | ((uint64_t)k[3] << 32), a2 = k[4] | ((uint64_t)k[5] << 32),
             a3 = k[6] | ((uint64_t)k[7] << 32);
    for (int i = 0; i < 256; i++) {
        uint32_t bit = (uint32_t)(a3 >> 63);
        a3 = (a3 << 1) | (a2 >> 63); a2 = (a2 << 1) | (a1 >> 63); a1 = (a1 << 1) | (a0 >> 63); a0 <<= 1;
        acc = xyzz_dbl(acc);
        if (bit) acc = xyzz_add_aff(acc, p);
    }
    return acc;
}

}  // namespace bp

namespace bp {

// ----------------------------------------------------------------------------------------------- lazy accumulator
// The bucket-accumulate loop (k_accumulate) keeps its XYZZ accumulator in the bounded domain of bp_field.cuh:
//     X < 8p,  Y < 4p,  ZZ < 2p,  ZZZ < 2p        (limbs normalised)
// and adds canonical affine points with madd-2008-s, every intermediate bound tracked in the types:
//     U2 = x2 ZZ            (1*2)         < 2p          S2 = y2 ZZZ          (1*2)        < 2p
//     P  = U2 - X + 8p                    < 10p         R  = S2 - Y + 4p                  < 6p
//     PP = P^2              (10*10)       < 2p          PPP = P PP           (10*2)       < 2p
//     Q  = X PP             (8*2)         < 2p
//     X3 = R^2 - PPP - Q - Q  (+2p each)  < 8p          (R^2: 6*6)
//     Y3 = R (Q - X3 + 8p) - Y PPP + 2p   < 4p          (6*10, 4*2)
//     ZZ3 = ZZ PP, ZZZ3 = ZZZ PPP         < 2p
// so the invariant is reproduced and no product exceeds kMaxProd.  Exceptional cases: P == 0 (mod p) is detected with
// feb_is_zero_mod_p; the (rare) doubling / cancellation / first-point paths go through the strict functions.
template <class C>
struct XyzzLazy {
    using Fp = typename C::Fp;
    FeB<Fp, 8> x;
    FeB<Fp, 4> y;
    FeB<Fp, 2> zz, zzz;
    bool inf;
};

template <class C> BP_HD XyzzLazy<C> xyzz_lazy_inf() {
    using Fp = typename C::Fp;
    XyzzLazy<C> r;
    for (int i = 0; i < Fp::NL; i++) { r.x.v[i] = 0; r.y.v[i] = 0; r.zz.v[i] = 0; r.zzz.v[i] = 0; }
    r.inf = true;
    return r;
}

template <class C> BP_HD XyzzLazy<C> xyzz_lazy_from_strict(const Xyzz<C>& p) {
    using Fp = typename C::Fp;
    XyzzLazy<C> r;
    r.x = feb_widen<8>(feb_from_strict<Fp>(p.x));
    r.y = feb_widen<4>(feb_from_strict<Fp>(p.y));
    r.zz = feb_widen<2>(feb_from_strict<Fp>(p.zz));
    r.zzz = feb_widen<2>(feb_from_strict<Fp>(p.zzz));
    r.inf = xyzz_is_inf(p);
    return r;
}

template <class C> BP_HD Xyzz<C> xyzz_lazy_to_strict(const XyzzLazy<C>& p) {
    if (p.inf) return xyzz_inf<C>();
    Xyzz<C> r;
    r.x = feb_to_strict(p.x);
    r.y = feb_to_strict(p.y);
    r.zz = feb_to_strict(p.zz);
    r.zzz = feb_to_strict(p.zzz);
    return r;
}

// acc += q  (q canonical affine, possibly the identity).  M = multiplier policy (bp_field.cuh): MulInline / MulCall.
template <class C, class M = MulInline> BP_HD void xyzz_lazy_add_aff(XyzzLazy<C>& a, const Aff<C>& q) {
    using Fp = typename C::Fp;
    if (aff_is_inf(q)) return;
    if (a.inf) { a = xyzz_lazy_from_strict(xyzz_from_aff(q)); return; }
    FeB<Fp, 1> qx = feb_from_strict<Fp>(q.x), qy = feb_from_strict<Fp>(q.y);
    FeB<Fp, 2> U2 = M::mul(qx, a.zz);
    FeB<Fp, 2> S2 = M::mul(qy, a.zzz);
    FeB<Fp, 10> Pp = feb_sub<8>(U2, a.x);
    FeB<Fp, 6> Rr = feb_sub<4>(S2, a.y);
    if (feb_is_zero_mod_p(Pp)) {
        if (feb_is_zero_mod_p(Rr)) a = xyzz_lazy_from_strict(xyzz_dbl_aff(q));     // acc == q
        else a = xyzz_lazy_inf<C>();                                                // acc == -q
        return;
    }
    FeB<Fp, 2> PP = M::sqr(Pp);
    FeB<Fp, 2> PPP = M::mul(Pp, PP);
    FeB<Fp, 2> Q = M::mul(a.x, PP);
    FeB<Fp, 2> R2 = M::sqr(Rr);
    FeB<Fp, 8> X3 = feb_sub<2>(feb_sub<2>(feb_sub<2>(R2, PPP), Q), Q);
    FeB<Fp, 10> QX = feb_sub<8>(Q, X3);
    // Y3 = R (Q - X3) - Y PPP as R (Q - X3) + (4p - Y) PPP with ONE reduction (6*10 + 4*2 <= kMaxProd)
    FeB<Fp, 4> Y3 = feb_widen<4>(M::mul_add_mul(Rr, QX, feb_neg<4>(a.y), PPP));
    a.zz = M::mul(a.zz, PP);
    a.zzz = M::mul(a.zzz, PPP);
    a.x = X3;
    a.y = Y3;
}

// The same addition for the inner loop of k_accumulate: the accumulator is known to hold a finite point, q is finite and arrives
// as bounded limbs (x canonical, y < 2p: the negation of a canonical y is p - y, no add-back pass).  Single path: when q.x equals
// the accumulator's x (doubling or cancellation) it returns false with the accumulator UNTOUCHED and the caller takes the general
// function above for that one point -- so the loop around this function has no merge of several definitions of the accumulator
// (the general form cost ~140 register copies per iteration at the loop's phi nodes).  X3 is one pass
// R^2 - PPP - 2Q + 6p instead of three subtractions.
template <class C, class M = MulInline>
BP_HD bool xyzz_lazy_add_aff_fast(XyzzLazy<C>& a, const FeB<typename C::Fp, 1>& qx, const FeB<typename C::Fp, 2>& qy) {
    using Fp = typename C::Fp;
    FeB<Fp, 2> U2 = M::mul(qx, a.zz);
    FeB<Fp, 2> S2 = M::mul(qy, a.zzz);
    FeB<Fp, 10> Pp = feb_sub<8>(U2, a.x);
    FeB<Fp, 6> Rr = feb_sub<4>(S2, a.y);
    if (feb_is_zero_mod_p(Pp)) return false;
    FeB<Fp, 2> PP = M::sqr(Pp);
    FeB<Fp, 2> PPP = M::mul(Pp, PP);
    FeB<Fp, 2> Q = M::mul(a.x, PP);
    FeB<Fp, 2> R2 = M::sqr(Rr);
    FeB<Fp, 8> X3 = feb_sub2k<6>(R2, PPP, feb_add(Q, Q));
    FeB<Fp, 10> QX = feb_sub<8>(Q, X3);
    FeB<Fp, 4> Y3 = feb_widen<4>(M::mul_add_mul(Rr, QX, feb_neg<4>(a.y), PPP));
    a.zz = M::mul(a.zz, PP);
    a.zzz = M::mul(a.zzz, PPP);
    a.x = X3;
    a.y = Y3;
    return true;
}

// ----------------------------------------------------------------------------------------------- lazy full addition / doubling
// The same bounded domain for bucket-sum + bucket-sum additions (tree sums, the digit-sum reduce, the small-MSM path):
//     X < 8p,  Y < 4p,  ZZ < 2p,  ZZZ < 2p   on both operands and on the result.
// add-2008-s:
//     U1 = X1 ZZ2 (8*2)  U2 = X2 ZZ1   S1 = Y1 ZZZ2 (4*2)  S2 = Y2 ZZZ1            all < 2p
//     P = U2 - U1 + 2p < 4p            R = S2 - S1 + 2p < 4p
//     PP = P^2 (16)   PPP = P PP (8)   Q = U1 PP (4)                                 all < 2p
//     X3 = R^2 - PPP - Q - Q (+2p each) < 8p        Y3 = R (Q - X3 + 8p) - S1 PPP + 2p   (4*10, 2*2)  < 4p
//     ZZ3 = (ZZ1 ZZ2) PP,  ZZZ3 = (ZZZ1 ZZZ2) PPP                                    < 2p
// dbl-2008-s-1 (a = 0):
//     U = 2Y < 8p   V = U^2 (64)   W = U V (16)   S = X V (16)   M = 3 X^2 (64 -> 3 * 2p = 6p)
//     X3 = M^2 - S - S (36; +2p each) < 6p          Y3 = M (S - X3 + 8p) - W Y + 2p   (6*10, 2*4)  < 4p
//     ZZ3 = V ZZ,  ZZZ3 = W ZZZ
// No conditional subtraction (no v_cndmask: 22 cycles per wave instruction on gfx950) anywhere on these paths.
// A lazy point is stored PACKED without canonicalisation when 8p < 2^(32 NW) (BLS12-381: 8p < 2^384); for BN254
// (4p < 2^256 < 8p) X is brought below 4p first.  The host tail reduces on entry (bp_host_tail.hpp).
template <class C, class M = MulInline> BP_HD XyzzLazy<C> xyzz_lazy_dbl(const XyzzLazy<C>& a) {
    using Fp = typename C::Fp;
    if (a.inf) return a;
    FeB<Fp, 8> U = feb_add(a.y, a.y);
    FeB<Fp, 2> V = M::sqr(U);
    FeB<Fp, 2> W = M::mul(U, V);
    FeB<Fp, 2> S = M::mul(a.x, V);
    FeB<Fp, 2> X2 = M::sqr(a.x);
    FeB<Fp, 6> Mm = feb_add(feb_add(X2, X2), X2);
    XyzzLazy<C> r;
    FeB<Fp, 6> X3 = feb_sub<2>(feb_sub<2>(M::sqr(Mm), S), S);
    FeB<Fp, 10> SX = feb_sub<8>(S, X3);
    r.y = feb_widen<4>(M::mul_add_mul(Mm, SX, W, feb_neg<4>(a.y)));      // 6*10 + 2*4
    r.x = feb_widen<8>(X3);
    r.zz = M::mul(V, a.zz);
    r.zzz = M::mul(W, a.zzz);
    r.inf = false;
    return r;
}

template <class C, class M = MulInline> BP_HD XyzzLazy<C> xyzz_lazy_add(const XyzzLazy<C>& a, const XyzzLazy<C>& b) {
    using Fp = typename C::Fp;
    if (b.inf) return a;
    if (a.inf) return b;
    FeB<Fp, 2> U1 = M::mul(a.x, b.zz);
    FeB<Fp, 2> U2 = M::mul(b.x, a.zz);
    FeB<Fp, 2> S1 = M::mul(a.y, b.zzz);
    FeB<Fp, 2> S2 = M::mul(b.y, a.zzz);
    FeB<Fp, 4> Pp = feb_sub<2>(U2, U1);
    FeB<Fp, 4> Rr = feb_sub<2>(S2, S1);
    if (feb_is_zero_mod_p(Pp)) {
        if (feb_is_zero_mod_p(Rr)) return xyzz_lazy_dbl<C, M>(a);
        return xyzz_lazy_inf<C>();
    }
    FeB<Fp, 2> PP = M::sqr(Pp);
    FeB<Fp, 2> PPP = M::mul(Pp, PP);
    FeB<Fp, 2> Q = M::mul(U1, PP);
    FeB<Fp, 8> X3 = feb_sub<2>(feb_sub<2>(feb_sub<2>(M::sqr(Rr), PPP), Q), Q);
    FeB<Fp, 10> QX = feb_sub<8>(Q, X3);
    XyzzLazy<C> r;
    r.y = feb_widen<4>(M::mul_add_mul(Rr, QX, feb_neg<2>(S1), PPP));     // 4*10 + 2*2
    r.x = X3;
    r.zz = M::mul(M::mul(a.zz, b.zz), PP);
    r.zzz = M::mul(M::mul(a.zzz, b.zzz), PPP);
    r.inf = false;
    return r;
}

// packed memory form of a lazy point (identity <=> ZZ = 0); values below 2^(32 NW) as explained above
template <class C> BP_HD XyzzPacked<C> xyzz_lazy_pack(const XyzzLazy<C>& p) {
    using Fp = typename C::Fp;
    XyzzPacked<C> r;
    if (p.inf) {
        for (int i = 0; i < Fp::NW; i++) { r.x.w[i] = 0; r.y.w[i] = 0; r.zz.w[i] = 0; r.zzz.w[i] = 0; }
        return r;
    }
    constexpr bool kFits8p = Fp::BITS + 3 <= 32 * Fp::NW;
    Fe<Fp> x;
    for (int i = 0; i < Fp::NL; i++) x.v[i] = p.x.v[i];
    if (!kFits8p) {   // one conditional subtraction of 4p
        uint32_t d[Fp::NL];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < Fp::NL; i++) { uint32_t t = x.v[i] - Lazy<Fp>::L.kp[2][i] - br; br = t >> 31; d[i] = t & LMASK; }
        const uint32_t keep = 0u - br;   // all ones: x < 4p, keep x
#pragma unroll
        for (int i = 0; i < Fp::NL; i++) x.v[i] = (x.v[i] & keep) | (d[i] & ~keep);
    }
    fe_pack_words<Fp>(r.x.w, x);
    Fe<Fp> t;
    for (int i = 0; i < Fp::NL; i++) t.v[i] = p.y.v[i];
    fe_pack_words<Fp>(r.y.w, t);
    for (int i = 0; i < Fp::NL; i++) t.v[i] = p.zz.v[i];
    fe_pack_words<Fp>(r.zz.w, t);
    for (int i = 0; i < Fp::NL; i++) t.v[i] = p.zzz.v[i];
    fe_pack_words<Fp>(r.zzz.w, t);
    // a non-identity point whose ZZ happens to be = 0 mod p cannot occur (ZZ is a product of non-zero differences), but its
    // packed words could all be zero only if ZZ = 0 exactly, which the bounded domain never produces for a finite point
    return r;
}

template <class C> BP_HD XyzzLazy<C> xyzz_lazy_unpack(const XyzzPacked<C>& p) {
    using Fp = typename C::Fp;
    XyzzLazy<C> r;
    Fe<Fp> x = fe_unpack_words<Fp>(p.x.w), y = fe_unpack_words<Fp>(p.y.w), zz = fe_unpack_words<Fp>(p.zz.w), zzz = fe_unpack_words<Fp>(p.zzz.w);
    uint32_t any = 0;
    for (int i = 0; i < Fp::NL; i++) { r.x.v[i] = x.v[i]; r.y.v[i] = y.v[i]; r.zz.v[i] = zz.v[i]; r.zzz.v[i] = zzz.v[i]; any |= zz.v[i]; }
    r.inf = any == 0;
    return r;
}

// ----------------------------------------------------------------------------------------------- one addition on FOUR lanes
// The tree sums that end every latency-bound kernel (k_small_msm, the heavy-bucket combine) add ever fewer points per level while
// the block's other lanes idle, and a dependent addition is ~5 300 wave instructions (17 us at one wave per SIMD) whether one lane
// or 64 run it.  Here the four lanes of a quad share ONE addition slot[ia] += slot[ib] of packed lazy points in LDS: the 14 products
// of add-2008-s run as four rounds of one product per lane,
//     round 1   U1 = X1 ZZ2        U2 = X2 ZZ1         S1 = Y1 ZZZ2        S2 = Y2 ZZZ1
//     round 2   PP = P^2           RR = R^2            ZZ12 = ZZ1 ZZ2      ZZZ12 = ZZZ1 ZZZ2          (P = U2 - U1, R = S2 - S1)
//     round 3   PPP = P PP         Q = U1 PP           ZZ3 = ZZ12 PP       --
//     round 4   ZZZ3 = ZZZ12 PPP   T2 = (-S1) PPP      T1 = R (Q - X3)     --                          (X3 = RR - PPP - 2Q, Y3 = T1 + T2)
// with results handed between lanes by quad-permute DPP moves and per-lane operands chosen by bit-select with lane masks: ~1 900
// wave instructions instead of ~5 300.  Same bounded domain as xyzz_lazy_add (every product within kMaxProd, X3 < 8p, Y3 < 4p,
// ZZ3, ZZZ3 < 2p), same results mod p; the rare cases (an identity operand, P = 0: doubling or cancellation) are decided per quad.
// All four lanes of the quad must be active and agree on (ia, ib).  Device only.
#if defined(__HIPCC__)
// lane l of a quad reads lane quad_perm[l] (CTRL = the four 2-bit selectors).  The empty asm pins the result in a VGPR: without it the
// compiler's DPP combiner folds the move into the consuming add / sub and -- when both operands of a subtraction are DPP moves of the
// same register -- produces wrong lanes (hipcc 7.2, gfx950; probe: perm[1,3,1,3](x) - perm[0,2,0,2](x) gave 3 0 -3 -6 for 3 3 3 3).
template <int CTRL> __device__ __forceinline__ uint32_t quad_perm_u32(uint32_t v) {
    int r = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
    asm volatile("" : "+v"(r));
    return (uint32_t)r;
}
template <int CTRL, class P, int B> __device__ __forceinline__ FeB<P, B> feb_quad_perm(const FeB<P, B>& a) {
    FeB<P, B> r;
#pragma unroll
    for (int i = 0; i < P::NL; i++) r.v[i] = quad_perm_u32<CTRL>(a.v[i]);
    return r;
}
// m = all ones: a, m = 0: b (per lane)
template <int B, class P, int B1, int B2> __device__ __forceinline__ FeB<P, B> feb_select(uint32_t m, const FeB<P, B1>& a, const FeB<P, B2>& b) {
    static_assert(B >= B1 && B >= B2, "the selection carries the larger bound");
    FeB<P, B> r;
#pragma unroll
    for (int i = 0; i < P::NL; i++) r.v[i] = (a.v[i] & m) | (b.v[i] & ~m);
    return r;
}
constexpr int kQuadBcast0 = 0x00, kQuadBcast1 = 0x55, kQuadBcast2 = 0xaa, kQuadBcast3 = 0xff;   // quad_perm:[k,k,k,k]
constexpr int kQuadPerm1313 = 1 | (3 << 2) | (1 << 4) | (3 << 6), kQuadPerm0202 = 0 | (2 << 2) | (0 << 4) | (2 << 6);

template <class C>
__device__ __forceinline__ void xyzz_lazy_add_quad(XyzzPacked<C>* slots, int ia, int ib, int q) {
    using Fp = typename C::Fp;
    constexpr int NW = Fp::NW;
    uint32_t* wa = (uint32_t*)&slots[ia];
    const uint32_t* wb = (const uint32_t*)&slots[ib];
    uint32_t m0 = q == 0 ? ~0u : 0u, m1 = q == 1 ? ~0u : 0u, m2 = q == 2 ? ~0u : 0u, m01 = q < 2 ? ~0u : 0u;
    // (opaque to the compiler: it otherwise recognises the selections below and emits v_cndmask_b32, ~5x the issue cost of the and / or pair)
    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m01));
    // fields: 0 = X, 1 = Y, 2 = ZZ, 3 = ZZZ.  Round-1 operands of lane q, and the round-2 operands of lanes 2, 3 (lanes 0, 1 load lane 2's)
    const int fa1 = q == 0 ? 0 : q == 1 ? 2 : q == 2 ? 1 : 3;          // of a:  X1   ZZ1   Y1    ZZZ1
    const int fb1 = q == 0 ? 2 : q == 1 ? 0 : q == 2 ? 3 : 1;          // of b:  ZZ2  X2    ZZZ2  Y2
    const int f2 = q == 3 ? 3 : 2;                                     // ZZ (lanes 0 .. 2), ZZZ (lane 3)
    FeB<Fp, 8> A1, B1;
    FeB<Fp, 2> A2, B2;
    { Fe<Fp> t = fe_unpack_words<Fp>(wa + fa1 * NW); for (int i = 0; i < Fp::NL; i++) A1.v[i] = t.v[i]; }
    { Fe<Fp> t = fe_unpack_words<Fp>(wb + fb1 * NW); for (int i = 0; i < Fp::NL; i++) B1.v[i] = t.v[i]; }
    { Fe<Fp> t = fe_unpack_words<Fp>(wa + f2 * NW); for (int i = 0; i < Fp::NL; i++) A2.v[i] = t.v[i]; }
    { Fe<Fp> t = fe_unpack_words<Fp>(wb + f2 * NW); for (int i = 0; i < Fp::NL; i++) B2.v[i] = t.v[i]; }
    // identity operands (ZZ = 0 exactly): lanes 0 .. 2 hold ZZ1, ZZ2 in A2, B2
    uint32_t za = 0, zb = 0;
#pragma unroll
    for (int i = 0; i < Fp::NL; i++) { za |= A2.v[i]; zb |= B2.v[i]; }
    za = quad_perm_u32<kQuadBcast0>(za);
    zb = quad_perm_u32<kQuadBcast0>(zb);
    if (zb == 0) return;                                               // a + identity
    if (za == 0) {                                                     // identity + b: lane q copies field q
        for (int i = 0; i < NW; i++) wa[q * NW + i] = wb[q * NW + i];
        return;
    }
    // ---- round 1
    const FeB<Fp, 2> r1 = feb_mul(A1, B1);                             // U1 | U2 | S1 | S2     (A1 is X or Y (< 8p), or ZZ / ZZZ; B1 likewise: <= 8 * 8)
    const FeB<Fp, 2> hi = feb_quad_perm<kQuadPerm1313>(r1), lo = feb_quad_perm<kQuadPerm0202>(r1);
    const FeB<Fp, 4> D = feb_sub<2>(hi, lo);                           // lanes 0, 2: P = U2 - U1; lanes 1, 3: R = S2 - S1
    uint32_t dz = feb_is_zero_mod_p(D) ? 1u : 0u;
    const uint32_t pz = quad_perm_u32<kQuadBcast0>(dz);
    if (pz) {                                                          // same x: doubling or cancellation -- lane 0 alone, the one-lane formulas
        if (q == 0) slots[ia] = xyzz_lazy_pack(xyzz_lazy_add(xyzz_lazy_unpack(slots[ia]), xyzz_lazy_unpack(slots[ib])));
        return;
    }
    // ---- round 2
    const FeB<Fp, 4> a2 = feb_select<4>(m01, D, A2), b2 = feb_select<4>(m01, D, B2);
    const FeB<Fp, 2> r2 = feb_mul(a2, b2);                             // PP | RR | ZZ12 | ZZZ12
    const FeB<Fp, 2> PP = feb_quad_perm<kQuadBcast0>(r2);
    // ---- round 3
    const FeB<Fp, 2> U1 = feb_quad_perm<kQuadBcast0>(r1);
    const FeB<Fp, 4> a3 = feb_select<4>(m0, D, feb_select<2>(m1, U1, r2));
    const FeB<Fp, 2> r3 = feb_mul(a3, PP);                             // PPP | Q | ZZ3 | (unused)
    const FeB<Fp, 2> PPP = feb_quad_perm<kQuadBcast0>(r3), Q = feb_quad_perm<kQuadBcast1>(r3), RR = feb_quad_perm<kQuadBcast1>(r2);
    const FeB<Fp, 8> X3 = feb_sub<2>(feb_sub<2>(feb_sub<2>(RR, PPP), Q), Q);
    const FeB<Fp, 10> QX = feb_sub<8>(Q, X3);
    // ---- round 4
    const FeB<Fp, 2> Z12 = feb_quad_perm<kQuadBcast3>(r2), S1 = feb_quad_perm<kQuadBcast2>(r1);
    const FeB<Fp, 4> R = feb_quad_perm<kQuadBcast1>(D);
    const FeB<Fp, 4> a4 = feb_select<4>(m0, Z12, feb_select<4>(m1, feb_neg<2>(S1), R));
    const FeB<Fp, 10> b4 = feb_select<10>(m2, QX, PPP);
    const FeB<Fp, 2> r4 = feb_mul(a4, b4);                             // ZZZ3 | T2 | T1 | (unused)
    const FeB<Fp, 2> T2 = feb_quad_perm<kQuadBcast1>(r4);
    const FeB<Fp, 4> Y3 = feb_add(r4, T2);                             // lane 2: T1 + T2
    // ---- the result, field by field: lane 0 X3 and ZZZ3, lane 2 Y3 and ZZ3
    if (q == 0) {
        XyzzLazy<C> t;                                                 // (packing X needs the conditional 4p of xyzz_lazy_pack: reuse it on a full record)
        t.x = X3; t.y = feb_widen<4>(r4); t.zz = r4; t.zzz = r4; t.inf = false;
        const XyzzPacked<C> pk = xyzz_lazy_pack(t);
        for (int i = 0; i < NW; i++) { wa[i] = pk.x.w[i]; wa[3 * NW + i] = pk.zzz.w[i]; }
    } else if (q == 2) {
        Fe<Fp> t;
        for (int i = 0; i < Fp::NL; i++) t.v[i] = Y3.v[i];
        fe_pack_words<Fp>(wa + NW, t);
        for (int i = 0; i < Fp::NL; i++) t.v[i] = r3.v[i];
        fe_pack_words<Fp>(wa + 2 * NW, t);
    }
}

// One DOUBLING on four lanes (round 4; the Horner chains of the generator compaction, bp_compact.cuh): slots[ia] = 2 slots[ia].
// dbl-2008-s-1 with W = U V never formed (W Y = (U Y) V, W ZZZ = (U ZZZ) V), so that the ten products have depth THREE:
//     round 1   V = U^2            X2 = X^2          A = U ZZZ         B = U Y                 (U = 2Y)
//     round 2   S = X V            MM = M^2          ZZZ3 = A V        BV = B V                (M = 3 X2)
//     round 3   T1 = M (S - X3)    --                ZZ3 = V ZZ        --                      (X3 = MM - 2S, Y3 = T1 - BV)
// Bounds as in xyzz_lazy_dbl: U < 8p (Y < 4p), M < 6p, X3 < 6p, S - X3 + 8p < 10p; every product <= 8 * 8; results X3 < 8p, Y3 < 4p,
// ZZ3, ZZZ3 < 2p.  ~1 450 wave instructions against ~3 400 for the one-lane doubling.  The identity stays the identity; y = 0
// cannot occur (odd group order).  All four lanes of the quad must be active.
template <class C>
__device__ __forceinline__ void xyzz_lazy_dbl_quad(XyzzPacked<C>* slots, int ia, int q) {
    using Fp = typename C::Fp;
    constexpr int NW = Fp::NW;
    uint32_t* wa = (uint32_t*)&slots[ia];
    uint32_t m0 = q == 0 ? ~0u : 0u, m1 = q == 1 ? ~0u : 0u;
    asm volatile("" : "+v"(m0), "+v"(m1));
    const int f1 = q == 1 ? 0 : 1;                                     // first operand:   Y   X   Y     Y
    const int f2 = q == 0 ? 0 : q == 2 ? 3 : f1;                       // second operand:  X   X   ZZZ   Y
    FeB<Fp, 8> L1, L2, U;
    FeB<Fp, 2> ZZ;
    { Fe<Fp> t = fe_unpack_words<Fp>(wa + f1 * NW); for (int i = 0; i < Fp::NL; i++) L1.v[i] = t.v[i]; }
    { Fe<Fp> t = fe_unpack_words<Fp>(wa + f2 * NW); for (int i = 0; i < Fp::NL; i++) L2.v[i] = t.v[i]; }
    { Fe<Fp> t = fe_unpack_words<Fp>(wa + 2 * NW); for (int i = 0; i < Fp::NL; i++) ZZ.v[i] = t.v[i]; }
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < Fp::NL; i++) nz |= ZZ.v[i];
    if (nz == 0) return;                                               // 2 * identity (every lane of the quad read the same ZZ)
    {   // U = 2Y on the lanes that loaded Y (< 4p, so U < 8p); lane 1 doubles its X here and never uses the result
        const FeB<Fp, 16> d = feb_add(L1, L1);
        for (int i = 0; i < Fp::NL; i++) U.v[i] = d.v[i];
    }
    // ---- round 1
    const FeB<Fp, 8> a1 = feb_select<8>(m1, L1, U), b1 = feb_select<8>(m0, U, L2);
    const FeB<Fp, 2> r1 = feb_mul(a1, b1);                             // V | X2 | A | B
    const FeB<Fp, 2> V = feb_quad_perm<kQuadBcast0>(r1);
    const FeB<Fp, 6> M3 = feb_add(feb_add(r1, r1), r1);                // lane 1: M = 3 X^2
    // ---- round 2
    const FeB<Fp, 8> a2 = feb_select<8>(m0, L2, feb_select<6>(m1, M3, r1));
    const FeB<Fp, 6> b2 = feb_select<6>(m1, M3, V);
    const FeB<Fp, 2> r2 = feb_mul(a2, b2);                             // S | MM | ZZZ3 | BV
    const FeB<Fp, 2> S = feb_quad_perm<kQuadBcast0>(r2), MM = feb_quad_perm<kQuadBcast1>(r2);
    const FeB<Fp, 6> X3 = feb_sub<2>(feb_sub<2>(MM, S), S);
    const FeB<Fp, 10> SX = feb_sub<8>(S, X3);
    // ---- round 3
    const FeB<Fp, 6> Mb = feb_quad_perm<kQuadBcast1>(M3);
    const FeB<Fp, 6> a3 = feb_select<6>(m0, Mb, V);
    const FeB<Fp, 10> b3 = feb_select<10>(m0, SX, ZZ);
    const FeB<Fp, 2> r3 = feb_mul(a3, b3);                             // T1 | -- | ZZ3 | --        (6 * 10)
    const FeB<Fp, 2> BV = feb_quad_perm<kQuadBcast3>(r2);
    const FeB<Fp, 4> Y3 = feb_sub<2>(r3, BV);
    // ---- the result: lane 0 X3 and Y3, lane 2 ZZ3 and ZZZ3
    if (q == 0) {
        XyzzLazy<C> t;                                                 // (packing X needs the conditional 4p of xyzz_lazy_pack)
        t.x = feb_widen<8>(X3); t.y = Y3; t.zz = r3; t.zzz = r3; t.inf = false;
        const XyzzPacked<C> pk = xyzz_lazy_pack(t);
        for (int i = 0; i < NW; i++) { wa[i] = pk.x.w[i]; wa[NW + i] = pk.y.w[i]; }
    } else if (q == 2) {
        Fe<Fp> t;
        for (int i = 0; i < Fp::NL; i++) t.v[i] = r3.v[i];
        fe_pack_words<Fp>(wa + 2 * NW, t);
        for (int i = 0; i < Fp::NL; i++) t.v[i] = r2.v[i];
        fe_pack_words<Fp>(wa + 3 * NW, t);
    }
}
#endif

}  // namespace bp
