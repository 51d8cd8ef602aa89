// bp_field.cuh -- prime-field arithmetic for BLS12-381 / AMCL-BN254, host + gfx950 device.
//
// What it replaces: the BIG/FP limb arithmetic that the reference reaches through
// amcl_wrapper::field_elem::FieldElement (`* + - negation square inverse`, /root/reference
// src/ipp.rs:113,116-117,179,182-183,297-298) and, one level down, the Fp arithmetic under every
// G1 operation.
//
// Design (MI355X, measured -- microbench/instr_rate.hip, microbench/fpmul_rate.hip):
//   * gfx950 issues v_mad_u64_u32 at half rate (~4.5 cyc / wave64) and EVERY carry instruction
//     (v_add_co/v_addc_co, v_lshl_add_u64) at that same half rate; only plain VOP2 ops are full rate.
//     A saturated 12x32-bit CIOS therefore spends as much time propagating carries as multiplying
//     (4.4e10 Fp-mul/s chip-wide), a column-wise mad+addc form 5.9e10.
//   * So: UNSATURATED 30-bit limbs in 32-bit VGPRs (13 limbs for the 381-bit Fp, 9 for the 254/255-bit
//     fields).  A column of up to 13 products of 30x30 bits fits a 64-bit accumulator, so a whole
//     column is a pure chain of v_mad_u64_u32 with zero carry instructions; one v_and + one 64-bit shift
//     per column extracts the limb.  6.6e10 Fp-mul/s chip-wide, and the least sensitive to occupancy.
//   * One element per lane, entirely in VGPRs.  No MFMA: in a*b both operands differ per lane, there
//     is no shared matrix operand to contract against (north_star: integer path, no MFMA).
//   * In HBM an element is PACKED: canonical residue (in Montgomery form, R = 2^(30 NL)) as NW
//     little-endian 32-bit words (12 words = 48 B for Fp381), so a G1 affine point is 96 B and is moved
//     with 16-byte vector loads.  pack()/unpack() convert at the load/store boundary (~2 ops/limb).
//
// Discipline ("strict"): every function takes and returns fully reduced elements (value in [0, p),
// every limb < 2^30).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BP_HD __host__ __device__ __forceinline__
#define BP_HD_NOINLINE __host__ __device__ __noinline__
#else
#define BP_HD inline
#define BP_HD_NOINLINE
#endif

namespace bp {

constexpr int LB = 30;                       // limb bits
constexpr uint32_t LMASK = (1u << LB) - 1;

// ----------------------------------------------------------------------------------------------- moduli
// Little-endian 32-bit words of each modulus (public parameters; cross-checked against
// tests/golden/curves.json by tests/test_capi_cpu.py through bp_curve_params()).

struct Bls381FpW {
    static constexpr int NW = 12, BITS = 381;
    static constexpr uint32_t MODW[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                          0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
};
struct Bls381FrW {
    static constexpr int NW = 8, BITS = 255;
    static constexpr uint32_t MODW[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
};
// AMCL "BN254" = Nogami curve, u = -(2^62 + 2^55 + 1)  
struct Bn254FpW {
    static constexpr int NW = 8, BITS = 254;
    static constexpr uint32_t MODW[8] = {0x00000013u, 0xa7000000u, 0x00000013u, 0x61210000u, 0x00000008u, 0xba344d80u, 0x40000001u, 0x25236482u};
};
struct Bn254FrW {
    static constexpr int NW = 8, BITS = 254;
    static constexpr uint32_t MODW[8] = {0x0000000du, 0xa1000000u, 0x00000010u, 0xff9f8000u, 0x00000007u, 0xba344d80u, 0x40000001u, 0x25236482u};
};

// ----------------------------------------------------------------------------------------------- constants
template <int NL>
struct LimbConsts {
    uint32_t mod[NL];   // p
    uint32_t one[NL];   // R mod p,   R = 2^(30 NL)
    uint32_t r2[NL];    // R^2 mod p
    uint32_t inv;       // -p^-1 mod 2^30
};

template <class W>
struct Field {
    using Words = W;
    static constexpr int NW = W::NW;                       // packed 32-bit words
    static constexpr int BITS = W::BITS;
    static constexpr int NL = (W::BITS + 2 + LB - 1) / LB;  // 30-bit limbs; R >= 4p
    static_assert(NL <= 15, "column accumulators would overflow 64 bits");
    static_assert(NL * LB <= NW * 32 + 32, "limb/word geometry");

    static constexpr LimbConsts<NL> make() {
        LimbConsts<NL> c{};
        for (int i = 0; i < NL; i++) {
            int bit = LB * i, w = bit / 32, o = bit % 32;
            uint64_t v = (w < NW ? (uint64_t)W::MODW[w] : 0) >> o;
            if (w + 1 < NW) v |= ((uint64_t)W::MODW[w + 1] << 32) >> o;
            c.mod[i] = (uint32_t)(v & LMASK);
        }
        uint32_t x = 1;
        for (int i = 0; i < 6; i++) x *= 2 - c.mod[0] * x;
        c.inv = (0u - x) & LMASK;
        // t = 2^k mod p by modular doubling
        uint32_t t[NL] = {};
        t[0] = 1;
        for (int k = 0; k < 2 * NL * LB; k++) {
            uint32_t carry = 0;
            for (int i = 0; i < NL; i++) { uint32_t v = (t[i] << 1) | carry; carry = v >> LB; t[i] = v & LMASK; }
            // t < 2p < 2^(30 NL): compare with p
            bool ge = true;
            for (int i = NL - 1; i >= 0; i--) { if (t[i] != c.mod[i]) { ge = t[i] > c.mod[i]; break; } }
            if (ge) { uint32_t br = 0; for (int i = 0; i < NL; i++) { uint32_t v = t[i] - c.mod[i] - br; br = v >> 31; t[i] = v & LMASK; } }
            if (k == NL * LB - 1) for (int i = 0; i < NL; i++) c.one[i] = t[i];
        }
        for (int i = 0; i < NL; i++) c.r2[i] = t[i];
        return c;
    }
    static constexpr LimbConsts<NL> C = make();
};

using Bls381Fp = Field<Bls381FpW>;
using Bls381Fr = Field<Bls381FrW>;
using Bn254Fp = Field<Bn254FpW>;
using Bn254Fr = Field<Bn254FrW>;

// ----------------------------------------------------------------------------------------------- element
template <class P>
struct Fe {
    uint32_t v[P::NL];
};

// Packed memory form: NW little-endian 32-bit words.
template <class P>
struct alignas(16) FePacked {
    uint32_t w[P::NW];
};

template <class P> BP_HD Fe<P> fe_zero() { Fe<P> r; for (int i = 0; i < P::NL; i++) r.v[i] = 0; return r; }
template <class P> BP_HD Fe<P> fe_one() { Fe<P> r; for (int i = 0; i < P::NL; i++) r.v[i] = P::C.one[i]; return r; }

template <class P> BP_HD bool fe_is_zero(const Fe<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) o |= a.v[i];
    return o == 0;
}

template <class P> BP_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// words (value < 2^(32 NW)) -> limbs.  No reduction.
template <class P> BP_HD Fe<P> fe_unpack_words(const uint32_t* w) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        const int bit = LB * i, k = bit / 32, o = bit % 32;
        uint64_t v = (k < P::NW ? (uint64_t)w[k] : 0);
        if (k + 1 < P::NW) v |= (uint64_t)w[k + 1] << 32;
        r.v[i] = (uint32_t)(v >> o) & LMASK;
    }
    return r;
}

// limbs (normalised) -> words
template <class P> BP_HD void fe_pack_words(uint32_t* w, const Fe<P>& a) {
#pragma unroll
    for (int k = 0; k < P::NW; k++) {
        const int bit = 32 * k, i = bit / LB, o = bit % LB;
        uint64_t v = (uint64_t)a.v[i] >> o;
        if (i + 1 < P::NL) v |= (uint64_t)a.v[i + 1] << (LB - o);
        if (i + 2 < P::NL && 2 * LB - o < 32) v |= (uint64_t)a.v[i + 2] << (2 * LB - o);
        w[k] = (uint32_t)v;
    }
}

template <class P> BP_HD Fe<P> fe_unpack(const FePacked<P>& p) { return fe_unpack_words<P>(p.w); }
template <class P> BP_HD FePacked<P> fe_pack(const Fe<P>& a) { FePacked<P> r; fe_pack_words<P>(r.w, a); return r; }

// a (< 2p, limbs normalised) -> a mod p
template <class P> BP_HD void fe_cond_sub_p(uint32_t* a) {
    uint32_t d[P::NL];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        uint32_t x = a[i] - P::C.mod[i] - br;
        br = x >> 31;
        d[i] = x & LMASK;
    }
#pragma unroll
    for (int i = 0; i < P::NL; i++) a[i] = br ? a[i] : d[i];
}

template <class P> BP_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        uint32_t x = a.v[i] + b.v[i] + c;
        r.v[i] = x & LMASK;
        c = x >> LB;
    }
    fe_cond_sub_p<P>(r.v);   // a + b < 2p < 2^(30 NL): no carry out of the top limb
    return r;
}

template <class P> BP_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        uint32_t x = a.v[i] - b.v[i] - br;
        br = x >> 31;
        r.v[i] = x & LMASK;
    }
    uint32_t mask = 0u - br;   // negative -> add p back
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        uint32_t x = r.v[i] + (P::C.mod[i] & mask) + c;
        r.v[i] = x & LMASK;
        c = x >> LB;
    }
    return r;
}

template <class P> BP_HD Fe<P> fe_neg(const Fe<P>& a) { return fe_sub<P>(fe_zero<P>(), a); }
template <class P> BP_HD Fe<P> fe_dbl(const Fe<P>& a) { return fe_add<P>(a, a); }

// The column sums are written as chains  acc = a*b + acc  starting from the carry of the previous column, which is exactly one
// v_mad_u64_u32 per product.  Left alone, the optimiser re-associates each column into "products first, carry last" and pays
// one or two extra 64-bit additions per column (44 half-rate instructions per product).  Showing every partial sum to an
// empty volatile asm gives it a second use, which makes it a leaf for the re-association and pins the order; nothing is
// emitted.  Used in the LAZY multipliers only (the MSM kernels): in the big strict-arithmetic kernels (hash to G1) the pinned
// order raises register pressure until the allocator spills (k_clear_cofactor: 1781 scratch loads, 3x slower).
// (An asm that also DEFINES the accumulator works too but draws an s_nop per step from the gfx940+ hazard
// recogniser -- "assume inline asm has a dst forwarding hazard" -- which costs lone waves more than the additions saved.)
// The reduction pass adds the limb t[k] of the double-length product into its column.  As a plain 64-bit addition that is a
// v_mov (zero high word) + v_lshl_add_u64; as  t[k] * 1 + acc  with a 1 the optimiser cannot see through it is one more
// v_mad_u64_u32 on the same chain (BP_OPAQUE_ONE: a scalar register holding 1).
#if defined(__HIP_DEVICE_COMPILE__)
#define BP_KEEP_ORDER(acc) asm volatile("" ::"v"(acc))
#define BP_OPAQUE_ONE(one) asm("" : "+s"(one))
#else
#define BP_KEEP_ORDER(acc) ((void)0)
#define BP_OPAQUE_ONE(one) ((void)0)
#endif

// ----------------------------------------------------------------------------------------------- fused Montgomery core
// Round 3.  The product a*b and the reduction m*p are accumulated in the SAME 64-bit column wherever that provably fits, so a
// column costs one limb extraction (and + 64-bit shift) instead of two plus the "t[k] * 1" re-injection of the two-pass form:
//     column k:  acc = carry + sum_{i+j=k} a_i b_j + sum_{i<k} m_i p_{k-i};   m_k = acc * (-p^-1) mod 2^30;   acc += m_k p_0;   acc >>= 30
// Round 2 rejected this ("26 products of 30x30 bits overflow 64 bits"), which is the bound for a column with 13 + 13 FULL-SIZE
// terms.  The real bound is smaller: column k < N has k+1 terms of each kind, and the m*p terms are m_i (< 2^30) times the ACTUAL
// limbs of p (on average half as large).  Evaluated at compile time below (worst-case operand limbs, the modulus' own limbs):
// for the 381-bit Fp only columns 10, 11, 12 of 26 exceed 2^64; for the 9-limb fields (BN254 Fp, both Fr) NO column does.
// The few middle columns that do not fit run as two chains (product chain + reduction chain, exactly the two-pass arithmetic)
// and the chains are joined by one 64-bit addition.  Per 381-bit product: 23 x (mad + and + shift) - 1 add fewer.
// Same value as the two-pass form in every case: (a b [+ t1] + m p) / R with the same m.
template <class P>
struct Fuse {
    static constexpr int N = P::NL;
    // largest top limb of a normalised value < 32 p (the widest bound FeB allows)
    static constexpr uint64_t top_limb_bound() {
        uint32_t t[N] = {};
        for (int k = 0; k < 32; k++) {                       // t = 32 p by repeated addition, limbs normalised except the top one
            uint32_t carry = 0;
            for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)t[i] + P::C.mod[i] + carry; if (i < N - 1) { carry = (uint32_t)(x >> LB); t[i] = (uint32_t)x & LMASK; } else t[i] = (uint32_t)x; }
        }
        return t[N - 1];
    }
    static constexpr bool fits(int k) {
        const uint64_t top = top_limb_bound();
        unsigned __int128 tot = (unsigned __int128)1 << 36;          // carry in (< 2^34), an injected limb (< 2^31), slack
        for (int i = 0; i < N; i++) {
            const int j = k - i;
            if (j < 0 || j >= N) continue;
            const uint64_t ai = i == N - 1 ? top : (uint64_t)LMASK, bj = j == N - 1 ? top : (uint64_t)LMASK;
            tot += (unsigned __int128)ai * bj;                        // a_i b_j (a doubled cross term of a square is two of these)
            tot += (unsigned __int128)LMASK * P::C.mod[j];            // m_i p_j
        }
        return tot < ((unsigned __int128)1 << 64);
    }
    static constexpr int lo() { for (int k = 0; k < 2 * N; k++) if (!fits(k)) return k; return 2 * N; }          // first column that does not fit
    static constexpr int hi() { for (int k = 2 * N - 1; k >= 0; k--) if (!fits(k)) return k; return -1; }       // last one
    static_assert(top_limb_bound() <= LMASK, "32 p must fit the limb vector");
};

// a*b terms of column k into acc.  SQR: b is ignored, a2 = 2a (cross terms once against the doubled operand).
// (Loops are written with exact bounds: a full-range loop with a condition inside blows the unroller's size budget before the
// outer column loop is unrolled and k becomes a constant.)
template <class P, bool SQR, bool PIN> BP_HD void mont_ab_terms(uint64_t& acc, const int k, const uint32_t* a, const uint32_t* b, const uint32_t* a2) {
    constexpr int N = P::NL;
    if (SQR) {
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); 2 * i < k; i++) { acc += (uint64_t)a2[i] * a[k - i]; if (PIN) BP_KEEP_ORDER(acc); }
        if ((k & 1) == 0 && k / 2 < N) { acc += (uint64_t)a[k / 2] * a[k / 2]; if (PIN) BP_KEEP_ORDER(acc); }
    } else {
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); i++) { acc += (uint64_t)a[i] * b[k - i]; if (PIN) BP_KEEP_ORDER(acc); }
    }
}
// m*p terms of column k that involve ALREADY KNOWN m (i < k for k < N; i = k-N+1 .. N-1 above)
template <class P, bool PIN> BP_HD void mont_mp_terms(uint64_t& acc, const int k, const uint32_t* m) {
    constexpr int N = P::NL;
#pragma unroll
    for (int i = (k < N ? 0 : k - N + 1); i < (k < N ? k : N); i++) { acc += (uint64_t)m[i] * P::C.mod[k - i]; if (PIN) BP_KEEP_ORDER(acc); }
}
// one fused column: everything into acc, then m_k (k < N) or the result limb (k >= N), then the carry
template <class P, bool SQR, bool ADD, bool PIN>
BP_HD void mont_fused_column(uint64_t& acc, const int k, const uint32_t* a, const uint32_t* b, const uint32_t* a2, const uint32_t* t1, uint32_t one, uint32_t* m, uint32_t* r) {
    constexpr int N = P::NL;
    mont_ab_terms<P, SQR, PIN>(acc, k, a, b, a2);
    if (ADD) { acc += (uint64_t)t1[k] * one; if (PIN) BP_KEEP_ORDER(acc); }
    mont_mp_terms<P, PIN>(acc, k, m);
    if (k < N) {
        m[k] = ((uint32_t)acc * P::C.inv) & LMASK;
        { acc += (uint64_t)m[k] * P::C.mod[0]; if (PIN) BP_KEEP_ORDER(acc); }
    } else {
        r[k - N] = (uint32_t)acc & LMASK;
    }
    acc >>= LB;
}

// r = (a b [+ t1] + m p) / R, limbs normalised, NO final subtraction (value < a b / R + t1 / R + p).
//   SQR  a*a with 91 instead of 169 product terms          ADD  t1 = 2N limbs (each < 2^31) added to the product
//   PIN  pin the order of every mad chain (BP_KEEP_ORDER; the lazy multipliers) or leave it to the compiler (strict functions)
template <class P, bool SQR, bool ADD, bool PIN> BP_HD void mont_core(const uint32_t* a, const uint32_t* b, const uint32_t* t1, uint32_t* r) {
    constexpr int N = P::NL;
    constexpr bool kSplit = Fuse<P>::lo() <= Fuse<P>::hi();
    constexpr int LO = kSplit ? Fuse<P>::lo() : 2 * N, HI = kSplit ? Fuse<P>::hi() : 2 * N - 1;   // no middle block: the first loop runs over every column
    uint32_t a2[N];
    if (SQR) {
#pragma unroll
        for (int i = 0; i < N; i++) a2[i] = a[i] << 1;
    }
    uint32_t m[N];
    uint32_t one = 1;
    if (ADD || kSplit) BP_OPAQUE_ONE(one);
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < LO; k++) mont_fused_column<P, SQR, ADD, PIN>(acc, k, a, b, a2, t1, one, m, r);
    if (kSplit) {
        // ---- middle block: product chain stays in acc, reduction chain in accB (the two-pass arithmetic) ----
        uint64_t accB = 0;
#pragma unroll
        for (int q = LO; q <= HI; q++) {
            mont_ab_terms<P, SQR, PIN>(acc, q, a, b, a2);
            if (ADD) { acc += (uint64_t)t1[q] * one; if (PIN) BP_KEEP_ORDER(acc); }
            const uint32_t tq = (uint32_t)acc & LMASK;
            acc >>= LB;
            { accB += (uint64_t)tq * one; if (PIN) BP_KEEP_ORDER(accB); }
            mont_mp_terms<P, PIN>(accB, q, m);
            if (q < N) {
                m[q] = ((uint32_t)accB * P::C.inv) & LMASK;
                { accB += (uint64_t)m[q] * P::C.mod[0]; if (PIN) BP_KEEP_ORDER(accB); }
            } else {
                r[q - N] = (uint32_t)accB & LMASK;
            }
            accB >>= LB;
        }
        acc += accB;                                           // both carries enter column HI + 1
#pragma unroll
        for (int k = HI + 1; k < 2 * N; k++) mont_fused_column<P, SQR, ADD, PIN>(acc, k, a, b, a2, t1, one, m, r);
    }
}

// Montgomery reduction of 2*NL normalised limbs t (value < p * R) -> t / R mod p, product scanning.
template <class P> BP_HD Fe<P> fe_mont_reduce(const uint32_t* t) {
    constexpr int N = P::NL;
    uint32_t m[N];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        acc += t[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::C.mod[k - i];
        m[k] = ((uint32_t)acc * P::C.inv) & LMASK;
        acc += (uint64_t)m[k] * P::C.mod[0];
        acc >>= LB;
    }
    Fe<P> r;
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
        acc += t[k];
#pragma unroll
        for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * P::C.mod[k - i];
        r.v[k - N] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    fe_cond_sub_p<P>(r.v);
    return r;
}

// Montgomery product a*b/R mod p (fused core, compiler-scheduled chains).
template <class P> BP_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    mont_core<P, false, false, false>(a.v, b.v, nullptr, r.v);
    fe_cond_sub_p<P>(r.v);   // a b / R + p < 2p
    return r;
}

// Montgomery square: cross products taken once against a doubled operand (2 a_i < 2^31).
template <class P> BP_HD Fe<P> fe_sqr(const Fe<P>& a) {
    Fe<P> r;
    mont_core<P, true, false, false>(a.v, a.v, nullptr, r.v);
    fe_cond_sub_p<P>(r.v);
    return r;
}

template <class P> BP_HD Fe<P> fe_to_mont(const Fe<P>& raw) {
    Fe<P> r2;
#pragma unroll
    for (int i = 0; i < P::NL; i++) r2.v[i] = P::C.r2[i];
    return fe_mul<P>(raw, r2);
}

template <class P> BP_HD Fe<P> fe_from_mont(const Fe<P>& a) {
    uint32_t t[2 * P::NL];
#pragma unroll
    for (int i = 0; i < P::NL; i++) { t[i] = a.v[i]; t[P::NL + i] = 0; }
    return fe_mont_reduce<P>(t);
}

// a^(p-2); inverse of 0 is 0.  Long and cold: kept out of line.
template <class P> BP_HD_NOINLINE Fe<P> fe_inv(const Fe<P>& a) {
    uint32_t e[P::NW];
    for (int i = 0; i < P::NW; i++) e[i] = P::Words::MODW[i];
    {   // e = p - 2 (BLS12-381's r ends in ...00000001, so the borrow can run)
        uint32_t br = 2;
        for (int i = 0; i < P::NW && br; i++) { uint32_t o = e[i]; e[i] = o - br; br = o < br; }
    }
    Fe<P> acc = fe_one<P>();
    for (int i = P::BITS - 1; i >= 0; i--) {
        acc = fe_sqr<P>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul<P>(acc, a);
    }
    return acc;
}

// canonical words < p ?
template <class P> BP_HD bool words_lt_mod(const uint32_t* a) {
    for (int i = P::NW - 1; i >= 0; i--) {
        if (a[i] < P::Words::MODW[i]) return true;
        if (a[i] > P::Words::MODW[i]) return false;
    }
    return false;
}

}  // namespace bp

namespace bp {

// =============================================================================================================
// Lazy ("bounded") arithmetic for hot loops.
//
// FeB<P, B> is a field element whose limbs are normalised (< 2^30) and whose VALUE is < B * p; B is part of the
// type, so every bound below is checked by the compiler.  No conditional subtraction happens anywhere:
//   mul / sqr : inputs < B1 p, B2 p with B1 * B2 <= kMaxProd  ->  output < 2p
//               ((x y + m p) / R < p (B1 B2 p / R + 1) and p / R < 2^-7 for every supported field; kMaxProd = 128 keeps
//               B1 B2 p / R < 1, hence output < 2p.)
//   add       : limb-wise add + one carry pass                         ->  < (B1 + B2) p
//   sub<K>    : a - b + K p with K >= B2 (K p precomputed in limbs)     ->  < (B1 + K) p
// All values stay < 2^(30 NL - 2) (B <= 32), so the top limb never overflows.  Leaving the lazy domain:
// to_strict() (binary conditional subtraction down to [0, p)).  is_zero_mod_p() tests x == 0 (mod p) for a bounded x.
// The strict functions above remain the reference semantics; tests compare the two on the golden vectors and on
// random data through the MSM kernels.
// =============================================================================================================
template <class P, int B>
struct FeB {
    static_assert(B >= 1 && B <= 32, "bound out of range");
    uint32_t v[P::NL];
};

constexpr int kMaxProd = 128;

template <class P>
struct LazyConsts {
    // k * p in normalised limbs for k = 1, 2, 4, 8, 16, 32
    uint32_t kp[6][P::NL];
    uint32_t pinv;   // p[0]^-1 mod 2^30  (for the multiple-of-p test)
};

template <class P>
constexpr LazyConsts<P> make_lazy() {
    LazyConsts<P> c{};
    for (int i = 0; i < P::NL; i++) c.kp[0][i] = P::C.mod[i];
    for (int k = 1; k < 6; k++) {
        uint32_t carry = 0;
        for (int i = 0; i < P::NL; i++) { uint32_t x = (c.kp[k - 1][i] << 1) | carry; carry = x >> LB; c.kp[k][i] = x & LMASK; }
    }
    uint32_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - P::C.mod[0] * x;
    c.pinv = x & LMASK;
    return c;
}
template <class P> struct Lazy { static constexpr LazyConsts<P> L = make_lazy<P>(); };

// k * p in normalised limbs for every k = 0 .. 32 (only the rows a kernel names become literals in its code): lets a subtraction
// add exactly the multiple its subtrahend needs (6p for PPP + 2Q) instead of the next power of two, which keeps the bounds of the
// curve formulas where they were when several subtrahends share one pass.
template <class P>
struct MultConsts { uint32_t mp[33][P::NL]; };
template <class P>
constexpr MultConsts<P> make_multiples() {
    MultConsts<P> c{};
    for (int k = 1; k <= 32; k++) {
        uint32_t carry = 0;
        for (int i = 0; i < P::NL; i++) { uint32_t x = c.mp[k - 1][i] + P::C.mod[i] + carry; carry = x >> LB; c.mp[k][i] = x & LMASK; }
    }
    return c;
}
template <class P> struct Mult { static constexpr MultConsts<P> M = make_multiples<P>(); };
static_assert(Bls381Fp::NL * LB - Bls381Fp::BITS >= 7 && Bls381Fr::NL * LB - Bls381Fr::BITS >= 7 && Bn254Fp::NL * LB - Bn254Fp::BITS >= 7,
              "p / R < 2^-7 is assumed by the lazy multiplication bound");

constexpr int log2_ceil_pow2(int k) { return k <= 1 ? 0 : k <= 2 ? 1 : k <= 4 ? 2 : k <= 8 ? 3 : k <= 16 ? 4 : 5; }

template <class P> BP_HD FeB<P, 1> feb_from_strict(const Fe<P>& a) { FeB<P, 1> r; for (int i = 0; i < P::NL; i++) r.v[i] = a.v[i]; return r; }
// widen the static bound (no-op on the data)
template <int B2, class P, int B1> BP_HD FeB<P, B2> feb_widen(const FeB<P, B1>& a) {
    static_assert(B2 >= B1, "cannot narrow a bound");
    FeB<P, B2> r;
    for (int i = 0; i < P::NL; i++) r.v[i] = a.v[i];
    return r;
}

template <class P, int B1, int B2> BP_HD FeB<P, 2> feb_mul(const FeB<P, B1>& a, const FeB<P, B2>& b) {
    static_assert(B1 * B2 <= kMaxProd, "operands too large for a lazy Montgomery product");
    FeB<P, 2> r;
    mont_core<P, false, false, true>(a.v, b.v, nullptr, r.v);
    return r;
}

// a1 b1 + a2 b2 with ONE Montgomery reduction: the two double-length products are added limb-wise (each limb < 2^31) and
// reduced together, saving one reduction pass (~45 % of a product).  Bound: (B1 B2 + B3 B4) p / R + 1 < 2 for kMaxProd.
template <class P, int B1, int B2, int B3, int B4>
BP_HD FeB<P, 2> feb_mul_add_mul(const FeB<P, B1>& a1, const FeB<P, B2>& b1, const FeB<P, B3>& a2, const FeB<P, B4>& b2) {
    static_assert(B1 * B2 + B3 * B4 <= kMaxProd, "operands too large for a shared lazy reduction");
    constexpr int N = P::NL;
    uint32_t t[2 * N];                      // a1 b1 as 2N normalised limbs; the fused core adds them to a2 b2 column by column
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); i++) { acc += (uint64_t)a1.v[i] * b1.v[k - i]; BP_KEEP_ORDER(acc); }
        t[k] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    t[2 * N - 1] = (uint32_t)acc;
    FeB<P, 2> r;
    mont_core<P, false, true, true>(a2.v, b2.v, t, r.v);
    return r;
}

// K p - a  (K a power of two >= B): the negation inside the bounded domain
template <int K, class P, int B> BP_HD FeB<P, K> feb_neg(const FeB<P, B>& a) {
    static_assert(K >= B && (K & (K - 1)) == 0 && K <= 32, "K p must dominate the operand");
    constexpr int ki = log2_ceil_pow2(K);
    FeB<P, K> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        int32_t x = (int32_t)Lazy<P>::L.kp[ki][i] - (int32_t)a.v[i] + c;
        r.v[i] = (uint32_t)x & LMASK;
        c = x >> LB;
    }
    return r;
}

template <class P, int B1> BP_HD FeB<P, 2> feb_sqr(const FeB<P, B1>& a) {
    static_assert(B1 * B1 <= kMaxProd, "operand too large for a lazy Montgomery square");
    FeB<P, 2> r;
    mont_core<P, true, false, true>(a.v, a.v, nullptr, r.v);
    return r;
}

template <class P, int B1, int B2> BP_HD FeB<P, B1 + B2> feb_add(const FeB<P, B1>& a, const FeB<P, B2>& b) {
    FeB<P, B1 + B2> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        uint32_t x = a.v[i] + b.v[i] + c;
        r.v[i] = x & LMASK;
        c = x >> LB;
    }
    return r;
}

// a - b + K p,  K a power of two >= B2
template <int K, class P, int B1, int B2> BP_HD FeB<P, B1 + K> feb_sub(const FeB<P, B1>& a, const FeB<P, B2>& b) {
    static_assert(K >= B2 && (K & (K - 1)) == 0 && K <= 32, "K p must dominate the subtrahend");
    constexpr int ki = log2_ceil_pow2(K);
    FeB<P, B1 + K> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        int32_t x = (int32_t)(a.v[i] + Lazy<P>::L.kp[ki][i]) - (int32_t)b.v[i] + c;   // in (-2^30, 2^31 + 2)
        r.v[i] = (uint32_t)x & LMASK;
        c = x >> LB;                                                                   // arithmetic shift: signed carry
    }
    return r;
}

// a - b + K p for ANY K in [B2, 32] (feb_sub above takes powers of two only)
template <int K, class P, int B1, int B2> BP_HD FeB<P, B1 + K> feb_subk(const FeB<P, B1>& a, const FeB<P, B2>& b) {
    static_assert(K >= B2 && K <= 32 && B1 + K <= 32, "K p must dominate the subtrahend");
    FeB<P, B1 + K> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        int32_t x = (int32_t)(a.v[i] + Mult<P>::M.mp[K][i]) - (int32_t)b.v[i] + c;
        r.v[i] = (uint32_t)x & LMASK;
        c = x >> LB;
    }
    return r;
}

// a - b - c + K p in ONE carry pass, K >= B2 + B3 (every limb sum stays inside a signed 32-bit word: a + Kp < 2^31, b + c < 2^31)
template <int K, class P, int B1, int B2, int B3> BP_HD FeB<P, B1 + K> feb_sub2k(const FeB<P, B1>& a, const FeB<P, B2>& b, const FeB<P, B3>& c3) {
    static_assert(K >= B2 + B3 && K <= 32 && B1 + K <= 32, "K p must dominate both subtrahends");
    FeB<P, B1 + K> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        int32_t x = (int32_t)(a.v[i] + Mult<P>::M.mp[K][i]) - (int32_t)b.v[i] - (int32_t)c3.v[i] + c;
        r.v[i] = (uint32_t)x & LMASK;
        c = x >> LB;
    }
    return r;
}

// p - a for a CANONICAL a (0 <= a <= p - 1; a = 0 gives p, which is 0 mod p and inside the bound): the negation of an affine
// coordinate without the add-back pass of the strict fe_neg
template <class P> BP_HD FeB<P, 2> feb_neg_canonical(const Fe<P>& a) {
    FeB<P, 2> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) {
        int32_t x = (int32_t)P::C.mod[i] - (int32_t)a.v[i] + c;
        r.v[i] = (uint32_t)x & LMASK;
        c = x >> LB;
    }
    return r;
}

// x < B p  ->  canonical [0, p): conditional subtraction of 2^j p, j descending
template <class P, int B> BP_HD Fe<P> feb_to_strict(const FeB<P, B>& a) {
    Fe<P> r;
    for (int i = 0; i < P::NL; i++) r.v[i] = a.v[i];
#pragma unroll
    for (int j = log2_ceil_pow2(B) - 1; j >= 0; j--) {
        uint32_t d[P::NL];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < P::NL; i++) {
            uint32_t x = r.v[i] - Lazy<P>::L.kp[j][i] - br;
            br = x >> 31;
            d[i] = x & LMASK;
        }
#pragma unroll
        for (int i = 0; i < P::NL; i++) r.v[i] = br ? r.v[i] : d[i];
    }
    return r;
}

// x == 0 (mod p) for x < B p.  If x = j p then j = x[0] * p[0]^-1 mod 2^30, and j < B: a two-instruction filter that
// almost never passes for a non-multiple; the full limb comparison runs only when it does.
template <class P, int B> BP_HD bool feb_is_zero_mod_p(const FeB<P, B>& a) {
    uint32_t j = (a.v[0] * Lazy<P>::L.pinv) & LMASK;
    if (j >= (uint32_t)B) return false;
    Fe<P> s = feb_to_strict<P, B>(a);
    return fe_is_zero(s);
}

// ----------------------------------------------------------------------------------------------- out-of-line multiplier
// A fully inlined point addition is ~50 KB of straight-line code (14 products of ~3.5 KB).  The hot accumulate loop wants
// exactly that, but every SHORT kernel around it -- bucket reduce, tree sums, the one-launch small MSM, IPP folds -- executes
// each inlined copy only a handful of times, and on gfx950 the first pass through cold code is paced by instruction fetch
// from HBM (measured: 0.2-0.4 ms per launch that do not depend on n, profiles/r02_*), i.e. longer than the arithmetic.
// Those kernels use the multiplier as a real function (one 3.5 KB body per field, shared by every call site, resident in
// the instruction cache after its first use); the call costs a few dozen register moves per product.
template <class P> struct LimbsV { uint32_t v[P::NL]; };

template <class P> BP_HD_NOINLINE LimbsV<P> feb_mul_outlined(LimbsV<P> a, LimbsV<P> b) {
    FeB<P, 8> x, y;
    for (int i = 0; i < P::NL; i++) { x.v[i] = a.v[i]; y.v[i] = b.v[i]; }
    FeB<P, 2> r = feb_mul(x, y);          // the bound is the caller's business (checked in MulCall below)
    LimbsV<P> o;
    for (int i = 0; i < P::NL; i++) o.v[i] = r.v[i];
    return o;
}
template <class P> BP_HD_NOINLINE LimbsV<P> feb_sqr_outlined(LimbsV<P> a) {
    FeB<P, 8> x;
    for (int i = 0; i < P::NL; i++) x.v[i] = a.v[i];
    FeB<P, 2> r = feb_sqr(x);
    LimbsV<P> o;
    for (int i = 0; i < P::NL; i++) o.v[i] = r.v[i];
    return o;
}

// multiplier policies for the curve formulas: same bounds, same results.  The library uses MulInline everywhere; MulCall (one
// out-of-line multiplier, ~30 % slower per dependent step, far less code) exists for microbench/point_latency.hip's comparison
struct MulInline {
    template <class P, int B1, int B2> static BP_HD FeB<P, 2> mul(const FeB<P, B1>& a, const FeB<P, B2>& b) { return feb_mul(a, b); }
    template <class P, int B1> static BP_HD FeB<P, 2> sqr(const FeB<P, B1>& a) { return feb_sqr(a); }
    template <class P, int B1, int B2, int B3, int B4>
    static BP_HD FeB<P, 2> mul_add_mul(const FeB<P, B1>& a1, const FeB<P, B2>& b1, const FeB<P, B3>& a2, const FeB<P, B4>& b2) { return feb_mul_add_mul(a1, b1, a2, b2); }
};
struct MulCall {
    template <class P, int B1, int B2> static BP_HD FeB<P, 2> mul(const FeB<P, B1>& a, const FeB<P, B2>& b) {
        static_assert(B1 * B2 <= kMaxProd, "operands too large for a lazy Montgomery product");
        LimbsV<P> x, y;
        for (int i = 0; i < P::NL; i++) { x.v[i] = a.v[i]; y.v[i] = b.v[i]; }
        LimbsV<P> o = feb_mul_outlined<P>(x, y);
        FeB<P, 2> r;
        for (int i = 0; i < P::NL; i++) r.v[i] = o.v[i];
        return r;
    }
    template <class P, int B1> static BP_HD FeB<P, 2> sqr(const FeB<P, B1>& a) {
        static_assert(B1 * B1 <= kMaxProd, "operand too large for a lazy Montgomery square");
        LimbsV<P> x;
        for (int i = 0; i < P::NL; i++) x.v[i] = a.v[i];
        LimbsV<P> o = feb_sqr_outlined<P>(x);
        FeB<P, 2> r;
        for (int i = 0; i < P::NL; i++) r.v[i] = o.v[i];
        return r;
    }
    template <class P, int B1, int B2, int B3, int B4>
    static BP_HD FeB<P, 2> mul_add_mul(const FeB<P, B1>& a1, const FeB<P, B2>& b1, const FeB<P, B3>& a2, const FeB<P, B4>& b2) { return feb_mul_add_mul(a1, b1, a2, b2); }   // no out-of-line form: used once per addition
};

}  // namespace bp
