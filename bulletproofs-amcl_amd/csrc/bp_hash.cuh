// bp_hash.cuh -- hash-to-G1 on the device: one lane per message.
//
// Replaces amcl_wrapper `G1::from_msg_hash(msg)` = `GroupG1::mapit(&hash_msg(msg))`, the call behind
// `get_generators(prefix, n)` (reference src/utils/mod.rs:16-23) and the `g`, `h` of the gadget tests
// (e.g. src/r1cs/gadgets/bound_check.rs:200-203).  The crates are not vendored; the map is restated from the
// published amcl algorithm [UNVERIFIED-RECALL]; the tests check it against two independent CPU restatements:
//     h = SHAKE256(msg)[0 .. MODBYTES)            x = BE(h) mod p
//     loop:  rhs = x^3 + b;  found = rhs is a non-zero square;  y = the EVEN square root;  x += 1
//            if found: P = cofactor * (x_old, y);  if P != O: return P
// Both base fields have p = 3 (mod 4), so sqrt(rhs) = rhs^((p+1)/4) and one exponentiation answers both questions.
//
// Cost per point (BLS12-381): ~2 tries * ~570 Fp-mul (exponentiation) + 126 doublings and ~60 additions for the
// cofactor + one inversion (~570) -- ALU-bound like the rest of the library; SHAKE256 is one Keccak-f per message.
#pragma once
#include "bp_curve.cuh"

namespace bp {

constexpr int kHashBlock = 256;

__device__ __constant__ const uint64_t kKeccakRC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// Keccak-f[1600], state in 25 named-by-index registers (every index below is a compile-time constant after unrolling).
__device__ __forceinline__ void keccak_f1600_dev(uint64_t (&a)[25]) {
    constexpr int rho[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    constexpr int pi[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; r++) {
        uint64_t bc[5];
#pragma unroll
        for (int i = 0; i < 5; i++) bc[i] = a[i] ^ a[i + 5] ^ a[i + 10] ^ a[i + 15] ^ a[i + 20];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
#pragma unroll
            for (int j = 0; j < 25; j += 5) a[j + i] ^= t;
        }
        uint64_t t = a[1];
#pragma unroll
        for (int i = 0; i < 24; i++) {
            uint64_t b = a[pi[i]];
            a[pi[i]] = rotl64(t, rho[i]);
            t = b;
        }
#pragma unroll
        for (int j = 0; j < 25; j += 5) {
            uint64_t c0 = a[j], c1 = a[j + 1], c2 = a[j + 2], c3 = a[j + 3], c4 = a[j + 4];
            a[j] = c0 ^ (~c1 & c2);
            a[j + 1] = c1 ^ (~c2 & c3);
            a[j + 2] = c2 ^ (~c3 & c4);
            a[j + 3] = c3 ^ (~c4 & c0);
            a[j + 4] = c4 ^ (~c0 & c1);
        }
        a[0] ^= kKeccakRC[r];
    }
}

// A message = prefix bytes followed by tail bytes (either the decimal digits of a counter, or nothing).
struct HashMsg {
    const uint8_t* head;
    uint32_t head_len;
    uint8_t tail[20];
    uint32_t tail_len;
    __device__ __forceinline__ uint32_t len() const { return head_len + tail_len; }
    // byte j of the padded SHAKE256 input (suffix 0x1f at len; the closing 0x80 is xored in by the caller)
    __device__ __forceinline__ uint64_t padded(uint32_t j) const {
        if (j < head_len) return head[j];
        uint32_t k = j - head_len;
        if (k < tail_len) return tail[k < 20 ? k : 0];
        return k == tail_len ? 0x1f : 0;
    }
};

constexpr int kShakeRate = 136;

// SHAKE256(msg) -> the first 64 output bytes as 8 little-endian lanes (MODBYTES <= 48 is all anyone asks for).
__device__ __forceinline__ void shake256_dev(const HashMsg& m, uint64_t (&out)[8]) {
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = 0;
    uint32_t nblocks = m.len() / kShakeRate + 1;
    for (uint32_t blk = 0; blk < nblocks; blk++) {
        uint32_t base = blk * kShakeRate;
#pragma unroll
        for (int l = 0; l < kShakeRate / 8; l++) {
            uint64_t w = 0;
            for (int k = 0; k < 8; k++) w |= m.padded(base + 8 * l + k) << (8 * k);
            a[l] ^= w;
        }
        if (blk + 1 == nblocks) a[kShakeRate / 8 - 1] ^= 0x8000000000000000ull;
        keccak_f1600_dev(a);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = a[i];
}

__device__ __forceinline__ uint64_t bswap64_dev(uint64_t v) { return __builtin_bswap64(v); }

// (p + 1) / 4 as canonical words, computed on the host from the modulus and passed by value.
struct SqrtExp { uint32_t w[12]; };

template <class P> __device__ __noinline__ Fe<P> fe_pow_words(const Fe<P>& a, const SqrtExp& e) {
    Fe<P> acc = fe_one<P>();
    for (int i = P::BITS - 1; i >= 0; i--) {
        acc = fe_sqr<P>(acc);
        if ((e.w[i >> 5] >> (i & 31)) & 1) acc = fe_mul<P>(acc, a);
    }
    return acc;
}

// x = BE(SHAKE256(msg)[0 .. MODBYTES)) mod p, Montgomery form
template <class C> __device__ Fe<typename C::Fp> hash_to_x(const HashMsg& m) {
    using Fp = typename C::Fp;
    uint64_t lanes[8];
    shake256_dev(m, lanes);
    // BE(h): the last hash byte is the least significant byte of x
    constexpr int NLANE = C::MODBYTES / 8;
    uint32_t xw[Fp::NW];
#pragma unroll
    for (int l = 0; l < NLANE; l++) {
        uint64_t v = bswap64_dev(lanes[NLANE - 1 - l]);
        xw[2 * l] = (uint32_t)v;
        xw[2 * l + 1] = (uint32_t)(v >> 32);
    }
    // x < 2^(8 MODBYTES) < R: the Montgomery product with R^2 mod p reduces it (pre-subtraction value < p + p/64)
    return fe_to_mont<Fp>(fe_unpack_words<Fp>(xw));
}

// One try of amcl's `new_bigint(x, 0)`: is x^3 + b a non-zero square?  If so cand = (x, even root).  x is incremented
// either way (mapit does `x.inc(1)` before looking at the result).
template <class C> __device__ __forceinline__ bool try_x(Fe<typename C::Fp>& x, const SqrtExp& e, Aff<C>& cand) {
    using Fp = typename C::Fp;
    Fe<Fp> braw = fe_zero<Fp>();
    braw.v[0] = C::B;
    Fe<Fp> rhs = fe_add(fe_mul(fe_sqr(x), x), fe_to_mont<Fp>(braw));
    Fe<Fp> s = fe_pow_words<Fp>(rhs, e);
    bool found = !fe_is_zero(rhs) && fe_eq(fe_sqr(s), rhs);
    cand.x = x;
    x = fe_add(x, fe_one<Fp>());
    if (found) {
        uint32_t sw[Fp::NW];
        fe_pack_words<Fp>(sw, fe_from_mont<Fp>(s));
        cand.y = (sw[0] & 1) ? fe_neg(s) : s;
    }
    return found;
}

__device__ __forceinline__ HashMsg make_msg(const uint8_t* bytes, const uint64_t* offs, uint32_t prefix_len, uint64_t first, size_t i) {
    HashMsg m;
    m.tail_len = 0;
    if (offs) {
        m.head = bytes + offs[i];
        m.head_len = (uint32_t)(offs[i + 1] - offs[i]);
    } else {
        m.head = bytes;
        m.head_len = prefix_len;
        uint64_t v = first + i;
        uint8_t rev[20];
        uint32_t nd = 0;
        do { rev[nd++] = (uint8_t)('0' + v % 10); v /= 10; } while (v);
        for (uint32_t k = 0; k < nd; k++) m.tail[k] = rev[nd - 1 - k];
        m.tail_len = nd;
    }
    return m;
}

// Stage 1: the x-search.  The number of tries is geometric (p = 1/2), so a wave that maps lane -> message statically runs
// as long as its unluckiest lane (~7 tries for 64 lanes against a mean of 2).  Lanes therefore PULL messages from an atomic
// counter: a lane that has found its point takes the next message while its neighbours are still searching.  Every lane
// leaves the loop once the counter passes n (each try succeeds with probability 1/2, so the loop terminates).
// out[i] = (x, even y) on the curve, cofactor not yet cleared; message i as in k_hash_to_g1 below.
template <class C>
__global__ void __launch_bounds__(kHashBlock, 1) k_hash_search(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offs, uint32_t prefix_len,
                                                               uint64_t first, size_t n, SqrtExp e, unsigned long long* __restrict__ next,
                                                               AffPacked<C>* __restrict__ out) {
    using Fp = typename C::Fp;
    bool have = false;
    size_t i = 0;
    Fe<Fp> x = fe_zero<Fp>();
    for (;;) {
        if (!have) {
            i = (size_t)atomicAdd(next, 1ull);
            if (i >= n) break;
            x = hash_to_x<C>(make_msg(bytes, offs, prefix_len, first, i));
            have = true;
        }
        Aff<C> cand;
        if (try_x<C>(x, e, cand)) {
            out[i] = aff_pack(cand);
            have = false;
        }
    }
}

// Stage 2 (curves with a cofactor): P <- h * P, uniform work for every lane.  h * P == O would need P in the h-torsion
// (probability ~ h / #E); amcl then goes on with the next x, and so does this loop.
template <class C>
__global__ void __launch_bounds__(kHashBlock, 2) k_clear_cofactor(size_t n, SqrtExp e, AffPacked<C>* __restrict__ pts) {
    using Fp = typename C::Fp;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<C> cand = aff_unpack(pts[i]);
    const uint32_t cof[8] = {C::COFACTOR[0], C::COFACTOR[1], C::COFACTOR[2], C::COFACTOR[3], 0, 0, 0, 0};
    for (;;) {
        Aff<C> r = xyzz_to_aff<C>(xyzz_mul_words<C>(cof, cand));
        if (!aff_is_inf(r)) { pts[i] = aff_pack(r); return; }
        Fe<Fp> x = fe_add(cand.x, fe_one<Fp>());
        while (!try_x<C>(x, e, cand)) {}
    }
}

// ---------------------------------------------------------------------------------------------- compressed points
// the proof structs derive serde (src/ipp.rs:13, src/r1cs/proof.rs:24) and amcl can write a point as one
// coordinate and a sign.  Wire form of THIS build (amcl's own compressed bytes cannot be checked here, so the format is not
// claimed to be amcl's): 1 + MODBYTES bytes per point,
//     tag 0x02 (y even) / 0x03 (y odd) || X big-endian          tag 0x00 || zeros = the identity
// (x = 0 IS an abscissa of y^2 = x^3 + 4 -- y = +-2 -- so the identity needs its own tag.)
// Decompression: x < p, rhs = x^3 + b a square, y = rhs^((p+1)/4) with the tagged parity; anything else sets bit 0 of *err
// and yields the identity.  Any other tag, or non-zero bytes behind tag 0, is an error too.
template <class C>
__global__ void __launch_bounds__(kHashBlock) k_g1_compress(const AffPacked<C>* __restrict__ pts, size_t n, uint8_t* __restrict__ out) {
    using Fp = typename C::Fp;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int MB = C::MODBYTES;
    Aff<C> a = aff_unpack(pts[i]);
    uint8_t* o = out + i * (size_t)(MB + 1);
    if (aff_is_inf(a)) { for (int k = 0; k <= MB; k++) o[k] = 0; return; }
    uint32_t xw[Fp::NW], yw[Fp::NW];
    fe_pack_words<Fp>(xw, fe_from_mont<Fp>(a.x));
    fe_pack_words<Fp>(yw, fe_from_mont<Fp>(a.y));
    o[0] = (uint8_t)(2 + (yw[0] & 1));
    for (int k = 0; k < MB; k++) o[1 + k] = (uint8_t)(xw[(MB - 1 - k) >> 2] >> (8 * ((MB - 1 - k) & 3)));
}

template <class C>
__global__ void __launch_bounds__(kHashBlock, 2) k_g1_decompress(const uint8_t* __restrict__ in, size_t n, SqrtExp e, AffPacked<C>* __restrict__ out,
                                                                 uint32_t* __restrict__ err) {
    using Fp = typename C::Fp;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int MB = C::MODBYTES;
    const uint8_t* p = in + i * (size_t)(MB + 1);
    uint32_t xw[Fp::NW];
    for (int k = 0; k < Fp::NW; k++) xw[k] = 0;
    uint32_t any = 0;
    for (int k = 0; k < MB; k++) { uint32_t b = p[1 + k]; any |= b; xw[(MB - 1 - k) >> 2] |= b << (8 * ((MB - 1 - k) & 3)); }
    const uint8_t tag = p[0];
    Aff<C> a;
    a.x = fe_zero<Fp>(); a.y = fe_zero<Fp>();
    bool ok = true;
    if (tag == 0) ok = any == 0;
    else if (tag != 2 && tag != 3) ok = false;
    else if (!words_lt_mod<Fp>(xw)) ok = false;
    else {
        Fe<Fp> x = fe_to_mont<Fp>(fe_unpack_words<Fp>(xw));
        Fe<Fp> braw = fe_zero<Fp>();
        braw.v[0] = C::B;
        Fe<Fp> rhs = fe_add(fe_mul(fe_sqr(x), x), fe_to_mont<Fp>(braw));
        Fe<Fp> s = fe_pow_words<Fp>(rhs, e);
        if (!fe_eq(fe_sqr(s), rhs) || fe_is_zero(s)) ok = false;      // not a square; y = 0 has even order 2 and is in neither group
        else {
            uint32_t sw[Fp::NW];
            fe_pack_words<Fp>(sw, fe_from_mont<Fp>(s));
            a.x = x;
            a.y = ((sw[0] & 1) == (uint32_t)(tag & 1)) ? s : fe_neg(s);
        }
    }
    if (!ok) { atomicOr(err, 1u); a.x = fe_zero<Fp>(); a.y = fe_zero<Fp>(); }
    out[i] = aff_pack(a);
}

}  // namespace bp
