// bp_host_tail.hpp -- the serial tail of an MSM on the host:  sum_r 2^(pos_r) Rec_r  and the affine normalisation.
//
// This is the one strictly sequential piece of the pipeline (~255 dependent point doublings; a single GPU lane needs
// ~2 ms for it).  It runs on the host with 64-bit limbs and unsigned __int128 products (the 30-bit-limb templates of
// bp_field.cuh are laid out for 32-bit VGPRs and are ~3x slower on a CPU).  Jacobian coordinates (doubling 2M + 5S).
//
// Round 3: the bucket reduce hands over one record per BIT PLANE of the reduce-thread index (bp_kernels.cuh:
// k_bucket_reduce) instead of multiplying by the weights on the device, so a 2^20-point MSM arrives as ~200 records instead of 16.
// The fold therefore had to become cheap per record:
//   * records arrive as the device's lazy XYZZ values in ITS Montgomery radix 2^(30 NL); one product by 2^(64 N - 30 NL) per
//     coordinate rescales (and reduces) them on entry, four more make a Jacobian point (Z = ZZZ);
//   * the Montgomery product is the "no-carry" CIOS for moduli with a clear top bit, fully unrolled (-25 %);
//   * the records are dealt round-robin to `chains` independent Horner walks (each one: all the doublings, 1/chains of the
//     additions) that can run on helper threads (bp_internal.hpp: HostPool) and are added at the end; with 4 chains the critical
//     path of a 208-record fold is 255 doublings + 55 additions instead of 255 + 208;
//   * the final inversion is a binary extended Euclid (~4x faster than a^(p-2)).
#pragma once
#include <cstdint>
#include <cstring>

#include "bp_curve.cuh"

namespace bp {
namespace host {

typedef unsigned __int128 u128;
constexpr int kTailMaxRecords = 4096;       // == kMaxRecords of bp_kernels.cuh (this header also builds without HIP)

template <class P>   // P = Field<...> of bp_field.cuh (for the modulus words)
struct F64 {
    static constexpr int N = (P::NW + 1) / 2;
    static_assert(P::BITS < 64 * N, "the no-carry product needs a clear top bit");
    uint64_t mod[N], one[N], r2[N], inv;
    F64() {
        for (int i = 0; i < N; i++) mod[i] = (uint64_t)P::Words::MODW[2 * i] | (2 * i + 1 < P::NW ? (uint64_t)P::Words::MODW[2 * i + 1] << 32 : 0);
        uint64_t x = 1;
        for (int i = 0; i < 6; i++) x *= 2 - mod[0] * x;
        inv = 0 - x;
        uint64_t t[N] = {};
        t[0] = 1;
        for (int k = 0; k < 2 * N * 64; k++) {
            uint64_t carry = t[N - 1] >> 63;
            for (int j = N - 1; j > 0; j--) t[j] = (t[j] << 1) | (t[j - 1] >> 63);
            t[0] <<= 1;
            if (carry || geq(t)) sub_mod(t);
            if (k == N * 64 - 1) memcpy(one, t, sizeof t);
        }
        memcpy(r2, t, sizeof t);
    }
    bool geq(const uint64_t* a) const {
        for (int i = N - 1; i >= 0; i--) { if (a[i] != mod[i]) return a[i] > mod[i]; }
        return true;
    }
    void sub_mod(uint64_t* a) const {
        uint64_t br = 0;
        for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - mod[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    }
    // r = a + b mod p   (inputs < p)
    inline void add(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
        uint64_t t[N], d[N], c = 0, br = 0;
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) { u128 s = (u128)a[i] + b[i] + c; t[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) { u128 s = (u128)t[i] - mod[i] - br; d[i] = (uint64_t)s; br = (uint64_t)(s >> 64) & 1; }
        const bool keep = br && !c;                        // t < p
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) r[i] = keep ? t[i] : d[i];
    }
    inline void sub(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
        uint64_t t[N], br = 0, c = 0;
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - b[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
        const uint64_t mask = 0 - br;                      // negative: add p back
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) { u128 s = (u128)t[i] + (mod[i] & mask) + c; r[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    }
    // Montgomery product, CIOS without the (N+1)-th and (N+2)-th accumulator words: valid because the modulus leaves its top bit
    // clear (El Housni / Botrel, "no-carry" optimisation).  Inputs < p (one of them may be any value < 2^(64N) whose product with
    // the other stays below p 2^(64N)); output < p.
    inline void mul(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
        uint64_t t[N] = {};
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) {
            u128 x = (u128)a[0] * b[i] + t[0];
            uint64_t A = (uint64_t)(x >> 64);
            const uint64_t m = (uint64_t)x * inv;
            u128 y = (u128)m * mod[0] + (uint64_t)x;
            uint64_t Cc = (uint64_t)(y >> 64);
#pragma GCC unroll 8
            for (int j = 1; j < N; j++) {
                x = (u128)a[j] * b[i] + t[j] + A; A = (uint64_t)(x >> 64);
                y = (u128)m * mod[j] + (uint64_t)x + Cc; Cc = (uint64_t)(y >> 64);
                t[j - 1] = (uint64_t)y;
            }
            t[N - 1] = Cc + A;
        }
        uint64_t d[N], br = 0;
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) { u128 s = (u128)t[i] - mod[i] - br; d[i] = (uint64_t)s; br = (uint64_t)(s >> 64) & 1; }
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) r[i] = br ? t[i] : d[i];
    }
    inline void sqr(uint64_t* r, const uint64_t* a) const { mul(r, a, a); }
    bool is_zero(const uint64_t* a) const { uint64_t o = 0; for (int i = 0; i < N; i++) o |= a[i]; return o == 0; }

    // ---- inversion: binary extended Euclid on plain integers (variable time: the operand is the Z of a public result) ----
    static bool is_one(const uint64_t* a) { uint64_t o = a[0] ^ 1; for (int i = 1; i < N; i++) o |= a[i]; return o == 0; }
    static bool ge(const uint64_t* a, const uint64_t* b) { for (int i = N - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; } return true; }
    static void shr1(uint64_t* a, uint64_t top = 0) { for (int i = 0; i < N - 1; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63); a[N - 1] = (a[N - 1] >> 1) | (top << 63); }
    static uint64_t add_n(uint64_t* a, const uint64_t* b) { uint64_t c = 0; for (int i = 0; i < N; i++) { u128 s = (u128)a[i] + b[i] + c; a[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } return c; }
    static void sub_n(uint64_t* a, const uint64_t* b) { uint64_t br = 0; for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - b[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
    void half_mod(uint64_t* x) const { if (x[0] & 1) { uint64_t c = add_n(x, mod); shr1(x, c); } else shr1(x); }   // x / 2 mod p
    void sub_modp(uint64_t* x, const uint64_t* y) const {      // x = x - y mod p, both < p
        if (ge(x, y)) { sub_n(x, y); return; }
        uint64_t t[N];
        memcpy(t, mod, sizeof t);
        sub_n(t, y);
        add_n(x, t);                                           // x + (p - y) < p
    }
    // r = a^-1 (Montgomery in, Montgomery out); 0 -> 0 (as a^(p-2) gives; the Euclid loop below would never leave u = 0)
    void inverse(uint64_t* r, const uint64_t* a) const {
        if (is_zero(a)) { memset(r, 0, sizeof(uint64_t) * N); return; }
        uint64_t u[N], v[N], x1[N] = {}, x2[N] = {};
        memcpy(u, a, sizeof u);
        memcpy(v, mod, sizeof v);
        x1[0] = 1;
        while (!is_one(u) && !is_one(v)) {
            while (!(u[0] & 1)) { shr1(u); half_mod(x1); }
            while (!(v[0] & 1)) { shr1(v); half_mod(x2); }
            if (ge(u, v)) { sub_n(u, v); sub_modp(x1, x2); }
            else { sub_n(v, u); sub_modp(x2, x1); }
        }
        const uint64_t* res = is_one(u) ? x1 : x2;     // (a R)^-1 = a^-1 R^-1 as a plain integer
        uint64_t t[N];
        mul(t, res, r2);                               // a^-1 R^-1 * R^2 / R = a^-1
        mul(r, t, r2);                                 // a^-1 * R^2 / R       = a^-1 R
    }
};

template <class C>
class Tail {
    using Fp = typename C::Fp;
    using F = F64<Fp>;
    static constexpr int N = F::N;
public:
    struct Jac { uint64_t x[N], y[N], z[N]; bool inf; };
private:
    F f;

    void dbl(Jac& p) const {   // dbl-2009-l, a = 0
        if (p.inf) return;
        uint64_t A[N], B[N], Cc[N], D[N], E[N], Fq[N], t[N];
        f.sqr(A, p.x); f.sqr(B, p.y); f.sqr(Cc, B);
        f.add(t, p.x, B); f.sqr(t, t); f.sub(t, t, A); f.sub(t, t, Cc); f.add(D, t, t);
        f.add(E, A, A); f.add(E, E, A);
        f.sqr(Fq, E);
        uint64_t x3[N], y3[N], z3[N];
        f.add(t, D, D); f.sub(x3, Fq, t);
        f.mul(z3, p.y, p.z); f.add(z3, z3, z3);
        f.sub(t, D, x3); f.mul(y3, E, t);
        f.add(Cc, Cc, Cc); f.add(Cc, Cc, Cc); f.add(Cc, Cc, Cc);
        f.sub(y3, y3, Cc);
        memcpy(p.x, x3, sizeof x3); memcpy(p.y, y3, sizeof y3); memcpy(p.z, z3, sizeof z3);
    }
    void add(Jac& p, const Jac& q) const {   // add-2007-bl, exceptional cases handled
        if (q.inf) return;
        if (p.inf) { p = q; return; }
        uint64_t z1z1[N], z2z2[N], u1[N], u2[N], s1[N], s2[N], h[N], i[N], j[N], rr[N], v[N], t[N];
        f.sqr(z1z1, p.z); f.sqr(z2z2, q.z);
        f.mul(u1, p.x, z2z2); f.mul(u2, q.x, z1z1);
        f.mul(s1, p.y, q.z); f.mul(s1, s1, z2z2);
        f.mul(s2, q.y, p.z); f.mul(s2, s2, z1z1);
        f.sub(h, u2, u1); f.sub(rr, s2, s1);
        if (f.is_zero(h)) { if (f.is_zero(rr)) { dbl(p); return; } p.inf = true; return; }
        f.add(rr, rr, rr);
        f.add(i, h, h); f.sqr(i, i);
        f.mul(j, h, i); f.mul(v, u1, i);
        uint64_t x3[N], y3[N], z3[N];
        f.sqr(x3, rr); f.sub(x3, x3, j); f.add(t, v, v); f.sub(x3, x3, t);
        f.sub(t, v, x3); f.mul(y3, rr, t); f.mul(t, s1, j); f.add(t, t, t); f.sub(y3, y3, t);
        f.add(z3, p.z, q.z); f.sqr(z3, z3); f.sub(z3, z3, z1z1); f.sub(z3, z3, z2z2); f.mul(z3, z3, h);
        memcpy(p.x, x3, sizeof x3); memcpy(p.y, y3, sizeof y3); memcpy(p.z, z3, sizeof z3);
    }
    uint64_t dev_to_host[N];   // 2^(128 N - 30 NL) mod p as a plain integer: mul(x_dev, k) = x * 2^(30 NL) * k / 2^(64 N) = x * 2^(64 N)

    // packed words of a device residue (x * 2^(30 NL), any value < 2^(32 NW)) -> this file's Montgomery form, canonical
    void load(uint64_t* out, const uint32_t* words) const {
        uint64_t t[N] = {};
        for (int i = 0; i < Fp::NW; i++) t[i / 2] |= (uint64_t)words[i] << (32 * (i & 1));
        f.mul(out, dev_to_host, t);
    }
    // lazy XYZZ record -> Jacobian with Z = ZZZ (then Z^2 = ZZ^3, Z^3 = ZZZ^3):  X_J = (X / ZZ) ZZ^3 = X ZZ^2,  Y_J = (Y / ZZZ) ZZZ^3 = Y ZZZ^2
    Jac from_record(const XyzzPacked<C>& r) const {
        Jac p;
        memset(&p, 0, sizeof p);
        uint64_t X[N], Y[N], ZZ[N], t[N];
        load(ZZ, r.zz.w);
        if (f.is_zero(ZZ)) { p.inf = true; return p; }
        load(X, r.x.w); load(Y, r.y.w); load(p.z, r.zzz.w);
        if (f.is_zero(p.z)) { p.inf = true; return p; }          // ZZZ = 0 with ZZ != 0 is no point (a malformed record from another rank): identity, not Z = 0
        f.sqr(t, ZZ); f.mul(p.x, X, t);
        f.sqr(t, p.z); f.mul(p.y, Y, t);
        p.inf = false;
        return p;
    }
    static bool is_identity_record(const XyzzPacked<C>& r) {
        uint32_t o = 0;
        for (int i = 0; i < Fp::NW; i++) o |= r.zz.w[i];
        return o == 0;
    }

public:
    static constexpr int kMaxChains = 16;
    Tail() {
        // k = 2^(128 N - 30 NL) mod p, plain integer
        uint64_t t[N] = {};
        t[0] = 1;
        const int e = 128 * N - LB * Fp::NL;
        for (int i = 0; i < e; i++) {
            uint64_t carry = t[N - 1] >> 63;
            for (int j = N - 1; j > 0; j--) t[j] = (t[j] << 1) | (t[j - 1] >> 63);
            t[0] <<= 1;
            if (carry || f.geq(t)) f.sub_mod(t);
        }
        memcpy(dev_to_host, t, sizeof t);
    }

    // an affine point (canonical LE words, plain integers) as a Jacobian point of this file: (x R, y R, R); all-zero = identity
    Jac jac_from_affine(const uint32_t* xw, const uint32_t* yw) const {
        Jac p;
        memset(&p, 0, sizeof p);
        uint32_t any = 0;
        for (int i = 0; i < Fp::NW; i++) any |= xw[i] | yw[i];
        p.inf = any == 0;
        if (p.inf) return p;
        uint64_t x[N] = {}, y[N] = {};
        for (int i = 0; i < Fp::NW; i++) { x[i / 2] |= (uint64_t)xw[i] << (32 * (i & 1)); y[i / 2] |= (uint64_t)yw[i] << (32 * (i & 1)); }
        f.mul(p.x, x, f.r2);
        f.mul(p.y, y, f.r2);
        memcpy(p.z, f.one, sizeof p.z);
        return p;
    }

    // 2^(c w) P for w = 0 .. W1-1 as canonical affine little-endian rows (x || y each; identity = zeros): the rows of ONE point in a
    // window-multiples table (the per-proof points Q / B_blinding next to the precomputed generators).  c (W1 - 1) Jacobian
    // doublings and one shared inversion: ~0.1 ms for 16 windows of 16 bits.
    void window_multiples(const uint8_t* p_le, int c, int W1, uint8_t* rows_le) const {
        const int fb = 4 * Fp::NW;
        memset(rows_le, 0, (size_t)W1 * 2 * fb);
        uint32_t xw[Fp::NW], yw[Fp::NW];
        memcpy(xw, p_le, fb); memcpy(yw, p_le + fb, fb);
        Jac acc = jac_from_affine(xw, yw);
        if (acc.inf || W1 <= 0) return;
        memcpy(rows_le, p_le, 2 * fb);
        if (W1 == 1) return;
        constexpr int kMaxW = 256;
        static_assert(kMaxW >= 128, "window count of a 2-bit table");
        if (W1 > kMaxW) W1 = kMaxW;
        Jac pts[kMaxW];
        uint64_t pre[kMaxW][N];                       // pre[w] = z_1 ... z_w
        uint64_t run[N];
        memcpy(run, f.one, sizeof run);
        for (int w = 1; w < W1; w++) {
            for (int k = 0; k < c; k++) dbl(acc);
            pts[w] = acc;
            f.mul(run, run, acc.z);
            memcpy(pre[w], run, sizeof run);
        }
        uint64_t inv[N];
        f.inverse(inv, run);
        uint64_t onep[N] = {};
        onep[0] = 1;
        for (int w = W1 - 1; w >= 1; w--) {
            uint64_t zi[N], zi2[N], zi3[N], x[N], y[N];
            if (w > 1) f.mul(zi, inv, pre[w - 1]); else memcpy(zi, inv, sizeof zi);     // 1 / z_w
            f.mul(inv, inv, pts[w].z);
            f.sqr(zi2, zi); f.mul(zi3, zi2, zi);
            f.mul(x, pts[w].x, zi2); f.mul(y, pts[w].y, zi3);
            f.mul(x, x, onep); f.mul(y, y, onep);                                         // out of Montgomery form
            memcpy(rows_le + (size_t)w * 2 * fb, x, fb); memcpy(rows_le + (size_t)w * 2 * fb + fb, y, fb);
        }
    }

    // k1 g + k2 h for two host-side points and canonical scalars (G1::binary_scalar_mul, /root/reference src/r1cs/prover.rs:496-500:
    // the five T_k commitments and Q = w g of an R1CS proof).  A shared doubling chain over the 256 bits (Shamir), ~0.2 ms: a GPU launch
    // for two terms is its host tail (the same 255 doublings) plus the launch, the upload and the synchronisation.
    // canonical coordinates (< p) and y^2 = x^3 + b, or the all-zero identity: what the device checks on upload (k_points_to_resident)
    bool valid_affine(const uint8_t* p_le) const {
        const int fb = 4 * Fp::NW;
        uint32_t xw[Fp::NW], yw[Fp::NW];
        memcpy(xw, p_le, fb); memcpy(yw, p_le + fb, fb);
        uint32_t any = 0;
        for (int i = 0; i < Fp::NW; i++) any |= xw[i] | yw[i];
        if (!any) return true;
        if (!words_lt_mod<Fp>(xw) || !words_lt_mod<Fp>(yw)) return false;
        uint64_t x[N] = {}, y[N] = {}, xm[N], ym[N], t[N], rhs[N], b[N] = {}, bm[N];
        for (int i = 0; i < Fp::NW; i++) { x[i / 2] |= (uint64_t)xw[i] << (32 * (i & 1)); y[i / 2] |= (uint64_t)yw[i] << (32 * (i & 1)); }
        b[0] = C::B;
        f.mul(xm, x, f.r2); f.mul(ym, y, f.r2); f.mul(bm, b, f.r2);
        f.sqr(t, xm); f.mul(rhs, t, xm); f.add(rhs, rhs, bm);
        f.sqr(t, ym);
        for (int i = 0; i < N; i++) if (t[i] != rhs[i]) return false;
        return true;
    }
    void mul2(const uint8_t* g_le, const uint8_t* h_le, const uint8_t* k1_le32, const uint8_t* k2_le32, uint8_t* out_le) const {
        const int fb = 4 * Fp::NW;
        uint32_t xw[Fp::NW], yw[Fp::NW];
        memcpy(xw, g_le, fb); memcpy(yw, g_le + fb, fb);
        const Jac g = jac_from_affine(xw, yw);
        memcpy(xw, h_le, fb); memcpy(yw, h_le + fb, fb);
        const Jac h = jac_from_affine(xw, yw);
        Jac gh = g;
        add(gh, h);
        Jac acc;
        memset(&acc, 0, sizeof acc);
        acc.inf = true;
        for (int bit = 255; bit >= 0; bit--) {
            dbl(acc);
            const int b1 = (k1_le32[bit >> 3] >> (bit & 7)) & 1, b2 = (k2_le32[bit >> 3] >> (bit & 7)) & 1;
            if (b1 & b2) add(acc, gh);
            else if (b1) add(acc, g);
            else if (b2) add(acc, h);
        }
        finish(&acc, 1, out_le);
    }

    // chain `k` of `chains`: Horner over every chains-th non-identity record (in descending bit position), all the way down to bit 0
    void fold_chain(const XyzzPacked<C>* rec, size_t sets, int nrec, const uint16_t* pos, int k, int chains, Jac* out) const {
        Jac acc;
        memset(&acc, 0, sizeof acc);
        acc.inf = true;
        // order of the records by descending position (bucket lists over the bit positions)
        constexpr int kMaxPos = 600;
        int head[kMaxPos + 1];
        for (int p = 0; p <= kMaxPos; p++) head[p] = -1;
        int next[kTailMaxRecords];                             // nrec <= kMaxRecords (msm_geom refuses more): no allocation on this path
        if (nrec > kTailMaxRecords) nrec = kTailMaxRecords;
        int top = 0;
        for (int r = 0; r < nrec; r++) { int p = pos[r] <= kMaxPos ? pos[r] : kMaxPos; next[r] = head[p]; head[p] = r; if (p > top) top = p; }   // positions are < 300 by construction (fr_bits + c)
        int cur = -1, seen = 0;
        for (int p = top; p >= 0; p--) {
            for (int r = head[p]; r >= 0; r = next[r]) {
                // (record, set) pairs are dealt round-robin among the chains (identity records are not dealt): with N shards' record
                // sets the additions -- N per bit position -- spread over the chains like those of a single set
                for (size_t s = 0; s < sets; s++) {
                    const XyzzPacked<C>& rs = rec[s * (size_t)nrec + r];
                    if (is_identity_record(rs)) continue;
                    if ((seen++ % chains) != k) continue;
                    if (cur >= 0) for (int i = 0; i < cur - p; i++) dbl(acc);
                    cur = p;
                    add(acc, from_record(rs));
                }
            }
        }
        for (int i = 0; i < cur; i++) dbl(acc);
        *out = acc;
    }

    // sum of the chains' partial results -> canonical little-endian x || y (all-zero = identity)
    void finish(const Jac* parts, int chains, uint8_t* out_le) const {
        Jac acc = parts[0];
        for (int k = 1; k < chains; k++) add(acc, parts[k]);
        const int fb = 4 * Fp::NW;
        memset(out_le, 0, 2 * fb);
        if (acc.inf) return;
        uint64_t zi[N], zi2[N], zi3[N], x[N], y[N], onep[N] = {};
        f.inverse(zi, acc.z);
        f.sqr(zi2, zi); f.mul(zi3, zi2, zi);
        f.mul(x, acc.x, zi2); f.mul(y, acc.y, zi3);
        onep[0] = 1;
        f.mul(x, x, onep); f.mul(y, y, onep);   // out of Montgomery form
        memcpy(out_le, x, fb); memcpy(out_le + fb, y, fb);
    }

    // result = sum_r 2^(pos[r]) (sum_s rec[s * nrec + r]): single chain on the calling thread
    void fold(const XyzzPacked<C>* rec, size_t sets, int nrec, const uint16_t* pos, uint8_t* out_le) const {
        Jac part;
        fold_chain(rec, sets, nrec, pos, 0, 1, &part);
        finish(&part, 1, out_le);
    }
};

}  // namespace host
}  // namespace bp
