// bp_host_tail.hpp -- the serial tail of an MSM on the host:  sum_w 2^(off_w) S_w  and the affine normalisation.
//
// This is the one strictly sequential piece of the pipeline (~255 dependent point doublings; a single GPU lane needs
// ~2 ms for it).  It runs on the host with 64-bit limbs and unsigned __int128 products (the 30-bit-limb templates of
// bp_field.cuh are laid out for 32-bit VGPRs and are ~3x slower on a CPU).  Jacobian coordinates (doubling 2M + 5S).
// Montgomery radix here is 2^(64 NL64); inputs arrive as packed XYZZ records in the DEVICE's Montgomery radix
// 2^(30 NL) and are rescaled on entry (one multiplication by 2^(64 NL64 - 30 NL) mod p ... folded into to_host()).
#pragma once
#include <cstdint>
#include <cstring>

#include "bp_curve.cuh"

namespace bp {
namespace host {

template <class P>   // P = Field<...> of bp_field.cuh (for the modulus words)
struct F64 {
    static constexpr int N = (P::NW + 1) / 2;
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
        unsigned __int128 br = 0;
        for (int i = 0; i < N; i++) { unsigned __int128 d = (unsigned __int128)a[i] - mod[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
    }
    void add(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
        unsigned __int128 c = 0;
        uint64_t t[N];
        for (int i = 0; i < N; i++) { c += (unsigned __int128)a[i] + b[i]; t[i] = (uint64_t)c; c >>= 64; }
        if (c || geq(t)) sub_mod(t);
        memcpy(r, t, sizeof t);
    }
    void sub(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
        uint64_t t[N], br = 0;
        for (int i = 0; i < N; i++) { unsigned __int128 d = (unsigned __int128)a[i] - b[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
        if (br) { unsigned __int128 c = 0; for (int i = 0; i < N; i++) { c += (unsigned __int128)t[i] + mod[i]; t[i] = (uint64_t)c; c >>= 64; } }
        memcpy(r, t, sizeof t);
    }
    void mul(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
        uint64_t t[N + 2] = {};
        for (int i = 0; i < N; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; j < N; j++) { c += (unsigned __int128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N] = (uint64_t)c; t[N + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * inv;
            c = (unsigned __int128)m * mod[0] + t[0]; c >>= 64;
            for (int j = 1; j < N; j++) { c += (unsigned __int128)m * mod[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N - 1] = (uint64_t)c; t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        if (t[N] || geq(t)) sub_mod(t);
        memcpy(r, t, N * 8);
    }
    bool is_zero(const uint64_t* a) const { uint64_t o = 0; for (int i = 0; i < N; i++) o |= a[i]; return o == 0; }
    void inverse(uint64_t* r, const uint64_t* a) const {   // a^(p-2)
        uint64_t e[N]; memcpy(e, mod, sizeof e);
        uint64_t br = 2;
        for (int i = 0; i < N && br; i++) { uint64_t o = e[i]; e[i] = o - br; br = o < br; }
        uint64_t acc[N]; memcpy(acc, one, sizeof acc);
        for (int i = P::BITS - 1; i >= 0; i--) {
            mul(acc, acc, acc);
            if ((e[i >> 6] >> (i & 63)) & 1) mul(acc, acc, a);
        }
        memcpy(r, acc, sizeof acc);
    }
};

template <class C>
class Tail {
    using Fp = typename C::Fp;
    using F = F64<Fp>;
    static constexpr int N = F::N;
    struct Jac { uint64_t x[N], y[N], z[N]; bool inf; };
    F f;
    uint64_t dev_to_host[N];   // 2^(64N) * 2^(64N) / 2^(30 NL)  in plain form: mul(x_devmont_plain, k) -> host Montgomery form

    // canonical words of a device-Montgomery residue (x * 2^(30 NL)) -> host Montgomery (x * 2^(64 N))
    void load(uint64_t* out, const uint32_t* words) const {
        uint64_t t[N] = {};
        for (int i = 0; i < Fp::NW; i++) t[i / 2] |= (uint64_t)words[i] << (32 * (i & 1));
        f.mul(out, t, dev_to_host);
    }
    void dbl(Jac& p) const {   // dbl-2009-l, a = 0
        if (p.inf) return;
        uint64_t A[N], B[N], Cc[N], D[N], E[N], Fq[N], t[N];
        f.mul(A, p.x, p.x); f.mul(B, p.y, p.y); f.mul(Cc, B, B);
        f.add(t, p.x, B); f.mul(t, t, t); f.sub(t, t, A); f.sub(t, t, Cc); f.add(D, t, t);
        f.add(E, A, A); f.add(E, E, A);
        f.mul(Fq, E, E);
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
        f.mul(z1z1, p.z, p.z); f.mul(z2z2, q.z, q.z);
        f.mul(u1, p.x, z2z2); f.mul(u2, q.x, z1z1);
        f.mul(s1, p.y, q.z); f.mul(s1, s1, z2z2);
        f.mul(s2, q.y, p.z); f.mul(s2, s2, z1z1);
        f.sub(h, u2, u1); f.sub(rr, s2, s1);
        if (f.is_zero(h)) { if (f.is_zero(rr)) { dbl(p); return; } p.inf = true; return; }
        f.add(rr, rr, rr);
        f.add(i, h, h); f.mul(i, i, i);
        f.mul(j, h, i); f.mul(v, u1, i);
        uint64_t x3[N], y3[N], z3[N];
        f.mul(x3, rr, rr); f.sub(x3, x3, j); f.add(t, v, v); f.sub(x3, x3, t);
        f.sub(t, v, x3); f.mul(y3, rr, t); f.mul(t, s1, j); f.add(t, t, t); f.sub(y3, y3, t);
        f.add(z3, p.z, q.z); f.mul(z3, z3, z3); f.sub(z3, z3, z1z1); f.sub(z3, z3, z2z2); f.mul(z3, z3, h);
        memcpy(p.x, x3, sizeof x3); memcpy(p.y, y3, sizeof y3); memcpy(p.z, z3, sizeof z3);
    }
    // XYZZ (x = X/ZZ, y = Y/ZZZ) -> Jacobian with Z = ZZZ/ZZ ... avoided: use X' = X*ZZ, Y' = Y*ZZZ, Z' = ZZ*... see below
    Jac from_record(const XyzzPacked<C>& r) const {
        Jac p; p.inf = false;
        uint64_t X[N], Y[N], ZZ[N], ZZZ[N];
        load(X, r.x.w); load(Y, r.y.w); load(ZZ, r.zz.w); load(ZZZ, r.zzz.w);
        if (f.is_zero(ZZ)) { p.inf = true; memset(p.x, 0, sizeof p.x); memset(p.y, 0, sizeof p.y); memset(p.z, 0, sizeof p.z); return p; }
        // With z^2 = ZZ, z^3 = ZZZ: choose Jacobian Z = ZZ * ZZZ (= z^5):  X_J = x Z^2 = X ZZ^4... simpler and exact:
        // Z := ZZZ ... Z^2 = ZZ^3, Z^3 = ZZZ^3.  X_J = (X/ZZ) ZZ^3 = X ZZ^2 ;  Y_J = (Y/ZZZ) ZZZ^3 = Y ZZZ^2.
        uint64_t t[N];
        f.mul(t, ZZ, ZZ); f.mul(p.x, X, t);
        f.mul(t, ZZZ, ZZZ); f.mul(p.y, Y, t);
        memcpy(p.z, ZZZ, sizeof ZZZ);
        return p;
    }

    static bool is_identity_record(const XyzzPacked<C>& r) {
        uint32_t o = 0;
        for (int i = 0; i < Fp::NW; i++) o |= r.zz.w[i];
        return o == 0;
    }

public:
    Tail() {
        // k = 2^(128 N - 30 NL) mod p, plain integer:  mul(a, k) = a * k / 2^(64N) = a * 2^(64N) / 2^(30NL)
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

    // result = sum_r 2^(pos[r]) (sum_s rec[s * nrec + r]) as canonical little-endian x || y (all-zero = identity): Horner over the
    // records in descending bit position, pos[r] - pos[next] doublings in between (~255 in all, whatever the record count).
    void fold(const XyzzPacked<C>* rec, size_t sets, int nrec, const uint16_t* pos, uint8_t* out_le) const {
        Jac acc; acc.inf = true;
        memset(acc.x, 0, sizeof acc.x); memset(acc.y, 0, sizeof acc.y); memset(acc.z, 0, sizeof acc.z);
        // order of the records by descending position (bucket lists over the bit positions)
        constexpr int kMaxPos = 600;
        int head[kMaxPos + 1];
        for (int p = 0; p <= kMaxPos; p++) head[p] = -1;
        int* next = new int[nrec > 0 ? nrec : 1];
        for (int r = 0; r < nrec; r++) { int p = pos[r] <= kMaxPos ? pos[r] : kMaxPos; next[r] = head[p]; head[p] = r; }
        int cur = -1;
        for (int p = kMaxPos; p >= 0; p--) {
            for (int r = head[p]; r >= 0; r = next[r]) {
                bool any = false;
                for (size_t s = 0; s < sets && !any; s++) any = !is_identity_record(rec[s * (size_t)nrec + r]);
                if (!any) continue;
                if (cur >= 0) for (int i = 0; i < cur - p; i++) dbl(acc);
                cur = p;
                for (size_t s = 0; s < sets; s++) add(acc, from_record(rec[s * (size_t)nrec + r]));
            }
        }
        delete[] next;
        for (int i = 0; i < cur; i++) dbl(acc);
        const int fb = 4 * Fp::NW;
        memset(out_le, 0, 2 * fb);
        if (acc.inf) return;
        uint64_t zi[N], zi2[N], zi3[N], x[N], y[N], onep[N] = {};
        f.inverse(zi, acc.z);
        f.mul(zi2, zi, zi); f.mul(zi3, zi2, zi);
        f.mul(x, acc.x, zi2); f.mul(y, acc.y, zi3);
        onep[0] = 1;
        f.mul(x, x, onep); f.mul(y, y, onep);   // out of Montgomery form
        memcpy(out_le, x, fb); memcpy(out_le + fb, y, fb);
    }
};

}  // namespace host
}  // namespace bp
