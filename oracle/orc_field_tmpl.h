/* oracle/orc_field_tmpl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Prime-field arithmetic "template": include once per field with
 *     #define NL   <number of 64-bit limbs>
 *     #define F(x) <prefix>_##x
 * Montgomery representation, 64-bit limbs, unsigned __int128 products (CIOS).  This is a
 * restatement of the arithmetic the reference obtains from the un-vendored `amcl` crate
 * (BIG/FP): amcl uses unsaturated 58-bit limbs; results are canonical
 * residues, which is all the reference's call sites observe (src/ipp.rs:113-130).
 */
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t l[NL]; } F(t);

typedef struct {
    uint64_t mod[NL];
    uint64_t one[NL];   /* R mod p     */
    uint64_t r2[NL];    /* R^2 mod p   */
    uint64_t inv;       /* -p^-1 mod 2^64 */
    int bits;
} F(params_t);

static F(params_t) F(P);

static inline int F(geq_mod)(const uint64_t* a) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a[i] > F(P).mod[i]) return 1;
        if (a[i] < F(P).mod[i]) return 0;
    }
    return 1;
}

static inline void F(sub_mod_raw)(uint64_t* a) {
    unsigned __int128 br = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 d = (unsigned __int128)a[i] - F(P).mod[i] - (uint64_t)br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}

static inline void F(add)(F(t)* r, const F(t)* a, const F(t)* b) {
    unsigned __int128 c = 0;
    uint64_t t[NL];
    for (int i = 0; i < NL; i++) { c += (unsigned __int128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (c || F(geq_mod)(t)) F(sub_mod_raw)(t);
    memcpy(r->l, t, sizeof t);
}

static inline void F(sub)(F(t)* r, const F(t)* a, const F(t)* b) {
    uint64_t t[NL];
    uint64_t br = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 d = (unsigned __int128)a->l[i] - b->l[i] - br;
        t[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    if (br) {
        unsigned __int128 c = 0;
        for (int i = 0; i < NL; i++) { c += (unsigned __int128)t[i] + F(P).mod[i]; t[i] = (uint64_t)c; c >>= 64; }
    }
    memcpy(r->l, t, sizeof t);
}

static inline int F(is_zero)(const F(t)* a) {
    uint64_t o = 0;
    for (int i = 0; i < NL; i++) o |= a->l[i];
    return o == 0;
}

static inline int F(eq)(const F(t)* a, const F(t)* b) { return memcmp(a->l, b->l, sizeof a->l) == 0; }

static inline void F(neg)(F(t)* r, const F(t)* a) {
    F(t) z; memset(&z, 0, sizeof z);
    F(sub)(r, &z, a);
}

static inline void F(dbl)(F(t)* r, const F(t)* a) { F(add)(r, a, a); }

static inline void F(mul)(F(t)* r, const F(t)* a, const F(t)* b) {
    uint64_t t[NL + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < NL; i++) {
        unsigned __int128 c = 0;
        uint64_t bi = b->l[i];
        for (int j = 0; j < NL; j++) {
            c += (unsigned __int128)a->l[j] * bi + t[j];
            t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[NL]; t[NL] = (uint64_t)c; t[NL + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F(P).inv;
        c = (unsigned __int128)m * F(P).mod[0] + t[0];
        c >>= 64;
        for (int j = 1; j < NL; j++) {
            c += (unsigned __int128)m * F(P).mod[j] + t[j];
            t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[NL]; t[NL - 1] = (uint64_t)c;
        t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
    }
    if (t[NL] || F(geq_mod)(t)) F(sub_mod_raw)(t);
    memcpy(r->l, t, sizeof r->l);
}

static inline void F(sqr)(F(t)* r, const F(t)* a) { F(mul)(r, a, a); }

/* r = a^e, e given as NL little-endian 64-bit words (plain integer, not Montgomery) */
static void F(pow)(F(t)* r, const F(t)* a, const uint64_t* e) {
    F(t) acc; memcpy(acc.l, F(P).one, sizeof acc.l);
    for (int i = NL * 64 - 1; i >= 0; i--) {
        F(sqr)(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) F(mul)(&acc, &acc, a);
    }
    *r = acc;
}

/* Fermat inverse; inv(0) = 0 */
static void F(inv)(F(t)* r, const F(t)* a) {
    uint64_t e[NL];
    memcpy(e, F(P).mod, sizeof e);
    /* e = p - 2 (p is odd and > 2, so no borrow past limb 0 unless limb0 < 2) */
    unsigned __int128 d = (unsigned __int128)e[0] - 2;
    e[0] = (uint64_t)d;
    if ((d >> 64) & 1) { int i = 1; while (i < NL && e[i]-- == 0) i++; }
    F(pow)(r, a, e);
}

/* canonical little-endian bytes <-> Montgomery form.  nbytes = 4 * (32-bit limb count of the C ABI) */
static void F(from_le)(F(t)* r, const uint8_t* in, int nbytes) {
    F(t) x; memset(&x, 0, sizeof x);
    for (int i = 0; i < nbytes && i < NL * 8; i++) x.l[i / 8] |= (uint64_t)in[i] << (8 * (i % 8));
    /* inputs are required canonical (< p); reduce defensively so that the oracle never sees junk */
    while (F(geq_mod)(x.l)) F(sub_mod_raw)(x.l);
    F(t) r2; memcpy(r2.l, F(P).r2, sizeof r2.l);
    F(mul)(r, &x, &r2);
}

static void F(to_le)(uint8_t* out, const F(t)* a, int nbytes) {
    F(t) one_raw; memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
    F(t) x; F(mul)(&x, a, &one_raw);
    memset(out, 0, nbytes);
    for (int i = 0; i < nbytes && i < NL * 8; i++) out[i] = (uint8_t)(x.l[i / 8] >> (8 * (i % 8)));
}

/* plain (non-Montgomery) integer value of a, little-endian 64-bit words */
static void F(to_raw)(uint64_t* out, const F(t)* a) {
    F(t) one_raw; memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
    F(t) x; F(mul)(&x, a, &one_raw);
    memcpy(out, x.l, sizeof x.l);
}

static void F(from_u64)(F(t)* r, uint64_t v) {
    F(t) x; memset(&x, 0, sizeof x); x.l[0] = v;
    F(t) r2; memcpy(r2.l, F(P).r2, sizeof r2.l);
    F(mul)(r, &x, &r2);
}

/* Reduce a big-endian byte string of any length modulo p (FieldElement::from(&[u8; MODBYTES]):
 * BIG::frombytes then rmod(CurveOrder), used by src/transcript.rs:55-60) -> Montgomery form. */
static void F(from_be_reduce)(F(t)* r, const uint8_t* in, int nbytes) {
    F(t) acc; memset(&acc, 0, sizeof acc);
    F(t) c256; F(from_u64)(&c256, 256);
    for (int i = 0; i < nbytes; i++) {
        F(t) d; F(from_u64)(&d, in[i]);
        F(mul)(&acc, &acc, &c256);
        F(add)(&acc, &acc, &d);
    }
    *r = acc;
}

/* Set up the parameter block from the modulus (little-endian 64-bit words). */
static void F(init)(const uint64_t* mod) {
    memcpy(F(P).mod, mod, sizeof F(P).mod);
    int bits = NL * 64;
    while (bits > 0 && !((mod[(bits - 1) / 64] >> ((bits - 1) % 64)) & 1)) bits--;
    F(P).bits = bits;
    /* inv = -p^-1 mod 2^64 by Newton iteration */
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - mod[0] * x;
    F(P).inv = (uint64_t)0 - x;
    /* one = 2^(64 NL) mod p, r2 = 2^(128 NL) mod p by repeated modular doubling of 1 */
    uint64_t t[NL]; memset(t, 0, sizeof t); t[0] = 1;
    for (int i = 0; i < 2 * NL * 64; i++) {
        uint64_t carry = t[NL - 1] >> 63;
        for (int j = NL - 1; j > 0; j--) t[j] = (t[j] << 1) | (t[j - 1] >> 63);
        t[0] <<= 1;
        if (carry || F(geq_mod)(t)) F(sub_mod_raw)(t);
        if (i == NL * 64 - 1) memcpy(F(P).one, t, sizeof t);
    }
    memcpy(F(P).r2, t, sizeof t);
}
