/* oracle/orc_curve_tmpl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Short-Weierstrass y^2 = x^3 + b (a = 0) G1 arithmetic "template": include once per curve with
 *     #define C(x)   <curve prefix>_##x        (this file's symbols)
 *     #define FP(x)  <base-field prefix>_##x    (orc_field_tmpl.h instance)
 *     #define FR(x)  <scalar-field prefix>_##x
 *     #define FP_NL / FR_NL  limb counts (64-bit)
 *     #define FP_LE_BYTES / FR_LE_BYTES   C-ABI canonical byte widths (4 * 32-bit limb count)
 *
 * Restates what the reference obtains from amcl_wrapper's G1 / G1Vector (not in tree):
 *   G1 + G1, scalar mul, binary_scalar_mul   (src/ipp.rs:119,125,185,187)
 *   G1Vector::multi_scalar_mul_var_time / inner_product_var_time_with_ref_vecs  (src/ipp.rs:91,104,158,170,251-253)
 * The result of each is a group element; parity is defined on its canonical affine bytes.
 */

typedef struct { FP(t) x, y; int inf; } C(aff_t);
typedef struct { FP(t) x, y, z; } C(jac_t);   /* z == 0  <=>  identity */

static FP(t) C(B);          /* curve constant b, Montgomery form */
static C(aff_t) C(GEN);     /* generator */

static inline void C(jac_set_inf)(C(jac_t)* r) { memset(r, 0, sizeof *r); memcpy(r->x.l, FP(P).one, sizeof r->x.l); memcpy(r->y.l, FP(P).one, sizeof r->y.l); }
static inline int C(jac_is_inf)(const C(jac_t)* p) { return FP(is_zero)(&p->z); }

static inline void C(jac_from_aff)(C(jac_t)* r, const C(aff_t)* p) {
    if (p->inf) { C(jac_set_inf)(r); return; }
    r->x = p->x; r->y = p->y; memcpy(r->z.l, FP(P).one, sizeof r->z.l);
}

/* dbl-2009-l (a = 0) */
static void C(jac_dbl)(C(jac_t)* r, const C(jac_t)* p) {
    if (C(jac_is_inf)(p)) { *r = *p; return; }
    FP(t) A, B, Cc, D, E, Fq, t;
    FP(sqr)(&A, &p->x);
    FP(sqr)(&B, &p->y);
    FP(sqr)(&Cc, &B);
    FP(add)(&t, &p->x, &B); FP(sqr)(&t, &t); FP(sub)(&t, &t, &A); FP(sub)(&t, &t, &Cc); FP(dbl)(&D, &t);
    FP(dbl)(&E, &A); FP(add)(&E, &E, &A);
    FP(sqr)(&Fq, &E);
    FP(t) x3, y3, z3;
    FP(dbl)(&t, &D); FP(sub)(&x3, &Fq, &t);
    FP(mul)(&z3, &p->y, &p->z); FP(dbl)(&z3, &z3);
    FP(sub)(&t, &D, &x3); FP(mul)(&y3, &E, &t);
    FP(dbl)(&Cc, &Cc); FP(dbl)(&Cc, &Cc); FP(dbl)(&Cc, &Cc);
    FP(sub)(&y3, &y3, &Cc);
    r->x = x3; r->y = y3; r->z = z3;
}

/* madd-2007-bl with the exceptional cases handled */
static void C(jac_add_aff)(C(jac_t)* r, const C(jac_t)* p, const C(aff_t)* q) {
    if (q->inf) { *r = *p; return; }
    if (C(jac_is_inf)(p)) { C(jac_from_aff)(r, q); return; }
    FP(t) z1z1, u2, s2, h, hh, i, j, rr, v, t;
    FP(sqr)(&z1z1, &p->z);
    FP(mul)(&u2, &q->x, &z1z1);
    FP(mul)(&s2, &q->y, &p->z); FP(mul)(&s2, &s2, &z1z1);
    FP(sub)(&h, &u2, &p->x);
    FP(sub)(&rr, &s2, &p->y);
    if (FP(is_zero)(&h)) {
        if (FP(is_zero)(&rr)) { C(jac_dbl)(r, p); return; }
        C(jac_set_inf)(r); return;
    }
    FP(dbl)(&rr, &rr);
    FP(sqr)(&hh, &h);
    FP(dbl)(&i, &hh); FP(dbl)(&i, &i);
    FP(mul)(&j, &h, &i);
    FP(mul)(&v, &p->x, &i);
    FP(t) x3, y3, z3;
    FP(sqr)(&x3, &rr); FP(sub)(&x3, &x3, &j); FP(dbl)(&t, &v); FP(sub)(&x3, &x3, &t);
    FP(sub)(&t, &v, &x3); FP(mul)(&y3, &rr, &t);
    FP(mul)(&t, &p->y, &j); FP(dbl)(&t, &t); FP(sub)(&y3, &y3, &t);
    FP(add)(&z3, &p->z, &h); FP(sqr)(&z3, &z3); FP(sub)(&z3, &z3, &z1z1); FP(sub)(&z3, &z3, &hh);
    r->x = x3; r->y = y3; r->z = z3;
}

/* add-2007-bl with the exceptional cases handled */
static void C(jac_add)(C(jac_t)* r, const C(jac_t)* p, const C(jac_t)* q) {
    if (C(jac_is_inf)(q)) { *r = *p; return; }
    if (C(jac_is_inf)(p)) { *r = *q; return; }
    FP(t) z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
    FP(sqr)(&z1z1, &p->z); FP(sqr)(&z2z2, &q->z);
    FP(mul)(&u1, &p->x, &z2z2); FP(mul)(&u2, &q->x, &z1z1);
    FP(mul)(&s1, &p->y, &q->z); FP(mul)(&s1, &s1, &z2z2);
    FP(mul)(&s2, &q->y, &p->z); FP(mul)(&s2, &s2, &z1z1);
    FP(sub)(&h, &u2, &u1);
    FP(sub)(&rr, &s2, &s1);
    if (FP(is_zero)(&h)) {
        if (FP(is_zero)(&rr)) { C(jac_dbl)(r, p); return; }
        C(jac_set_inf)(r); return;
    }
    FP(dbl)(&rr, &rr);
    FP(dbl)(&i, &h); FP(sqr)(&i, &i);
    FP(mul)(&j, &h, &i);
    FP(mul)(&v, &u1, &i);
    FP(t) x3, y3, z3;
    FP(sqr)(&x3, &rr); FP(sub)(&x3, &x3, &j); FP(dbl)(&t, &v); FP(sub)(&x3, &x3, &t);
    FP(sub)(&t, &v, &x3); FP(mul)(&y3, &rr, &t);
    FP(mul)(&t, &s1, &j); FP(dbl)(&t, &t); FP(sub)(&y3, &y3, &t);
    FP(add)(&z3, &p->z, &q->z); FP(sqr)(&z3, &z3); FP(sub)(&z3, &z3, &z1z1); FP(sub)(&z3, &z3, &z2z2); FP(mul)(&z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

static void C(jac_neg)(C(jac_t)* r, const C(jac_t)* p) { r->x = p->x; r->z = p->z; FP(neg)(&r->y, &p->y); }
static void C(aff_neg)(C(aff_t)* r, const C(aff_t)* p) { *r = *p; if (!p->inf) FP(neg)(&r->y, &p->y); }

static void C(jac_to_aff)(C(aff_t)* r, const C(jac_t)* p) {
    if (C(jac_is_inf)(p)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
    FP(t) zi, zi2, zi3;
    FP(inv)(&zi, &p->z);
    FP(sqr)(&zi2, &zi); FP(mul)(&zi3, &zi2, &zi);
    FP(mul)(&r->x, &p->x, &zi2); FP(mul)(&r->y, &p->y, &zi3);
    r->inf = 0;
}

/* Batch normalisation (Montgomery's trick): out[i] = affine(in[i]) */
static void C(jac_batch_to_aff)(C(aff_t)* out, const C(jac_t)* in, size_t n) {
    if (!n) return;
    FP(t)* pre = (FP(t)*)malloc(n * sizeof(FP(t)));
    FP(t) acc; memcpy(acc.l, FP(P).one, sizeof acc.l);
    for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!C(jac_is_inf)(&in[i])) FP(mul)(&acc, &acc, &in[i].z); }
    FP(t) inv; FP(inv)(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (C(jac_is_inf)(&in[i])) { memset(&out[i], 0, sizeof out[i]); out[i].inf = 1; continue; }
        FP(t) zi, zi2, zi3;
        FP(mul)(&zi, &inv, &pre[i]);
        FP(mul)(&inv, &inv, &in[i].z);
        FP(sqr)(&zi2, &zi); FP(mul)(&zi3, &zi2, &zi);
        FP(mul)(&out[i].x, &in[i].x, &zi2); FP(mul)(&out[i].y, &in[i].y, &zi3); out[i].inf = 0;
    }
    free(pre);
}

static int C(aff_on_curve)(const C(aff_t)* p) {
    if (p->inf) return 1;
    FP(t) l, r;
    FP(sqr)(&l, &p->y);
    FP(sqr)(&r, &p->x); FP(mul)(&r, &r, &p->x); FP(add)(&r, &r, &C(B));
    return FP(eq)(&l, &r);
}

/* ---- byte formats (C-ABI BP_FMT_LE: x || y little-endian canonical; all-zero = identity) ------------ */
static void C(aff_from_le)(C(aff_t)* r, const uint8_t* in) {
    int allz = 1;
    for (int i = 0; i < 2 * FP_LE_BYTES; i++) if (in[i]) { allz = 0; break; }
    if (allz) { memset(r, 0, sizeof *r); r->inf = 1; return; }
    FP(from_le)(&r->x, in, FP_LE_BYTES); FP(from_le)(&r->y, in + FP_LE_BYTES, FP_LE_BYTES); r->inf = 0;
}
static void C(aff_to_le)(uint8_t* out, const C(aff_t)* p) {
    if (p->inf) { memset(out, 0, 2 * FP_LE_BYTES); return; }
    FP(to_le)(out, &p->x, FP_LE_BYTES); FP(to_le)(out + FP_LE_BYTES, &p->y, FP_LE_BYTES);
}
/* amcl ECP::tobytes(.., false): 04 || X || Y big-endian MODBYTES each; identity = 04 || 0 || 1
 * [UNVERIFIED-RECALL, SURVEY 8c]; used only for the transcript (src/transcript.rs:51-53). */
static void C(aff_to_amcl)(uint8_t* out, const C(aff_t)* p, int modbytes) {
    uint8_t le[2 * FP_LE_BYTES];
    memset(out, 0, 2 * modbytes + 1);
    out[0] = 0x04;
    if (p->inf) { out[2 * modbytes] = 1; return; }
    C(aff_to_le)(le, p);
    for (int i = 0; i < modbytes && i < FP_LE_BYTES; i++) { out[modbytes - i] = le[i]; out[2 * modbytes - i] = le[FP_LE_BYTES + i]; }
}

/* ---- scalar multiplication ------------------------------------------------------------------------------ */
/* k as plain little-endian 64-bit words (FR_NL of them) */
static void C(jac_mul_raw)(C(jac_t)* r, const uint64_t* k, const C(aff_t)* p) {
    C(jac_t) acc; C(jac_set_inf)(&acc);
    for (int i = FR_NL * 64 - 1; i >= 0; i--) {
        C(jac_dbl)(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) C(jac_add_aff)(&acc, &acc, p);
    }
    *r = acc;
}

/* width-w NAF of a plain integer; returns number of digits written (<= FR_NL*64+1) */
static int C(wnaf)(int8_t* out, const uint64_t* k_in, int w) {
    uint64_t k[FR_NL + 1];
    memcpy(k, k_in, FR_NL * 8); k[FR_NL] = 0;
    int len = 0;
    for (;;) {
        int nz = 0; for (int i = 0; i <= FR_NL; i++) if (k[i]) { nz = 1; break; }
        if (!nz) break;
        int d = 0;
        if (k[0] & 1) {
            d = (int)(k[0] & ((1u << w) - 1));
            if (d >= (1 << (w - 1))) d -= (1 << w);
            /* k -= d */
            if (d >= 0) {
                uint64_t b = (uint64_t)d;
                for (int i = 0; i <= FR_NL && b; i++) { uint64_t o = k[i]; k[i] = o - b; b = o < b; }
            } else {
                uint64_t c = (uint64_t)(-d);
                for (int i = 0; i <= FR_NL && c; i++) { uint64_t o = k[i]; k[i] = o + c; c = k[i] < o; }
            }
        }
        out[len++] = (int8_t)d;
        for (int i = 0; i < FR_NL; i++) k[i] = (k[i] >> 1) | (k[i + 1] << 63);
        k[FR_NL] >>= 1;
    }
    return len;
}

/* ---- MSM: three algorithms, same group element ---------------------------------------------------------- */

/* (1) naive: sum of independent double-and-add products */
static void C(msm_naive)(C(jac_t)* r, const C(aff_t)* pts, const uint64_t* ks, size_t n) {
    C(jac_t) acc; C(jac_set_inf)(&acc);
    for (size_t i = 0; i < n; i++) {
        C(jac_t) t; C(jac_mul_raw)(&t, ks + i * FR_NL, &pts[i]);
        C(jac_add)(&acc, &acc, &t);
    }
    *r = acc;
}

/* (2) Strauss / interleaved wNAF-5, single thread: the algorithm class amcl_wrapper's
 *     multi_scalar_mul_var_time is believed to use ([UNVERIFIED-RECALL]).  This is the
 *     "reference-like" CPU baseline. */
static void C(msm_strauss)(C(jac_t)* r, const C(aff_t)* pts, const uint64_t* ks, size_t n) {
    enum { W = 5, TBL = 1 << (W - 2) };
    if (!n) { C(jac_set_inf)(r); return; }
    C(jac_t)* tj = (C(jac_t)*)malloc(n * TBL * sizeof(C(jac_t)));
    C(aff_t)* ta = (C(aff_t)*)malloc(n * TBL * sizeof(C(aff_t)));
    int8_t* naf = (int8_t*)calloc(n, FR_NL * 64 + 2);
    int maxlen = 0;
    for (size_t i = 0; i < n; i++) {
        C(jac_t) p2, p1; C(jac_from_aff)(&p1, &pts[i]); C(jac_dbl)(&p2, &p1);
        tj[i * TBL] = p1;
        for (int j = 1; j < TBL; j++) C(jac_add)(&tj[i * TBL + j], &tj[i * TBL + j - 1], &p2);
        int l = C(wnaf)(naf + i * (FR_NL * 64 + 2), ks + i * FR_NL, W);
        if (l > maxlen) maxlen = l;
    }
    C(jac_batch_to_aff)(ta, tj, n * TBL);
    C(jac_t) acc; C(jac_set_inf)(&acc);
    for (int b = maxlen - 1; b >= 0; b--) {
        C(jac_dbl)(&acc, &acc);
        for (size_t i = 0; i < n; i++) {
            int d = naf[i * (FR_NL * 64 + 2) + b];
            if (!d) continue;
            if (d > 0) C(jac_add_aff)(&acc, &acc, &ta[i * TBL + (d >> 1)]);
            else { C(aff_t) m; C(aff_neg)(&m, &ta[i * TBL + ((-d) >> 1)]); C(jac_add_aff)(&acc, &acc, &m); }
        }
    }
    free(tj); free(ta); free(naf);
    *r = acc;
}

/* (3) Pippenger bucket method, unsigned c-bit windows, optionally multi-threaded over (window, slice). */
static inline unsigned C(get_window)(const uint64_t* k, int bit, int c) {
    int w = bit / 64, o = bit % 64;
    uint64_t v = k[w] >> o;
    if (o + c > 64 && w + 1 < FR_NL) v |= k[w + 1] << (64 - o);
    return (unsigned)(v & ((1ull << c) - 1));
}

static void C(pip_window)(C(jac_t)* out, const C(aff_t)* pts, const uint64_t* ks, size_t n, int bit, int c) {
    size_t nb = ((size_t)1 << c) - 1;
    C(jac_t)* bk = (C(jac_t)*)malloc(nb * sizeof(C(jac_t)));
    for (size_t i = 0; i < nb; i++) C(jac_set_inf)(&bk[i]);
    for (size_t i = 0; i < n; i++) {
        unsigned d = C(get_window)(ks + i * FR_NL, bit, c);
        if (d) C(jac_add_aff)(&bk[d - 1], &bk[d - 1], &pts[i]);
    }
    C(jac_t) run, sum; C(jac_set_inf)(&run); C(jac_set_inf)(&sum);
    for (size_t i = nb; i-- > 0;) { C(jac_add)(&run, &run, &bk[i]); C(jac_add)(&sum, &sum, &run); }
    free(bk);
    *out = sum;
}

typedef struct { const C(aff_t)* pts; const uint64_t* ks; size_t n; int c, nwin, nslice; C(jac_t)* partial; volatile int* next; } C(pip_job_t);

static void* C(pip_worker)(void* arg) {
    C(pip_job_t)* j = (C(pip_job_t)*)arg;
    for (;;) {
        int t = __sync_fetch_and_add(j->next, 1);
        if (t >= j->nwin * j->nslice) break;
        int w = t / j->nslice, s = t % j->nslice;
        size_t lo = j->n * (size_t)s / j->nslice, hi = j->n * (size_t)(s + 1) / j->nslice;
        C(pip_window)(&j->partial[t], j->pts + lo, j->ks + lo * FR_NL, hi - lo, w * j->c, j->c);
    }
    return 0;
}

static int C(pip_pick_c)(size_t n) {
    int c = 1; while (((size_t)1 << (c + 3)) < n + 1 && c < 16) c++;   /* ~ log2(n) - 3 */
    if (c < 2) c = 2;
    return c;
}

static void C(msm_pippenger)(C(jac_t)* r, const C(aff_t)* pts, const uint64_t* ks, size_t n, int nthreads) {
    if (!n) { C(jac_set_inf)(r); return; }
    if (nthreads < 1) nthreads = 1;
    int nslice = 1;
    size_t per = n / (nthreads > 1 ? (size_t)nthreads : 1);
    int c = C(pip_pick_c)(nthreads > 1 && per > 64 ? per : n);
    int nwin = (FR(P).bits + c - 1) / c;
    if (nthreads > 1) { nslice = (nthreads + nwin - 1) / nwin; if ((size_t)nslice > n) nslice = 1; }
    C(jac_t)* partial = (C(jac_t)*)malloc((size_t)nwin * nslice * sizeof(C(jac_t)));
    volatile int next = 0;
    C(pip_job_t) job = { pts, ks, n, c, nwin, nslice, partial, &next };
    if (nthreads == 1) C(pip_worker)(&job);
    else {
        pthread_t* th = (pthread_t*)malloc(nthreads * sizeof(pthread_t));
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], 0, C(pip_worker), &job);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], 0);
        free(th);
    }
    C(jac_t) acc; C(jac_set_inf)(&acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int i = 0; i < c; i++) C(jac_dbl)(&acc, &acc);
        for (int s = 0; s < nslice; s++) C(jac_add)(&acc, &acc, &partial[w * nslice + s]);
    }
    free(partial);
    *r = acc;
}
