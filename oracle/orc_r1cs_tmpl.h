/* oracle/orc_r1cs_tmpl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * The R1CS layer around the inner-product argument, restated from the reference for constraint systems without
 * deferred (second-phase) constraints:
 *   Prover::prove            src/r1cs/prover.rs:322-593
 *   flattened_constraints    src/r1cs/prover.rs:142-184, src/r1cs/verifier.rs:149-193
 *   Verifier::verify         src/r1cs/verifier.rs:267-457
 *   VecPoly3 / Poly6         src/utils/vector_poly.rs:79-120
 * A constraint system arrives as flat term arrays (constraint q, variable kind 0/1/2 = MultiplierLeft/Right/Output,
 * 3 = Committed, 4 = One, index, coefficient); the transcript already holds r1cs_domain_sep and the V commitments
 * (Prover::new / commit, prover.rs:84-127).  Included once per curve after orc_ipp_tmpl.h.
 */

typedef struct {
    size_t n_terms, n_constraints;
    const uint32_t* con;
    const uint8_t* kind;
    const uint32_t* idx;
    const FR(t)* coeff;
} C(r1cs_terms_t);

/* wL, wR, wO (n), wV (m), *wc */
static void C(r1cs_flatten)(const C(r1cs_terms_t)* cs, const FR(t)* z, size_t n, size_t m, FR(t)* wL, FR(t)* wR, FR(t)* wO, FR(t)* wV, FR(t)* wc) {
    memset(wL, 0, n * sizeof *wL); memset(wR, 0, n * sizeof *wR); memset(wO, 0, n * sizeof *wO); memset(wV, 0, (m ? m : 1) * sizeof *wV);
    memset(wc, 0, sizeof *wc);
    FR(t)* zp = (FR(t)*)malloc((cs->n_constraints ? cs->n_constraints : 1) * sizeof *zp);     /* exp_z of constraint q = z^(q+1) */
    if (cs->n_constraints) zp[0] = *z;
    for (size_t q = 1; q < cs->n_constraints; q++) FR(mul)(&zp[q], &zp[q - 1], z);
    for (size_t k = 0; k < cs->n_terms; k++) {
        FR(t) v; FR(mul)(&v, &zp[cs->con[k]], &cs->coeff[k]);
        switch (cs->kind[k]) {
            case 0: FR(add)(&wL[cs->idx[k]], &wL[cs->idx[k]], &v); break;
            case 1: FR(add)(&wR[cs->idx[k]], &wR[cs->idx[k]], &v); break;
            case 2: FR(add)(&wO[cs->idx[k]], &wO[cs->idx[k]], &v); break;
            case 3: FR(sub)(&wV[cs->idx[k]], &wV[cs->idx[k]], &v); break;
            default: FR(sub)(wc, wc, &v); break;
        }
    }
    free(zp);
}

static void C(commit2)(C(aff_t)* out, const C(aff_t)* g, const C(aff_t)* h, const FR(t)* m_, const FR(t)* r_) {   /* commit_to_field_element */
    C(binary_scalar_mul)(out, g, h, m_, r_);
}

/* <a, G> + <b, H> + c h   (commit_to_field_element_vectors; b may be NULL) */
static void C(commit_vectors)(C(aff_t)* out, const C(aff_t)* G, const C(aff_t)* H, const C(aff_t)* h, const FR(t)* a, const FR(t)* b, const FR(t)* c, size_t n) {
    size_t k = b ? 2 * n + 1 : n + 1;
    C(aff_t)* pts = (C(aff_t)*)malloc(k * sizeof *pts);
    FR(t)* sc = (FR(t)*)malloc(k * sizeof *sc);
    memcpy(pts, G, n * sizeof *pts); memcpy(sc, a, n * sizeof *sc);
    if (b) { memcpy(pts + n, H, n * sizeof *pts); memcpy(sc + n, b, n * sizeof *sc); }
    pts[k - 1] = *h; sc[k - 1] = *c;
    C(msm_fr)(out, pts, sc, k);
    free(pts); free(sc);
}

static size_t C(next_pow2)(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

typedef struct {
    C(aff_t) pts[11];                /* A_I1 A_O1 S1 A_I2 A_O2 S2 T_1 T_3 T_4 T_5 T_6 */
    FR(t) t_x, t_x_blinding, e_blinding;
    C(aff_t) L[64], R[64];
    int lg;
    FR(t) a, b;
} C(r1cs_proof_t);

static void C(poly6_eval)(FR(t)* out, const FR(t) t[6], const FR(t)* x) {   /* vector_poly.rs:116-119 */
    FR(t) acc = t[5];
    for (int k = 4; k >= 0; k--) { FR(mul)(&acc, &acc, x); FR(add)(&acc, &acc, &t[k]); }
    FR(mul)(out, &acc, x);
}

/* blind: i, o, s, t1, t3, t4, t5, t6.  0 ok, 1 = InvalidGeneratorsLength */
static int C(r1cs_prove)(orc_transcript* tr, const C(r1cs_terms_t)* cs, size_t n, size_t m, const C(aff_t)* g, const C(aff_t)* h,
                         const C(aff_t)* G, const C(aff_t)* H, size_t ngens, const FR(t)* aL, const FR(t)* aR, const FR(t)* aO,
                         const FR(t)* v_blinding, const FR(t)* sL, const FR(t)* sR, const FR(t) blind[8], C(r1cs_proof_t)* pf) {
    orc_transcript_append_u64(tr, (const uint8_t*)"m", 1, (uint64_t)m);                         /* :327 */
    const size_t n1 = n;
    if (ngens < n1) return 1;                                                                  /* :332 */
    C(commit_vectors)(&pf->pts[0], G, H, h, aL, aR, &blind[0], n1);                            /* A_I1 :347 */
    C(commit_vectors)(&pf->pts[1], G, H, h, aO, NULL, &blind[1], n1);                          /* A_O1 :358 */
    C(commit_vectors)(&pf->pts[2], G, H, h, sL, sR, &blind[2], n1);                            /* S1   :361 */
    C(t_commit_point)(tr, "A_I1", &pf->pts[0]); C(t_commit_point)(tr, "A_O1", &pf->pts[1]); C(t_commit_point)(tr, "S1", &pf->pts[2]);
    orc_transcript_append(tr, (const uint8_t*)"dom-sep", 7, (const uint8_t*)"r1cs-1phase", 11);   /* :369 -> :304-306 */
    const size_t padded_n = C(next_pow2)(n), pad = padded_n - n;                               /* :374-377 */
    if (ngens < padded_n) return 1;                                                            /* :379 */
    for (int k = 3; k < 6; k++) { memset(&pf->pts[k], 0, sizeof pf->pts[k]); pf->pts[k].inf = 1; }   /* :429 identity */
    C(t_commit_point)(tr, "A_I2", &pf->pts[3]); C(t_commit_point)(tr, "A_O2", &pf->pts[4]); C(t_commit_point)(tr, "S2", &pf->pts[5]);
    FR(t) y, z, y_inv, one;
    memcpy(one.l, FR(P).one, sizeof one.l);
    C(t_challenge_scalar)(tr, "y", &y); C(t_challenge_scalar)(tr, "z", &z);                    /* :438-439 */
    FR(t)* w = (FR(t)*)malloc((3 * n + m + 1) * sizeof *w);
    FR(t) *wL = w, *wR = w + n, *wO = w + 2 * n, *wV = w + 3 * n, wc;
    C(r1cs_flatten)(cs, &z, n, m, wL, wR, wO, wV, &wc);                                        /* :441 */
    FR(t)* pl = (FR(t)*)malloc(6 * (n ? n : 1) * sizeof *pl);
    FR(t) *l1 = pl, *l2 = pl + n, *l3 = pl + 2 * n, *r0 = pl + 3 * n, *r1 = pl + 4 * n, *r3 = pl + 5 * n;
    FR(t)* yinv = (FR(t)*)malloc(padded_n * sizeof *yinv);
    FR(inv)(&y_inv, &y);
    yinv[0] = one;
    for (size_t i = 1; i < padded_n; i++) FR(mul)(&yinv[i], &yinv[i - 1], &y_inv);              /* :463 */
    FR(t) exp_y = one, t_;
    for (size_t i = 0; i < n; i++) {                                                           /* :469-486 */
        FR(mul)(&t_, &yinv[i], &wR[i]); FR(add)(&l1[i], &aL[i], &t_);
        l2[i] = aO[i];
        l3[i] = sL[i];
        FR(sub)(&r0[i], &wO[i], &exp_y);
        FR(mul)(&t_, &exp_y, &aR[i]); FR(add)(&r1[i], &t_, &wL[i]);
        FR(mul)(&r3[i], &exp_y, &sR[i]);
        FR(mul)(&exp_y, &exp_y, &y);
    }
    FR(t) tc[6], u1, u2;                                                                       /* special_inner_product, vector_poly.rs:79-97 */
    C(fr_inner)(&tc[0], l1, r0, n);
    C(fr_inner)(&u1, l1, r1, n); C(fr_inner)(&u2, l2, r0, n); FR(add)(&tc[1], &u1, &u2);
    C(fr_inner)(&u1, l2, r1, n); C(fr_inner)(&u2, l3, r0, n); FR(add)(&tc[2], &u1, &u2);
    C(fr_inner)(&u1, l1, r3, n); C(fr_inner)(&u2, l3, r1, n); FR(add)(&tc[3], &u1, &u2);
    C(fr_inner)(&tc[4], l2, r3, n);
    C(fr_inner)(&tc[5], l3, r3, n);
    static const int tk[5] = {0, 2, 3, 4, 5};                                                  /* T_1 T_3 T_4 T_5 T_6 :496-500 */
    static const char* tl[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
    for (int k = 0; k < 5; k++) { C(commit2)(&pf->pts[6 + k], g, h, &tc[tk[k]], &blind[3 + k]); C(t_commit_point)(tr, tl[k], &pf->pts[6 + k]); }
    FR(t) u, x;
    C(t_challenge_scalar)(tr, "u", &u); C(t_challenge_scalar)(tr, "x", &x);                    /* :508-509 */
    FR(t) tb[6];
    tb[0] = blind[3]; C(fr_inner)(&tb[1], wV, v_blinding, m); tb[2] = blind[4]; tb[3] = blind[5]; tb[4] = blind[6]; tb[5] = blind[7];   /* :513-522 */
    C(poly6_eval)(&pf->t_x, tc, &x); C(poly6_eval)(&pf->t_x_blinding, tb, &x);                 /* :524-525 */
    FR(t)* lv = (FR(t)*)malloc(2 * padded_n * sizeof *lv);
    FR(t)* rv = lv + padded_n;
    memset(lv, 0, 2 * padded_n * sizeof *lv);
    for (size_t i = 0; i < n; i++) {                                                           /* VecPoly3::eval, vector_poly.rs:99-106 */
        FR(mul)(&t_, &x, &l3[i]); FR(add)(&t_, &t_, &l2[i]); FR(mul)(&t_, &t_, &x); FR(add)(&t_, &t_, &l1[i]); FR(mul)(&lv[i], &t_, &x);
        FR(mul)(&t_, &x, &r3[i]); FR(mul)(&t_, &t_, &x); FR(add)(&t_, &t_, &r1[i]); FR(mul)(&t_, &t_, &x); FR(add)(&rv[i], &t_, &r0[i]);
    }
    for (size_t i = n; i < padded_n; i++) { FR(neg)(&rv[i], &exp_y); FR(mul)(&exp_y, &exp_y, &y); }   /* :532-535 */
    FR(mul)(&t_, &x, &blind[2]); FR(add)(&t_, &t_, &blind[1]); FR(mul)(&t_, &t_, &x); FR(add)(&t_, &t_, &blind[0]); FR(mul)(&pf->e_blinding, &t_, &x);   /* :537-541, second phase = 0 */
    uint8_t sb[MODBYTES];
    const FR(t)* three[3] = {&pf->t_x, &pf->t_x_blinding, &pf->e_blinding};
    static const char* sl[3] = {"t_x", "t_x_blinding", "e_blinding"};
    for (int k = 0; k < 3; k++) { C(fr_to_be)(sb, three[k]); orc_transcript_append(tr, (const uint8_t*)sl[k], strlen(sl[k]), sb, MODBYTES); }   /* :543-546 */
    FR(t) wch;
    C(t_challenge_scalar)(tr, "w", &wch);                                                      /* :549 */
    C(aff_t) Q; { uint64_t raw[FR_NL]; FR(to_raw)(raw, &wch); C(jac_t) j; C(jac_mul_raw)(&j, raw, g); C(jac_to_aff)(&Q, &j); }   /* :550 */
    FR(t)* Gf = (FR(t)*)malloc(2 * padded_n * sizeof *Gf);
    FR(t)* Hf = Gf + padded_n;
    for (size_t i = 0; i < padded_n; i++) { Gf[i] = i < n1 ? one : u; FR(mul)(&Hf[i], &yinv[i], &Gf[i]); }   /* :552-563 */
    pf->lg = C(ipp_create)(tr, &Q, Gf, Hf, G, H, lv, rv, padded_n, pf->L, pf->R, &pf->a, &pf->b);   /* :565-574 */
    (void)pad;
    free(w); free(pl); free(yinv); free(lv); free(Gf);
    return 0;
}

/* 0 accepted, 3 = VerificationError, 1 = InvalidGeneratorsLength.  rnd = the verifier's random r (:392). */
static int C(r1cs_verify)(orc_transcript* tr, const C(r1cs_terms_t)* cs, size_t n, size_t m, const C(aff_t)* V, const C(r1cs_proof_t)* pf,
                          const C(aff_t)* g, const C(aff_t)* h, const C(aff_t)* G, const C(aff_t)* H, size_t ngens, const FR(t)* rnd) {
    orc_transcript_append_u64(tr, (const uint8_t*)"m", 1, (uint64_t)m);                         /* :279 */
    const size_t n1 = n;
    C(t_commit_point)(tr, "A_I1", &pf->pts[0]); C(t_commit_point)(tr, "A_O1", &pf->pts[1]); C(t_commit_point)(tr, "S1", &pf->pts[2]);   /* :282-284 */
    orc_transcript_append(tr, (const uint8_t*)"dom-sep", 7, (const uint8_t*)"r1cs-1phase", 11);   /* :287 */
    const size_t padded_n = C(next_pow2)(n), pad = padded_n - n;
    if (ngens < padded_n) return 1;                                                            /* :297-299 */
    C(t_commit_point)(tr, "A_I2", &pf->pts[3]); C(t_commit_point)(tr, "A_O2", &pf->pts[4]); C(t_commit_point)(tr, "S2", &pf->pts[5]);
    FR(t) y, z, u, x, wch, one;
    memcpy(one.l, FR(P).one, sizeof one.l);
    C(t_challenge_scalar)(tr, "y", &y); C(t_challenge_scalar)(tr, "z", &z);
    static const char* tl[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
    for (int k = 0; k < 5; k++) C(t_commit_point)(tr, tl[k], &pf->pts[6 + k]);                 /* :308-312 */
    C(t_challenge_scalar)(tr, "u", &u); C(t_challenge_scalar)(tr, "x", &x);
    uint8_t sb[MODBYTES];
    const FR(t)* three[3] = {&pf->t_x, &pf->t_x_blinding, &pf->e_blinding};
    static const char* sl[3] = {"t_x", "t_x_blinding", "e_blinding"};
    for (int k = 0; k < 3; k++) { C(fr_to_be)(sb, three[k]); orc_transcript_append(tr, (const uint8_t*)sl[k], strlen(sl[k]), sb, MODBYTES); }   /* :317-321 */
    C(t_challenge_scalar)(tr, "w", &wch);                                                      /* :323 */
    FR(t)* w = (FR(t)*)malloc((3 * n + m + 1) * sizeof *w);
    FR(t) *wL = w, *wR = w + n, *wO = w + 2 * n, *wV = w + 3 * n, wc;
    C(r1cs_flatten)(cs, &z, n, m, wL, wR, wO, wV, &wc);                                        /* :325 */
    FR(t) y_inv, delta, t_, t2;
    FR(inv)(&y_inv, &y);
    FR(t)* yinv = (FR(t)*)malloc(2 * padded_n * sizeof *yinv);
    FR(t)* yinv_wR = yinv + padded_n;
    yinv[0] = one;
    for (size_t i = 1; i < padded_n; i++) FR(mul)(&yinv[i], &yinv[i - 1], &y_inv);              /* :342 */
    memset(yinv_wR, 0, padded_n * sizeof *yinv_wR);
    for (size_t i = 0; i < n; i++) FR(mul)(&yinv_wR[i], &wR[i], &yinv[i]);                      /* :343-348 */
    C(fr_inner)(&delta, yinv_wR, wL, n);                                                       /* :350-352 */
    const size_t lg = (size_t)pf->lg;
    FR(t)* s = (FR(t)*)malloc(padded_n * sizeof *s);
    FR(t) u_sq[64], u_inv_sq[64];
    int rc = C(ipp_verification_scalars)(tr, pf->L, pf->R, lg, padded_n, u_sq, u_inv_sq, s);    /* :354-360 */
    if (rc) { free(w); free(yinv); free(s); return 3; }
    const size_t total = 6 + m + 5 + 2 + 2 * padded_n + 2 * lg;                                /* :431-446 */
    FR(t)* sc = (FR(t)*)malloc(total * sizeof *sc);
    C(aff_t)* pt = (C(aff_t)*)malloc(total * sizeof *pt);
    FR(t) x2, x3, rx2;
    FR(sqr)(&x2, &x); FR(mul)(&x3, &x, &x2); FR(mul)(&rx2, rnd, &x2);                           /* :394-396 */
    size_t k = 0;
    sc[0] = x; sc[1] = x2; sc[2] = x3; FR(mul)(&sc[3], &u, &x); FR(mul)(&sc[4], &u, &x2); FR(mul)(&sc[5], &u, &x3);   /* :410-415 */
    for (int j = 0; j < 6; j++) pt[k++] = pf->pts[j];
    for (size_t j = 0; j < m; j++) { FR(mul)(&sc[k], &wV[j], &rx2); pt[k++] = V[j]; }           /* :416-418 */
    FR(mul)(&sc[k], rnd, &x); FR(mul)(&sc[k + 1], rnd, &x3);                                    /* rx, rx^3 :400-401 */
    FR(mul)(&sc[k + 2], &sc[k + 1], &x); FR(mul)(&sc[k + 3], &sc[k + 2], &x); FR(mul)(&sc[k + 4], &sc[k + 3], &x);   /* rx^4..rx^6 :403-408 */
    for (int j = 0; j < 5; j++) pt[k++] = pf->pts[6 + j];
    /* w (t_x - a b) + r (x^2 (wc + delta) - t_x)  on g   :421-422 */
    FR(mul)(&t_, &pf->a, &pf->b); FR(sub)(&t_, &pf->t_x, &t_); FR(mul)(&t_, &wch, &t_);
    FR(add)(&t2, &wc, &delta); FR(mul)(&t2, &x2, &t2); FR(sub)(&t2, &t2, &pf->t_x); FR(mul)(&t2, rnd, &t2);
    FR(add)(&sc[k], &t_, &t2); pt[k++] = *g;
    FR(mul)(&t_, rnd, &pf->t_x_blinding); FR(add)(&t_, &t_, &pf->e_blinding); FR(neg)(&sc[k], &t_); pt[k++] = *h;   /* :424-425 */
    for (size_t i = 0; i < padded_n; i++) {                                                    /* g_scalars :368-373 */
        FR(mul)(&t_, &x, &yinv_wR[i]); FR(mul)(&t2, &pf->a, &s[i]); FR(sub)(&t_, &t_, &t2);
        if (i >= n1) FR(mul)(&t_, &t_, &u);
        sc[k] = t_; pt[k++] = G[i];
    }
    for (size_t i = 0; i < padded_n; i++) {                                                    /* h_scalars :375-390 */
        FR(t) acc; memset(&acc, 0, sizeof acc);
        if (i < n) { FR(mul)(&acc, &x, &wL[i]); FR(add)(&acc, &acc, &wO[i]); }
        FR(mul)(&t2, &pf->b, &s[padded_n - 1 - i]); FR(sub)(&acc, &acc, &t2);
        FR(mul)(&acc, &acc, &yinv[i]); FR(sub)(&acc, &acc, &one);
        if (i >= n1) FR(mul)(&acc, &acc, &u);
        sc[k] = acc; pt[k++] = H[i];
    }
    for (size_t j = 0; j < lg; j++) { sc[k] = u_sq[j]; pt[k++] = pf->L[j]; }                    /* :428, :445 */
    for (size_t j = 0; j < lg; j++) { sc[k] = u_inv_sq[j]; pt[k++] = pf->R[j]; }                /* :429, :446 */
    C(aff_t) res;
    C(msm_fr)(&res, pt, sc, total);                                                            /* :451 */
    (void)pad;
    free(w); free(yinv); free(s); free(sc); free(pt);
    return res.inf ? 0 : 3;                                                                    /* :452-454 */
}
