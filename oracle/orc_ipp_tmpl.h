/* oracle/orc_ipp_tmpl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Inner-product argument, restated line by line from the reference:
 *   create_ipp            src/ipp.rs:35-202
 *   verification_scalars  src/ipp.rs:262-315
 *   verify_ipp            src/ipp.rs:204-260
 *   transcript protocol   src/transcript.rs:29-61
 * Included once per curve after orc_curve_tmpl.h with the same C()/FP()/FR() macros plus MODBYTES.
 */

static void C(t_commit_point)(orc_transcript* t, const char* label, const C(aff_t)* p) {   /* transcript.rs:51-53 */
    uint8_t buf[2 * MODBYTES + 1];
    C(aff_to_amcl)(buf, p, MODBYTES);
    orc_transcript_append(t, (const uint8_t*)label, strlen(label), buf, sizeof buf);
}

/* FieldElement::to_bytes: MODBYTES big-endian (commit_scalar, transcript.rs:47-49) */
static void C(fr_to_be)(uint8_t* out, const FR(t)* x) {
    uint8_t le[FR_LE_BYTES];
    FR(to_le)(le, x, FR_LE_BYTES);
    memset(out, 0, MODBYTES);
    for (int i = 0; i < FR_LE_BYTES; i++) out[MODBYTES - 1 - i] = le[i];
}

static void C(t_challenge_scalar)(orc_transcript* t, const char* label, FR(t)* out) {        /* transcript.rs:55-60 */
    uint8_t buf[MODBYTES];
    orc_transcript_challenge(t, (const uint8_t*)label, strlen(label), buf, MODBYTES);
    FR(from_be_reduce)(out, buf, MODBYTES);
}

static void C(t_ipp_domain_sep)(orc_transcript* t, uint64_t n) {                              /* transcript.rs:30-33 */
    orc_transcript_append(t, (const uint8_t*)"dom-sep", 7, (const uint8_t*)"ipp v1", 6);
    orc_transcript_append_u64(t, (const uint8_t*)"n", 1, n);
}

static void C(fr_inner)(FR(t)* r, const FR(t)* a, const FR(t)* b, size_t n) {
    FR(t) acc; memset(&acc, 0, sizeof acc);
    for (size_t i = 0; i < n; i++) { FR(t) t; FR(mul)(&t, &a[i], &b[i]); FR(add)(&acc, &acc, &t); }
    *r = acc;
}

static void C(msm_fr)(C(aff_t)* out, const C(aff_t)* pts, const FR(t)* ks, size_t n) {
    uint64_t* raw = (uint64_t*)malloc((n ? n : 1) * FR_NL * 8);
    for (size_t i = 0; i < n; i++) FR(to_raw)(raw + i * FR_NL, &ks[i]);
    C(jac_t) j;
    if (n <= 32) C(msm_naive)(&j, pts, raw, n); else C(msm_pippenger)(&j, pts, raw, n, orc_threads);
    C(jac_to_aff)(out, &j);
    free(raw);
}

/* self*r1 + h*r2  (G1::binary_scalar_mul, src/ipp.rs:119,125,185,187) */
static void C(binary_scalar_mul)(C(aff_t)* out, const C(aff_t)* p, const C(aff_t)* h, const FR(t)* r1, const FR(t)* r2) {
    uint64_t k1[FR_NL], k2[FR_NL];
    FR(to_raw)(k1, r1); FR(to_raw)(k2, r2);
    C(jac_t) a, b;
    C(jac_mul_raw)(&a, k1, p); C(jac_mul_raw)(&b, k2, h);
    C(jac_add)(&a, &a, &b);
    C(jac_to_aff)(out, &a);
}

/* the fold loop of one round (src/ipp.rs:115-130 first round, :181-188 later rounds) over [lo, hi): independent per i */
typedef struct { FR(t) *aL, *aR, *bL, *bR; C(aff_t) *GL, *GR, *HL, *HR; const FR(t) *Gf, *Hf; size_t n; int first; FR(t) u, ui; } C(fold_job_t);
static void C(fold_range)(size_t lo, size_t hi, void* arg) {
    C(fold_job_t)* j = (C(fold_job_t)*)arg;
    const size_t n = j->n;
    const FR(t) *u = &j->u, *ui = &j->ui;
    for (size_t i = lo; i < hi; i++) {
        FR(t) t1, t2;
        FR(mul)(&t1, &j->aL[i], u); FR(mul)(&t2, ui, &j->aR[i]); FR(add)(&j->aL[i], &t1, &t2);
        FR(mul)(&t1, &j->bL[i], ui); FR(mul)(&t2, u, &j->bR[i]); FR(add)(&j->bL[i], &t1, &t2);
        if (j->first) {
            FR(mul)(&t1, ui, &j->Gf[i]); FR(mul)(&t2, u, &j->Gf[n + i]);
            C(binary_scalar_mul)(&j->GL[i], &j->GL[i], &j->GR[i], &t1, &t2);
            FR(mul)(&t1, u, &j->Hf[i]); FR(mul)(&t2, ui, &j->Hf[n + i]);
            C(binary_scalar_mul)(&j->HL[i], &j->HL[i], &j->HR[i], &t1, &t2);
        } else {
            C(binary_scalar_mul)(&j->GL[i], &j->GL[i], &j->GR[i], ui, u);
            C(binary_scalar_mul)(&j->HL[i], &j->HL[i], &j->HR[i], u, ui);
        }
    }
}

/* Returns lg n.  L_out/R_out have room for lg n points. */
static int C(ipp_create)(orc_transcript* tr, const C(aff_t)* Q, const FR(t)* Gf, const FR(t)* Hf,
                         const C(aff_t)* G_in, const C(aff_t)* H_in, const FR(t)* a_in, const FR(t)* b_in, size_t n,
                         C(aff_t)* L_out, C(aff_t)* R_out, FR(t)* a_out, FR(t)* b_out) {
    C(aff_t)* G = (C(aff_t)*)malloc(n * sizeof *G); memcpy(G, G_in, n * sizeof *G);     /* :57-60 */
    C(aff_t)* H = (C(aff_t)*)malloc(n * sizeof *H); memcpy(H, H_in, n * sizeof *H);
    FR(t)* a = (FR(t)*)malloc(n * sizeof *a); memcpy(a, a_in, n * sizeof *a);
    FR(t)* b = (FR(t)*)malloc(n * sizeof *b); memcpy(b, b_in, n * sizeof *b);
    C(aff_t)* mp = (C(aff_t)*)malloc((n + 1) * sizeof *mp);
    FR(t)* ms = (FR(t)*)malloc((n + 1) * sizeof *ms);
    C(t_ipp_domain_sep)(tr, n);                                                           /* :62 */
    int rounds = 0, first = 1;
    while (n != 1) {
        n /= 2;
        FR(t) *aL = a, *aR = a + n, *bL = b, *bR = b + n;
        C(aff_t) *GL = G, *GR = G + n, *HL = H, *HR = H + n;
        FR(t) cL, cR;
        C(fr_inner)(&cL, aL, bR, n);                                                      /* :77 / :145 */
        C(fr_inner)(&cR, aR, bL, n);                                                      /* :78 / :146 */
        /* L = <a_L (.Gf_R), G_R> + <b_R (.Hf_L), H_L> + c_L Q                           :80-91 / :148-158 */
        for (size_t i = 0; i < n; i++) {
            mp[i] = GR[i]; mp[n + i] = HL[i];
            if (first) { FR(mul)(&ms[i], &aL[i], &Gf[n + i]); FR(mul)(&ms[n + i], &bR[i], &Hf[i]); }
            else { ms[i] = aL[i]; ms[n + i] = bR[i]; }
        }
        mp[2 * n] = *Q; ms[2 * n] = cL;
        C(msm_fr)(&L_out[rounds], mp, ms, 2 * n + 1);
        /* R = <a_R (.Gf_L), G_L> + <b_L (.Hf_R), H_R> + c_R Q                           :93-104 / :160-170 */
        for (size_t i = 0; i < n; i++) {
            mp[i] = GL[i]; mp[n + i] = HR[i];
            if (first) { FR(mul)(&ms[i], &aR[i], &Gf[i]); FR(mul)(&ms[n + i], &bL[i], &Hf[n + i]); }
            else { ms[i] = aR[i]; ms[n + i] = bL[i]; }
        }
        mp[2 * n] = *Q; ms[2 * n] = cR;
        C(msm_fr)(&R_out[rounds], mp, ms, 2 * n + 1);
        C(t_commit_point)(tr, "L", &L_out[rounds]);                                       /* :106-107 / :172-173 */
        C(t_commit_point)(tr, "R", &R_out[rounds]);
        FR(t) u, ui;
        C(t_challenge_scalar)(tr, "u", &u);                                               /* :112 / :178 */
        FR(inv)(&ui, &u);                                                                 /* :113 / :179 */
        C(fold_job_t) fj = {aL, aR, bL, bR, GL, GR, HL, HR, Gf, Hf, n, first, u, ui};
        orc_parallel_for(n, C(fold_range), &fj);                                          /* :115-130 / :181-188 */
        first = 0; rounds++;
    }
    *a_out = a[0]; *b_out = b[0];                                                         /* :196-201 */
    free(G); free(H); free(a); free(b); free(mp); free(ms);
    return rounds;
}

/* 0 on success, 3 (verification error) otherwise; s has room for n, u_sq/u_inv_sq for lg_n */
static int C(ipp_verification_scalars)(orc_transcript* tr, const C(aff_t)* L, const C(aff_t)* R, size_t lg_n, size_t n,
                                       FR(t)* u_sq, FR(t)* u_inv_sq, FR(t)* s) {
    if (lg_n >= 32) return 3;                                                             /* :269-273 */
    if (n != ((size_t)1 << lg_n)) return 3;                                               /* :274-276 */
    C(t_ipp_domain_sep)(tr, n);                                                           /* :278 */
    FR(t) prod_inv; memcpy(prod_inv.l, FR(P).one, sizeof prod_inv.l);
    for (size_t j = 0; j < lg_n; j++) {                                                   /* :283-288 */
        C(t_commit_point)(tr, "L", &L[j]);
        C(t_commit_point)(tr, "R", &R[j]);
        FR(t) u, ui;
        C(t_challenge_scalar)(tr, "u", &u);
        FR(inv)(&ui, &u);                                                                 /* batch_invert :295 (same values) */
        FR(mul)(&prod_inv, &prod_inv, &ui);
        FR(sqr)(&u_sq[j], &u); FR(sqr)(&u_inv_sq[j], &ui);                                /* :296-299 */
    }
    s[0] = prod_inv;                                                                      /* :304 */
    for (size_t i = 1; i < n; i++) {                                                      /* :305-312 */
        int lg_i = 63 - __builtin_clzll((unsigned long long)i);
        size_t k = (size_t)1 << lg_i;
        FR(mul)(&s[i], &s[i - k], &u_sq[(lg_n - 1) - lg_i]);
    }
    return 0;
}

static int C(ipp_verify)(orc_transcript* tr, size_t n, const FR(t)* Gf, const FR(t)* Hf, const C(aff_t)* P, const C(aff_t)* Q,
                         const C(aff_t)* G, const C(aff_t)* H, const FR(t)* a, const FR(t)* b,
                         const C(aff_t)* L, const C(aff_t)* R, size_t lg_n) {
    if (lg_n >= 32 || n != ((size_t)1 << lg_n)) return 3;
    FR(t)* s = (FR(t)*)malloc(n * sizeof *s);
    FR(t)* u_sq = (FR(t)*)malloc((lg_n + 1) * sizeof *u_sq);
    FR(t)* u_inv_sq = (FR(t)*)malloc((lg_n + 1) * sizeof *u_inv_sq);
    int rc = C(ipp_verification_scalars)(tr, L, R, lg_n, n, u_sq, u_inv_sq, s);           /* :218 */
    if (rc) { free(s); free(u_sq); free(u_inv_sq); return rc; }
    size_t m = 1 + 2 * n + 2 * lg_n;
    FR(t)* sc = (FR(t)*)malloc(m * sizeof *sc);
    C(aff_t)* pt = (C(aff_t)*)malloc(m * sizeof *pt);
    FR(mul)(&sc[0], a, b); pt[0] = *Q;                                                    /* :237, :245 */
    for (size_t i = 0; i < n; i++) {
        FR(t) t;
        FR(mul)(&t, a, &s[i]); FR(mul)(&sc[1 + i], &t, &Gf[i]); pt[1 + i] = G[i];         /* :220-224 */
        FR(mul)(&t, b, &s[n - 1 - i]); FR(mul)(&sc[1 + n + i], &t, &Hf[i]); pt[1 + n + i] = H[i];  /* :226-232 */
    }
    for (size_t j = 0; j < lg_n; j++) {
        FR(neg)(&sc[1 + 2 * n + j], &u_sq[j]); pt[1 + 2 * n + j] = L[j];                  /* :234, :248 */
        FR(neg)(&sc[1 + 2 * n + lg_n + j], &u_inv_sq[j]); pt[1 + 2 * n + lg_n + j] = R[j]; /* :235, :249 */
    }
    C(aff_t) expect;
    C(msm_fr)(&expect, pt, sc, m);                                                        /* :251-253 */
    int ok = (expect.inf && P->inf) || (!expect.inf && !P->inf && FP(eq)(&expect.x, &P->x) && FP(eq)(&expect.y, &P->y));
    free(s); free(u_sq); free(u_inv_sq); free(sc); free(pt);
    return ok ? 0 : 3;                                                                    /* :255-259 */
}
