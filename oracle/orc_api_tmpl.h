/* oracle/orc_api_tmpl.h -- TEST INFRASTRUCTURE ONLY.  Byte-level wrappers (C-ABI canonical LE format)
 * around the per-curve template code; one instance per curve, dispatched in oracle.c. */

static void C(load_pts)(C(aff_t)* out, const uint8_t* in, size_t n) { for (size_t i = 0; i < n; i++) C(aff_from_le)(&out[i], in + i * 2 * FP_LE_BYTES); }
static void C(load_frs)(FR(t)* out, const uint8_t* in, size_t n) { for (size_t i = 0; i < n; i++) FR(from_le)(&out[i], in + i * FR_LE_BYTES, FR_LE_BYTES); }
static void C(load_raw)(uint64_t* out, const uint8_t* in, size_t n) {
    memset(out, 0, n * FR_NL * 8);
    for (size_t i = 0; i < n; i++) for (int j = 0; j < FR_LE_BYTES; j++) out[i * FR_NL + j / 8] |= (uint64_t)in[i * FR_LE_BYTES + j] << (8 * (j % 8));
}

static int C(api_field_op)(int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    if (which == 0) {
        FP(t) x, y, z; FP(from_le)(&x, a, FP_LE_BYTES); FP(from_le)(&y, b, FP_LE_BYTES);
        switch (op) { case 0: FP(add)(&z, &x, &y); break; case 1: FP(sub)(&z, &x, &y); break; case 2: FP(mul)(&z, &x, &y); break; case 3: FP(inv)(&z, &x); break; default: return 2; }
        FP(to_le)(out, &z, FP_LE_BYTES);
    } else {
        FR(t) x, y, z; FR(from_le)(&x, a, FR_LE_BYTES); FR(from_le)(&y, b, FR_LE_BYTES);
        switch (op) { case 0: FR(add)(&z, &x, &y); break; case 1: FR(sub)(&z, &x, &y); break; case 2: FR(mul)(&z, &x, &y); break; case 3: FR(inv)(&z, &x); break; default: return 2; }
        FR(to_le)(out, &z, FR_LE_BYTES);
    }
    return 0;
}

static int C(api_on_curve)(const uint8_t* p) { C(aff_t) a; C(aff_from_le)(&a, p); return C(aff_on_curve)(&a); }

static int C(api_g1_add)(const uint8_t* p, const uint8_t* q, uint8_t* out) {
    C(aff_t) a, b, r; C(aff_from_le)(&a, p); C(aff_from_le)(&b, q);
    C(jac_t) j; C(jac_from_aff)(&j, &a); C(jac_add_aff)(&j, &j, &b); C(jac_to_aff)(&r, &j); C(aff_to_le)(out, &r);
    return 0;
}

static int C(api_g1_mul)(const uint8_t* k, const uint8_t* p, uint8_t* out) {
    C(aff_t) a, r; C(aff_from_le)(&a, p);
    uint64_t raw[FR_NL]; C(load_raw)(raw, k, 1);
    C(jac_t) j; C(jac_mul_raw)(&j, raw, &a); C(jac_to_aff)(&r, &j); C(aff_to_le)(out, &r);
    return 0;
}

static int C(api_binary_scalar_mul)(const uint8_t* p, const uint8_t* h, const uint8_t* r1, const uint8_t* r2, uint8_t* out) {
    C(aff_t) a, b, r; C(aff_from_le)(&a, p); C(aff_from_le)(&b, h);
    FR(t) k1, k2; FR(from_le)(&k1, r1, FR_LE_BYTES); FR(from_le)(&k2, r2, FR_LE_BYTES);
    C(binary_scalar_mul)(&r, &a, &b, &k1, &k2); C(aff_to_le)(out, &r);
    return 0;
}

/* out[i] = k[i] * G (i < n), batch-normalised; threaded by contiguous slices */
typedef struct { const uint64_t* ks; C(aff_t)* out; size_t lo, hi; } C(fb_job_t);
static C(aff_t)* C(fb_table);   /* [64 windows of 4 bits][15] affine multiples of G, built once */
static void C(fb_build)(void) {
    if (C(fb_table)) return;
    int nw = (FR(P).bits + 3) / 4;
    C(jac_t)* tj = (C(jac_t)*)malloc((size_t)nw * 15 * sizeof(C(jac_t)));
    C(jac_t) base; C(jac_from_aff)(&base, &C(GEN));
    for (int w = 0; w < nw; w++) {
        tj[w * 15] = base;
        for (int d = 1; d < 15; d++) C(jac_add)(&tj[w * 15 + d], &tj[w * 15 + d - 1], &base);
        for (int i = 0; i < 4; i++) C(jac_dbl)(&base, &base);
    }
    C(aff_t)* ta = (C(aff_t)*)malloc((size_t)nw * 15 * sizeof(C(aff_t)));
    C(jac_batch_to_aff)(ta, tj, (size_t)nw * 15);
    free(tj);
    C(fb_table) = ta;
}
static void* C(fb_worker)(void* arg) {
    C(fb_job_t)* j = (C(fb_job_t)*)arg;
    size_t n = j->hi - j->lo;
    int nw = (FR(P).bits + 3) / 4;
    C(jac_t)* acc = (C(jac_t)*)malloc((n ? n : 1) * sizeof(C(jac_t)));
    for (size_t i = 0; i < n; i++) {
        const uint64_t* k = j->ks + (j->lo + i) * FR_NL;
        C(jac_set_inf)(&acc[i]);
        for (int w = 0; w < nw; w++) {
            unsigned d = (unsigned)(k[(4 * w) / 64] >> ((4 * w) % 64)) & 15;
            if (d) C(jac_add_aff)(&acc[i], &acc[i], &C(fb_table)[w * 15 + d - 1]);
        }
    }
    C(jac_batch_to_aff)(j->out + j->lo, acc, n);
    free(acc);
    return 0;
}
static int C(api_fixed_base_batch)(const uint8_t* ks, size_t n, int nthreads, uint8_t* out) {
    C(fb_build)();
    uint64_t* raw = (uint64_t*)malloc((n ? n : 1) * FR_NL * 8); C(load_raw)(raw, ks, n);
    C(aff_t)* res = (C(aff_t)*)malloc((n ? n : 1) * sizeof(C(aff_t)));
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t th[64]; C(fb_job_t) jobs[64];
    if (nthreads > 64) nthreads = 64;
    for (int t = 0; t < nthreads; t++) {
        jobs[t].ks = raw; jobs[t].out = res; jobs[t].lo = n * (size_t)t / nthreads; jobs[t].hi = n * (size_t)(t + 1) / nthreads;
        if (nthreads == 1) C(fb_worker)(&jobs[t]); else pthread_create(&th[t], 0, C(fb_worker), &jobs[t]);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    for (size_t i = 0; i < n; i++) C(aff_to_le)(out + i * 2 * FP_LE_BYTES, &res[i]);
    free(raw); free(res);
    return 0;
}

/* algo: 0 naive, 1 Strauss wNAF-5 (reference-like, single thread), 2 Pippenger (nthreads) */
static int C(api_msm)(int algo, const uint8_t* pts, const uint8_t* ks, size_t n, int nthreads, uint8_t* out) {
    C(aff_t)* P = (C(aff_t)*)malloc((n ? n : 1) * sizeof(C(aff_t))); C(load_pts)(P, pts, n);
    uint64_t* raw = (uint64_t*)malloc((n ? n : 1) * FR_NL * 8); C(load_raw)(raw, ks, n);
    C(jac_t) j; C(aff_t) r;
    switch (algo) {
        case 0: C(msm_naive)(&j, P, raw, n); break;
        case 1: C(msm_strauss)(&j, P, raw, n); break;
        case 2: C(msm_pippenger)(&j, P, raw, n, nthreads); break;
        default: free(P); free(raw); return 2;
    }
    C(jac_to_aff)(&r, &j); C(aff_to_le)(out, &r);
    free(P); free(raw);
    return 0;
}

/* Timed variant for bench.py's cpu_baseline leg: conversion from bytes is excluded from the timing,
 * matching the GPU leg, whose inputs are already resident in Montgomery form. */
static int C(api_msm_timed)(int algo, const uint8_t* pts, const uint8_t* ks, size_t n, int nthreads, uint8_t* out, double* seconds) {
    C(aff_t)* P = (C(aff_t)*)malloc((n ? n : 1) * sizeof(C(aff_t))); C(load_pts)(P, pts, n);
    uint64_t* raw = (uint64_t*)malloc((n ? n : 1) * FR_NL * 8); C(load_raw)(raw, ks, n);
    C(jac_t) j; C(aff_t) r;
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    switch (algo) {
        case 0: C(msm_naive)(&j, P, raw, n); break;
        case 1: C(msm_strauss)(&j, P, raw, n); break;
        case 2: C(msm_pippenger)(&j, P, raw, n, nthreads); break;
        default: free(P); free(raw); return 2;
    }
    C(jac_to_aff)(&r, &j);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    C(aff_to_le)(out, &r);
    free(P); free(raw);
    return 0;
}

static int C(api_fr_inner)(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
    FR(t)* x = (FR(t)*)malloc((n ? n : 1) * sizeof(FR(t))); FR(t)* y = (FR(t)*)malloc((n ? n : 1) * sizeof(FR(t)));
    C(load_frs)(x, a, n); C(load_frs)(y, b, n);
    FR(t) r; C(fr_inner)(&r, x, y, n); FR(to_le)(out, &r, FR_LE_BYTES);
    free(x); free(y);
    return 0;
}

/* sum_i a_i * b_i mod r with a_i, b_i canonical LE; used for the full-size linearity check
 * MSM(s, k.G) == (sum s_i k_i) G */
static int C(api_ipp_create)(orc_transcript* tr, const uint8_t* Q, const uint8_t* Gf, const uint8_t* Hf, const uint8_t* G, const uint8_t* H,
                             const uint8_t* a, const uint8_t* b, size_t n, uint8_t* L_out, uint8_t* R_out, uint8_t* a_out, uint8_t* b_out) {
    if (n == 0 || (n & (n - 1))) return 2;                                                /* assert!(n.is_power_of_two()) :48 */
    C(aff_t) q; C(aff_from_le)(&q, Q);
    C(aff_t)* g = (C(aff_t)*)malloc(n * sizeof *g); C(aff_t)* h = (C(aff_t)*)malloc(n * sizeof *h);
    FR(t)* gf = (FR(t)*)malloc(n * sizeof *gf); FR(t)* hf = (FR(t)*)malloc(n * sizeof *hf);
    FR(t)* av = (FR(t)*)malloc(n * sizeof *av); FR(t)* bv = (FR(t)*)malloc(n * sizeof *bv);
    C(load_pts)(g, G, n); C(load_pts)(h, H, n); C(load_frs)(gf, Gf, n); C(load_frs)(hf, Hf, n); C(load_frs)(av, a, n); C(load_frs)(bv, b, n);
    C(aff_t) L[64], R[64]; FR(t) a0, b0;
    int lg = C(ipp_create)(tr, &q, gf, hf, g, h, av, bv, n, L, R, &a0, &b0);
    for (int i = 0; i < lg; i++) { C(aff_to_le)(L_out + i * 2 * FP_LE_BYTES, &L[i]); C(aff_to_le)(R_out + i * 2 * FP_LE_BYTES, &R[i]); }
    FR(to_le)(a_out, &a0, FR_LE_BYTES); FR(to_le)(b_out, &b0, FR_LE_BYTES);
    free(g); free(h); free(gf); free(hf); free(av); free(bv);
    return 0;
}

static int C(api_ipp_verify)(orc_transcript* tr, size_t n, const uint8_t* Gf, const uint8_t* Hf, const uint8_t* P, const uint8_t* Q,
                             const uint8_t* G, const uint8_t* H, const uint8_t* a, const uint8_t* b, const uint8_t* L, const uint8_t* R, size_t lg_n) {
    if (lg_n >= 32 || n != ((size_t)1 << lg_n)) return 3;
    C(aff_t) p, q; C(aff_from_le)(&p, P); C(aff_from_le)(&q, Q);
    C(aff_t)* g = (C(aff_t)*)malloc(n * sizeof *g); C(aff_t)* h = (C(aff_t)*)malloc(n * sizeof *h);
    FR(t)* gf = (FR(t)*)malloc(n * sizeof *gf); FR(t)* hf = (FR(t)*)malloc(n * sizeof *hf);
    C(load_pts)(g, G, n); C(load_pts)(h, H, n); C(load_frs)(gf, Gf, n); C(load_frs)(hf, Hf, n);
    C(aff_t) Lp[64], Rp[64]; C(load_pts)(Lp, L, lg_n); C(load_pts)(Rp, R, lg_n);
    FR(t) av, bv; FR(from_le)(&av, a, FR_LE_BYTES); FR(from_le)(&bv, b, FR_LE_BYTES);
    int rc = C(ipp_verify)(tr, n, gf, hf, &p, &q, g, h, &av, &bv, Lp, Rp, lg_n);
    free(g); free(h); free(gf); free(hf);
    return rc;
}

static int C(api_verification_scalars)(orc_transcript* tr, const uint8_t* L, const uint8_t* R, size_t lg_n, size_t n,
                                       uint8_t* u_sq, uint8_t* u_inv_sq, uint8_t* s) {
    if (lg_n >= 32 || n != ((size_t)1 << lg_n)) return 3;
    C(aff_t) Lp[64], Rp[64]; C(load_pts)(Lp, L, lg_n); C(load_pts)(Rp, R, lg_n);
    FR(t)* sv = (FR(t)*)malloc(n * sizeof *sv); FR(t) us[64], uis[64];
    int rc = C(ipp_verification_scalars)(tr, Lp, Rp, lg_n, n, us, uis, sv);
    if (!rc) {
        for (size_t j = 0; j < lg_n; j++) { FR(to_le)(u_sq + j * FR_LE_BYTES, &us[j], FR_LE_BYTES); FR(to_le)(u_inv_sq + j * FR_LE_BYTES, &uis[j], FR_LE_BYTES); }
        for (size_t i = 0; i < n; i++) FR(to_le)(s + i * FR_LE_BYTES, &sv[i], FR_LE_BYTES);
    }
    free(sv);
    return rc;
}

static int C(api_challenge_scalar)(orc_transcript* tr, const char* label, uint8_t* out) {
    FR(t) u; C(t_challenge_scalar)(tr, label, &u); FR(to_le)(out, &u, FR_LE_BYTES); return 0;
}
static int C(api_commit_point)(orc_transcript* tr, const char* label, const uint8_t* p) {
    C(aff_t) a; C(aff_from_le)(&a, p); C(t_commit_point)(tr, label, &a); return 0;
}

static int C(api_commit_scalar)(orc_transcript* tr, const char* label, const uint8_t* x) {   /* transcript.rs:47-49 */
    FR(t) v; FR(from_le)(&v, x, FR_LE_BYTES);
    uint8_t be[MODBYTES]; C(fr_to_be)(be, &v);
    orc_transcript_append(tr, (const uint8_t*)label, strlen(label), be, MODBYTES);
    return 0;
}

/* ---- R1CS (orc_r1cs_tmpl.h).  Proof bytes: 11 points | t_x t_x_blinding e_blinding | L[lg] R[lg] | a b (include/bpmsm.h layout). */
static int C(r1cs_load_terms)(C(r1cs_terms_t)* cs, size_t n_terms, const uint32_t* con, const uint8_t* kind, const uint32_t* idx, const uint8_t* coeff,
                              size_t n_constraints, size_t n, size_t m, FR(t)** owned) {
    for (size_t k = 0; k < n_terms; k++) {
        if (con[k] >= n_constraints || kind[k] > 4) return 2;
        if (kind[k] <= 2 && idx[k] >= n) return 2;
        if (kind[k] == 3 && idx[k] >= m) return 2;
    }
    FR(t)* cf = (FR(t)*)malloc((n_terms ? n_terms : 1) * sizeof *cf);
    C(load_frs)(cf, coeff, n_terms);
    *owned = cf;
    *cs = (C(r1cs_terms_t)){n_terms, n_constraints, con, kind, idx, cf};
    return 0;
}
static size_t C(r1cs_lg)(size_t n) { size_t lg = 0; while (((size_t)1 << lg) < n) lg++; return lg; }
static size_t C(r1cs_proof_bytes)(size_t n) { return 11 * 2 * FP_LE_BYTES + 3 * FR_LE_BYTES + 2 * C(r1cs_lg)(n) * 2 * FP_LE_BYTES + 2 * FR_LE_BYTES; }

static int C(api_r1cs_prove)(orc_transcript* tr, size_t n_terms, const uint32_t* con, const uint8_t* kind, const uint32_t* idx, const uint8_t* coeff,
                             size_t n_constraints, size_t n, size_t m, const uint8_t* g, const uint8_t* h, const uint8_t* G, const uint8_t* H, size_t ngens,
                             const uint8_t* aL, const uint8_t* aR, const uint8_t* aO, const uint8_t* vb, const uint8_t* sL, const uint8_t* sR,
                             const uint8_t* blindings, uint8_t* out) {
    C(r1cs_terms_t) cs; FR(t)* cf;
    int rc = C(r1cs_load_terms)(&cs, n_terms, con, kind, idx, coeff, n_constraints, n, m, &cf);
    if (rc) return rc;
    C(aff_t) gp, hp; C(aff_from_le)(&gp, g); C(aff_from_le)(&hp, h);
    C(aff_t)* Gp = (C(aff_t)*)malloc(2 * (ngens ? ngens : 1) * sizeof *Gp); C(aff_t)* Hp = Gp + ngens;
    C(load_pts)(Gp, G, ngens); C(load_pts)(Hp, H, ngens);
    FR(t)* wit = (FR(t)*)malloc((5 * n + m + 8 + 1) * sizeof *wit);
    C(load_frs)(wit, aL, n); C(load_frs)(wit + n, aR, n); C(load_frs)(wit + 2 * n, aO, n); C(load_frs)(wit + 3 * n, sL, n); C(load_frs)(wit + 4 * n, sR, n);
    C(load_frs)(wit + 5 * n, vb, m); C(load_frs)(wit + 5 * n + m, blindings, 8);
    C(r1cs_proof_t)* pf = (C(r1cs_proof_t)*)malloc(sizeof *pf);
    rc = C(r1cs_prove)(tr, &cs, n, m, &gp, &hp, Gp, Hp, ngens, wit, wit + n, wit + 2 * n, wit + 5 * n, wit + 3 * n, wit + 4 * n, wit + 5 * n + m, pf);
    if (!rc) {
        uint8_t* o = out;
        for (int k = 0; k < 11; k++, o += 2 * FP_LE_BYTES) C(aff_to_le)(o, &pf->pts[k]);
        FR(to_le)(o, &pf->t_x, FR_LE_BYTES); o += FR_LE_BYTES; FR(to_le)(o, &pf->t_x_blinding, FR_LE_BYTES); o += FR_LE_BYTES;
        FR(to_le)(o, &pf->e_blinding, FR_LE_BYTES); o += FR_LE_BYTES;
        for (int k = 0; k < pf->lg; k++, o += 2 * FP_LE_BYTES) C(aff_to_le)(o, &pf->L[k]);
        for (int k = 0; k < pf->lg; k++, o += 2 * FP_LE_BYTES) C(aff_to_le)(o, &pf->R[k]);
        FR(to_le)(o, &pf->a, FR_LE_BYTES); o += FR_LE_BYTES; FR(to_le)(o, &pf->b, FR_LE_BYTES);
    }
    free(cf); free(Gp); free(wit); free(pf);
    return rc;
}

static int C(api_r1cs_verify)(orc_transcript* tr, size_t n_terms, const uint32_t* con, const uint8_t* kind, const uint32_t* idx, const uint8_t* coeff,
                              size_t n_constraints, size_t n, size_t m, const uint8_t* V, const uint8_t* proof, size_t proof_len, const uint8_t* g,
                              const uint8_t* h, const uint8_t* G, const uint8_t* H, size_t ngens, const uint8_t* rnd) {
    if (proof_len != C(r1cs_proof_bytes)(n)) return 3;
    C(r1cs_terms_t) cs; FR(t)* cf;
    int rc = C(r1cs_load_terms)(&cs, n_terms, con, kind, idx, coeff, n_constraints, n, m, &cf);
    if (rc) return rc;
    C(aff_t) gp, hp; C(aff_from_le)(&gp, g); C(aff_from_le)(&hp, h);
    C(aff_t)* Gp = (C(aff_t)*)malloc((2 * (ngens ? ngens : 1) + m + 1) * sizeof *Gp); C(aff_t)* Hp = Gp + ngens; C(aff_t)* Vp = Hp + ngens;
    C(load_pts)(Gp, G, ngens); C(load_pts)(Hp, H, ngens); C(load_pts)(Vp, V, m);
    C(r1cs_proof_t)* pf = (C(r1cs_proof_t)*)malloc(sizeof *pf);
    const uint8_t* o = proof;
    pf->lg = (int)C(r1cs_lg)(n);
    for (int k = 0; k < 11; k++, o += 2 * FP_LE_BYTES) C(aff_from_le)(&pf->pts[k], o);
    FR(from_le)(&pf->t_x, o, FR_LE_BYTES); o += FR_LE_BYTES; FR(from_le)(&pf->t_x_blinding, o, FR_LE_BYTES); o += FR_LE_BYTES;
    FR(from_le)(&pf->e_blinding, o, FR_LE_BYTES); o += FR_LE_BYTES;
    for (int k = 0; k < pf->lg; k++, o += 2 * FP_LE_BYTES) C(aff_from_le)(&pf->L[k], o);
    for (int k = 0; k < pf->lg; k++, o += 2 * FP_LE_BYTES) C(aff_from_le)(&pf->R[k], o);
    FR(from_le)(&pf->a, o, FR_LE_BYTES); o += FR_LE_BYTES; FR(from_le)(&pf->b, o, FR_LE_BYTES);
    FR(t) rr; FR(from_le)(&rr, rnd, FR_LE_BYTES);
    rc = C(r1cs_verify)(tr, &cs, n, m, Vp, pf, &gp, &hp, Gp, Hp, ngens, &rr);
    free(cf); free(Gp); free(pf);
    return rc;
}

static int C(api_r1cs_flatten)(size_t n_terms, const uint32_t* con, const uint8_t* kind, const uint32_t* idx, const uint8_t* coeff, size_t n_constraints,
                               size_t n, size_t m, const uint8_t* z, uint8_t* wL, uint8_t* wR, uint8_t* wO, uint8_t* wV, uint8_t* wc) {
    C(r1cs_terms_t) cs; FR(t)* cf;
    int rc = C(r1cs_load_terms)(&cs, n_terms, con, kind, idx, coeff, n_constraints, n, m, &cf);
    if (rc) return rc;
    FR(t)* w = (FR(t)*)malloc((3 * n + m + 1) * sizeof *w);
    FR(t) zz, c; FR(from_le)(&zz, z, FR_LE_BYTES);
    C(r1cs_flatten)(&cs, &zz, n, m, w, w + n, w + 2 * n, w + 3 * n, &c);
    for (size_t i = 0; i < n; i++) { FR(to_le)(wL + i * FR_LE_BYTES, &w[i], FR_LE_BYTES); FR(to_le)(wR + i * FR_LE_BYTES, &w[n + i], FR_LE_BYTES); FR(to_le)(wO + i * FR_LE_BYTES, &w[2 * n + i], FR_LE_BYTES); }
    for (size_t j = 0; j < m; j++) FR(to_le)(wV + j * FR_LE_BYTES, &w[3 * n + j], FR_LE_BYTES);
    FR(to_le)(wc, &c, FR_LE_BYTES);
    free(cf); free(w);
    return 0;
}

/* ---- hash to G1 ---------------------------------------------------------------------------------------------
 * amcl_wrapper `G1::from_msg_hash(msg)` = `GroupG1::mapit(&hash_msg(msg))`, the call behind get_generators
 * (src/utils/mod.rs:16-23).  [UNVERIFIED-RECALL] of amcl `ECP::mapit` / `new_bigint(x, 0)` / `cfp` (crate not vendored):
 * x = BE(SHAKE256(msg)[0..MODBYTES]) mod p; try-and-increment on x; y = the even square root; multiply by the cofactor. */
static const uint64_t C(COF)[FR_NL] = COFACTOR_WORDS;
static void C(from_msg_hash)(C(aff_t)* out, const uint8_t* msg, size_t len) {
    uint8_t h[MODBYTES];
    orc_shake256(msg, len, h, MODBYTES);
    FP(t) x, one; FP(from_be_reduce)(&x, h, MODBYTES);
    memcpy(one.l, FP(P).one, sizeof one.l);
    uint64_t e[FP_NL];                                   /* (p + 1) / 4; p = 3 mod 4 for both curves */
    memcpy(e, FP(P).mod, sizeof e);
    for (int i = 0; i < FP_NL; i++) { if (++e[i]) break; }
    for (int i = 0; i < FP_NL; i++) e[i] = (e[i] >> 2) | (i + 1 < FP_NL ? e[i + 1] << 62 : 0);
    for (;;) {
        FP(t) rhs, s, s2;
        FP(sqr)(&rhs, &x); FP(mul)(&rhs, &rhs, &x); FP(add)(&rhs, &rhs, &C(B));
        FP(pow)(&s, &rhs, e); FP(sqr)(&s2, &s);
        int ok = !FP(is_zero)(&rhs) && FP(eq)(&s2, &rhs);
        C(aff_t) cand; cand.inf = 0; cand.x = x;
        FP(add)(&x, &x, &one);
        if (!ok) continue;
        uint64_t raw[FP_NL]; FP(to_raw)(raw, &s);
        if (raw[0] & 1) FP(neg)(&s, &s);
        cand.y = s;
        C(jac_t) j; C(jac_mul_raw)(&j, C(COF), &cand);
        if (C(jac_is_inf)(&j)) continue;
        C(jac_to_aff)(out, &j);
        return;
    }
}
static int C(api_from_msg_hash)(const uint8_t* msg, size_t len, uint8_t* out) {
    C(aff_t) r; C(from_msg_hash)(&r, msg, len); C(aff_to_le)(out, &r); return 0;
}
/* out[i] = from_msg_hash(prefix || decimal(first + i)), i < n  (get_generators uses first = 1) */
typedef struct { const uint8_t* prefix; size_t plen; uint64_t first; size_t lo, hi; uint8_t* out; } C(gg_job_t);
static void* C(gg_worker)(void* arg) {
    C(gg_job_t)* jb = (C(gg_job_t)*)arg;
    uint8_t* buf = (uint8_t*)malloc(jb->plen + 24);
    memcpy(buf, jb->prefix, jb->plen);
    for (size_t i = jb->lo; i < jb->hi; i++) {
        int nd = snprintf((char*)buf + jb->plen, 24, "%llu", (unsigned long long)(jb->first + i));
        C(aff_t) r; C(from_msg_hash)(&r, buf, jb->plen + (size_t)nd);
        C(aff_to_le)(jb->out + i * 2 * FP_LE_BYTES, &r);
    }
    free(buf);
    return NULL;
}
static int C(api_get_generators)(const uint8_t* prefix, size_t plen, uint64_t first, size_t n, int nthreads, uint8_t* out) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    C(gg_job_t)* jobs = (C(gg_job_t)*)malloc(sizeof(C(gg_job_t)) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (C(gg_job_t)){prefix, plen, first, n * t / nthreads, n * (t + 1) / nthreads, out};
        pthread_create(&th[t], NULL, C(gg_worker), &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}

static void C(api_init)(const uint64_t* p, const uint64_t* r, uint64_t b, const uint64_t* gx, const uint64_t* gy) {
    FP(init)(p); FR(init)(r);
    FP(from_u64)(&C(B), b);
    uint8_t buf[2 * FP_LE_BYTES]; memset(buf, 0, sizeof buf);
    for (int i = 0; i < FP_LE_BYTES; i++) { buf[i] = (uint8_t)(gx[i / 8] >> (8 * (i % 8))); buf[FP_LE_BYTES + i] = (uint8_t)(gy[i / 8] >> (8 * (i % 8))); }
    C(aff_from_le)(&C(GEN), buf);
}
