// bp_capi_r1cs.hip -- host orchestration of the two callers of the hot path in the R1CS layer, written against the PUBLIC C ABI
// of this library (include/bpmsm.h) plus host field arithmetic: `Prover::prove` (/root/reference src/r1cs/prover.rs:323-560)
// and `Verifier::verify` (src/r1cs/verifier.rs:265-452) for single-phase constraint systems (n2 = 0: A_I2 = A_O2 = S2 = O).
// The host keeps the transcript and a handful of scalars, exactly the reference's split; everything with a vector or a group
// element in it is a bp_* call.  tests/r1cs_twin.py is the same orchestration in the Python mirror; the tests
// require both to produce the same proof bytes.
#include <chrono>
#include <string>
#include <vector>

#include "bp_internal.hpp"

using namespace bp;

namespace {

template <class F> Fe<F> fr_in(const uint8_t* le32) {
    uint32_t w[8];
    memcpy(w, le32, 32);
    return fe_to_mont<F>(fe_unpack_words<F>(w));
}
template <class F> void fr_out(const Fe<F>& x_mont, uint8_t* le32) {
    uint32_t w[8];
    fe_pack_words<F>(w, fe_from_mont<F>(x_mont));
    memcpy(le32, w, 32);
}

// owns every temporary handle of one call
struct Temps {
    std::vector<bp_frvec*> fr;
    std::vector<bp_g1vec*> g1;
    bp_frvec* keep(bp_frvec* v) { fr.push_back(v); return v; }
    bp_g1vec* keep(bp_g1vec* v) { g1.push_back(v); return v; }
    ~Temps() {
        for (auto* v : fr) bp_frvec_free(v);
        for (auto* v : g1) bp_g1vec_free(v);
    }
};

#define RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
// uploads of PROOF data in the verifier: a point off the curve / a non-canonical encoding is a failed verification
// (the reference fails while deserialising), not a caller error
#define RCV(expr) do { int rc_ = (expr); if (rc_) return rc_ == BP_ERR_ARG ? BP_ERR_VERIFY : rc_; } while (0)

inline size_t padded_len(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
inline size_t lg_of(size_t p) { size_t l = 0; while (((size_t)1 << l) < p) l++; return l; }

// scalars held on the host -> one resident vector
int upload_scalars(bp_ctx* ctx, Temps& T, const std::vector<uint8_t>& le, bp_frvec** out) {
    RC(bp_internal_frvec_upload_trusted(ctx, le.data(), le.size() / 32, out));      // canonical by construction (fr_out of reduced values): no check, no wait
    T.keep(*out);
    return BP_OK;
}

// sum_i scalars[i] * points[i] for a few host-side points and scalars (T_k, Q, ...)
int small_msm(bp_ctx* ctx, Temps& T, const std::vector<uint8_t>& pts_le, const std::vector<uint8_t>& sc_le, size_t pb, uint8_t* out_le) {
    bp_g1vec* p = nullptr;
    bp_frvec* s = nullptr;
    RC(bp_g1vec_upload(ctx, pts_le.data(), pts_le.size() / pb, BP_FMT_LE, &p));
    T.keep(p);
    RC(upload_scalars(ctx, T, sc_le, &s));
    return bp_msm_g1(ctx, p, s, out_le);
}

template <class C>
struct R1cs {
    using F = typename C::Fr;
    static constexpr size_t pb = 2 * 4 * C::Fp::NW;
    static constexpr size_t row = sizeof(AffPacked<C>);

    // [G[off..off+n) | H[off..off+n) | h] as one resident vector (device-to-device; the generators stay where they are)
    static int cat_GHh(bp_ctx* ctx, Temps& T, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* h_le, size_t off, size_t n, bp_g1vec** out) {
        bp_g1vec *v = nullptr, *hv = nullptr;
        RC(bp_g1vec_alloc(ctx, 2 * n + 1, &v));
        T.keep(v);
        RC(bp_g1vec_upload(ctx, h_le, 1, BP_FMT_LE, &hv));
        T.keep(hv);
        hipStream_t s = ctx->stream;
        if (n) {
            HIPCHK(hipMemcpyAsync(v->d, (const uint8_t*)G->d + off * row, n * row, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync((uint8_t*)v->d + n * row, (const uint8_t*)H->d + off * row, n * row, hipMemcpyDeviceToDevice, s));
        }
        HIPCHK(hipMemcpyAsync((uint8_t*)v->d + 2 * n * row, hv->d, row, hipMemcpyDeviceToDevice, s));
        if (2 * n + 1 > kSmallMsmMax) RC(bp_internal_table_concat(ctx, G, off, H, off, n, h_le, &v->table));   // G, H precomputed: merged-window MSMs (freed with v)
        *out = v;
        return BP_OK;
    }

    // scalars [a[off..off+n) | b[off..off+n) | c] of  <a, G> + <b, H> + c h  (commit_to_field_element_vectors, prover.rs:346-361,
    // 404-427); b == nullptr: <a, G> + c h
    static int commit_scalars(bp_ctx* ctx, Temps& T, size_t off, size_t n, const bp_frvec* a, const bp_frvec* b, const Fe<F>& c, bp_frvec** out) {
        bp_frvec *sc = nullptr, *one = nullptr;
        RC(bp_frvec_alloc(ctx, 2 * n + 1, &sc));
        T.keep(sc);
        if (n) RC(bp_frvec_copy(ctx, sc, 0, a, off, n));
        if (n && b) RC(bp_frvec_copy(ctx, sc, n, b, off, n));
        std::vector<uint8_t> cle(32);
        fr_out<F>(c, cle.data());
        RC(upload_scalars(ctx, T, cle, &one));
        RC(bp_frvec_copy(ctx, sc, 2 * n, one, 0, 1));
        *out = sc;
        return BP_OK;
    }
    // A_I, A_O, S of one phase: the witness slice [off, off + n) against G[off..), H[off..) and h, three MSMs in flight together
    static int phase_commitments(bp_ctx* ctx, Temps& T, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* h_le, size_t off, size_t n, const bp_frvec* aL,
                                 const bp_frvec* aR, const bp_frvec* aO, const bp_frvec* sL, const bp_frvec* sR, const Fe<F>& i_bl, const Fe<F>& o_bl,
                                 const Fe<F>& s_bl, uint8_t* P3) {
        bp_g1vec* GHh = nullptr;
        RC(cat_GHh(ctx, T, G, H, h_le, off, n, &GHh));
        bp_frvec* sc3[3] = {nullptr, nullptr, nullptr};
        RC(commit_scalars(ctx, T, off, n, aL, aR, i_bl, &sc3[0]));
        RC(commit_scalars(ctx, T, off, n, aO, nullptr, o_bl, &sc3[1]));
        RC(commit_scalars(ctx, T, off, n, sL, sR, s_bl, &sc3[2]));
        // A_I and S in ONE pipeline pass over [G | H | h] (two scalar sets, bp_msm_g1_pair), A_O -- all-zero a_O in a gadget circuit, one live
        // term -- on a sibling context beside it.  (Three separate MSMs in flight took 1.9 ms at 2^16 gates; the uniform S alone is 0.9.)
        bp_ctx* side = bp_internal_helper(ctx, 0);
        if (!side) {
            uint8_t* out3[3] = {P3, P3 + pb, P3 + 2 * pb};
            return commit_vectors_concurrent(ctx, GHh, sc3, out3, 3);
        }
        RC(bp_internal_fork(ctx, side));
        RC(bp_msm_g1_begin(side, GHh, sc3[1]));
        const int rc_pair = bp_msm_g1_pair(ctx, GHh, sc3[0], sc3[2], P3, P3 + 2 * pb);
        const int rc_o = bp_msm_g1_end(side, P3 + pb);
        return rc_pair ? rc_pair : rc_o;
    }
    // k independent MSMs over the same point vector, three in flight at a time: on the context and its two siblings (the
    // latency-bound bucket reduce and host tail of one hide behind the accumulate of the others).  Every MSM that was begun is
    // ended before returning, whatever failed, because the vectors in T are released on return.
    static int commit_vectors_concurrent(bp_ctx* ctx, const bp_g1vec* pts, bp_frvec* const sc[], uint8_t* const out_le[], int k) {
        bp_ctx* ex[3] = {ctx, bp_internal_helper(ctx, 0), bp_internal_helper(ctx, 1)};
        if (!ex[1] || !ex[2]) { for (int i = 0; i < k; i++) RC(bp_msm_g1(ctx, pts, sc[i], out_le[i])); return BP_OK; }
        int rc = BP_OK;
        for (int i0 = 0; i0 < k && rc == BP_OK; i0 += 3) {
            const int kb = k - i0 < 3 ? k - i0 : 3;
            int begun = 0;
            for (int i = 0; i < kb && rc == BP_OK; i++) {
                if (i > 0) rc = bp_internal_fork(ctx, ex[i]);
                if (rc == BP_OK) rc = bp_msm_g1_begin(ex[i], pts, sc[i0 + i]);
                if (rc == BP_OK) begun++;
            }
            // the host tails (~0.13 ms each) on the siblings' helper threads beside this one
            int r_end[3] = {BP_OK, BP_OK, BP_OK};
            bool queued[3] = {false, false, false};
            for (int i = 1; i < begun; i++) queued[i] = ex[i]->worker.submit([&, i]() { r_end[i] = bp_msm_g1_end(ex[i], out_le[i0 + i]); });
            for (int i = 0; i < begun; i++) if (!queued[i]) r_end[i] = bp_msm_g1_end(ex[i], out_le[i0 + i]);
            for (int i = 1; i < begun; i++) if (queued[i]) ex[i]->worker.wait();
            for (int i = 0; i < begun; i++) if (rc == BP_OK) rc = r_end[i];
        }
        return rc;
    }

    static Fe<F> challenge(bp_transcript* t, int curve, const char* label) {
        uint8_t b[32];
        bp_transcript_challenge_scalar(t, curve, label, b);
        return fr_in<F>(b);
    }

    // proof layout: A_I1 A_O1 S1 A_I2 A_O2 S2 T_1 T_3 T_4 T_5 T_6 | t_x t_x_blinding e_blinding | L[lg] R[lg] | a b
    static size_t proof_bytes(size_t n) { return 11 * pb + 96 + 2 * lg_of(padded_len(n)) * pb + 64; }

    // Prover::prove up to and including the first-phase commitments (prover.rs:323-366): "m", A_I1, A_O1, S1 on the transcript
    static int prove_phase1(bp_ctx* ctx, Temps& T, bp_transcript* t, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* h_le, size_t m, size_t n1,
                            const bp_frvec* aL, const bp_frvec* aR, const bp_frvec* aO, const bp_frvec* sL, const bp_frvec* sR, const uint8_t* bl3, uint8_t* P3) {
        const int cv = ctx->curve;
        RC(bp_transcript_append_u64(t, (const uint8_t*)"m", 1, (uint64_t)m));                                  // prover.rs:328
        RC(phase_commitments(ctx, T, G, H, h_le, 0, n1, aL, aR, aO, sL, sR, fr_in<F>(bl3), fr_in<F>(bl3 + 32), fr_in<F>(bl3 + 64), P3));   // :346-361
        RC(bp_transcript_commit_point(t, cv, "A_I1", P3));
        RC(bp_transcript_commit_point(t, cv, "A_O1", P3 + pb));
        RC(bp_transcript_commit_point(t, cv, "S1", P3 + 2 * pb));
        return BP_OK;
    }

    // BP_PROFILE: wall clock of the prover's phases on stderr (diagnostic)
    struct Phase {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        std::string log;
        const char* who = "prove";
        void lap(bp_ctx* ctx, const char* what) {
            if (!bp_profile_on()) return;
            (void)hipStreamSynchronize(ctx->stream);
            auto t1 = std::chrono::steady_clock::now();
            char b[96];
            snprintf(b, sizeof b, "  %s %.0f", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
            log += b;
            t0 = t1;
        }
        ~Phase() { if (bp_profile_on() && !log.empty()) fprintf(stderr, "[bpmsm profile] r1cs %s us:%s\n", who, log.c_str()); }
    };

    // single phase: bl = i1 o1 s1 t1 t3 t4 t5 t6
    static int prove(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                     const uint8_t* h_le, const bp_frvec* aL, const bp_frvec* aR, const bp_frvec* aO, const bp_frvec* v_blinding, const bp_frvec* sL,
                     const bp_frvec* sR, const uint8_t* bl, uint8_t* proof) {
        Temps T;
        const size_t n = aL->n, m = v_blinding ? v_blinding->n : 0;
        memset(proof, 0, proof_bytes(n));
        Phase ph;
        RC(prove_phase1(ctx, T, t, G, H, h_le, m, n, aL, aR, aO, sL, sR, bl, proof));
        ph.lap(ctx, "commitments A_I A_O S");
        RC(bp_transcript_append_message(t, (const uint8_t*)"dom-sep", 7, (const uint8_t*)"r1cs-1phase", 11));  // :304-306
        uint8_t zero3[96] = {0};
        return prove_tail(ctx, T, t, plan, G, H, g_le, h_le, n, aL, aR, aO, v_blinding, sL, sR, bl, zero3, bl + 96, proof);
    }

    // Prover::prove after create_randomized_constraints (prover.rs:371-593).  n1 = first-phase multipliers, the vectors hold all
    // n = n1 + n2; bl1 = i1 o1 s1, bl2 = i2 o2 s2 (ignored when n2 = 0), tbl = t1 t3 t4 t5 t6.  proof[0..3) already holds A_I1 A_O1 S1.
    static int prove_tail(bp_ctx* ctx, Temps& T, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                          const uint8_t* h_le, size_t n1, const bp_frvec* aL, const bp_frvec* aR, const bp_frvec* aO, const bp_frvec* v_blinding,
                          const bp_frvec* sL, const bp_frvec* sR, const uint8_t* bl1, const uint8_t* bl2, const uint8_t* tbl, uint8_t* proof) {
        const int cv = ctx->curve;
        const size_t n = aL->n, n2 = n - n1, m = v_blinding ? v_blinding->n : 0, pn = padded_len(n), lg = lg_of(pn);
        uint8_t* P = proof;                                       // the 11 points
        uint8_t* S = proof + 11 * pb;                             // the 3 scalars
        uint8_t* Lp = S + 96;
        uint8_t* Rp = Lp + lg * pb;
        uint8_t* ab = Rp + lg * pb;
        const Fe<F> i_bl1 = fr_in<F>(bl1), o_bl1 = fr_in<F>(bl1 + 32), s_bl1 = fr_in<F>(bl1 + 64);
        Fe<F> i_bl2 = fe_zero<F>(), o_bl2 = fe_zero<F>(), s_bl2 = fe_zero<F>();                                 // :398-402
        Phase ph;
        Fe<F> tb[7];
        tb[1] = fr_in<F>(tbl); tb[3] = fr_in<F>(tbl + 32); tb[4] = fr_in<F>(tbl + 64); tb[5] = fr_in<F>(tbl + 96); tb[6] = fr_in<F>(tbl + 128);
        if (n2) {                                                                                              // :385-427: A_I2, A_O2, S2 over G[n1..n), H[n1..n)
            i_bl2 = fr_in<F>(bl2); o_bl2 = fr_in<F>(bl2 + 32); s_bl2 = fr_in<F>(bl2 + 64);
            RC(phase_commitments(ctx, T, G, H, h_le, n1, n2, aL, aR, aO, sL, sR, i_bl2, o_bl2, s_bl2, P + 3 * pb));
        }
        RC(bp_transcript_commit_point(t, cv, "A_I2", P + 3 * pb));                                             // identity when n2 = 0, :429-434
        RC(bp_transcript_commit_point(t, cv, "A_O2", P + 4 * pb));
        RC(bp_transcript_commit_point(t, cv, "S2", P + 5 * pb));
        const Fe<F> y = challenge(t, cv, "y"), z = challenge(t, cv, "z");
        uint8_t yle[32], zle[32];
        fr_out<F>(y, yle); fr_out<F>(z, zle);
        bp_frvec* w[4] = {};
        RC(bp_r1cs_flattened_constraints(ctx, plan, zle, w, nullptr));                                         // :438
        for (auto* v : w) T.keep(v);
        ph.lap(ctx, "flattened_constraints");
        const bp_frvec* in8[8] = {aL, aR, aO, sL, sR, w[0], w[1], w[2]};
        bp_frvec* lr[6] = {};
        RC(bp_r1cs_prover_polys(ctx, in8, yle, lr));                                                           // :465-486
        for (auto* v : lr) T.keep(v);
        bp_frvec* zero = nullptr;
        RC(bp_frvec_alloc(ctx, n, &zero));
        T.keep(zero);
        const bp_frvec* lpoly[4] = {zero, lr[0], lr[1], lr[2]};
        const bp_frvec* rpoly[4] = {lr[3], lr[4], zero, lr[5]};
        uint8_t tcoef[6 * 32];
        RC(bp_vecpoly3_special_inner_product(ctx, lpoly, rpoly, tcoef));                                       // t1..t6, :488
        Fe<F> tc[7];
        for (int k = 1; k <= 6; k++) tc[k] = fr_in<F>(tcoef + 32 * (k - 1));
        ph.lap(ctx, "polys + special_inner_product");
        std::vector<uint8_t> gh(2 * pb);
        memcpy(gh.data(), g_le, pb);
        memcpy(gh.data() + pb, h_le, pb);
        const int tk[5] = {1, 3, 4, 5, 6};
        static const char* const tlabel[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
        {   // T_k = t_k g + t_k_blinding h, k = 1, 3, 4, 5, 6 (:496-500): five two-term commitments of host-side points and scalars -- on the
            // host, side by side on the helper threads (0.2 ms; as five GPU launches, three in flight at a time, they took 0.77 ms: each is
            // its upload, its launch and the same 255-doubling tail)
            uint8_t k1[5 * 32], k2[5 * 32];
            uint8_t* tout[5] = {};
            for (int j = 0; j < 5; j++) {
                fr_out<F>(tc[tk[j]], k1 + 32 * j);
                fr_out<F>(tb[tk[j]], k2 + 32 * j);
                tout[j] = P + (6 + j) * pb;
            }
            RC(bp_internal_host_mul2(ctx, g_le, h_le, k1, k2, 5, tout));
        }
        for (int j = 0; j < 5; j++) RC(bp_transcript_commit_point(t, cv, tlabel[j], P + (6 + j) * pb));
        ph.lap(ctx, "T commitments");
        const Fe<F> u = challenge(t, cv, "u"), x = challenge(t, cv, "x");
        uint8_t ule[32], xle[32];
        fr_out<F>(u, ule); fr_out<F>(x, xle);
        tb[2] = fe_zero<F>();
        if (m) {                                                                                               // :513
            uint8_t ip[32];
            RC(bp_fr_inner_product(ctx, w[3], 0, v_blinding, 0, m, ip));
            tb[2] = fr_in<F>(ip);
        }
        Fe<F> t_x = fe_zero<F>(), t_xb = fe_zero<F>(), xp = fe_one<F>();
        for (int k = 1; k <= 6; k++) {
            xp = fe_mul(xp, x);
            t_x = fe_add(t_x, fe_mul(tc[k], xp));
            t_xb = fe_add(t_xb, fe_mul(tb[k], xp));
        }
        bp_frvec *l_eval = nullptr, *r_eval = nullptr;
        RC(bp_vecpoly_eval(ctx, lpoly, 3, xle, &l_eval));                                                      // :522-523
        T.keep(l_eval);
        RC(bp_vecpoly_eval(ctx, rpoly, 3, xle, &r_eval));
        T.keep(r_eval);
        bp_frvec* ippin[4] = {};
        RC(bp_r1cs_ipp_inputs(ctx, l_eval, r_eval, yle, ule, n1, pn, ippin));                                  // :526-563
        for (auto* v : ippin) T.keep(v);
        const Fe<F> i_bl = fe_add(i_bl1, fe_mul(u, i_bl2)), o_bl = fe_add(o_bl1, fe_mul(u, o_bl2)), s_bl = fe_add(s_bl1, fe_mul(u, s_bl2));   // :537-539
        const Fe<F> e_bl = fe_mul(x, fe_add(i_bl, fe_mul(x, fe_add(o_bl, fe_mul(x, s_bl)))));                  // :541
        fr_out<F>(t_x, S); fr_out<F>(t_xb, S + 32); fr_out<F>(e_bl, S + 64);
        RC(bp_transcript_commit_scalar(t, cv, "t_x", S));
        RC(bp_transcript_commit_scalar(t, cv, "t_x_blinding", S + 32));
        RC(bp_transcript_commit_scalar(t, cv, "e_blinding", S + 64));
        const Fe<F> wch = challenge(t, cv, "w");
        uint8_t Q[pb];
        {
            uint8_t k1[32], k2[32] = {0};
            uint8_t* qo[1] = {Q};
            fr_out<F>(wch, k1);
            RC(bp_internal_host_mul2(ctx, g_le, h_le, k1, k2, 1, qo));                                         // Q = w g, :552
        }
        bp_g1vec *Gp = nullptr, *Hp = nullptr;                                                                 // G[0..pn), H[0..pn): views
        RC(bp_g1vec_wrap_device(ctx, G->d, pn, &Gp));
        T.keep(Gp);
        RC(bp_g1vec_wrap_device(ctx, H->d, pn, &Hp));
        T.keep(Hp);
        Gp->tview = G->table ? G->table : G->tview; Gp->tview_off = G->table ? 0 : G->tview_off;      // precomputed generators: the IPP's rounds inherit their tables
        Hp->tview = H->table ? H->table : H->tview; Hp->tview_off = H->table ? 0 : H->tview_off;
        Gp->cview = G->ctable ? G->ctable : G->cview;
        Hp->cview = H->ctable ? H->ctable : H->cview;
        size_t lg_out = 0;
        ph.lap(ctx, "evals, ipp inputs, Q");
        RC(bp_ipp_create(ctx, t, Q, ippin[2], ippin[3], Gp, Hp, ippin[0], ippin[1], Lp, Rp, &lg_out, ab, ab + 32));   // :567-576
        ph.lap(ctx, "ipp");
        return lg_out == lg ? BP_OK : BP_ERR_DEVICE;
    }

    // Verifier::verify up to create_randomized_constraints (verifier.rs:276-284): transcript only
    static int verify_phase1(bp_transcript* t, int cv, size_t m, const uint8_t* P) {
        RC(bp_transcript_append_u64(t, (const uint8_t*)"m", 1, (uint64_t)m));                                  // verifier.rs:278
        RC(bp_transcript_commit_point(t, cv, "A_I1", P + 0 * pb));
        RC(bp_transcript_commit_point(t, cv, "A_O1", P + 1 * pb));
        RC(bp_transcript_commit_point(t, cv, "S1", P + 2 * pb));
        return BP_OK;
    }
    static int verify(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                      const uint8_t* h_le, const uint8_t* V_le, size_t n, size_t m, const uint8_t* proof, const uint8_t* r_le32) {
        RC(verify_phase1(t, ctx->curve, m, proof));
        RC(bp_transcript_append_message(t, (const uint8_t*)"dom-sep", 7, (const uint8_t*)"r1cs-1phase", 11));
        return verify_tail(ctx, t, plan, G, H, g_le, h_le, V_le, n, n, m, proof, r_le32);
    }
    // Verifier::verify after create_randomized_constraints (verifier.rs:289-457); n1 = first-phase multipliers of the n
    static int verify_tail(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                           const uint8_t* h_le, const uint8_t* V_le, size_t n1, size_t n, size_t m, const uint8_t* proof, const uint8_t* r_le32) {
        Temps T;
        Phase vph;
        vph.who = "verify";
        const int cv = ctx->curve;
        const size_t pn = padded_len(n), lg = lg_of(pn);
        const uint8_t* P = proof;
        const uint8_t* S = proof + 11 * pb;
        const uint8_t* Lp = S + 96;
        const uint8_t* Rp = Lp + lg * pb;
        const uint8_t* ab = Rp + lg * pb;
        RC(bp_transcript_commit_point(t, cv, "A_I2", P + 3 * pb));
        RC(bp_transcript_commit_point(t, cv, "A_O2", P + 4 * pb));
        RC(bp_transcript_commit_point(t, cv, "S2", P + 5 * pb));
        const Fe<F> y = challenge(t, cv, "y"), z = challenge(t, cv, "z");
        static const char* const tlabel[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
        for (int j = 0; j < 5; j++) RC(bp_transcript_commit_point(t, cv, tlabel[j], P + (6 + j) * pb));
        const Fe<F> u = challenge(t, cv, "u"), x = challenge(t, cv, "x");
        RC(bp_transcript_commit_scalar(t, cv, "t_x", S));
        RC(bp_transcript_commit_scalar(t, cv, "t_x_blinding", S + 32));
        RC(bp_transcript_commit_scalar(t, cv, "e_blinding", S + 64));
        const Fe<F> wch = challenge(t, cv, "w");
        uint8_t zle[32], xle[32], ule[32], yile[32], wcle[32];
        fr_out<F>(z, zle); fr_out<F>(x, xle); fr_out<F>(u, ule);
        bp_frvec* w[4] = {};
        vph.lap(ctx, "transcript + challenges");
        RC(bp_r1cs_flattened_constraints(ctx, plan, zle, w, wcle));                                            // :329
        for (auto* v : w) T.keep(v);
        vph.lap(ctx, "flattened_constraints");
        const Fe<F> wc = fr_in<F>(wcle), a = fr_in<F>(ab), b = fr_in<F>(ab + 32);
        const Fe<F> y_inv = fe_inv<F>(y);
        fr_out<F>(y_inv, yile);
        bp_frvec *yv = nullptr, *ywr = nullptr;
        RC(bp_fr_vandermonde(ctx, yile, n, &yv));
        T.keep(yv);
        RC(bp_fr_hadamard(ctx, w[1], yv, &ywr));
        T.keep(ywr);
        uint8_t dle[32];
        RC(bp_fr_inner_product(ctx, ywr, 0, w[0], 0, n, dle));                                                 // delta, :344-352
        const Fe<F> delta = fr_in<F>(dle);
        vph.lap(ctx, "delta");
        std::vector<uint8_t> usq(lg * 32 + 32), uisq(lg * 32 + 32);
        bp_frvec *g_sc = nullptr, *h_sc = nullptr;
        RC(bp_r1cs_verifier_scalars(ctx, t, Lp, Rp, lg, pn, n1, w[0], w[1], w[2], yile, xle, ule, ab, ab + 32, usq.data(), uisq.data(), &g_sc, &h_sc));   // :354-390
        T.keep(g_sc);
        T.keep(h_sc);
        vph.lap(ctx, "verifier scalars");
        const Fe<F> r = fr_in<F>(r_le32);                                                                      // :392
        const Fe<F> x2 = fe_sqr(x), x3 = fe_mul(x2, x);
        const Fe<F> tx = fr_in<F>(S), txb = fr_in<F>(S + 32), eb = fr_in<F>(S + 64);
        // The reference's terms (:409-446: A_I1 A_O1 S1 A_I2 A_O2 S2 | V | T_1.. | g h | G | H | L | R) in the order
        //   [A_I1 A_O1 S1 A_I2 A_O2 S2 | V | T_1.. | g h | L | R] ++ [G | H]
        // (a sum does not depend on it): the first nx = head + 2 lg points come from the proof and the caller, one upload; the generators are
        // resident, and when both carry a window table their part runs over the tables without being copied at all.
        const size_t head = 6 + m + 5 + 2, nx = head + 2 * lg, total = nx + 2 * pn;
        bool tabled = false;
        RC(bp_internal_gh_ready(ctx, G, H, pn, &tabled));
        bp_frvec* sc = nullptr;
        bp_g1vec* pts = nullptr;
        RC(bp_frvec_alloc(ctx, total, &sc));
        T.keep(sc);
        std::vector<uint8_t> xs(nx * 32);
        const Fe<F> hv[6] = {x, x2, x3, fe_mul(u, x), fe_mul(u, x2), fe_mul(u, x3)};
        for (int k = 0; k < 6; k++) fr_out<F>(hv[k], xs.data() + 32 * k);
        uint8_t* ts = xs.data() + (6 + m) * 32;
        Fe<F> rx = fe_mul(r, x);
        fr_out<F>(rx, ts);                                                                                     // r x
        Fe<F> acc = fe_mul(r, x3);
        for (int k = 1; k < 5; k++) { fr_out<F>(acc, ts + 32 * k); acc = fe_mul(acc, x); }                     // r x^3 .. r x^6
        const Fe<F> wg = fe_add(fe_mul(wch, fe_sub(tx, fe_mul(a, b))), fe_mul(r, fe_sub(fe_mul(x2, fe_add(wc, delta)), tx)));   // :422
        const Fe<F> ph = fe_neg(fe_add(eb, fe_mul(r, txb)));                                                   // :425
        fr_out<F>(wg, ts + 5 * 32);
        fr_out<F>(ph, ts + 6 * 32);
        if (lg) {
            memcpy(xs.data() + head * 32, usq.data(), lg * 32);
            memcpy(xs.data() + (head + lg) * 32, uisq.data(), lg * 32);
        }
        bp_frvec* d_xs = nullptr;
        RC(upload_scalars(ctx, T, xs, &d_xs));                                                                 // (the V slots are overwritten below)
        RC(bp_frvec_copy(ctx, sc, 0, d_xs, 0, nx));
        if (m) {
            uint8_t rx2[32];
            fr_out<F>(fe_mul(r, x2), rx2);
            bp_frvec* wvs = nullptr;
            RC(bp_fr_scaled_by(ctx, w[3], rx2, &wvs));                                                         // :416
            T.keep(wvs);
            RC(bp_frvec_copy(ctx, sc, 6, wvs, 0, m));
        }
        RC(bp_frvec_copy(ctx, sc, nx, g_sc, 0, pn));
        RC(bp_frvec_copy(ctx, sc, nx + pn, h_sc, 0, pn));
        std::vector<uint8_t> hp(nx * pb);
        memcpy(hp.data(), P, 6 * pb);
        if (m) memcpy(hp.data() + 6 * pb, V_le, m * pb);
        memcpy(hp.data() + (6 + m) * pb, P + 6 * pb, 5 * pb);
        memcpy(hp.data() + (11 + m) * pb, g_le, pb);
        memcpy(hp.data() + (12 + m) * pb, h_le, pb);
        if (lg) {
            memcpy(hp.data() + head * pb, Lp, lg * pb);
            memcpy(hp.data() + (head + lg) * pb, Rp, lg * pb);
        }
        bp_g1vec* d_hp = nullptr;
        RCV(bp_g1vec_upload(ctx, hp.data(), nx, BP_FMT_LE, &d_hp));
        T.keep(d_hp);
        vph.lap(ctx, "term assembly + uploads");
        uint8_t res[pb];
        bool done = false;
        if (tabled) RC(bp_internal_msm_extras_gh(ctx, d_hp->d, sc->d, nx, (const uint8_t*)sc->d + nx * 32, G, H, pn, res, &done));
        if (!done) {
            RC(bp_g1vec_alloc(ctx, total, &pts));
            T.keep(pts);
            hipStream_t s = ctx->stream;
            HIPCHK(hipMemcpyAsync(pts->d, d_hp->d, nx * row, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync((uint8_t*)pts->d + nx * row, G->d, pn * row, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync((uint8_t*)pts->d + (nx + pn) * row, H->d, pn * row, hipMemcpyDeviceToDevice, s));
            RC(bp_msm_g1(ctx, pts, sc, res));                                                                  // :448
        }
        vph.lap(ctx, "msm");
        for (size_t k = 0; k < pb; k++) if (res[k]) return BP_ERR_VERIFY;                                       // !res.is_identity(), :449-451
        return BP_OK;
    }
};

}  // namespace

extern "C" {

size_t bp_r1cs_proof_bytes(int curve_id, size_t n) {
    if (!curve_ok(curve_id) || n == 0) return 0;
    return curve_id == BP_CURVE_BLS12_381 ? R1cs<Bls381>::proof_bytes(n) : R1cs<Bn254>::proof_bytes(n);
}

int bp_r1cs_prove(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                  const uint8_t* h_le, const bp_frvec* a_L, const bp_frvec* a_R, const bp_frvec* a_O, const bp_frvec* v_blinding, const bp_frvec* s_L,
                  const bp_frvec* s_R, const uint8_t* blindings_le32, uint8_t* proof_out, size_t proof_cap) {
    return bp_guard([&]() -> int {
    if (!ctx || !t || !plan || !G || !H || !g_le || !h_le || !a_L || !a_R || !a_O || !s_L || !s_R || !blindings_le32 || !proof_out) return BP_ERR_ARG;
    const size_t n = a_L->n;
    if (n == 0 || a_R->n != n || a_O->n != n || s_L->n != n || s_R->n != n) return BP_ERR_LENGTH;
    size_t pn = 1;
    while (pn < n) pn <<= 1;
    if (G->n < pn || H->n < pn) return BP_ERR_LENGTH;                      // R1CSError::InvalidGeneratorsLength, prover.rs:333,382
    if (proof_cap < bp_r1cs_proof_bytes(ctx->curve, n)) return BP_ERR_LENGTH;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    if (ctx->curve == BP_CURVE_BLS12_381) return R1cs<Bls381>::prove(ctx, t, plan, G, H, g_le, h_le, a_L, a_R, a_O, v_blinding, s_L, s_R, blindings_le32, proof_out);
    return R1cs<Bn254>::prove(ctx, t, plan, G, H, g_le, h_le, a_L, a_R, a_O, v_blinding, s_L, s_R, blindings_le32, proof_out);
    });
}

int bp_r1cs_verify(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                   const uint8_t* h_le, const uint8_t* V_le, size_t n, size_t m, const uint8_t* proof, size_t proof_len, const uint8_t* r_le32) {
    return bp_guard([&]() -> int {
    if (!ctx || !t || !plan || !G || !H || !g_le || !h_le || (m && !V_le) || !proof || n == 0) return BP_ERR_ARG;
    if (proof_len != bp_r1cs_proof_bytes(ctx->curve, n)) return BP_ERR_VERIFY;
    size_t pn = 1;
    while (pn < n) pn <<= 1;
    if (G->n < pn || H->n < pn) return BP_ERR_LENGTH;                      // verifier.rs:296-298
    // The verifier's weight r (verifier.rs:392, FieldElement::random()): drawn here unless the caller supplies one (tests).
    // r = 0 would drop the t(x) / constraint check from the combined MSM, so it is refused, as is a non-canonical value.
    uint8_t rbuf[32];
    if (!r_le32) {
        int rcr = bp_fr_random(ctx->curve, rbuf, 1);
        if (rcr) return rcr;
        r_le32 = rbuf;
    } else if (!bp_fr_is_canonical_nonzero(ctx->curve, r_le32)) return BP_ERR_ARG;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    if (ctx->curve == BP_CURVE_BLS12_381) return R1cs<Bls381>::verify(ctx, t, plan, G, H, g_le, h_le, V_le, n, m, proof, r_le32);
    return R1cs<Bn254>::verify(ctx, t, plan, G, H, g_le, h_le, V_le, n, m, proof, r_le32);
    });
}

// ---- two-phase (randomised) constraint systems: the same functions split where the reference runs the deferred callbacks -------
namespace {
struct Phase1Blob {          // what bp_r1cs_prove_finish needs from bp_r1cs_prove_begin; plain bytes in caller memory
    uint32_t magic, curve;
    uint64_t n1, m;
    uint8_t points[3 * 96];  // A_I1 A_O1 S1 (BP_FMT_LE, 2 * fp_bytes each)
    uint8_t blindings[96];   // i1 o1 s1
};
constexpr uint32_t kPhase1Magic = 0x31504842u;   // "BHP1"
}  // namespace

size_t bp_r1cs_phase1_bytes(void) { return sizeof(Phase1Blob); }

int bp_r1cs_prove_begin(bp_ctx* ctx, bp_transcript* t, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* h_le, size_t m, const bp_frvec* a_L1,
                        const bp_frvec* a_R1, const bp_frvec* a_O1, const bp_frvec* s_L1, const bp_frvec* s_R1, const uint8_t* blindings3_le32,
                        uint8_t* phase1_out, size_t phase1_cap) {
    return bp_guard([&]() -> int {
    if (!ctx || !t || !G || !H || !h_le || !blindings3_le32 || !phase1_out) return BP_ERR_ARG;
    if (phase1_cap < sizeof(Phase1Blob)) return BP_ERR_LENGTH;
    const size_t n1 = a_L1 ? a_L1->n : 0;
    if (n1 && (!a_R1 || !a_O1 || !s_L1 || !s_R1)) return BP_ERR_ARG;
    if (n1 && (a_R1->n != n1 || a_O1->n != n1 || s_L1->n != n1 || s_R1->n != n1)) return BP_ERR_LENGTH;
    if (G->n < n1 || H->n < n1) return BP_ERR_LENGTH;                      // prover.rs:333
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    Phase1Blob b;
    memset(&b, 0, sizeof b);
    b.magic = kPhase1Magic; b.curve = (uint32_t)ctx->curve; b.n1 = n1; b.m = m;
    memcpy(b.blindings, blindings3_le32, 96);
    Temps T;
    uint8_t P3[3 * 96] = {0};
    if (ctx->curve == BP_CURVE_BLS12_381) rc = R1cs<Bls381>::prove_phase1(ctx, T, t, G, H, h_le, m, n1, a_L1, a_R1, a_O1, s_L1, s_R1, blindings3_le32, P3);
    else rc = R1cs<Bn254>::prove_phase1(ctx, T, t, G, H, h_le, m, n1, a_L1, a_R1, a_O1, s_L1, s_R1, blindings3_le32, P3);
    if (rc) return rc;
    memcpy(b.points, P3, sizeof P3);
    RC(bp_transcript_append_message(t, (const uint8_t*)"dom-sep", 7, (const uint8_t*)"r1cs-2phase", 11));      // prover.rs:308
    memcpy(phase1_out, &b, sizeof b);
    return BP_OK;
    });
}

int bp_r1cs_prove_finish(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                         const uint8_t* h_le, const uint8_t* phase1, const bp_frvec* a_L, const bp_frvec* a_R, const bp_frvec* a_O,
                         const bp_frvec* v_blinding, const bp_frvec* s_L, const bp_frvec* s_R, const uint8_t* blindings8_le32, uint8_t* proof_out,
                         size_t proof_cap) {
    return bp_guard([&]() -> int {
    if (!ctx || !t || !plan || !G || !H || !g_le || !h_le || !phase1 || !a_L || !a_R || !a_O || !s_L || !s_R || !blindings8_le32 || !proof_out) return BP_ERR_ARG;
    Phase1Blob b;
    memcpy(&b, phase1, sizeof b);
    if (b.magic != kPhase1Magic || b.curve != (uint32_t)ctx->curve) return BP_ERR_ARG;
    const size_t n = a_L->n, m = v_blinding ? v_blinding->n : 0;
    if (n == 0 || b.n1 > n || b.m != m) return BP_ERR_ARG;
    if (a_R->n != n || a_O->n != n || s_L->n != n || s_R->n != n) return BP_ERR_LENGTH;
    size_t pn = 1;
    while (pn < n) pn <<= 1;
    if (G->n < pn || H->n < pn) return BP_ERR_LENGTH;                      // prover.rs:382
    if (proof_cap < bp_r1cs_proof_bytes(ctx->curve, n)) return BP_ERR_LENGTH;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    memset(proof_out, 0, bp_r1cs_proof_bytes(ctx->curve, n));
    Temps T;
    if (ctx->curve == BP_CURVE_BLS12_381) {
        memcpy(proof_out, b.points, 3 * R1cs<Bls381>::pb);
        return R1cs<Bls381>::prove_tail(ctx, T, t, plan, G, H, g_le, h_le, (size_t)b.n1, a_L, a_R, a_O, v_blinding, s_L, s_R, b.blindings, blindings8_le32,
                                        blindings8_le32 + 96, proof_out);
    }
    memcpy(proof_out, b.points, 3 * R1cs<Bn254>::pb);
    return R1cs<Bn254>::prove_tail(ctx, T, t, plan, G, H, g_le, h_le, (size_t)b.n1, a_L, a_R, a_O, v_blinding, s_L, s_R, b.blindings, blindings8_le32,
                                   blindings8_le32 + 96, proof_out);
    });
}

int bp_r1cs_verify_begin(bp_transcript* t, int curve_id, size_t m, const uint8_t* proof, size_t proof_len) {
    return bp_guard([&]() -> int {
    if (!t || !proof || !curve_ok(curve_id)) return BP_ERR_ARG;
    const size_t pb = curve_id == BP_CURVE_BLS12_381 ? R1cs<Bls381>::pb : R1cs<Bn254>::pb;
    if (proof_len < 3 * pb) return BP_ERR_VERIFY;
    RC(curve_id == BP_CURVE_BLS12_381 ? R1cs<Bls381>::verify_phase1(t, curve_id, m, proof) : R1cs<Bn254>::verify_phase1(t, curve_id, m, proof));
    return bp_transcript_append_message(t, (const uint8_t*)"dom-sep", 7, (const uint8_t*)"r1cs-2phase", 11);  // verifier.rs:253
    });
}

int bp_r1cs_verify_finish(bp_ctx* ctx, bp_transcript* t, const bp_r1cs_plan* plan, const bp_g1vec* G, const bp_g1vec* H, const uint8_t* g_le,
                          const uint8_t* h_le, const uint8_t* V_le, size_t n1, size_t n, size_t m, const uint8_t* proof, size_t proof_len,
                          const uint8_t* r_le32) {
    return bp_guard([&]() -> int {
    if (!ctx || !t || !plan || !G || !H || !g_le || !h_le || (m && !V_le) || !proof || n == 0 || n1 > n) return BP_ERR_ARG;
    if (proof_len != bp_r1cs_proof_bytes(ctx->curve, n)) return BP_ERR_VERIFY;
    size_t pn = 1;
    while (pn < n) pn <<= 1;
    if (G->n < pn || H->n < pn) return BP_ERR_LENGTH;                      // verifier.rs:296-298
    uint8_t rbuf[32];
    if (!r_le32) {
        int rcr = bp_fr_random(ctx->curve, rbuf, 1);
        if (rcr) return rcr;
        r_le32 = rbuf;
    } else if (!bp_fr_is_canonical_nonzero(ctx->curve, r_le32)) return BP_ERR_ARG;
    int rc = bp_internal_set_device(ctx); if (rc) return rc;
    if (ctx->curve == BP_CURVE_BLS12_381) return R1cs<Bls381>::verify_tail(ctx, t, plan, G, H, g_le, h_le, V_le, n1, n, m, proof, r_le32);
    return R1cs<Bn254>::verify_tail(ctx, t, plan, G, H, g_le, h_le, V_le, n1, n, m, proof, r_le32);
    });
}

}  // extern "C"
