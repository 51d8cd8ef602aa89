// include/bpmsm.hpp -- C++ host-side mirror of the reference's interface for the hot path, header-only, over the C ABI
// of include/bpmsm.h.  (The reference is Rust; there is no Rust toolchain in the build image, so the compiled-language
// host mirror is C++.)  Names, argument meaning and error behaviour follow the reference:
//
//   bp::FieldElementVector        amcl_wrapper::field_elem::FieldElementVector    inner_product, hadamard_product, scaled_by,
//                                                                                 new_vandermonde_vector
//   bp::G1Vector                  amcl_wrapper::group_elem_g1::G1Vector           multi_scalar_mul_var_time,
//                                                                                 inner_product_var_time / _const_time
//   bp::Transcript                merlin::Transcript + TranscriptProtocol         /root/reference src/transcript.rs:12-61
//   bp::InnerProductArgumentProof /root/reference src/ipp.rs:13-20
//   bp::IPP::create_ipp / verify_ipp / verification_scalars                       /root/reference src/ipp.rs:35-315
//
// Errors: amcl_wrapper's ValueError (length mismatch) -> bp::ValueError; the assert!s of create_ipp -> bp::ArgError
// (the reference panics); R1CSError::VerificationError -> bp::VerificationError; HIP failures / no GPU -> bp::DeviceError.
// Points are BP_FMT_LE byte strings (x || y little-endian), scalars 32-byte little-endian.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bpmsm.h"

namespace bp {

using Bytes = std::vector<uint8_t>;

struct Error : std::runtime_error { int code; Error(const std::string& w, int c) : std::runtime_error(w + " failed with status " + std::to_string(c)), code(c) {} };
struct ValueError : Error { using Error::Error; };
struct ArgError : Error { using Error::Error; };
struct VerificationError : Error { using Error::Error; };
struct DeviceError : Error { using Error::Error; };

inline void check(int rc, const char* what) {
    switch (rc) {
        case BP_OK: return;
        case BP_ERR_LENGTH: throw ValueError(what, rc);
        case BP_ERR_ARG: throw ArgError(what, rc);
        case BP_ERR_VERIFY: throw VerificationError(what, rc);
        default: throw DeviceError(what, rc);
    }
}

class Context {
public:
    explicit Context(int curve_id = BP_CURVE_BLS12_381, int device = 0) : curve_(curve_id) {
        check(bp_ctx_create(curve_id, device, &h_), "bp_ctx_create");
        bp_curve_info info;
        check(bp_curve_params(curve_id, &info), "bp_curve_params");
        point_bytes_ = 2 * (size_t)info.fp_bytes;
    }
    ~Context() { bp_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    bp_ctx* handle() const { return h_; }
    int curve() const { return curve_; }
    size_t point_bytes() const { return point_bytes_; }

private:
    bp_ctx* h_ = nullptr;
    int curve_;
    size_t point_bytes_;
};

class FieldElementVector {
public:
    FieldElementVector(Context& ctx, const Bytes& scalars_le32) : ctx_(&ctx) {
        check(bp_frvec_upload(ctx.handle(), scalars_le32.data(), scalars_le32.size() / 32, &h_), "bp_frvec_upload");
    }
    FieldElementVector(Context& ctx, bp_frvec* owned) : ctx_(&ctx), h_(owned) {}
    FieldElementVector(FieldElementVector&& o) noexcept : ctx_(o.ctx_), h_(o.h_) { o.h_ = nullptr; }
    FieldElementVector(const FieldElementVector&) = delete;
    ~FieldElementVector() { bp_frvec_free(h_); }
    static FieldElementVector new_vandermonde_vector(Context& ctx, const Bytes& e_le32, size_t n) {
        bp_frvec* h = nullptr;
        check(bp_fr_vandermonde(ctx.handle(), e_le32.data(), n, &h), "bp_fr_vandermonde");
        return FieldElementVector(ctx, h);
    }
    size_t len() const { return bp_frvec_len(h_); }
    Bytes to_bytes() const {
        Bytes out(len() * 32);
        check(bp_frvec_download(ctx_->handle(), h_, 0, len(), out.data()), "bp_frvec_download");
        return out;
    }
    Bytes inner_product(const FieldElementVector& rhs) const {
        if (len() != rhs.len()) throw ValueError("inner_product", BP_ERR_LENGTH);
        Bytes out(32);
        check(bp_fr_inner_product(ctx_->handle(), h_, 0, rhs.h_, 0, len(), out.data()), "bp_fr_inner_product");
        return out;
    }
    FieldElementVector hadamard_product(const FieldElementVector& rhs) const {
        bp_frvec* h = nullptr;
        check(bp_fr_hadamard(ctx_->handle(), h_, rhs.h_, &h), "bp_fr_hadamard");
        return FieldElementVector(*ctx_, h);
    }
    FieldElementVector scaled_by(const Bytes& s_le32) const {
        bp_frvec* h = nullptr;
        check(bp_fr_scaled_by(ctx_->handle(), h_, s_le32.data(), &h), "bp_fr_scaled_by");
        return FieldElementVector(*ctx_, h);
    }
    bp_frvec* handle() const { return h_; }

private:
    Context* ctx_;
    bp_frvec* h_ = nullptr;
};

class G1Vector {
public:
    G1Vector(Context& ctx, const Bytes& points_le) : ctx_(&ctx) {
        check(bp_g1vec_upload(ctx.handle(), points_le.data(), points_le.size() / ctx.point_bytes(), BP_FMT_LE, &h_), "bp_g1vec_upload");
    }
    G1Vector(Context& ctx, bp_g1vec* owned) : ctx_(&ctx), h_(owned) {}
    G1Vector(G1Vector&& o) noexcept : ctx_(o.ctx_), h_(o.h_) { o.h_ = nullptr; }
    G1Vector(const G1Vector&) = delete;
    ~G1Vector() { bp_g1vec_free(h_); }
    // [k_i * G]: stand-in generator vectors (the reference's get_generators hashes to the curve, src/utils/mod.rs:16-23)
    static G1Vector fixed_base(Context& ctx, const FieldElementVector& k) {
        bp_g1vec* h = nullptr;
        check(bp_g1vec_fixed_base_mul(ctx.handle(), k.handle(), &h), "bp_g1vec_fixed_base_mul");
        return G1Vector(ctx, h);
    }
    // utils::get_generators(prefix, n) (src/utils/mod.rs:16-23): from_msg_hash(prefix || decimal(i)), i = 1..n
    static G1Vector get_generators(Context& ctx, const std::string& prefix, size_t n) {
        bp_g1vec* h = nullptr;
        check(bp_get_generators(ctx.handle(), reinterpret_cast<const uint8_t*>(prefix.data()), prefix.size(), 1, n, &h), "bp_get_generators");
        return G1Vector(ctx, h);
    }
    // [G1::from_msg_hash(m) for m in messages]
    static G1Vector from_msg_hash(Context& ctx, const std::vector<std::string>& messages) {
        std::vector<uint64_t> offs(messages.size() + 1, 0);
        std::string all;
        for (size_t i = 0; i < messages.size(); i++) { all += messages[i]; offs[i + 1] = all.size(); }
        bp_g1vec* h = nullptr;
        check(bp_g1vec_from_msg_hash(ctx.handle(), reinterpret_cast<const uint8_t*>(all.data()), offs.data(), messages.size(), &h), "bp_g1vec_from_msg_hash");
        return G1Vector(ctx, h);
    }
    // this build's compressed wire form (tag || X big-endian, bp_g1_compressed_bytes per point); invalid encodings -> ArgError
    static G1Vector from_compressed(Context& ctx, const Bytes& compressed) {
        bp_g1vec* h = nullptr;
        check(bp_g1vec_decompress(ctx.handle(), compressed.data(), compressed.size() / bp_g1_compressed_bytes(ctx.curve()), &h), "bp_g1vec_decompress");
        return G1Vector(ctx, h);
    }
    Bytes to_compressed() const {
        Bytes out(len() * bp_g1_compressed_bytes(ctx_->curve()));
        check(bp_g1vec_compress(ctx_->handle(), h_, 0, len(), out.data()), "bp_g1vec_compress");
        return out;
    }
    // window-multiples table for a fixed generator vector (bp_g1vec_precompute): every later MSM over the whole vector, and the
    // IPP / R1CS provers that take it as G or H, run merged-window; same bytes with and without
    void precompute(int window_bits = 0) { check(bp_g1vec_precompute(ctx_->handle(), h_, window_bits), "bp_g1vec_precompute"); }
    void drop_table() { check(bp_g1vec_drop_table(h_), "bp_g1vec_drop_table"); }
    size_t len() const { return bp_g1vec_len(h_); }
    Bytes to_bytes() const {
        Bytes out(len() * ctx_->point_bytes());
        check(bp_g1vec_download(ctx_->handle(), h_, 0, len(), BP_FMT_LE, out.data()), "bp_g1vec_download");
        return out;
    }
    // multi_scalar_mul_var_time over index-range shards that live in several contexts (one per device; several contexts on one
    // device work too): every shard's device stage runs concurrently, one host fold (bp_msm_g1_multi)
    static Bytes multi_scalar_mul_var_time_sharded(const std::vector<Context*>& ctxs, const std::vector<const G1Vector*>& points,
                                                   const std::vector<const FieldElementVector*>& scalars) {
        if (ctxs.empty() || ctxs.size() != points.size() || ctxs.size() != scalars.size()) throw ValueError("multi_scalar_mul_var_time_sharded", BP_ERR_LENGTH);
        std::vector<bp_ctx*> c;
        std::vector<const bp_g1vec*> p;
        std::vector<const bp_frvec*> k;
        for (size_t i = 0; i < ctxs.size(); i++) { c.push_back(ctxs[i]->handle()); p.push_back(points[i]->handle()); k.push_back(scalars[i]->handle()); }
        Bytes out(ctxs[0]->point_bytes());
        check(bp_msm_g1_multi(c.data(), p.data(), k.data(), c.size(), out.data()), "bp_msm_g1_multi");
        return out;
    }
    Bytes multi_scalar_mul_var_time(const FieldElementVector& scalars) const {
        Bytes out(ctx_->point_bytes());
        check(bp_msm_g1(ctx_->handle(), h_, scalars.handle(), out.data()), "bp_msm_g1");
        return out;
    }
    Bytes inner_product_var_time(const FieldElementVector& s) const { return multi_scalar_mul_var_time(s); }
    Bytes inner_product_const_time(const FieldElementVector& s) const { return multi_scalar_mul_var_time(s); }
    bp_g1vec* handle() const { return h_; }

private:
    Context* ctx_;
    bp_g1vec* h_ = nullptr;
};

class Transcript {
public:
    explicit Transcript(const std::string& label) { check(bp_transcript_new((const uint8_t*)label.data(), label.size(), &h_), "bp_transcript_new"); }
    ~Transcript() { bp_transcript_free(h_); }
    Transcript(const Transcript&) = delete;
    void append_message(const std::string& label, const Bytes& msg) {
        check(bp_transcript_append_message(h_, (const uint8_t*)label.data(), label.size(), msg.data(), msg.size()), "append_message");
    }
    Bytes challenge_bytes(const std::string& label, size_t n) {
        Bytes out(n);
        check(bp_transcript_challenge_bytes(h_, (const uint8_t*)label.data(), label.size(), out.data(), n), "challenge_bytes");
        return out;
    }
    void commit_point(int curve, const char* label, const Bytes& p) { check(bp_transcript_commit_point(h_, curve, label, p.data()), "commit_point"); }
    void commit_scalar(int curve, const char* label, const Bytes& s) { check(bp_transcript_commit_scalar(h_, curve, label, s.data()), "commit_scalar"); }
    Bytes challenge_scalar(int curve, const char* label) {
        Bytes out(32);
        check(bp_transcript_challenge_scalar(h_, curve, label, out.data()), "challenge_scalar");
        return out;
    }
    bp_transcript* handle() const { return h_; }

private:
    bp_transcript* h_ = nullptr;
};

struct InnerProductArgumentProof {   // src/ipp.rs:13-20
    Bytes L, R;                      // lg n points each, BP_FMT_LE
    Bytes a, b;                      // 32-byte LE scalars
};

struct IPP {
    // src/ipp.rs:35-202
    static InnerProductArgumentProof create_ipp(Context& ctx, Transcript& transcript, const Bytes& Q, const FieldElementVector& G_factors,
                                                const FieldElementVector& H_factors, const G1Vector& G_vec, const G1Vector& H_vec,
                                                const FieldElementVector& a_vec, const FieldElementVector& b_vec) {
        size_t n = G_vec.len(), lg = 0;
        InnerProductArgumentProof p;
        p.L.resize(64 * ctx.point_bytes());
        p.R.resize(64 * ctx.point_bytes());
        p.a.resize(32);
        p.b.resize(32);
        check(bp_ipp_create(ctx.handle(), transcript.handle(), Q.data(), G_factors.handle(), H_factors.handle(), G_vec.handle(), H_vec.handle(),
                            a_vec.handle(), b_vec.handle(), p.L.data(), p.R.data(), &lg, p.a.data(), p.b.data()),
              "bp_ipp_create");
        (void)n;
        p.L.resize(lg * ctx.point_bytes());
        p.R.resize(lg * ctx.point_bytes());
        return p;
    }
    // create_ipp with the generators sharded by index range over several contexts / devices (bp_ipp_create_multi); a, b as n x 32-byte
    // host scalars.  Same proof bytes as create_ipp.
    static InnerProductArgumentProof create_ipp_sharded(const std::vector<Context*>& ctxs, Transcript& transcript, const Bytes& Q,
                                                        const std::vector<const FieldElementVector*>& G_factors,
                                                        const std::vector<const FieldElementVector*>& H_factors, const std::vector<const G1Vector*>& G_vecs,
                                                        const std::vector<const G1Vector*>& H_vecs, const Bytes& a_le32, const Bytes& b_le32) {
        const size_t k = ctxs.size();
        if (k == 0 || G_factors.size() != k || H_factors.size() != k || G_vecs.size() != k || H_vecs.size() != k) throw ValueError("create_ipp_sharded", BP_ERR_LENGTH);
        std::vector<bp_ctx*> c;
        std::vector<const bp_frvec*> gf, hf;
        std::vector<const bp_g1vec*> g, h;
        size_t n = 0, lg = 0;
        for (size_t i = 0; i < k; i++) {
            c.push_back(ctxs[i]->handle()); gf.push_back(G_factors[i]->handle()); hf.push_back(H_factors[i]->handle());
            g.push_back(G_vecs[i]->handle()); h.push_back(H_vecs[i]->handle());
            n += G_vecs[i]->len();
        }
        if (a_le32.size() != 32 * n || b_le32.size() != 32 * n) throw ValueError("create_ipp_sharded", BP_ERR_LENGTH);
        InnerProductArgumentProof p;
        p.L.resize(64 * ctxs[0]->point_bytes());
        p.R.resize(64 * ctxs[0]->point_bytes());
        p.a.resize(32);
        p.b.resize(32);
        check(bp_ipp_create_multi(c.data(), k, transcript.handle(), Q.data(), gf.data(), hf.data(), g.data(), h.data(), a_le32.data(), b_le32.data(), n,
                                  p.L.data(), p.R.data(), &lg, p.a.data(), p.b.data()),
              "bp_ipp_create_multi");
        p.L.resize(lg * ctxs[0]->point_bytes());
        p.R.resize(lg * ctxs[0]->point_bytes());
        return p;
    }
    // src/ipp.rs:204-260: returns on success, throws VerificationError otherwise (Result<(), R1CSError>)
    static void verify_ipp(Context& ctx, size_t n, Transcript& transcript, const FieldElementVector& G_factors, const FieldElementVector& H_factors,
                           const Bytes& P, const Bytes& Q, const G1Vector& G, const G1Vector& H, const Bytes& a, const Bytes& b, const Bytes& L_vec,
                           const Bytes& R_vec) {
        check(bp_ipp_verify(ctx.handle(), transcript.handle(), n, G_factors.handle(), H_factors.handle(), P.data(), Q.data(), G.handle(), H.handle(),
                            a.data(), b.data(), L_vec.data(), R_vec.data(), L_vec.size() / ctx.point_bytes()),
              "bp_ipp_verify");
    }
    // m proofs over the same generators in ONE MSM (random linear combination across proofs, bp_ipp_verify_batch).
    // weights: m 32-byte LE scalars drawn by the caller after the proofs are fixed.  Throws VerificationError if the
    // combination is not the identity (it does not say which proof is bad: fall back to verify_ipp per proof).
    struct BatchItem {
        Transcript* transcript;
        const Bytes* P; const Bytes* Q; const InnerProductArgumentProof* proof;
    };
    static void verify_batch(Context& ctx, size_t n, const FieldElementVector& G_factors, const FieldElementVector& H_factors, const G1Vector& G,
                             const G1Vector& H, const std::vector<BatchItem>& items, const Bytes& weights) {
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        std::vector<bp_ipp_proof_ref> refs(items.size());
        for (size_t i = 0; i < items.size(); i++) {
            const InnerProductArgumentProof& p = *items[i].proof;
            if (p.L.size() != lg * ctx.point_bytes() || p.R.size() != lg * ctx.point_bytes()) throw VerificationError("bp_ipp_verify_batch", BP_ERR_VERIFY);
            refs[i] = bp_ipp_proof_ref{items[i].transcript->handle(), items[i].P->data(), items[i].Q->data(), p.a.data(), p.b.data(), p.L.data(), p.R.data()};
        }
        if (weights.size() != 32 * items.size()) throw ArgError("bp_ipp_verify_batch", BP_ERR_ARG);
        check(bp_ipp_verify_batch(ctx.handle(), n, lg, G_factors.handle(), H_factors.handle(), G.handle(), H.handle(), refs.data(), refs.size(), weights.data()),
              "bp_ipp_verify_batch");
    }
    // src/ipp.rs:262-315: (u_sq, u_inv_sq, s)
    static void verification_scalars(int curve, size_t point_bytes, const Bytes& L_vec, const Bytes& R_vec, size_t n, Transcript& transcript, Bytes& u_sq,
                                     Bytes& u_inv_sq, Bytes& s) {
        size_t lg = L_vec.size() / point_bytes;
        u_sq.assign(lg * 32 + 32, 0); u_inv_sq.assign(lg * 32 + 32, 0); s.assign(n * 32 + 32, 0);
        check(bp_ipp_verification_scalars(curve, transcript.handle(), L_vec.data(), R_vec.data(), lg, n, u_sq.data(), u_inv_sq.data(), s.data()),
              "bp_ipp_verification_scalars");
        u_sq.resize(lg * 32); u_inv_sq.resize(lg * 32); s.resize(n * 32);
    }
};

// ---- R1CS layer: the callers of the hot path (reference src/r1cs/prover.rs, src/r1cs/verifier.rs) ------------------------
// A circuit arrives as flat terms (the constraint-system builder and the gadgets are not mirrored).
struct Term { uint32_t constraint; uint8_t kind; uint32_t index; Bytes coeff_le32; };   // kind = BP_VAR_*

class R1CSPlan {   // the terms regrouped once per circuit for flattened_constraints (prover.rs:142-184, verifier.rs:149-193)
public:
    R1CSPlan(Context& ctx, const std::vector<Term>& terms, size_t n_constraints, size_t n, size_t m) : n_(n), m_(m) {
        std::vector<uint32_t> q(terms.size() + 1), idx(terms.size() + 1);
        std::vector<uint8_t> kind(terms.size() + 1), coeff(32 * terms.size() + 32);
        for (size_t t = 0; t < terms.size(); t++) {
            q[t] = terms[t].constraint; kind[t] = terms[t].kind; idx[t] = terms[t].index;
            for (int k = 0; k < 32; k++) coeff[32 * t + k] = terms[t].coeff_le32[k];
        }
        check(bp_r1cs_plan_create(ctx.handle(), terms.size(), q.data(), kind.data(), idx.data(), coeff.data(), n_constraints, n, m, &h_), "bp_r1cs_plan_create");
    }
    R1CSPlan(const R1CSPlan&) = delete;
    ~R1CSPlan() { bp_r1cs_plan_free(h_); }
    bp_r1cs_plan* handle() const { return h_; }
    size_t n() const { return n_; }
    size_t m() const { return m_; }

private:
    bp_r1cs_plan* h_ = nullptr;
    size_t n_, m_;
};

namespace r1cs {

// Prover::commit for all values at once: [v_j g + r_j h]  (prover.rs:118-127; batched commit_to_field_element)
inline Bytes commit(Context& ctx, const Bytes& g, const Bytes& h, const FieldElementVector& v, const FieldElementVector& blinding) {
    bp_g1vec* out = nullptr;
    check(bp_g1vec_commit_pairs(ctx.handle(), g.data(), h.data(), v.handle(), blinding.handle(), &out), "bp_g1vec_commit_pairs");
    return G1Vector(ctx, out).to_bytes();
}

// Prover::new + commit on the transcript side: r1cs_domain_sep, then one commit_point("V") per commitment
inline void start_transcript(Context& ctx, Transcript& t, const Bytes& V) {
    t.append_message("dom-sep", Bytes{'r', '1', 'c', 's', ' ', 'v', '1'});
    for (size_t j = 0; j * ctx.point_bytes() < V.size(); j++)
        t.commit_point(ctx.curve(), "V", Bytes(V.begin() + j * ctx.point_bytes(), V.begin() + (j + 1) * ctx.point_bytes()));
}

// Prover::prove (prover.rs:323-560), single phase.  blindings = i, o, s, t1, t3, t4, t5, t6 (8 x 32 bytes).
inline Bytes prove(Context& ctx, Transcript& t, const R1CSPlan& plan, const G1Vector& G, const G1Vector& H, const Bytes& g, const Bytes& h,
                   const FieldElementVector& a_L, const FieldElementVector& a_R, const FieldElementVector& a_O, const FieldElementVector* v_blinding,
                   const FieldElementVector& s_L, const FieldElementVector& s_R, const Bytes& blindings) {
    Bytes proof(bp_r1cs_proof_bytes(ctx.curve(), a_L.len()));
    check(bp_r1cs_prove(ctx.handle(), t.handle(), plan.handle(), G.handle(), H.handle(), g.data(), h.data(), a_L.handle(), a_R.handle(), a_O.handle(),
                        v_blinding ? v_blinding->handle() : nullptr, s_L.handle(), s_R.handle(), blindings.data(), proof.data(), proof.size()),
          "bp_r1cs_prove");
    return proof;
}

// Verifier::verify (verifier.rs:265-452): returns on success, throws VerificationError otherwise.
// r_weight_le32: the verifier's random combination weight (verifier.rs:392); empty = drawn inside the library (the reference's behaviour).
inline void verify(Context& ctx, Transcript& t, const R1CSPlan& plan, const G1Vector& G, const G1Vector& H, const Bytes& g, const Bytes& h, const Bytes& V,
                   const Bytes& proof, const Bytes& r_weight_le32 = Bytes()) {
    check(bp_r1cs_verify(ctx.handle(), t.handle(), plan.handle(), G.handle(), H.handle(), g.data(), h.data(), V.empty() ? nullptr : V.data(), plan.n(),
                         V.size() / ctx.point_bytes(), proof.data(), proof.size(), r_weight_le32.empty() ? nullptr : r_weight_le32.data()),
          "bp_r1cs_verify");
}

// Randomised (two-phase) systems: Prover::prove / Verifier::verify split where the reference runs the callbacks of
// specify_randomized_constraints (prover.rs:298-319, verifier.rs:245-263).  Between begin and finish the caller draws the
// callbacks' challenges from the same transcript (Transcript::challenge_scalar) and builds the plan of the complete system.
inline Bytes prove_begin(Context& ctx, Transcript& t, const G1Vector& G, const G1Vector& H, const Bytes& h, size_t m, const FieldElementVector* a_L1,
                         const FieldElementVector* a_R1, const FieldElementVector* a_O1, const FieldElementVector* s_L1, const FieldElementVector* s_R1,
                         const Bytes& blindings3) {
    Bytes phase1(bp_r1cs_phase1_bytes());
    auto hd = [](const FieldElementVector* v) { return v ? v->handle() : nullptr; };
    check(bp_r1cs_prove_begin(ctx.handle(), t.handle(), G.handle(), H.handle(), h.data(), m, hd(a_L1), hd(a_R1), hd(a_O1), hd(s_L1), hd(s_R1), blindings3.data(),
                              phase1.data(), phase1.size()),
          "bp_r1cs_prove_begin");
    return phase1;
}
// blindings8 = i2, o2, s2, t1, t3, t4, t5, t6; the vectors hold all n1 + n2 multipliers
inline Bytes prove_finish(Context& ctx, Transcript& t, const R1CSPlan& plan, const G1Vector& G, const G1Vector& H, const Bytes& g, const Bytes& h,
                          const Bytes& phase1, const FieldElementVector& a_L, const FieldElementVector& a_R, const FieldElementVector& a_O,
                          const FieldElementVector* v_blinding, const FieldElementVector& s_L, const FieldElementVector& s_R, const Bytes& blindings8) {
    Bytes proof(bp_r1cs_proof_bytes(ctx.curve(), a_L.len()));
    check(bp_r1cs_prove_finish(ctx.handle(), t.handle(), plan.handle(), G.handle(), H.handle(), g.data(), h.data(), phase1.data(), a_L.handle(), a_R.handle(),
                               a_O.handle(), v_blinding ? v_blinding->handle() : nullptr, s_L.handle(), s_R.handle(), blindings8.data(), proof.data(), proof.size()),
          "bp_r1cs_prove_finish");
    return proof;
}
inline void verify_begin(Context& ctx, Transcript& t, size_t m, const Bytes& proof) {
    check(bp_r1cs_verify_begin(t.handle(), ctx.curve(), m, proof.data(), proof.size()), "bp_r1cs_verify_begin");
}
inline void verify_finish(Context& ctx, Transcript& t, const R1CSPlan& plan, const G1Vector& G, const G1Vector& H, const Bytes& g, const Bytes& h, const Bytes& V,
                          size_t n1, const Bytes& proof, const Bytes& r_weight_le32 = Bytes()) {
    check(bp_r1cs_verify_finish(ctx.handle(), t.handle(), plan.handle(), G.handle(), H.handle(), g.data(), h.data(), V.empty() ? nullptr : V.data(), n1, plan.n(),
                                V.size() / ctx.point_bytes(), proof.data(), proof.size(), r_weight_le32.empty() ? nullptr : r_weight_le32.data()),
          "bp_r1cs_verify_finish");
}

// serde of R1CSProof (src/r1cs/proof.rs:24) in this build's compressed point form; a bad encoding -> VerificationError
inline Bytes compress_proof(Context& ctx, size_t n_gates, const Bytes& proof) {
    Bytes out(bp_r1cs_proof_compressed_bytes(ctx.curve(), n_gates));
    check(bp_r1cs_proof_compress(ctx.handle(), n_gates, proof.data(), proof.size(), out.data(), out.size()), "bp_r1cs_proof_compress");
    return out;
}
inline Bytes decompress_proof(Context& ctx, size_t n_gates, const Bytes& compressed) {
    Bytes out(bp_r1cs_proof_bytes(ctx.curve(), n_gates));
    check(bp_r1cs_proof_decompress(ctx.handle(), n_gates, compressed.data(), compressed.size(), out.data(), out.size()), "bp_r1cs_proof_decompress");
    return out;
}

}  // namespace r1cs

}  // namespace bp
