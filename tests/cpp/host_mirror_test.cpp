// C++ host-mirror test (include/bpmsm.hpp).  `cpu` mode needs no GPU: Merlin conformance vector + error mapping.
// `gpu` mode reproduces the reference's own unit test shape (/root/reference src/ipp.rs:325-390): n = 4, a = [1..4],
// b = [5..8], G_factors = 1, H_factors = vandermonde(y_inv); P = <a,G> + <b.y^i,H> + <a,b> Q; create, verify, tamper.
#include <cstdio>
#include <cstring>
#include <string>

#include "bpmsm.hpp"

static bp::Bytes scalar(uint64_t v) { bp::Bytes b(32, 0); for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); return b; }
static bp::Bytes concat(std::initializer_list<bp::Bytes> xs) { bp::Bytes o; for (auto& x : xs) o.insert(o.end(), x.begin(), x.end()); return o; }
static std::string hex(const bp::Bytes& b) { static const char* d = "0123456789abcdef"; std::string s; for (uint8_t c : b) { s += d[c >> 4]; s += d[c & 15]; } return s; }

static int cpu_mode() {
    bp::Transcript t("test protocol");
    t.append_message("some label", bp::Bytes{'s', 'o', 'm', 'e', ' ', 'd', 'a', 't', 'a'});
    std::string got = hex(t.challenge_bytes("challenge", 32));
    if (got != "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615") { printf("merlin mismatch %s\n", got.c_str()); return 1; }
    try { bp::Context ctx(7, 0); printf("bad curve accepted\n"); return 1; } catch (const bp::ArgError&) {}
    printf("cpp cpu ok\n");
    return 0;
}

static int gpu_mode(int curve) {
    bp::Context ctx(curve, 0);
    const size_t n = 4;
    auto ints = [&](std::initializer_list<uint64_t> v) { bp::Bytes o; for (uint64_t x : v) { auto s = scalar(x); o.insert(o.end(), s.begin(), s.end()); } return o; };
    bp::FieldElementVector gk(ctx, ints({11, 12, 13, 14})), hk(ctx, ints({21, 22, 23, 24})), qk(ctx, ints({31}));
    bp::G1Vector G = bp::G1Vector::fixed_base(ctx, gk), H = bp::G1Vector::fixed_base(ctx, hk);
    bp::Bytes Q = bp::G1Vector::fixed_base(ctx, qk).to_bytes();
    bp::FieldElementVector a(ctx, ints({1, 2, 3, 4})), b(ctx, ints({5, 6, 7, 8}));
    bp::FieldElementVector G_factors(ctx, ints({1, 1, 1, 1}));
    bp::FieldElementVector H_factors = bp::FieldElementVector::new_vandermonde_vector(ctx, scalar(0x1234567), n);
    bp::Transcript t1("innerproduct");
    bp::InnerProductArgumentProof proof = bp::IPP::create_ipp(ctx, t1, Q, G_factors, H_factors, G, H, a, b);
    if (proof.L.size() != 2 * ctx.point_bytes()) { printf("wrong proof size\n"); return 1; }
    // P = G^a * H^(b*y^i) * Q^c   (src/ipp.rs:353-372)
    bp::FieldElementVector b_prime = b.hadamard_product(H_factors);
    bp::Bytes c = a.inner_product(b);
    bp::G1Vector pts(ctx, concat({G.to_bytes(), H.to_bytes(), Q}));
    bp::FieldElementVector sc(ctx, concat({a.to_bytes(), b_prime.to_bytes(), c}));
    bp::Bytes P = pts.multi_scalar_mul_var_time(sc);
    bp::Transcript t2("innerproduct");
    bp::IPP::verify_ipp(ctx, n, t2, G_factors, H_factors, P, Q, G, H, proof.a, proof.b, proof.L, proof.R);
    bp::Bytes bad = proof.a; bad[0] ^= 1;
    try { bp::Transcript t3("innerproduct"); bp::IPP::verify_ipp(ctx, n, t3, G_factors, H_factors, P, Q, G, H, bad, proof.b, proof.L, proof.R); printf("tampered proof accepted\n"); return 1; }
    catch (const bp::VerificationError&) {}
    try { bp::FieldElementVector three(ctx, ints({1, 2, 3})); G.multi_scalar_mul_var_time(three); printf("length mismatch accepted\n"); return 1; }
    catch (const bp::ValueError&) {}
    try { bp::Transcript t4("x"); bp::FieldElementVector three(ctx, ints({1, 2, 3})); bp::IPP::create_ipp(ctx, t4, Q, G_factors, H_factors, G, H, a, three); printf("unequal lengths accepted\n"); return 1; }
    catch (const bp::ArgError&) {}
    // the reference's own generator construction (src/ipp.rs:340-342) through the mirror, then the same proof again and
    // both proofs verified in one batch
    bp::G1Vector Gh = bp::G1Vector::get_generators(ctx, "g", n), Hh = bp::G1Vector::get_generators(ctx, "h", n);
    bp::Bytes Qh = bp::G1Vector::from_msg_hash(ctx, {"Q"}).to_bytes();
    if (Gh.len() != n || Qh.size() != ctx.point_bytes() || Gh.to_bytes() == G.to_bytes()) { printf("get_generators wrong\n"); return 1; }
    bp::Transcript t5("innerproduct");
    bp::InnerProductArgumentProof proof2 = bp::IPP::create_ipp(ctx, t5, Qh, G_factors, H_factors, Gh, Hh, a, b);
    bp::G1Vector pts2(ctx, concat({Gh.to_bytes(), Hh.to_bytes(), Qh}));
    bp::Bytes P2 = pts2.multi_scalar_mul_var_time(sc);
    bp::InnerProductArgumentProof proof3 = proof2;      // a second proof over the same generators: other Q
    bp::Bytes Q3 = bp::G1Vector::from_msg_hash(ctx, {"Q3"}).to_bytes();
    { bp::Transcript t6("innerproduct"); proof3 = bp::IPP::create_ipp(ctx, t6, Q3, G_factors, H_factors, Gh, Hh, a, b); }
    bp::G1Vector pts3(ctx, concat({Gh.to_bytes(), Hh.to_bytes(), Q3}));
    bp::Bytes P3 = pts3.multi_scalar_mul_var_time(sc);
    bp::Bytes weights = concat({scalar(0x9e3779b97f4a7c15ull), scalar(0xc2b2ae3d27d4eb4full)});
    {
        bp::Transcript ta("innerproduct"), tb("innerproduct");
        bp::IPP::verify_batch(ctx, n, G_factors, H_factors, Gh, Hh, {{&ta, &P2, &Qh, &proof2}, {&tb, &P3, &Q3, &proof3}}, weights);
    }
    try {
        bp::Transcript ta("innerproduct"), tb("innerproduct");
        bp::IPP::verify_batch(ctx, n, G_factors, H_factors, Gh, Hh, {{&ta, &P2, &Qh, &proof2}, {&tb, &P2, &Q3, &proof3}}, weights);   // wrong commitment
        printf("bad batch accepted\n"); return 1;
    } catch (const bp::VerificationError&) {}
    if (curve == BP_CURVE_BLS12_381) {
        // R1CS layer through the mirror: two multiplication gates (3 * 5 = 15, 2 * 7 = 14), one committed value v = 15 and the
        // constraint a_O[0] - V_0 = 0 (plus a trivially true one); bp_r1cs_prove / bp_r1cs_verify (prover.rs:323-560, verifier.rs:265-452)
        const bp::Bytes minus_one = {0x00, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0x02, 0xa4, 0xbd, 0x53,
                                     0x05, 0xd8, 0xa1, 0x09, 0x08, 0xd8, 0x39, 0x33, 0x48, 0x7d, 0x9d, 0x29, 0x53, 0xa7, 0xed, 0x73};   // r - 1
        std::vector<bp::Term> terms = {{0, BP_VAR_MUL_OUTPUT, 0, scalar(1)}, {0, BP_VAR_COMMITTED, 0, minus_one},
                                       {1, BP_VAR_MUL_LEFT, 1, scalar(1)}, {1, BP_VAR_MUL_LEFT, 1, minus_one}};
        bp::R1CSPlan plan(ctx, terms, 2, 2, 1);
        bp::Bytes gq = bp::G1Vector::from_msg_hash(ctx, {"g"}).to_bytes(), hq = bp::G1Vector::from_msg_hash(ctx, {"h"}).to_bytes();
        bp::G1Vector GG = bp::G1Vector::get_generators(ctx, "G", 2), HH = bp::G1Vector::get_generators(ctx, "H", 2);
        bp::FieldElementVector aL(ctx, ints({3, 2})), aR(ctx, ints({5, 7})), aO(ctx, ints({15, 14})), vv(ctx, ints({15})), vb(ctx, ints({424242}));
        bp::FieldElementVector sL(ctx, ints({1001, 1002})), sR(ctx, ints({2001, 2002}));
        bp::Bytes V = bp::r1cs::commit(ctx, gq, hq, vv, vb);
        bp::Bytes blindings = ints({11, 12, 13, 14, 15, 16, 17, 18});
        bp::Transcript tp("cpp r1cs");
        bp::r1cs::start_transcript(ctx, tp, V);
        bp::Bytes rproof = bp::r1cs::prove(ctx, tp, plan, GG, HH, gq, hq, aL, aR, aO, &vb, sL, sR, blindings);
        if (rproof.size() != bp_r1cs_proof_bytes(curve, 2)) { printf("wrong r1cs proof size\n"); return 1; }
        {
            bp::Transcript tv("cpp r1cs");
            bp::r1cs::start_transcript(ctx, tv, V);
            bp::r1cs::verify(ctx, tv, plan, GG, HH, gq, hq, V, rproof, scalar(0x5eed5eed));
        }
        {   // library-drawn weight, and the proof through its compressed wire form
            bp::Bytes wire = bp::r1cs::compress_proof(ctx, 2, rproof);
            if (wire.size() >= rproof.size() || bp::r1cs::decompress_proof(ctx, 2, wire) != rproof) { printf("compressed proof round trip failed\n"); return 1; }
            bp::Transcript tv("cpp r1cs");
            bp::r1cs::start_transcript(ctx, tv, V);
            bp::r1cs::verify(ctx, tv, plan, GG, HH, gq, hq, V, bp::r1cs::decompress_proof(ctx, 2, wire));
        }
        try {
            bp::Bytes badp = rproof;
            badp[11 * ctx.point_bytes()] ^= 1;                   // t_x
            bp::Transcript tv("cpp r1cs");
            bp::r1cs::start_transcript(ctx, tv, V);
            bp::r1cs::verify(ctx, tv, plan, GG, HH, gq, hq, V, badp, scalar(0x5eed5eed));
            printf("tampered r1cs proof accepted\n"); return 1;
        } catch (const bp::VerificationError&) {}
        try {                                                    // a witness that breaks the gate: 3 * 5 != 16
            bp::FieldElementVector aObad(ctx, ints({16, 14})), v16(ctx, ints({16}));
            bp::Bytes V16 = bp::r1cs::commit(ctx, gq, hq, v16, vb);
            bp::Transcript tp2("cpp r1cs");
            bp::r1cs::start_transcript(ctx, tp2, V16);
            bp::Bytes p2 = bp::r1cs::prove(ctx, tp2, plan, GG, HH, gq, hq, aL, aR, aObad, &vb, sL, sR, blindings);
            bp::Transcript tv("cpp r1cs");
            bp::r1cs::start_transcript(ctx, tv, V16);
            bp::r1cs::verify(ctx, tv, plan, GG, HH, gq, hq, V16, p2, scalar(77));
            printf("unsatisfied circuit accepted\n"); return 1;
        } catch (const bp::VerificationError&) {}
    }
    {   // compressed generators round-trip; the MSM over two index-range shards held by two contexts equals the single-context MSM
        if (bp::G1Vector::from_compressed(ctx, Gh.to_compressed()).to_bytes() != Gh.to_bytes()) { printf("compressed points differ\n"); return 1; }
        bp::Context ctx2(curve, 0);
        const size_t tot = pts2.len(), half = tot / 2, pb = ctx.point_bytes();
        bp::Bytes pb_all = pts2.to_bytes(), sc_all = sc.to_bytes();
        bp::G1Vector p_lo(ctx, bp::Bytes(pb_all.begin(), pb_all.begin() + half * pb)), p_hi(ctx2, bp::Bytes(pb_all.begin() + half * pb, pb_all.end()));
        bp::FieldElementVector s_lo(ctx, bp::Bytes(sc_all.begin(), sc_all.begin() + half * 32)), s_hi(ctx2, bp::Bytes(sc_all.begin() + half * 32, sc_all.end()));
        if (bp::G1Vector::multi_scalar_mul_var_time_sharded({&ctx, &ctx2}, {&p_lo, &p_hi}, {&s_lo, &s_hi}) != P2) { printf("sharded MSM differs\n"); return 1; }
        // round 3: a window-multiples table changes no byte of an MSM ...
        pts2.precompute(7);
        if (pts2.multi_scalar_mul_var_time(sc) != P2) { printf("MSM over a table differs\n"); return 1; }
        pts2.drop_table();
        // ... and create_ipp with the generators sharded over two contexts is create_ipp
        const size_t ng = Gh.len(), cut = ng / 2 ? ng / 2 : 1;
        if (ng >= 2) {
            bp::Bytes gb = Gh.to_bytes(), hb = Hh.to_bytes(), gfb = G_factors.to_bytes(), hfb = H_factors.to_bytes();
            auto pt = [&](const bp::Bytes& v, size_t lo, size_t hi, size_t w) { return bp::Bytes(v.begin() + lo * w, v.begin() + hi * w); };
            bp::G1Vector g0(ctx, pt(gb, 0, cut, pb)), g1(ctx2, pt(gb, cut, ng, pb)), h0(ctx, pt(hb, 0, cut, pb)), h1(ctx2, pt(hb, cut, ng, pb));
            bp::FieldElementVector gf0(ctx, pt(gfb, 0, cut, 32)), gf1(ctx2, pt(gfb, cut, ng, 32)), hf0(ctx, pt(hfb, 0, cut, 32)), hf1(ctx2, pt(hfb, cut, ng, 32));
            bp::Transcript ts("innerproduct");
            bp::InnerProductArgumentProof ps = bp::IPP::create_ipp_sharded({&ctx, &ctx2}, ts, Qh, {&gf0, &gf1}, {&hf0, &hf1}, {&g0, &g1}, {&h0, &h1}, a.to_bytes(), b.to_bytes());
            if (ps.L != proof2.L || ps.R != proof2.R || ps.a != proof2.a || ps.b != proof2.b) { printf("sharded IPP differs\n"); return 1; }
        }
    }
    printf("cpp gpu ok curve=%d a=%s\n", curve, hex(proof.a).c_str());
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "gpu")) return gpu_mode(0) + gpu_mode(1);
    return cpu_mode();
}
