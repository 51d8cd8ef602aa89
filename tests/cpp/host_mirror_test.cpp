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
    printf("cpp gpu ok curve=%d a=%s\n", curve, hex(proof.a).c_str());
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "gpu")) return gpu_mode(0) + gpu_mode(1);
    return cpu_mode();
}
