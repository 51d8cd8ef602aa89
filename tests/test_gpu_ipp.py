"""GPU parity tests for the inner-product argument: FieldElementVector kernels, the round/fold kernels and the whole
create_ipp / verify_ipp flow through the C ABI, against the golden vectors (Python-int) and the C oracle."""
import pytest

import __graft_entry__ as G
import _oracle as O

pytestmark = pytest.mark.gpu
CURVES = ["bls12_381", "bn254"]


def hx(s):
    return bytes.fromhex(s)


@pytest.fixture(scope="module")
def bp():
    return G.load_package()


@pytest.fixture(scope="module")
def ctxs(bp):
    c = {name: bp.Context(cid, 0) for name, cid in bp.CURVE_IDS.items()}
    yield c
    for x in c.values():
        x.close()


def ints(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


@pytest.mark.parametrize("name", CURVES)
def test_fr_vector_kernels(bp, ctxs, name):
    ctx = ctxs[name]
    r = ctx.r
    for n in (1, 2, 255, 256, 257, 5000, 70001):
        ab, bb = O.random_scalars(ctx.curve, 300 + n, n), O.random_scalars(ctx.curve, 400 + n, n)
        a, b = bp.FieldElementVector.from_bytes(ctx, ab, n), bp.FieldElementVector.from_bytes(ctx, bb, n)
        assert a.inner_product(b) == O.fr_inner(ctx.curve, ab, bb, n)
        if n > 4:
            assert a.inner_product(b, aoff=1, boff=3, n=n - 3) == O.fr_inner(ctx.curve, ab[32:], bb[96:], n - 3)
        ai, bi = ints(ab), ints(bb)
        if n <= 5000:
            assert ints(a.hadamard_product(b).to_bytes()) == [x * y % r for x, y in zip(ai, bi)]
            s = bi[0]
            assert ints(a.scaled_by(bb[:32]).to_bytes()) == [x * s % r for x in ai]
            assert ints(bp.FieldElementVector.new_vandermonde_vector(ctx, bb[:32], n).to_bytes()) == [pow(s, i, r) for i in range(n)]
    a3 = bp.FieldElementVector.from_ints(ctx, [1, 2, 3])
    a4 = bp.FieldElementVector.from_ints(ctx, [1, 2, 3, 4])
    with pytest.raises(bp.ValueError_):
        a3.inner_product(a4)
    with pytest.raises(bp.ValueError_):
        a3.hadamard_product(a4)
    assert bp.FieldElementVector.from_ints(ctx, [0, 1, r - 1]).inner_product(bp.FieldElementVector.from_ints(ctx, [5, r - 1, r - 1])) == \
        ((r - 1 + (r - 1) * (r - 1)) % r).to_bytes(32, "little")


@pytest.mark.parametrize("name", CURVES)
def test_vector_poly_kernels(bp, ctxs, name):
    """src/utils/vector_poly.rs: VecPoly1/3 inner products and evals against Python-int arithmetic."""
    ctx = ctxs[name]
    r = ctx.r
    for n in (1, 300, 4099):
        vecs = [ints(O.random_scalars(ctx.curve, 900 + 10 * n + j, n)) for j in range(8)]
        dev = [bp.FieldElementVector.from_ints(ctx, v) for v in vecs]
        zero = bp.FieldElementVector.new(ctx, n)
        l, rr = vecs[0:4], vecs[4:8]
        lhs = bp.VecPoly3(zero, dev[1], dev[2], dev[3])
        rhs = bp.VecPoly3(dev[4], dev[5], zero, dev[7])
        ip = lambda a, b: sum(x * y for x, y in zip(a, b)) % r
        want = [ip(l[1], rr[0]), (ip(l[1], rr[1]) + ip(l[2], rr[0])) % r, (ip(l[2], rr[1]) + ip(l[3], rr[0])) % r,
                (ip(l[1], rr[3]) + ip(l[3], rr[1])) % r, ip(l[2], rr[3]), ip(l[3], rr[3])]
        got = [int.from_bytes(t, "little") for t in bp.VecPoly3.special_inner_product(lhs, rhs)]
        assert got == want
        x = vecs[0][0]
        xb = x.to_bytes(32, "little")
        p3 = bp.VecPoly3(dev[0], dev[1], dev[2], dev[3])
        assert ints(p3.eval(xb).to_bytes()) == [(a + x * (b + x * (c + x * d))) % r for a, b, c, d in zip(*vecs[0:4])]
        p1a, p1b = bp.VecPoly1(dev[0], dev[1]), bp.VecPoly1(dev[4], dev[5])
        assert ints(p1a.eval(xb).to_bytes()) == [(a + b * x) % r for a, b in zip(vecs[0], vecs[1])]
        t0, t1, t2 = (int.from_bytes(t, "little") for t in p1a.inner_product(p1b))
        assert (t0, t1, t2) == (ip(vecs[0], vecs[4]), (ip(vecs[0], vecs[5]) + ip(vecs[1], vecs[4])) % r, ip(vecs[1], vecs[5]))
    with pytest.raises(bp.ValueError_):
        bp.VecPoly1(dev[0], bp.FieldElementVector.from_ints(ctx, [1, 2])).eval(xb)


@pytest.mark.parametrize("name", CURVES)
def test_r1cs_vector_pipeline(bp, ctxs, name):
    """src/r1cs/prover.rs:465-563 and src/r1cs/verifier.rs:342-390 restated with Python ints."""
    ctx = ctxs[name]
    cid, r = ctx.curve, ctx.r
    n, n1, padded_n, lg = 11, 7, 16, 4
    iv = lambda seed, k=n: ints(O.random_scalars(cid, seed, k))
    aL, aR, aO, sL, sR, wL, wR, wO = (iv(1300 + j) for j in range(8))
    y, x, u, a, b = iv(1400, 5)
    yb, xb, ub, ab, bb = (v.to_bytes(32, "little") for v in (y, x, u, a, b))
    dev = lambda v: bp.FieldElementVector.from_ints(ctx, v)
    d = [dev(v) for v in (aL, aR, aO, sL, sR, wL, wR, wO)]
    yi = pow(y, -1, r)
    lp, rp = bp.r1cs_prover_polys(ctx, *d, yb)
    got = [ints(v.to_bytes()) for v in (lp.v[1], lp.v[2], lp.v[3], rp.v[0], rp.v[1], rp.v[3])]
    want = [[(aL[i] + pow(yi, i, r) * wR[i]) % r for i in range(n)], aO, sL, [(wO[i] - pow(y, i, r)) % r for i in range(n)],
            [(pow(y, i, r) * aR[i] + wL[i]) % r for i in range(n)], [pow(y, i, r) * sR[i] % r for i in range(n)]]
    assert got == want
    assert ints(lp.v[0].to_bytes()) == [0] * n and ints(rp.v[2].to_bytes()) == [0] * n
    # t(x) coefficients and evaluations through the vector-poly kernels
    l_eval, r_eval = lp.eval(xb), rp.eval(xb)
    lx = [(x * (want[0][i] + x * (want[1][i] + x * want[2][i]))) % r for i in range(n)]
    rx = [(want[3][i] + x * (want[4][i] + x * x * want[5][i])) % r for i in range(n)]
    assert ints(l_eval.to_bytes()) == lx and ints(r_eval.to_bytes()) == rx
    lv, rv, gf, hf = bp.r1cs_ipp_inputs(ctx, l_eval, r_eval, yb, ub, n1, padded_n)
    assert ints(lv.to_bytes()) == lx + [0] * (padded_n - n)
    assert ints(rv.to_bytes()) == rx + [(-pow(y, i, r)) % r for i in range(n, padded_n)]
    gfw = [1] * n1 + [u] * (padded_n - n1)
    assert ints(gf.to_bytes()) == gfw
    assert ints(hf.to_bytes()) == [pow(yi, i, r) * gfw[i] % r for i in range(padded_n)]
    # verifier scalars: any lg points serve as L, R for the transcript replay
    gen = O.generator(cid)
    L = b"".join(O.g1_mul(cid, (3 + j).to_bytes(32, "little"), gen) for j in range(lg))
    R = b"".join(O.g1_mul(cid, (30 + j).to_bytes(32, "little"), gen) for j in range(lg))
    us, uis, gs, hs = bp.r1cs_verifier_scalars(ctx, bp.Transcript(b"R1CSTest"), L, R, padded_n, n1, d[5], d[6], d[7], yi.to_bytes(32, "little"), xb, ub, ab, bb)
    rc, (us_w, uis_w, s_w) = O.ipp_verification_scalars(cid, O.Transcript(b"R1CSTest"), L, R, lg, padded_n)
    assert rc == 0 and us == us_w and uis == uis_w
    s = ints(s_w)
    pad0 = lambda v: v + [0] * (padded_n - n)
    wLp, wRp, wOp = pad0(wL), pad0(wR), pad0(wO)
    g_w = [gfw[i] * (x * pow(yi, i, r) * wRp[i] - a * s[i]) % r for i in range(padded_n)]
    h_w = [gfw[i] * (pow(yi, i, r) * (x * wLp[i] + wOp[i] - b * s[padded_n - 1 - i]) - 1) % r for i in range(padded_n)]
    assert ints(gs.to_bytes()) == g_w and ints(hs.to_bytes()) == h_w
    with pytest.raises(bp.VerificationError):
        bp.r1cs_verifier_scalars(ctx, bp.Transcript(b"R1CSTest"), L, R, 2 * padded_n, n1, d[5], d[6], d[7], yb, xb, ub, ab, bb)


def load_case(bp, ctx, c):
    n = c["n"]
    cat = lambda k: b"".join(hx(x) for x in c[k])
    Gv = bp.G1Vector.from_bytes(ctx, cat("G"), n)
    Hv = bp.G1Vector.from_bytes(ctx, cat("H"), n)
    Gf = bp.FieldElementVector.from_bytes(ctx, cat("G_factors"), n)
    Hf = bp.FieldElementVector.from_bytes(ctx, cat("H_factors"), n)
    a = bp.FieldElementVector.from_bytes(ctx, cat("a"), n)
    b = bp.FieldElementVector.from_bytes(ctx, cat("b"), n)
    return n, Gv, Hv, Gf, Hf, a, b


@pytest.fixture(params=[0, 1], ids=["msm-over-originals", "fold-generators"])
def prover_mode(request, ctxs):
    """Both prover strategies (include/bpmsm.h: bp_ctx_set_ipp_fold_generators) must give bit-identical proofs."""
    for c in ctxs.values():
        c.set_ipp_fold_generators(request.param)
    yield request.param
    for c in ctxs.values():
        c.set_ipp_fold_generators(0)


@pytest.mark.parametrize("name", CURVES)
def test_ipp_golden_create_and_verify(bp, ctxs, golden, name, prover_mode):
    """Includes the reference's own two unit tests (src/ipp.rs:325-390 n=4, :393-489 n=8 padded) as fixtures."""
    ctx = ctxs[name]
    for c in golden("ipp")[name]:
        n, Gv, Hv, Gf, Hf, a, b = load_case(bp, ctx, c)
        Q, P = hx(c["Q"]), hx(c["P"])
        tr = bp.Transcript(b"innerproduct")
        proof = bp.IPP.create_ipp(ctx, tr, Q, Gf, Hf, Gv, Hv, a, b)
        assert proof.L == b"".join(hx(x) for x in c["L"]), c["name"]
        assert proof.R == b"".join(hx(x) for x in c["R"]), c["name"]
        assert proof.a == hx(c["a_out"]) and proof.b == hx(c["b_out"])
        assert tr.challenge_bytes(b"after", 32).hex() == c["transcript_after"]
        # inputs are borrowed and cloned (src/ipp.rs:57-60): the caller's vectors are untouched
        assert Gv.to_bytes() == b"".join(hx(x) for x in c["G"]) and a.to_bytes() == b"".join(hx(x) for x in c["a"])
        bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
        bad_a = ((int.from_bytes(proof.a, "little") + 1) % ctx.r).to_bytes(32, "little")
        with pytest.raises(bp.VerificationError):
            bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, bad_a, proof.b, proof.L, proof.R)
        if proof.L:
            badL = proof.R[:ctx.point_bytes] + proof.L[ctx.point_bytes:]
            with pytest.raises(bp.VerificationError):
                bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, badL, proof.R)
        with pytest.raises(bp.VerificationError):       # wrong n (src/ipp.rs:274-276)
            bp.IPP.verify_ipp(ctx, 2 * n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)


@pytest.mark.parametrize("name", CURVES)
def test_ipp_argument_checks(bp, ctxs, name):
    """create_ipp's assert!/assert_eq! (src/ipp.rs:48-55) surface as ArgError."""
    ctx = ctxs[name]
    g = O.generator(ctx.curve)
    G3, G4 = bp.G1Vector.from_bytes(ctx, g * 3, 3), bp.G1Vector.from_bytes(ctx, g * 4, 4)
    f3, f4 = bp.FieldElementVector.from_ints(ctx, [1, 2, 3]), bp.FieldElementVector.from_ints(ctx, [1, 2, 3, 4])
    with pytest.raises(bp.ArgError):
        bp.IPP.create_ipp(ctx, bp.Transcript(b"x"), g, f3, f3, G3, G3, f3, f3)       # not a power of two
    with pytest.raises(bp.ArgError):
        bp.IPP.create_ipp(ctx, bp.Transcript(b"x"), g, f4, f4, G4, G4, f4, f3)       # unequal lengths


def make_instance(bp, ctx, n, seed, unit_gf=True):
    cid = ctx.curve
    gk, hk = O.random_scalars(cid, seed, n), O.random_scalars(cid, seed + 1, n)
    Gv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, gk, n))
    Hv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, hk, n))
    Q = O.g1_mul(cid, O.random_scalars(cid, seed + 2, 1), O.generator(cid))
    ab, bb = O.random_scalars(cid, seed + 3, n), O.random_scalars(cid, seed + 4, n)
    y_inv = O.random_scalars(cid, seed + 5, 1)
    Hf = bp.FieldElementVector.new_vandermonde_vector(ctx, y_inv, n)                 # as in the reference's tests
    Gf = bp.FieldElementVector.from_ints(ctx, [1] * n) if unit_gf else bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, seed + 6, n), n)
    a, b = bp.FieldElementVector.from_bytes(ctx, ab, n), bp.FieldElementVector.from_bytes(ctx, bb, n)
    return Gv, Hv, Q, Gf, Hf, a, b


def commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b):
    """P = <a.Gf, G> + <b.Hf, H> + <a,b> Q  (the reference's test construction, src/ipp.rs:353-372)."""
    n = len(Gv)
    pts = bp.G1Vector.from_bytes(ctx, Gv.to_bytes() + Hv.to_bytes() + Q, 2 * n + 1)
    sc = bp.FieldElementVector.from_bytes(ctx, a.hadamard_product(Gf).to_bytes() + b.hadamard_product(Hf).to_bytes() + a.inner_product(b), 2 * n + 1)
    return pts.multi_scalar_mul_var_time(sc)


# (n = 16 .. 4096: rounds on the single-launch path over the state's digit multiples; above 255 two blocks per window)
@pytest.mark.parametrize("name,n,unit_gf", [("bls12_381", 64, True), ("bls12_381", 32, False), ("bn254", 64, True), ("bn254", 256, False),
                                            ("bls12_381", 8, False), ("bls12_381", 16, False), ("bls12_381", 512, False), ("bls12_381", 2048, True), ("bn254", 1024, False)])
def test_ipp_random_vs_oracle(bp, ctxs, name, n, unit_gf, prover_mode):
    """BASELINE config 1 shape (n = 64): GPU proof bit-for-bit equal to the oracle's, accepted by both verifiers."""
    ctx = ctxs[name]
    cid = ctx.curve
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, ctx, n, 7000 + n, unit_gf)
    proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    rc, want = O.ipp_create(cid, O.Transcript(b"innerproduct"), Q, Gf.to_bytes(), Hf.to_bytes(), Gv.to_bytes(), Hv.to_bytes(), a.to_bytes(),
                            b.to_bytes(), n)
    assert rc == 0
    assert (proof.L, proof.R, proof.a, proof.b) == want
    P = commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b)
    bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
    assert O.ipp_verify(cid, O.Transcript(b"innerproduct"), n, Gf.to_bytes(), Hf.to_bytes(), P, Q, Gv.to_bytes(), Hv.to_bytes(), proof.a, proof.b,
                        proof.L, proof.R, proof.lg_n) == 0


@pytest.mark.parametrize("name,n,at,unit_gf", [("bls12_381", 64, 16, True), ("bls12_381", 256, 16, False), ("bn254", 128, 32, False), ("bn254", 512, 16, True),
                                               ("bls12_381", 1024, 512, False), ("bn254", 64, 32, False)])
def test_ipp_generator_compaction_vs_oracle(bp, ctxs, name, n, at, unit_gf):
    """Generator compaction (bp_compact.cuh; /root/reference src/ipp.rs:181-188 done once instead of never): with BP_TUNE_COMPACT_AT the
    prover materialises the folded generators when the live length reaches `at` (n / at = 2 .. 16 originals per output) and finishes over
    their digit multiples.  Proof bytes = the oracle's = the bytes without compaction; the round API crosses the switch as well."""
    ctx = ctxs[name]
    cid = ctx.curve
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, ctx, n, 41000 + n + at, unit_gf)
    rc, want = O.ipp_create(cid, O.Transcript(b"innerproduct"), Q, Gf.to_bytes(), Hf.to_bytes(), Gv.to_bytes(), Hv.to_bytes(), a.to_bytes(), b.to_bytes(), n)
    assert rc == 0
    try:
        ctx.set_tuning(bp.TUNE_COMPACT_AT, 1)            # never
        plain = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
        ctx.set_tuning(bp.TUNE_COMPACT_AT, at)
        proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
        ctx.set_tuning(bp.TUNE_GLV, 1)                    # without the GLV split of the scalars (the only form BN254 has)
        noglv = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
        ctx.set_tuning(bp.TUNE_GLV, 0)
        st = bp.IPPState(ctx, Gv, Hv, Q, Gf, Hf, a, b)    # the same through bp_ipp_round / bp_ipp_fold with the caller's transcript
        tr = bp.Transcript(b"innerproduct")
        tr.append_message(b"dom-sep", b"ipp v1")
        tr.append_u64(b"n", n)
        Ls, Rs = b"", b""
        while len(st) > 1:
            L, R = st.round()
            tr.commit_point(cid, b"L", L); tr.commit_point(cid, b"R", R)
            u = tr.challenge_scalar(cid, b"u")
            st.fold(u, bp.fr_inverse(cid, u))
            Ls += L; Rs += R
        fa, fb = st.finish()
    finally:
        ctx.set_tuning(bp.TUNE_COMPACT_AT, 0)
        ctx.set_tuning(bp.TUNE_GLV, 0)
    assert (plain.L, plain.R, plain.a, plain.b) == want
    assert (proof.L, proof.R, proof.a, proof.b) == want
    assert (noglv.L, noglv.R, noglv.a, noglv.b) == want
    assert (Ls, Rs, fa, fb) == want
    # ... and over the vectors' COMPACTION TABLES (bp_g1vec_precompute with a width that divides 64: rows 2^(64 k) P, Horner chain of 60 doublings)
    try:
        Gv.precompute(16); Hv.precompute(16)
        ctx.set_tuning(bp.TUNE_COMPACT_AT, at)
        tabled = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
        Hv.precompute(8)                                 # tables of different widths: both still have a compaction table
        mixed = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
        Hv.precompute(12)                                # 12 does not divide 64: H has no compaction table, the prover builds the multiples itself
        none = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    finally:
        ctx.set_tuning(bp.TUNE_COMPACT_AT, 0)
        Gv.drop_table(); Hv.drop_table()
    for pr in (tabled, mixed, none):
        assert (pr.L, pr.R, pr.a, pr.b) == want
    P = commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b)
    bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)


@pytest.mark.parametrize("name,n", [("bls12_381", 4096), ("bn254", 4096), ("bls12_381", 8192), ("bn254", 8192), ("bn254", 16384)])
def test_ipp_single_launch_boundary_and_compaction_sizes_vs_oracle(bp, ctxs, name, n):
    """kSmallDigitMax = 8193 makes n = 4096 the last proof whose every round is a single launch over digit multiples
    (BASELINE config 5's size) and n = 8192 the first that starts on the bucket pipeline -- and, since round 4, the first that COMPACTS
    its generators (after one round; n = 16384 after two).  Proof bytes = the (threaded) C oracle's, with and without window /
    compaction tables, compaction on and off."""
    import os
    ctx = ctxs[name]
    cid = ctx.curve
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, ctx, n, 51000 + n, unit_gf=False)
    O.set_threads(min(16, os.cpu_count() or 1))
    try:
        rc, want = O.ipp_create(cid, O.Transcript(b"innerproduct"), Q, Gf.to_bytes(), Hf.to_bytes(), Gv.to_bytes(), Hv.to_bytes(), a.to_bytes(), b.to_bytes(), n)
    finally:
        O.set_threads(1)
    assert rc == 0
    proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    assert (proof.L, proof.R, proof.a, proof.b) == want
    try:
        ctx.set_tuning(bp.TUNE_COMPACT_AT, 1)
        off = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
        ctx.set_tuning(bp.TUNE_COMPACT_AT, 0)
        Gv.precompute(16); Hv.precompute(16)
        tabled = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    finally:
        ctx.set_tuning(bp.TUNE_COMPACT_AT, 0)
        Gv.drop_table(); Hv.drop_table()
    assert (off.L, off.R, off.a, off.b) == want
    assert (tabled.L, tabled.R, tabled.a, tabled.b) == want
    P = commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b)
    bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)


@pytest.mark.parametrize("name,n", [("bls12_381", 32), ("bn254", 64), ("bls12_381", 1024)])
def test_ipp_degenerate_generators_vs_oracle(bp, ctxs, name, n):
    """Generators with structure the digit-multiples table and the four-lane tree must survive: identity points, a repeated point,
    a point and its negative, a generator equal to Q -- and scalar vectors with zeros and repeats.  Proof bytes = the oracle's."""
    ctx = ctxs[name]
    cid = ctx.curve
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, ctx, n, 9100 + n, unit_gf=False)
    pb = ctx.point_bytes
    g, h = bytearray(Gv.to_bytes()), bytearray(Hv.to_bytes())
    P0 = bytes(g[0:pb])
    negP0 = O.g1_mul(cid, (ctx.r - 1).to_bytes(32, "little"), P0)
    g[pb * 1:pb * 2] = bytes(pb)                     # identity
    g[pb * 2:pb * 3] = P0                            # repeated
    g[pb * 3:pb * 4] = negP0                         # P and -P in one vector
    h[pb * 0:pb * 1] = P0                            # shared between G and H
    h[pb * 5:pb * 6] = bytes(pb)
    h[pb * 6:pb * 7] = Q                             # a generator equal to Q
    Gv = bp.G1Vector.from_bytes(ctx, bytes(g), n)
    Hv = bp.G1Vector.from_bytes(ctx, bytes(h), n)
    ab, bb = bytearray(a.to_bytes()), bytearray(b.to_bytes())
    ab[32 * 4:32 * 5] = bytes(32); ab[32 * 6:32 * 7] = ab[0:32]
    bb[32 * 0:32 * 1] = bytes(32); bb[32 * 9:32 * 10] = bb[32:64]
    a, b = bp.FieldElementVector.from_bytes(ctx, bytes(ab), n), bp.FieldElementVector.from_bytes(ctx, bytes(bb), n)
    proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    rc, want = O.ipp_create(cid, O.Transcript(b"innerproduct"), Q, Gf.to_bytes(), Hf.to_bytes(), Gv.to_bytes(), Hv.to_bytes(), a.to_bytes(), b.to_bytes(), n)
    assert rc == 0 and (proof.L, proof.R, proof.a, proof.b) == want
    P = commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b)
    bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)


@pytest.mark.parametrize("name,n,c", [("bls12_381", 256, 0), ("bls12_381", 1024, 11), ("bn254", 512, 16), ("bn254", 2048, 9)])
def test_ipp_with_precomputed_generators(bp, ctxs, name, n, c):
    """bp_g1vec_precompute on G and H: every round's L / R is a merged-window MSM over [G | H | Q] rows -- the proof must be the
    oracle's byte for byte (and the one produced without tables), mismatched table widths fall back to the plain pipeline."""
    ctx = ctxs[name]
    cid = ctx.curve
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, ctx, n, 21000 + n, unit_gf=False)
    rc, want = O.ipp_create(cid, O.Transcript(b"innerproduct"), Q, Gf.to_bytes(), Hf.to_bytes(), Gv.to_bytes(), Hv.to_bytes(), a.to_bytes(),
                            b.to_bytes(), n)
    assert rc == 0
    plain = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    assert (plain.L, plain.R, plain.a, plain.b) == want
    Gv.precompute(c)
    Hv.precompute(c)
    assert Gv.table_info()[0] == Hv.table_info()[0] != 0
    tabled = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    assert (tabled.L, tabled.R, tabled.a, tabled.b) == want
    Hv.precompute(12 if Gv.table_info()[0] != 12 else 10)             # widths differ: no table for the concatenation
    mixed = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    assert (mixed.L, mixed.R, mixed.a, mixed.b) == want
    P = commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b)
    bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, tabled.a, tabled.b, tabled.L, tabled.R)


@pytest.mark.parametrize("name,n,cuts,tables", [("bls12_381", 256, (0, 100, 256), False), ("bls12_381", 1024, (0, 300, 301, 1024), True),
                                                   ("bn254", 512, (0, 256, 512), True), ("bn254", 64, (0, 1, 33, 64), False), ("bls12_381", 2, (0, 1, 2), False)])
def test_ipp_sharded_over_contexts(bp, ctxs, name, n, cuts, tables):
    """SURVEY 8e, second sentence: the generators sharded by index range over several contexts (bp_ipp_create_multi; here the
    contexts share one device, on a multi-GPU box they would not).  Ragged shards, with and without window tables: the proof must be
    the single-context proof and the oracle's, byte for byte."""
    main = ctxs[name]
    cid = main.curve
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, main, n, 31000 + n, unit_gf=False)
    rc, want = O.ipp_create(cid, O.Transcript(b"innerproduct"), Q, Gf.to_bytes(), Hf.to_bytes(), Gv.to_bytes(), Hv.to_bytes(), a.to_bytes(),
                            b.to_bytes(), n)
    assert rc == 0
    single = bp.IPP.create_ipp(main, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    assert (single.L, single.R, single.a, single.b) == want
    shards = [bp.Context(cid, 0) for _ in range(len(cuts) - 1)]
    pb = main.point_bytes
    gb, hb, gfb, hfb = Gv.to_bytes(), Hv.to_bytes(), Gf.to_bytes(), Hf.to_bytes()
    Gs, Hs, Gfs, Hfs = [], [], [], []
    for c, lo, hi in zip(shards, cuts[:-1], cuts[1:]):
        Gs.append(bp.G1Vector.from_bytes(c, gb[lo * pb:hi * pb], hi - lo))
        Hs.append(bp.G1Vector.from_bytes(c, hb[lo * pb:hi * pb], hi - lo))
        Gfs.append(bp.FieldElementVector.from_bytes(c, gfb[lo * 32:hi * 32], hi - lo))
        Hfs.append(bp.FieldElementVector.from_bytes(c, hfb[lo * 32:hi * 32], hi - lo))
        if tables:
            Gs[-1].precompute(16)
            Hs[-1].precompute(16)
    proof = bp.IPP.create_ipp_multi(shards, bp.Transcript(b"innerproduct"), Q, Gfs, Hfs, Gs, Hs, a.to_bytes(), b.to_bytes())
    assert (proof.L, proof.R, proof.a, proof.b) == want
    P = commitment_P(bp, main, Gv, Hv, Q, Gf, Hf, a, b)
    bp.IPP.verify_ipp(main, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
    # argument checks: sizes that do not add up to a power of two, a context used twice, a non-canonical scalar
    rest = cuts[-2]
    if rest & (rest - 1):
        with pytest.raises(bp.ArgError):
            bp.IPP.create_ipp_multi(shards[:-1], bp.Transcript(b"innerproduct"), Q, Gfs[:-1], Hfs[:-1], Gs[:-1], Hs[:-1], a.to_bytes(), b.to_bytes())
    if len(shards) > 1:
        with pytest.raises(bp.ArgError):
            bp.IPP.create_ipp_multi([shards[0]] * len(shards), bp.Transcript(b"innerproduct"), Q, Gfs, Hfs, Gs, Hs, a.to_bytes(), b.to_bytes())
    with pytest.raises(bp.ArgError):
        bp.IPP.create_ipp_multi(shards, bp.Transcript(b"innerproduct"), Q, Gfs, Hfs, Gs, Hs, b"\xff" * 32 + a.to_bytes()[32:], b.to_bytes())
    for c in shards:
        c.close()


@pytest.mark.parametrize("name", CURVES)
def test_ipp_round_api_with_external_transcript(bp, ctxs, name, prover_mode):
    """The low-level state API driven by a transcript the caller owns (here: the oracle's), as a Rust host would."""
    ctx = ctxs[name]
    cid = ctx.curve
    n = 16
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, ctx, n, 9100)
    st = bp.IPPState(ctx, Gv, Hv, Q, Gf, Hf, a, b)
    tr = O.Transcript(b"innerproduct")
    tr.append_message(b"dom-sep", b"ipp v1")
    tr.append_message(b"n", n.to_bytes(8, "little"))
    Ls, Rs = b"", b""
    while len(st) > 1:
        L, R = st.round()
        tr.commit_point(cid, b"L", L)
        tr.commit_point(cid, b"R", R)
        u = tr.challenge_scalar(cid, b"u")
        st.fold(u, bp.fr_inverse(cid, u))
        Ls, Rs = Ls + L, Rs + R
    a0, b0 = st.finish()
    rc, want = O.ipp_create(cid, O.Transcript(b"innerproduct"), Q, Gf.to_bytes(), Hf.to_bytes(), Gv.to_bytes(), Hv.to_bytes(), a.to_bytes(),
                            b.to_bytes(), n)
    assert (Ls, Rs, a0, b0) == want


def test_ipp_bn254_n4096_config5(bp, ctxs):
    """BASELINE config 5: BN254 IPP at n = 2^12 -- proof bytes equal the oracle's; verify on the GPU, cross-verify with the oracle."""
    ctx = ctxs["bn254"]
    cid = ctx.curve
    n = 4096
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, ctx, n, 12000)
    proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    assert proof.lg_n == 12
    rc, want = O.ipp_create(cid, O.Transcript(b"innerproduct"), Q, Gf.to_bytes(), Hf.to_bytes(), Gv.to_bytes(), Hv.to_bytes(), a.to_bytes(), b.to_bytes(), n)
    assert rc == 0 and (proof.L, proof.R, proof.a, proof.b) == want          # bit-exact at config 5's size
    P = commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b)
    bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
    assert O.ipp_verify(cid, O.Transcript(b"innerproduct"), n, Gf.to_bytes(), Hf.to_bytes(), P, Q, Gv.to_bytes(), Hv.to_bytes(), proof.a, proof.b,
                        proof.L, proof.R, 12) == 0
    bad = bytearray(proof.b); bad[0] ^= 1
    with pytest.raises(bp.VerificationError):
        bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, bytes(bad), proof.L, proof.R)


@pytest.mark.parametrize("name,n,unit_gf,m", [("bls12_381", 64, True, 9), ("bls12_381", 1, True, 3), ("bn254", 32, False, 5),
                                               ("bls12_381", 2, False, 1)])
def test_ipp_verify_batch(bp, ctxs, name, n, unit_gf, m):
    """bp_ipp_verify_batch: m proofs over the same generators pass together iff each passes alone."""
    ctx = ctxs[name]
    cid = ctx.curve
    Gv, Hv, _, Gf, Hf, _, _ = make_instance(bp, ctx, n, 9100 + n, unit_gf)
    items = []
    for j in range(m):
        Q = O.g1_mul(cid, O.random_scalars(cid, 9200 + j, 1), O.generator(cid))
        a = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, 9300 + j, n), n)
        b = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, 9400 + j, n), n)
        label = b"proof-%d" % j
        proof = bp.IPP.create_ipp(ctx, bp.Transcript(label), Q, Gf, Hf, Gv, Hv, a, b)
        P = commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b)
        bp.IPP.verify_ipp(ctx, n, bp.Transcript(label), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
        items.append((label, P, Q, proof))

    def refs(mutate=None):
        out = []
        for j, (label, P, Q, pr) in enumerate(items):
            f = {"P": P, "Q": Q, "a": pr.a, "b": pr.b, "L": pr.L, "R": pr.R}
            if mutate and mutate[0] == j:
                f[mutate[1]] = mutate[2](f[mutate[1]])
            out.append((bp.Transcript(label), f["P"], f["Q"], f["a"], f["b"], f["L"], f["R"]))
        return out

    good = refs()
    bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, good)                                 # fresh random 128-bit weights
    # every transcript ends where the single-proof verifier leaves it
    single = bp.Transcript(items[0][0])
    bp.IPP.verify_ipp(ctx, n, single, Gf, Hf, items[0][1], items[0][2], Gv, Hv, items[0][3].a, items[0][3].b, items[0][3].L, items[0][3].R)
    assert good[0][0].challenge_bytes(b"after", 32) == single.challenge_bytes(b"after", 32)
    fixed = b"".join(O.random_scalars(cid, 9500 + j, 1) for j in range(m))            # full-width weights work as well
    bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, refs(), weights=fixed)
    bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, [])                                    # nothing to check

    def bump(x):                                                                       # scalar + 1 mod r
        return ((int.from_bytes(x, "little") + 1) % ctx.r).to_bytes(32, "little")

    def swap_first(pts):                                                               # replace the first point by the generator
        return O.generator(cid) + pts[ctx.point_bytes:]

    bad_cases = [(m - 1, "a", bump), (0, "b", bump), (m // 2, "P", swap_first), (m // 2, "Q", swap_first)]
    if n > 1:
        bad_cases += [(0, "L", swap_first), (m - 1, "R", swap_first)]
    for mut in bad_cases:
        with pytest.raises(bp.VerificationError):
            bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, refs(mut))
    with pytest.raises(bp.VerificationError):                                          # wrong n for these proofs
        bp.IPP.verify_batch(ctx, 2 * n, Gf, Hf, Gv, Hv, refs())


def _random_constraint_system(rng, r, nq, n, m, hub=None):
    """nq linear combinations of 0..6 terms each over MultiplierLeft/Right/Output(i < n), Committed(i < m) and One();
    `hub` = a variable that appears in most constraints (a destination with hundreds of terms)."""
    cons = []
    for q in range(nq):
        terms = []
        for _ in range(rng.randrange(7)):
            kind = rng.choice([0, 0, 1, 1, 2, 3, 4])
            idx = rng.randrange(n) if kind <= 2 else (rng.randrange(m) if kind == 3 else 0)
            coeff = rng.choice([1, r - 1, 2, rng.randrange(r), rng.randrange(r), 0])
            terms.append((kind, idx, coeff))
        if hub is not None and rng.random() < 0.8:
            terms.append((hub[0], hub[1], rng.randrange(r)))
        cons.append(terms)
    return cons


@pytest.mark.parametrize("name", CURVES)
def test_glv_split_of_scalars(bp, ctxs, name):
    """bp_fr_glv_split = the decomposition the prover's kernels use (bp_compact.cuh: glv_decompose; BLS12-381 by division by LAMBDA = z^2 - 1,
    BN254 by the reduced lattice basis with a signed second half): s = s1 + s2 LAMBDA mod r with |halves| < 2^128 on boundary values and
    random scalars, and LAMBDA is the eigenvalue of (x, y) -> (BETA x, y) on the oracle's curve (Python integers: oracle/pyref.py)."""
    import random
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    ctx = ctxs[name]
    c = R.CURVES[name]
    r, p = c.r, c.p
    if name == "bls12_381":
        z = -0xd201000000010000
        lam = z * z - 1
    else:
        u = -0x4080000000000001
        lam = (36 * u ** 4 - 1) % r
        assert lam == 0x9366c48000000005b696800000000013a700000000000016 == (-(36 * u ** 3 + 18 * u ** 2 + 6 * u + 2)) % r
    assert (lam * lam + lam + 1) % r == 0
    P = c.mul(0x1234567, c.g)
    lp = c.mul(lam, P)
    assert lp[1] == P[1] and pow(lp[0] * pow(P[0], -1, p) % p, 3, p) == 1 and lp[0] != P[0]            # (x, y) -> (beta x, y), beta^3 = 1
    rnd = random.Random(17)
    vals = [0, 1, 2, r - 1, r - 2, r // 2, r // 2 + 1, r // 3, lam % r, (lam - 1) % r, (lam + 1) % r, (r - lam) % r, (1 << 128) - 1, 1 << 128, (1 << 128) + 1,
            (1 << 253) - 1, 1 << 253] + [rnd.randrange(r) for _ in range(4000)] + [(k * r) // 257 for k in range(1, 257)] + [(k * r) // 257 + 1 for k in range(1, 257)]
    vals += [(k * lam) % r for k in range(1, 200)] + [(r - k * lam) % r for k in range(1, 200)]
    sv = bp.FieldElementVector.from_bytes(ctx, b"".join(v.to_bytes(32, "little") for v in vals), len(vals))
    out = sv.glv_split().to_bytes()
    neg_seen = 0
    for i, v in enumerate(vals):
        s1 = int.from_bytes(out[32 * i:32 * i + 16], "little")
        s2 = int.from_bytes(out[32 * i + 16:32 * i + 32], "little")
        if name == "bn254" and s2 >> 66:                       # a negative second half, mod 2^128
            s2 -= 1 << 128
            neg_seen += 1
        assert (s1 + s2 * lam - v) % r == 0, (name, hex(v))
        assert 0 <= s1 < 1 << 128 and abs(s2) < 1 << 128
    assert name == "bls12_381" or neg_seen > 4000


@pytest.mark.parametrize("name,n,c,lim", [("bls12_381", 64, 8, 16), ("bn254", 256, 16, 2), ("bls12_381", 4096, 16, 4096), ("bn254", 8192, 13, 4096)])
def test_ipp_verify_over_generator_tables(bp, name, n, c, lim):
    """VERDICT r3 #8 (verify_ipp, /root/reference src/ipp.rs:204-260): with window tables on G and H the verifier's [G | H] terms run over
    the tables (a side-by-side table the context keeps) and [Q | L | R] as a small MSM beside them.  Same accept / reject matrix as the
    plain path, single and batched, for a prefix of longer generator vectors, after the tables change, and when they cannot be used."""
    cid = bp.CURVE_IDS[name]
    ctx = bp.Context(cid, 0)                                    # its own context: the knob and the kept table are per context
    with pytest.raises(bp.ArgError):
        ctx.set_tuning(bp.TUNE_VERIFY_TABLES, 1)
    Gv, Hv, Q, Gf, Hf, a, b = make_instance(bp, ctx, n, 77000 + n, unit_gf=False)
    label = b"verify over tables"
    proof = bp.IPP.create_ipp(ctx, bp.Transcript(label), Q, Gf, Hf, Gv, Hv, a, b)
    P = commitment_P(bp, ctx, Gv, Hv, Q, Gf, Hf, a, b)
    pb = ctx.point_bytes
    ver = lambda **kw: bp.IPP.verify_ipp(ctx, kw.get("n", n), bp.Transcript(label), kw.get("Gf", Gf), kw.get("Hf", Hf), kw.get("P", P), kw.get("Q", Q),
                                         kw.get("G", Gv), kw.get("H", Hv), kw.get("a", proof.a), kw.get("b", proof.b), kw.get("L", proof.L), kw.get("R", proof.R))
    ver()
    Gv.precompute(c)
    Hv.precompute(c)
    ver()
    assert ctx.verify_table_info() == (0, 0)                    # off by default: the plain path, tables or not
    ctx.set_tuning(bp.TUNE_VERIFY_TABLES, lim)
    after_plain = bp.Transcript(label)
    bp.IPP.verify_ipp(ctx, n, after_plain, Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
    kept = ctx.verify_table_info()
    assert kept == (n, Gv.table_info()[1] * 2 * n * pb), kept   # W x 2n rows
    plain_t = bp.Transcript(label)
    ctx.set_tuning(bp.TUNE_VERIFY_TABLES, 0)                    # off again: same verdict, same transcript state
    bp.IPP.verify_ipp(ctx, n, plain_t, Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
    assert plain_t.challenge_bytes(b"after", 32) == after_plain.challenge_bytes(b"after", 32)
    ctx.set_tuning(bp.TUNE_VERIFY_TABLES, lim)
    assert O.ipp_verify(cid, O.Transcript(label), n, Gf.to_bytes(), Hf.to_bytes(), P, Q, Gv.to_bytes(), Hv.to_bytes(), proof.a, proof.b, proof.L, proof.R,
                        proof.lg_n) == 0

    def bump(x):
        return ((int.from_bytes(x, "little") + 1) % ctx.r).to_bytes(32, "little")

    gen = O.generator(cid)
    off_curve = gen[:pb - 1] + bytes([gen[pb - 1] ^ 1])
    for kw in ({"a": bump(proof.a)}, {"b": bump(proof.b)}, {"P": gen}, {"Q": gen}, {"L": gen + proof.L[pb:]}, {"R": proof.R[:-pb] + gen},
               {"L": off_curve + proof.L[pb:]}, {"Q": off_curve}):
        with pytest.raises((bp.VerificationError, bp.ArgError)):
            ver(**kw)
    ver()                                                       # and the kept table is still good
    # batched over the same tables: accept iff each verifies
    items = [(label, P, Q, proof)]
    for j in range(2):
        a2 = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, 77100 + j, n), n)
        b2 = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, 77200 + j, n), n)
        Q2 = O.g1_mul(cid, O.random_scalars(cid, 77300 + j, 1), gen)
        pr2 = bp.IPP.create_ipp(ctx, bp.Transcript(b"b%d" % j), Q2, Gf, Hf, Gv, Hv, a2, b2)
        items.append((b"b%d" % j, commitment_P(bp, ctx, Gv, Hv, Q2, Gf, Hf, a2, b2), Q2, pr2))
    refs = lambda bad=None: [(bp.Transcript(lb), Pj, Qj, bump(pr.a) if bad == k else pr.a, pr.b, pr.L, pr.R) for k, (lb, Pj, Qj, pr) in enumerate(items)]
    bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, refs())
    for k in range(3):
        with pytest.raises(bp.VerificationError):
            bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, refs(k))
    # a proof over the first half of the generators, verified against the long (precomputed) vectors
    h = n // 2
    if h >= lim:
        gb, hb = Gv.to_bytes(), Hv.to_bytes()
        G2, H2 = bp.G1Vector.from_bytes(ctx, gb[:h * pb], h), bp.G1Vector.from_bytes(ctx, hb[:h * pb], h)
        fh = lambda v: bp.FieldElementVector.from_bytes(ctx, v.to_bytes()[:h * 32], h)
        pr = bp.IPP.create_ipp(ctx, bp.Transcript(label), Q, fh(Gf), fh(Hf), G2, H2, fh(a), fh(b))
        Ph = commitment_P(bp, ctx, G2, H2, Q, fh(Gf), fh(Hf), fh(a), fh(b))
        ver(n=h, Gf=fh(Gf), Hf=fh(Hf), P=Ph, a=pr.a, b=pr.b, L=pr.L, R=pr.R)
        assert ctx.verify_table_info()[0] == h                  # the kept table follows the generators last verified against
        with pytest.raises(bp.VerificationError):
            ver(n=h, Gf=fh(Gf), Hf=fh(Hf), P=Ph, a=bump(pr.a), b=pr.b, L=pr.L, R=pr.R)
        ver()
        assert ctx.verify_table_info()[0] == n
    # the tables are rebuilt (new rows, same points): the kept table must not be mistaken for the new vectors'
    Gv.precompute(c)
    ver()
    with pytest.raises(bp.VerificationError):
        ver(a=bump(proof.a))
    # widths differ: no side-by-side table, the plain path decides
    Hv.precompute(c - 1)
    ver()
    with pytest.raises(bp.VerificationError):
        ver(b=bump(proof.b))
    Hv.precompute(c)
    ver()
    assert ctx.verify_table_info()[0] == n
    ctx.drop_verify_table()
    assert ctx.verify_table_info() == (0, 0)
    ver()
    Gv.free(); Hv.free()
    ctx.close()


@pytest.mark.parametrize("name", CURVES)
@pytest.mark.parametrize("nq,n,m", [(0, 4, 1), (1, 1, 1), (37, 5, 3), (700, 64, 8), (5000, 300, 0)])
def test_r1cs_flattened_constraints(bp, ctxs, name, nq, n, m):
    """Prover/Verifier::flattened_constraints (src/r1cs/prover.rs:142-184, verifier.rs:149-193) against the Python-int
    restatement that walks the constraints in the reference's order; includes destinations with hundreds of terms (the
    block-per-destination kernel) and constraints with no terms at all."""
    import random
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    ctx = ctxs[name]
    r = ctx.r
    curve = R.CURVES[name]
    rng = random.Random(1000 * nq + n)
    cons = _random_constraint_system(rng, r, nq, n, max(m, 1) if m else 1, hub=(1, n // 2) if nq >= 100 else None)
    if m == 0:   # no committed variables: drop those terms
        cons = [[t for t in terms if t[0] != 3] for terms in cons]
    terms = [(q, kind, idx, coeff) for q, ts in enumerate(cons) for kind, idx, coeff in ts]
    rng.shuffle(terms)                                     # the plan does not depend on the order the terms arrive in
    plan = bp.R1CSPlan(ctx, terms, nq, n, m)
    for z in (rng.randrange(1, r), 1, 0):
        wL, wR, wO, wV, wc = plan.flattened_constraints(z.to_bytes(32, "little"))
        eL, eR, eO, eV, ec = R.r1cs_flattened_constraints(curve, cons, z, n, m)
        assert ints(wL.to_bytes()) == eL and ints(wR.to_bytes()) == eR and ints(wO.to_bytes()) == eO
        assert ints(wV.to_bytes()) == eV and int.from_bytes(wc, "little") == ec
        assert plan.flattened_constraints(z.to_bytes(32, "little"), want_constant=False)[4] is None
    plan.free()


def test_r1cs_plan_argument_checks(bp, ctxs):
    ctx = ctxs["bls12_381"]
    for bad in ([(5, 0, 0, 1)],            # constraint index out of range
                [(0, 0, 9, 1)],            # multiplier index >= n
                [(0, 3, 2, 1)],            # committed index >= m
                [(0, 7, 0, 1)]):           # unknown variable kind
        with pytest.raises(bp.ArgError):
            bp.R1CSPlan(ctx, bad, 5, 4, 2)
    plan = bp.R1CSPlan(ctx, [(0, 4, 0, 3)], 1, 4, 2)       # a lone constant term
    other = bp.Context(bp.BN254, 0)
    with pytest.raises(bp.ArgError):                        # a plan belongs to its curve
        import ctypes
        out = (ctypes.c_void_p * 4)()
        rc = bp.lib().bp_r1cs_flattened_constraints(other.h, plan.h, (1).to_bytes(32, "little"), out, None)
        if rc == bp.BP_ERR_ARG:
            raise bp.ArgError("bp_r1cs_flattened_constraints failed with status 2")
    other.close()
    wL, wR, wO, wV, wc = plan.flattened_constraints((5).to_bytes(32, "little"))
    assert int.from_bytes(wc, "little") == (ctx.r - 15) and ints(wL.to_bytes()) == [0] * 4


@pytest.mark.parametrize("name", CURVES)
def test_commit_pairs_vs_oracle(bp, ctxs, name):
    """bp_g1vec_commit_pairs = batched `commit_to_field_element(g, h, m, r)` / `g.binary_scalar_mul(h, m, r)`
    (src/r1cs/prover.rs:123,496-500) against the oracle's binary_scalar_mul, including the degenerate point pairs."""
    ctx = ctxs[name]
    cid, r, pb = ctx.curve, ctx.r, ctx.point_bytes
    gen = O.generator(cid)
    g = O.g1_mul(cid, (7).to_bytes(32, "little"), gen)
    h = O.g1_mul(cid, (11).to_bytes(32, "little"), gen)
    neg_g = O.g1_mul(cid, (r - 7).to_bytes(32, "little"), gen)
    specials = [0, 1, 2, r - 1, r - 2, (1 << 128) - 1, 1 << 200]
    k1 = specials + ints(O.random_scalars(cid, 7700, 60))
    k2 = list(reversed(specials)) + ints(O.random_scalars(cid, 7701, 60))
    dev = lambda xs: bp.FieldElementVector.from_ints(ctx, xs)
    for P, Q in ((g, h), (g, g), (g, neg_g), (bytes(pb), h), (g, bytes(pb))):
        got = bp.G1Vector.commit_pairs(ctx, P, Q, dev(k1), dev(k2)).to_bytes()
        for j, (a, b) in enumerate(zip(k1, k2)):
            want = O.binary_scalar_mul(cid, P, Q, a.to_bytes(32, "little"), b.to_bytes(32, "little"))
            assert got[j * pb:(j + 1) * pb] == want, (j, a, b)
    assert len(bp.G1Vector.commit_pairs(ctx, g, h, dev([]), dev([]))) == 0
    with pytest.raises(bp.ValueError_):
        bp.G1Vector.commit_pairs(ctx, g, h, dev([1, 2]), dev([1]))
