"""CPU: the C oracle (oracle/liboracle.so) against the committed golden vectors (tests/golden/*.json,
written by oracle/gen_golden.py from Python-int arithmetic) and the public KATs."""
import pytest

import _oracle as O

CURVES = ["bls12_381", "bn254"]


def hx(s):
    return bytes.fromhex(s)


@pytest.mark.parametrize("name", CURVES)
def test_curve_constants(golden, name):
    g = golden("curves")[name]
    cid = O.CURVE_IDS[name]
    assert g["curve_id"] == cid
    G = O.generator(cid)
    assert G == hx(g["G"])
    assert O.on_curve(cid, G)
    assert O.g1_add(cid, G, G) == hx(g["G2"])
    assert O.g1_add(cid, hx(g["G2"]), G) == hx(g["G3"])
    rm1 = (int(g["r"], 16) - 1).to_bytes(32, "little")
    assert O.g1_mul(cid, rm1, G) == hx(g["G_rm1"])
    assert O.g1_add(cid, hx(g["G_rm1"]), G) == bytes(len(G))          # r*G = O
    assert O.g1_to_amcl(cid, G) == hx(g["G_amcl"])
    assert O.g1_to_amcl(cid, bytes(len(G))) == hx(g["identity_amcl"])
    if name == "bls12_381":   # public constants of the curve (IETF pairing-friendly-curves draft / zkcrypto), then the published 2G
        assert int(g["p"], 16) == 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
        assert int(g["r"], 16) == 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
        assert int(g["gx"], 16) == 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
        assert int(g["gy"], 16) == 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
        z = -0xd201000000010000            # the BLS parameter: r = z^4 - z^2 + 1, p = (z - 1)^2 r / 3 + z, cofactor = (z - 1)^2 / 3
        assert int(g["r"], 16) == z**4 - z**2 + 1 and int(g["p"], 16) == (z - 1) ** 2 * int(g["r"], 16) // 3 + z
        x2 = int("0572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e", 16)
        assert O.g1_add(cid, G, G)[:48] == x2.to_bytes(48, "little")


@pytest.mark.parametrize("name", CURVES)
def test_field_ops(golden, name):
    cid = O.CURVE_IDS[name]
    for which, key in ((0, "fp"), (1, "fr")):
        for c in golden("field")[name][key]:
            a, b = hx(c["a"]), hx(c["b"])
            assert O.field_op(cid, which, 0, a, b) == hx(c["add"])
            assert O.field_op(cid, which, 1, a, b) == hx(c["sub"])
            assert O.field_op(cid, which, 2, a, b) == hx(c["mul"])
            assert O.field_op(cid, which, 3, a, b) == hx(c["inv_a"])


@pytest.mark.parametrize("name", CURVES)
def test_g1_ops(golden, name):
    cid = O.CURVE_IDS[name]
    g = golden("g1")[name]
    for c in g["add"]:
        assert O.g1_add(cid, hx(c["p"]), hx(c["q"])) == hx(c["sum"])
    for c in g["mul"]:
        assert O.g1_mul(cid, hx(c["k"]), hx(c["p"])) == hx(c["kp"])
    for c in g["binary_scalar_mul"]:
        assert O.binary_scalar_mul(cid, hx(c["p"]), hx(c["q"]), hx(c["k1"]), hx(c["k2"])) == hx(c["out"])


@pytest.mark.parametrize("name", CURVES)
@pytest.mark.parametrize("algo", [O.NAIVE, O.STRAUSS, O.PIPPENGER])
def test_msm_golden(golden, name, algo):
    cid = O.CURVE_IDS[name]
    for c in golden("msm")[name]:
        pts = b"".join(hx(p) for p in c["points"])
        sc = b"".join(hx(s) for s in c["scalars"])
        assert O.msm(cid, pts, sc, c["n"], algo=algo) == hx(c["out"]), c["name"]
    c = golden("msm")[name][5]
    pts = b"".join(hx(p) for p in c["points"])
    sc = b"".join(hx(s) for s in c["scalars"])
    assert O.msm(cid, pts, sc, c["n"], algo=O.PIPPENGER, nthreads=4) == hx(c["out"])


def test_merlin_golden(golden):
    for c in golden("merlin"):
        t = O.Transcript(hx(c["label"]))
        got = []
        for op in c["ops"]:
            if op[0] == "append":
                t.append_message(hx(op[1]), hx(op[2]))
            else:
                got.append(t.challenge_bytes(hx(op[1]), op[2]).hex())
        assert got == c["challenges"], c["name"]
    assert golden("merlin")[0]["challenges"][0] == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


@pytest.mark.parametrize("name", CURVES)
def test_ipp_golden(golden, name):
    cid = O.CURVE_IDS[name]
    for c in golden("ipp")[name]:
        n = c["n"]
        cat = lambda k: b"".join(hx(x) for x in c[k])
        tr = O.Transcript(b"innerproduct")
        rc, out = O.ipp_create(cid, tr, hx(c["Q"]), cat("G_factors"), cat("H_factors"), cat("G"), cat("H"), cat("a"), cat("b"), n)
        assert rc == 0
        L, R, a0, b0 = out
        assert L == cat("L") and R == cat("R"), c["name"]
        assert a0 == hx(c["a_out"]) and b0 == hx(c["b_out"])
        assert tr.challenge_bytes(b"after", 32).hex() == c["transcript_after"]
        lg = len(c["L"])
        for i, lam in enumerate(c["L_amcl"]):
            assert O.g1_to_amcl(cid, hx(c["L"][i])).hex() == lam
        ok = O.ipp_verify(cid, O.Transcript(b"innerproduct"), n, cat("G_factors"), cat("H_factors"), hx(c["P"]), hx(c["Q"]),
                          cat("G"), cat("H"), a0, b0, L, R, lg)
        assert ok == 0
        bad_a = ((int.from_bytes(a0, "little") + 1) % int(golden("curves")[name]["r"], 16)).to_bytes(32, "little")
        assert O.ipp_verify(cid, O.Transcript(b"innerproduct"), n, cat("G_factors"), cat("H_factors"), hx(c["P"]), hx(c["Q"]),
                            cat("G"), cat("H"), bad_a, b0, L, R, lg) == 3
        # verification_scalars error exits (src/ipp.rs:269-276)
        assert O.ipp_verify(cid, O.Transcript(b"innerproduct"), 2 * n, cat("G_factors"), cat("H_factors"), hx(c["P"]), hx(c["Q"]),
                            cat("G"), cat("H"), a0, b0, L, R, lg) == 3


def test_ipp_create_rejects_non_power_of_two():
    cid = 0
    G = O.generator(cid)
    z = bytes(32)
    rc, _ = O.ipp_create(cid, O.Transcript(b"innerproduct"), G, z * 3, z * 3, G * 3, G * 3, z * 3, z * 3, 3)
    assert rc == 2


@pytest.mark.parametrize("name", CURVES)
def test_fixed_base_and_linearity(golden, name):
    """MSM(s, k.G) == (sum s_i k_i) G -- the size-independent property the full-size GPU tests use."""
    cid = O.CURVE_IDS[name]
    r = int(golden("curves")[name]["r"], 16)
    n = 200
    ks = O.random_scalars(cid, 11, n)
    ss = O.random_scalars(cid, 12, n)
    pts = O.fixed_base_batch(cid, ks, n, nthreads=3)
    G = O.generator(cid)
    pb = O.pt_bytes(cid)
    for i in (0, 1, n - 1):
        assert pts[i * pb:(i + 1) * pb] == O.g1_mul(cid, ks[i * 32:(i + 1) * 32], G)
    dot = O.fr_inner(cid, ks, ss, n)
    exp = sum(int.from_bytes(ks[i * 32:(i + 1) * 32], "little") * int.from_bytes(ss[i * 32:(i + 1) * 32], "little") for i in range(n)) % r
    assert dot == exp.to_bytes(32, "little")
    assert O.msm(cid, pts, ss, n, algo=O.PIPPENGER, nthreads=2) == O.g1_mul(cid, dot, G)


def test_random_scalars_match_pyref(golden):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref
    for c in (pyref.BLS12_381, pyref.BN254):
        rng = pyref.SplitMix64(77)
        exp = b"".join(c.fr_to_le(rng.scalar(c)) for _ in range(20))
        assert O.random_scalars(c.curve_id, 77, 20) == exp


def test_shake256_matches_hashlib_and_golden(golden):
    """The hash half of G1::from_msg_hash is pinned by an independent implementation (hashlib's SHAKE256)."""
    import ctypes
    import hashlib
    L = O.lib()
    for c in golden("hash_to_g1")["shake256"]:
        msg = bytes.fromhex(c["msg"])
        out = ctypes.create_string_buffer(48)
        L.orc_shake256(msg, ctypes.c_size_t(len(msg)), out, ctypes.c_size_t(48))
        assert out.raw.hex() == c["digest48"] == hashlib.shake_256(msg).hexdigest(48)
    for n in (0, 1, 135, 136, 137, 271, 272, 273, 1000):
        msg = bytes((7 * i + n) & 0xFF for i in range(n))
        out = ctypes.create_string_buffer(200)
        L.orc_shake256(msg, ctypes.c_size_t(n), out, ctypes.c_size_t(200))   # squeeze past one rate block
        assert out.raw == hashlib.shake_256(msg).digest(200)
    # long message / long output (hundreds of absorb and squeeze blocks), and the two public FIPS 202 answers for the empty message
    msg = bytes((i * i + 3 * i) & 0xFF for i in range(100003))
    out = ctypes.create_string_buffer(10007)
    L.orc_shake256(msg, ctypes.c_size_t(len(msg)), out, ctypes.c_size_t(10007))
    assert out.raw == hashlib.shake_256(msg).digest(10007)
    out = ctypes.create_string_buffer(32)
    L.orc_shake256(b"", ctypes.c_size_t(0), out, ctypes.c_size_t(32))
    assert out.raw.hex() == "46b9dd2b0ba88d13233b3feb743eeb243fcd52ea62b81b82b50c27646ed5762f"   # SHAKE256(""), FIPS 202 / NIST example


@pytest.mark.parametrize("name", ["bls12_381", "bn254"])
def test_hash_to_g1_golden(golden, name):
    """C oracle vs the Python-int restatement of amcl's mapit (tests/golden/hash_to_g1.json); outputs lie on the curve."""
    cid = O.CURVE_IDS[name]
    g = golden("hash_to_g1")["curves"][name]
    for c in g["from_msg_hash"]:
        p = O.g1_from_msg_hash(cid, bytes.fromhex(c["msg"]))
        assert p.hex() == c["point"]
        assert O.on_curve(cid, p)
    for prefix, pts in g["get_generators"].items():
        got = O.get_generators(cid, prefix, len(pts), nthreads=3)
        assert got.hex() == "".join(pts)
    # a shifted counter continues the same sequence
    pts = g["get_generators"]["G"]
    assert O.get_generators(cid, "G", 4, first=5).hex() == "".join(pts[4:8])
    # r * P == O: the cofactor was cleared
    r = int(golden("curves")[name]["r"], 16)
    p = bytes.fromhex(pts[0])
    rm1 = O.g1_mul(cid, (r - 1).to_bytes(32, "little"), p)
    assert O.g1_add(cid, rm1, p) == bytes(len(p))


def r1cs_case_inputs(c):
    """Fixture case (tests/golden/r1cs.json) -> the byte arguments of O.r1cs_prove / O.r1cs_verify and of the C ABI."""
    cat = lambda key: b"".join(hx(x) for x in c[key])
    terms = [(q, k, i, hx(cf)) for q, k, i, cf in c["terms"]]
    return {"terms": terms, "label": hx(c["label"]), "g": hx(c["g"]), "h": hx(c["h"]), "G": cat("G"), "H": cat("H"), "V": [hx(v) for v in c["V"]],
            "aL": cat("a_L"), "aR": cat("a_R"), "aO": cat("a_O"), "vb": cat("v_blinding"), "sL": cat("s_L"), "sR": cat("s_R"),
            "blind": cat("blindings"), "proof": hx(c["proof"]), "r": hx(c["verifier_r"])}


@pytest.mark.parametrize("name", CURVES)
@pytest.mark.parametrize("threads", [1, 3])
def test_r1cs_golden(golden, name, threads):
    """Prover::prove / Verifier::verify in the C oracle (oracle/orc_r1cs_tmpl.h) against whole proofs made by the Python-int
    restatement (oracle/pyref.py r1cs_prove): same bytes, accepted, and rejected after any single change."""
    cid = O.CURVE_IDS[name]
    O.set_threads(threads)
    try:
        for c in golden("r1cs")[name]:
            a = r1cs_case_inputs(c)
            cs = O.R1CSTerms(a["terms"], c["n_constraints"], c["n"], c["m"])
            ng = c["n_generators"]
            rc, proof = O.r1cs_prove(cid, O.r1cs_start_transcript(cid, a["label"], a["V"]), cs, a["g"], a["h"], a["G"], a["H"], ng,
                                     a["aL"], a["aR"], a["aO"], a["vb"], a["sL"], a["sR"], a["blind"])
            assert rc == 0 and proof == a["proof"], c["name"]
            Vb = b"".join(a["V"])
            verify = lambda pf, V=Vb, r=a["r"]: O.r1cs_verify(cid, O.r1cs_start_transcript(cid, a["label"], a["V"]), cs, V, pf, a["g"], a["h"],
                                                               a["G"], a["H"], ng, r)
            assert verify(proof) == 0, c["name"]
            assert verify(proof, r=(12345).to_bytes(32, "little")) == 0          # any weight accepts an honest proof
            pb = O.pt_bytes(cid)
            for off in (11 * pb, 11 * pb + 32, 11 * pb + 64, len(proof) - 1, len(proof) - 33):     # t_x, t_x_blinding, e_blinding, b, a
                bad = bytearray(proof)
                bad[off] ^= 1
                assert verify(bytes(bad)) == 3, (c["name"], off)
            swapped = proof[pb:2 * pb] + proof[:pb] + proof[2 * pb:]                 # A_I1 <-> A_O1
            assert verify(swapped) == 3
            assert verify(proof[:-1]) == 3                                           # wrong length
            if c["m"]:
                V2 = Vb[pb:2 * pb] + Vb[:pb] + Vb[2 * pb:] if c["m"] > 1 else O.generator(cid)
                if V2 != Vb:
                    assert O.r1cs_verify(cid, O.r1cs_start_transcript(cid, a["label"], a["V"]), cs, V2, proof, a["g"], a["h"], a["G"], a["H"],
                                         ng, a["r"]) == 3
            assert O.r1cs_prove(cid, O.r1cs_start_transcript(cid, a["label"], a["V"]), cs, a["g"], a["h"], a["G"], a["H"], max(1, c["n"]) - 1,
                                a["aL"], a["aR"], a["aO"], a["vb"], a["sL"], a["sR"], a["blind"])[0] == (1 if c["n"] > 0 else 0)   # InvalidGeneratorsLength
    finally:
        O.set_threads(1)


@pytest.mark.parametrize("name", CURVES)
def test_r1cs_flatten_vs_pyref(name):
    """flattened_constraints (src/r1cs/verifier.rs:149-193): C oracle against the Python-int restatement on a random system."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    cv = R.CURVES[name]
    cid = O.CURVE_IDS[name]
    rng = R.SplitMix64(77 + cid)
    n, m, nq = 13, 4, 29
    cons = []
    for q in range(nq):
        lc = []
        for _ in range(1 + rng.next() % 5):
            kind = rng.next() % 5
            lc.append(((kind, rng.next() % (m if kind == 3 else n)), rng.scalar(cv)))
        cons.append(lc)
    z = rng.scalar(cv)
    want = R._flatten(cv, cons, z, n, m)
    cs = O.R1CSTerms(R.constraints_to_terms(cons), nq, n, m)
    got = O.r1cs_flattened_constraints(cid, cs, z.to_bytes(32, "little"))
    ints = lambda b: [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]
    assert [ints(g) for g in got[:4]] == [list(w) for w in want[:4]] and int.from_bytes(got[4], "little") == want[4]


@pytest.mark.parametrize("name", CURVES)
def test_r1cs_two_phase_fixture_verifies_in_pyref(golden, name):
    """tests/golden/r1cs2.json against the committed oracle/pyref.py: the fixture's proofs are decoded from their C-ABI bytes and checked
    by the Python-int Verifier::verify with the shuffle gadget's deferred callback (verifier.rs:245-263) -- accepted as made, rejected
    with another commitment; the challenge the callback draws is the one the fixture records."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    c = R.CURVES[name]
    fb = c.modbytes if name == "bn254" else 48

    def pt(b):
        if b == bytes(len(b)):
            return None
        return int.from_bytes(b[:fb], "little"), int.from_bytes(b[fb:], "little")

    for case in golden("r1cs2")[name]:
        pb = 2 * fb
        raw = hx(case["proof"])
        n = case["n1"] + case["n2"]
        lg = max(0, (n - 1).bit_length())
        keys = ("A_I1", "A_O1", "S1", "A_I2", "A_O2", "S2", "T_1", "T_3", "T_4", "T_5", "T_6")
        proof = {k: pt(raw[i * pb:(i + 1) * pb]) for i, k in enumerate(keys)}
        o = 11 * pb
        for k in ("t_x", "t_x_blinding", "e_blinding"):
            proof[k] = int.from_bytes(raw[o:o + 32], "little"); o += 32
        proof["L"] = [pt(raw[o + i * pb:o + (i + 1) * pb]) for i in range(lg)]; o += lg * pb
        proof["R"] = [pt(raw[o + i * pb:o + (i + 1) * pb]) for i in range(lg)]; o += lg * pb
        proof["a"], proof["b"] = int.from_bytes(raw[o:o + 32], "little"), int.from_bytes(raw[o + 32:o + 64], "little")
        g, h_ = pt(hx(case["g"])), pt(hx(case["h"]))
        Gs, Hs = [pt(hx(x)) for x in case["G"]], [pt(hx(x)) for x in case["H"]]
        V = [pt(hx(x)) for x in case["V"]]
        rv = int.from_bytes(hx(case["verifier_r"]), "little")

        def run(Vs):
            tr = R.Transcript(hx(case["label"]))
            seen = {}
            real = tr.challenge_scalar
            def spy(curve, lbl):
                v = real(curve, lbl)
                seen[lbl] = v
                return v
            tr.challenge_scalar = spy
            ver = R.R1CSVerifier(c, tr)
            vars_ = [ver.commit(P) for P in Vs]
            k = len(vars_) // 2
            if case["name"] == "product_then_shuffle_3":
                _, _, o_ = ver.multiply([(vars_[0], 1)], [(vars_[1], 1)])
                ver.constrain(R.lc_sub(c, [(o_, 1)], [(vars_[6], 1)]))
                k = 3
            R.shuffle_gadget(ver, vars_[:k], vars_[k:2 * k])
            ok = R.r1cs_verify(ver, proof, g, h_, Gs, Hs, rv)
            return ok, seen.get(hx(case["challenge_label"]))

        ok, z = run(V)
        assert ok and z == int.from_bytes(hx(case["challenge"]), "little")
        assert not run([c.g] + V[1:])[0]
