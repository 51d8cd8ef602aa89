#!/usr/bin/env python3
"""Writes tests/golden/*.json from oracle/pyref.py (Python-int arithmetic, seeded).

TEST INFRASTRUCTURE ONLY.  Run from the repo root:  python3 oracle/gen_golden.py
The fixtures are data (inputs + expected outputs); both the C oracle and the HIP path
are checked against them byte-for-byte.  All values are hex strings; field elements and
points use the C-ABI's canonical little-endian limb format (BP_FMT_LE) unless the key
says `_amcl` (amcl big-endian `04||X||Y` / MODBYTES big-endian).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyref as R  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SEED = 0xB0117E7


def h(b):
    return bytes(b).hex()


def fp_le(c, x):
    return h(x.to_bytes(4 * c.fp_limbs32, "little"))


def fr_le(c, x):
    return h(c.fr_to_le(x))


def pt_le(c, P):
    return h(c.g1_to_le(P))


def rand_fp(rng, c):
    bits = c.p.bit_length()
    while True:
        v = 0
        for i in range((bits + 63) // 64):
            v |= rng.next() << (64 * i)
        v &= (1 << bits) - 1
        if v < c.p:
            return v


def gen_curves():
    out = {}
    for c in R.CURVES.values():
        G = c.g
        n32 = c.fp_limbs32
        Rm = 1 << (32 * n32)
        Rr = 1 << (32 * c.fr_limbs32)
        out[c.name] = {
            "curve_id": c.curve_id,
            "p": hex(c.p), "r": hex(c.r), "b": c.b,
            "gx": hex(G[0]), "gy": hex(G[1]),
            "modbytes": c.modbytes, "fp_limbs32": n32, "fr_limbs32": c.fr_limbs32,
            "fp_inv32": hex((-pow(c.p, -1, 1 << 32)) % (1 << 32)),
            "fr_inv32": hex((-pow(c.r, -1, 1 << 32)) % (1 << 32)),
            "fp_inv64": hex((-pow(c.p, -1, 1 << 64)) % (1 << 64)),
            "fr_inv64": hex((-pow(c.r, -1, 1 << 64)) % (1 << 64)),
            "fp_R_mod_p": hex(Rm % c.p), "fp_R2_mod_p": hex(Rm * Rm % c.p),
            "fr_R_mod_r": hex(Rr % c.r), "fr_R2_mod_r": hex(Rr * Rr % c.r),
            "G": pt_le(c, G), "G2": pt_le(c, c.add(G, G)), "G3": pt_le(c, c.mul(3, G)),
            "G_rm1": pt_le(c, c.mul(c.r - 1, G)),
            "G_amcl": h(c.g1_to_bytes(G)), "identity_amcl": h(c.g1_to_bytes(None)),
        }
    return out


def gen_field():
    out = {}
    for c in R.CURVES.values():
        rng = R.SplitMix64(SEED + 100 + c.curve_id)
        fp_cases, fr_cases = [], []
        specials_p = [0, 1, 2, c.p - 1, c.p - 2, (c.p - 1) // 2]
        specials_r = [0, 1, 2, c.r - 1, c.r - 2, (c.r - 1) // 2]
        pairs_p = [(a, b) for a in specials_p for b in specials_p] + [(rand_fp(rng, c), rand_fp(rng, c)) for _ in range(64)]
        pairs_r = [(a, b) for a in specials_r for b in specials_r] + [(rng.scalar(c), rng.scalar(c)) for _ in range(64)]
        for a, b in pairs_p:
            fp_cases.append({"a": fp_le(c, a), "b": fp_le(c, b), "add": fp_le(c, (a + b) % c.p), "sub": fp_le(c, (a - b) % c.p),
                             "mul": fp_le(c, a * b % c.p), "inv_a": fp_le(c, pow(a, -1, c.p) if a else 0)})
        for a, b in pairs_r:
            fr_cases.append({"a": fr_le(c, a), "b": fr_le(c, b), "add": fr_le(c, (a + b) % c.r), "sub": fr_le(c, (a - b) % c.r),
                             "mul": fr_le(c, a * b % c.r), "inv_a": fr_le(c, pow(a, -1, c.r) if a else 0)})
        out[c.name] = {"fp": fp_cases, "fr": fr_cases}
    return out


def gen_g1():
    out = {}
    for c in R.CURVES.values():
        rng = R.SplitMix64(SEED + 200 + c.curve_id)
        G = c.g
        pts = [c.mul(rng.scalar(c), G) for _ in range(6)]
        adds = []
        cases = [(None, None), (None, pts[0]), (pts[0], None), (pts[0], pts[0]), (pts[0], c.neg(pts[0])),
                 (G, G), (G, c.neg(G))] + [(pts[i], pts[j]) for i in range(6) for j in range(i + 1, 6)]
        for P, Q in cases:
            adds.append({"p": pt_le(c, P), "q": pt_le(c, Q), "sum": pt_le(c, c.add(P, Q))})
        muls = []
        ks = [0, 1, 2, 3, 15, 16, 17, 2**32 - 1, 2**32, 2**64 + 1, c.r - 1, c.r - 2, (c.r + 1) // 2] + [rng.scalar(c) for _ in range(12)]
        for k in ks:
            for P in (G, pts[1]):
                muls.append({"k": fr_le(c, k), "p": pt_le(c, P), "kp": pt_le(c, c.mul(k, P))})
        muls.append({"k": fr_le(c, 5), "p": pt_le(c, None), "kp": pt_le(c, None)})
        # a1*P + a2*Q  (binary_scalar_mul, src/ipp.rs:119,125,185,187)
        bins = []
        for _ in range(8):
            k1, k2 = rng.scalar(c), rng.scalar(c)
            P, Q = pts[rng.next() % 6], pts[rng.next() % 6]
            bins.append({"k1": fr_le(c, k1), "k2": fr_le(c, k2), "p": pt_le(c, P), "q": pt_le(c, Q),
                         "out": pt_le(c, c.add(c.mul(k1, P), c.mul(k2, Q)))})
        P = pts[2]
        bins.append({"k1": fr_le(c, 7), "k2": fr_le(c, c.r - 7), "p": pt_le(c, P), "q": pt_le(c, P), "out": pt_le(c, None)})
        bins.append({"k1": fr_le(c, 0), "k2": fr_le(c, 0), "p": pt_le(c, P), "q": pt_le(c, G), "out": pt_le(c, None)})
        bins.append({"k1": fr_le(c, 1), "k2": fr_le(c, 1), "p": pt_le(c, P), "q": pt_le(c, P), "out": pt_le(c, c.add(P, P))})
        out[c.name] = {"add": adds, "mul": muls, "binary_scalar_mul": bins}
    return out


def gen_msm():
    out = {}
    for c in R.CURVES.values():
        rng = R.SplitMix64(SEED + 300 + c.curve_id)
        G = c.g
        pool = [c.mul(rng.scalar(c), G) for _ in range(300)]
        cases = []

        def case(name, scalars, points):
            cases.append({"name": name, "n": len(scalars), "scalars": [fr_le(c, s) for s in scalars],
                          "points": [pt_le(c, P) for P in points], "out": pt_le(c, c.msm(scalars, points))})
            print("  msm", c.name, name, len(scalars), file=sys.stderr)

        case("empty", [], [])
        for n in (1, 2, 3, 17, 64, 257):
            case("random_%d" % n, [rng.scalar(c) for _ in range(n)], pool[:n])
        case("all_zero_scalars", [0] * 9, pool[:9])
        case("some_zero_scalars", [0 if i % 3 == 0 else rng.scalar(c) for i in range(40)], pool[:40])
        case("all_ones", [1] * 33, pool[:33])
        case("bits", [rng.next() & 1 for _ in range(100)], pool[:100])                    # config-3-like a_L
        case("small_scalars", [rng.next() % 1000 for _ in range(64)], pool[:64])
        case("r_minus_1", [c.r - 1] * 5, pool[:5])
        case("identity_points", [rng.scalar(c) for _ in range(6)], [None, pool[0], None, pool[1], pool[2], None])
        case("all_same_point", [rng.scalar(c) for _ in range(50)], [pool[7]] * 50)
        case("same_point_same_scalar", [12345] * 40, [pool[8]] * 40)                       # forces P+P in buckets
        case("p_and_minus_p", [5, 5, 7, 7], [pool[3], c.neg(pool[3]), pool[4], c.neg(pool[4])])
        s = rng.scalar(c)
        case("cancel_to_identity", [s, c.r - s], [pool[5], pool[5]])
        # digits that land in the same signed-window bucket with opposite signs, every window size 1..16
        case("window_edge_scalars", [(1 << k) - 1 for k in range(1, 33)] + [1 << k for k in range(1, 33)] + [(1 << 255) % c.r, c.r >> 1],
             (pool * 2)[:66])
        case("duplicates_mixed", [rng.scalar(c) for _ in range(30)], [pool[i % 4] for i in range(30)])
        out[c.name] = cases
    return out


def gen_merlin():
    cases = []
    t = R.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    cases.append({"name": "merlin_conformance_simple", "label": h(b"test protocol"),
                  "ops": [["append", h(b"some label"), h(b"some data")], ["challenge", h(b"challenge"), 32]],
                  "challenges": [h(t.challenge_bytes(b"challenge", 32))]})
    assert cases[0]["challenges"][0] == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    rng = R.SplitMix64(SEED + 400)
    t = R.Transcript(b"innerproduct")
    ops, chals = [], []
    for i in range(12):
        ln = [0, 1, 7, 48, 97, 164, 165, 166, 167, 200, 400, 1024][i]
        msg = bytes(rng.next() & 0xFF for _ in range(ln))
        lab = b"L" if i % 2 else b"some-longer-label"
        t.append_message(lab, msg)
        ops.append(["append", h(lab), h(msg)])
        n = [1, 32, 48, 166, 167, 400][i % 6]
        ops.append(["challenge", h(b"u"), n])
        chals.append(h(t.challenge_bytes(b"u", n)))
    cases.append({"name": "long_mixed", "label": h(b"innerproduct"), "ops": ops, "challenges": chals})
    return cases


def gen_ipp():
    out = {}
    for c in R.CURVES.values():
        rng = R.SplitMix64(SEED + 500 + c.curve_id)
        G0 = c.g
        cases = []
        for name, n, a, b, unit_gf in (
            ("n1", 1, [3], [9], True),
            ("n2", 2, [1, 2], [5, 6], True),
            ("test_ipp_n4", 4, [1, 2, 3, 4], [5, 6, 7, 8], True),                     # src/ipp.rs:325-390
            ("test_ipp_non_power_of_2_n8", 8, [1, 2, 3, 4, 9, 0, 0, 0], [5, 6, 7, 8, 10, 0, 0, 0], True),  # :393-489
            ("random_n8_gfactors", 8, None, None, False),
            ("random_n16", 16, None, None, True),
            # the same unit test with its own generator construction (src/ipp.rs:340-342): get_generators("g"/"h", n),
            # Q = G1::from_msg_hash("Q") -- kept last so that the cases above keep their RNG stream
            ("test_ipp_n4_hashed_generators", 4, [1, 2, 3, 4], [5, 6, 7, 8], True),
            # BASELINE config 1's size with inputs anybody can rebuild from the reference's API alone (integration/rust/pin_fixtures.rs,
            # pin_9): a = 1..64, b = 65..128, get_generators("g"/"h", 64), Q = from_msg_hash("Q"), H_factors = vandermonde(y_inv)
            ("pin9_n64_hashed_generators", 64, list(range(1, 65)), list(range(65, 129)), True),
        ):
            if a is None:
                a = [rng.scalar(c) for _ in range(n)]
                b = [rng.scalar(c) for _ in range(n)]
            if name.endswith("hashed_generators"):
                G, H, Q = R.get_generators(c, "g", n), R.get_generators(c, "h", n), R.g1_from_msg_hash(c, b"Q")
            else:
                G = [c.mul(rng.scalar(c), G0) for _ in range(n)]
                H = [c.mul(rng.scalar(c), G0) for _ in range(n)]
                Q = c.mul(rng.scalar(c), G0)
            y_inv = rng.scalar(c)
            Gf = [1] * n if unit_gf else [rng.scalar(c) for _ in range(n)]
            Hf = [pow(y_inv, i, c.r) for i in range(n)]                                # new_vandermonde_vector
            t = R.Transcript(b"innerproduct")
            Lv, Rv, a0, b0 = R.ipp_create(c, t, Q, Gf, Hf, G, H, a, b)
            after = t.challenge_bytes(b"after", 32)
            # P = <a, G*Gf> + <b, H*Hf> + <a,b> Q  (src/ipp.rs:353-372, generalised to G_factors != 1)
            ip = sum(x * y for x, y in zip(a, b)) % c.r
            P = c.msm([x * g % c.r for x, g in zip(a, Gf)] + [x * y % c.r for x, y in zip(b, Hf)] + [ip], G + H + [Q])
            assert R.ipp_verify(c, n, R.Transcript(b"innerproduct"), Gf, Hf, P, Q, G, H, a0, b0, Lv, Rv)
            assert not R.ipp_verify(c, n, R.Transcript(b"innerproduct"), Gf, Hf, P, Q, G, H, (a0 + 1) % c.r, b0, Lv, Rv)
            cases.append({
                "name": name, "n": n,
                "a": [fr_le(c, x) for x in a], "b": [fr_le(c, x) for x in b],
                "G": [pt_le(c, X) for X in G], "H": [pt_le(c, X) for X in H], "Q": pt_le(c, Q),
                "G_factors": [fr_le(c, x) for x in Gf], "H_factors": [fr_le(c, x) for x in Hf],
                "L": [pt_le(c, X) for X in Lv], "R": [pt_le(c, X) for X in Rv],
                "L_amcl": [h(c.g1_to_bytes(X)) for X in Lv],
                "a_out": fr_le(c, a0), "b_out": fr_le(c, b0), "P": pt_le(c, P),
                "transcript_after": h(after),
            })
            print("  ipp", c.name, name, file=sys.stderr)
        out[c.name] = cases
    return out


def gen_hash_to_g1():
    """G1::from_msg_hash / get_generators (src/utils/mod.rs:16-23) restated in pyref.g1_from_msg_hash; SHAKE256 digests are
    cross-checked against hashlib here so that at least the hash half is pinned by an independent implementation."""
    import hashlib
    msgs = [b"", b"g", b"h", b"G1", b"H1", b"G4096", b"bulletproofs", b"a" * 135, b"b" * 136, b"c" * 137, b"d" * 300]
    out = {"shake256": [], "curves": {}}
    for m in msgs:
        d = R.shake256(m, 48)
        assert d == hashlib.shake_256(m).digest(48)
        out["shake256"].append({"msg": h(m), "digest48": h(d)})
    for c in (R.BLS12_381, R.BN254):
        pts = []
        for m in msgs:
            P = R.g1_from_msg_hash(c, m)
            assert c.on_curve(P) and c.mul_raw(c.r, P) is None
            pts.append({"msg": h(m), "point": pt_le(c, P)})
        gens = {}
        for prefix, n in (("G", 12), ("H", 12), ("x" * 131, 3)):
            gens[prefix] = [pt_le(c, P) for P in R.get_generators(c, prefix, n)]
        out["curves"][c.name] = {"from_msg_hash": pts, "get_generators": gens}
    return out


def gen_r1cs():
    """Whole R1CS proofs from pyref.r1cs_prove / r1cs_verify (restated from src/r1cs/prover.rs:322-593, verifier.rs:267-457):
    the reference's own test shapes at small sizes, with the randomness the reference draws from its RNG fixed by the seed.
      factors_2          tests/r1cs.rs:16-70   (two p * q = r statements via `multiply`, 8 generators)
      bound_check_3bit   src/r1cs/gadgets/bound_check.rs:188-230 shrunk to 3 bits (6 gates -> padded to 8)
      bound_chain_2x4    tests/multiple_constraint_systems.rs:25-97 shape: two chained bound checks of 4 bits in ONE prover
                         (16 gates: no padding) -- BASELINE config 3 is this with 1024 checks of 32 bits
      one_gate           a single multiplier (padded_n = 1: the inner-product argument has zero rounds)
      no_commitments     m = 0
    Each case: the circuit as flat terms, generators, witness, randomness, commitments V, the proof bytes (C-ABI layout) and the
    scalars of the verifier's single MSM for a fixed verifier weight r."""
    out = {}
    for c in (R.BLS12_381, R.BN254):
        rng = R.SplitMix64(SEED + 900 + c.curve_id)
        g, hh = R.g1_from_msg_hash(c, b"g"), R.g1_from_msg_hash(c, b"h")
        cases = []

        def build_factors(cs, comms):
            outs = []
            for (p_, q_, r_) in ((17, 19, 323), (7, 5, 35)):
                if comms is None:
                    Vp, vp = cs.commit(p_, rng.scalar(c))
                    Vq, vq = cs.commit(q_, rng.scalar(c))
                    outs += [Vp, Vq]
                else:
                    vp, vq = cs.commit(comms.pop(0)), cs.commit(comms.pop(0))
                _, _, o = cs.multiply([(vp, 1)], [(vq, 1)])
                cs.constrain(R.lc_sub(c, [(o, 1)], R.lc_scalar(c, r_)))
            return outs

        def build_bound3(cs, comms):
            if comms is None:
                return R.prove_bounded_num(cs, 13, rng.scalar(c), 10, 17, 3, rng.scalar(c), rng.scalar(c))
            R.verify_bounded_num(cs, 10, 17, 3, comms)

        def build_chain(cs, comms):
            outs = []
            for val, lo, hi in ((9, 3, 18), (200, 190, 205)):
                if comms is None:
                    outs += R.prove_bounded_num(cs, val, rng.scalar(c), lo, hi, 4, rng.scalar(c), rng.scalar(c))
                else:
                    R.verify_bounded_num(cs, lo, hi, 4, [comms.pop(0) for _ in range(3)])
            return outs

        def build_one_gate(cs, comms):
            if comms is None:
                Vr, vr = cs.commit(12, rng.scalar(c))
                a, b, o = cs.allocate_multiplier(3, 4)
                outs = [Vr]
            else:
                vr = cs.commit(comms.pop(0))
                a, b, o = cs.allocate_multiplier()
                outs = None
            cs.constrain(R.lc_sub(c, [(o, 1)], [(vr, 1)]))
            cs.constrain(R.lc_sub(c, [(a, 1)], R.lc_scalar(c, 3)))
            return outs

        def build_no_commitments(cs, comms):
            vals = ((2, 3), (5, 7), (11, 13))
            for l_, r_ in vals:
                a, b, o = cs.allocate_multiplier(l_, r_) if comms is None else cs.allocate_multiplier()
                cs.constrain(R.lc_sub(c, [(o, 1)], R.lc_scalar(c, l_ * r_)))
                cs.constrain(R.lc_sub(c, [(a, 2), (b, c.r - 1)], R.lc_scalar(c, 2 * l_ - r_)))
            return []

        for name, label, ngens, build in (("factors_2", b"Factors", 8, build_factors), ("bound_check_3bit", b"BoundsTest", 8, build_bound3),
                                          ("bound_chain_2x4", b"BoundsChain", 16, build_chain), ("one_gate", b"OneGate", 2, build_one_gate),
                                          ("no_commitments", b"NoComm", 4, build_no_commitments)):
            G, H = R.get_generators(c, "G", ngens), R.get_generators(c, "H", ngens)
            prover = R.R1CSProver(c, g, hh, R.Transcript(label))
            V = build(prover, None)
            n, m = len(prover.aL), len(prover.v)
            rand = {"i_blinding1": rng.scalar(c), "o_blinding1": rng.scalar(c), "s_blinding1": rng.scalar(c),
                    "s_L1": [rng.scalar(c) for _ in range(n)], "s_R1": [rng.scalar(c) for _ in range(n)]}
            for k in (1, 3, 4, 5, 6):
                rand["t_%d_blinding" % k] = rng.scalar(c)
            assert all(prover.eval(lc) == 0 for lc in prover.constraints), "fixture circuit is not satisfied"
            proof = R.r1cs_prove(prover, G, H, rand)
            rv = rng.scalar(c)
            ver = R.R1CSVerifier(c, R.Transcript(label))
            build(ver, list(V))
            assert ver.constraints == prover.constraints and ver.num_vars == n
            sc, pts = R.r1cs_verifier_msm(ver, proof, g, hh, G, H, rv)
            assert c.msm(sc, pts) is None, "fixture proof does not verify"
            bad = dict(proof, t_x=(proof["t_x"] + 1) % c.r)
            ver2 = R.R1CSVerifier(c, R.Transcript(label))
            build(ver2, list(V))
            assert not R.r1cs_verify(ver2, bad, g, hh, G, H, rv)
            terms = R.constraints_to_terms(prover.constraints)
            cases.append({
                "name": name, "label": h(label), "n": n, "m": m, "n_constraints": len(prover.constraints), "n_generators": ngens,
                "terms": [[q, k, i, fr_le(c, cf)] for q, k, i, cf in terms],
                "g": pt_le(c, g), "h": pt_le(c, hh), "G": [pt_le(c, P) for P in G], "H": [pt_le(c, P) for P in H],
                "V": [pt_le(c, P) for P in V],
                "a_L": [fr_le(c, x) for x in prover.aL], "a_R": [fr_le(c, x) for x in prover.aR], "a_O": [fr_le(c, x) for x in prover.aO],
                "v": [fr_le(c, x) for x in prover.v], "v_blinding": [fr_le(c, x) for x in prover.v_blinding],
                "s_L": [fr_le(c, x) for x in rand["s_L1"]], "s_R": [fr_le(c, x) for x in rand["s_R1"]],
                "blindings": [fr_le(c, rand[k]) for k in ("i_blinding1", "o_blinding1", "s_blinding1", "t_1_blinding", "t_3_blinding",
                                                          "t_4_blinding", "t_5_blinding", "t_6_blinding")],
                "proof": h(R.r1cs_proof_to_le(c, proof)),
                "verifier_r": fr_le(c, rv), "verifier_msm_scalars": [fr_le(c, x) for x in sc],
            })
            print("  r1cs", c.name, name, "gates", n, file=sys.stderr)
        out[c.name] = cases
    return out


def gen_r1cs2():
    """Two-phase (randomised) R1CS proofs: pyref.r1cs_prove with deferred constraints (src/r1cs/prover.rs:298-319,383-431,
    verifier.rs:245-263).  The circuit is pyref.shuffle_gadget (a challenge z drawn after A_I1 / A_O1 / S1, then
    prod (x_i - z) = prod (y_i - z)); the reference ships the mechanism but no gadget using it.
      shuffle_2               no first-phase multipliers at all (n1 = 0: A_I1 = i_blinding1 h), two in the second phase
      product_then_shuffle_3  one first-phase gate (x0 x1 = p), four second-phase gates, 5 gates padded to 8
    Each case holds what the split C ABI (bp_r1cs_prove_begin / _finish, bp_r1cs_verify_begin / _finish) needs: the first-phase part,
    the challenge the circuit draws (label and expected value) and the complete system as flat terms."""
    out = {}
    for c in (R.BLS12_381, R.BN254):
        rng = R.SplitMix64(SEED + 950 + c.curve_id)
        g, hh = R.g1_from_msg_hash(c, b"g"), R.g1_from_msg_hash(c, b"h")
        cases = []

        def build_shuffle2(cs, comms):
            xs_val, ys_val = (5, 9), (9, 5)
            if comms is None:
                cm = [cs.commit(v, rng.scalar(c)) for v in xs_val + ys_val]
                V, vars_ = [a for a, _ in cm], [b for _, b in cm]
            else:
                V, vars_ = None, [cs.commit(P) for P in comms]
            R.shuffle_gadget(cs, vars_[:2], vars_[2:])
            return V

        def build_product_shuffle3(cs, comms):
            xs_val, ys_val, pv = (3, 11, 7), (7, 3, 11), 33
            if comms is None:
                cm = [cs.commit(v, rng.scalar(c)) for v in xs_val + ys_val + (pv,)]
                V, vars_ = [a for a, _ in cm], [b for _, b in cm]
            else:
                V, vars_ = None, [cs.commit(P) for P in comms]
            _, _, o = cs.multiply([(vars_[0], 1)], [(vars_[1], 1)])                  # first phase: x0 x1 = p
            cs.constrain(R.lc_sub(c, [(o, 1)], [(vars_[6], 1)]))
            R.shuffle_gadget(cs, vars_[:3], vars_[3:6])
            return V

        for name, label, ngens, build in (("shuffle_2", b"Shuffle2", 2, build_shuffle2), ("product_then_shuffle_3", b"ProductShuffle3", 8, build_product_shuffle3)):
            G, H = R.get_generators(c, "G", ngens), R.get_generators(c, "H", ngens)
            tr = R.Transcript(label)
            prover = R.R1CSProver(c, g, hh, tr)
            V = build(prover, None)
            n1, m, q1 = len(prover.aL), len(prover.v), len(prover.constraints)
            seen = {}
            real_challenge = tr.challenge_scalar
            def spy(curve, lbl, _real=real_challenge, _seen=seen):
                v = _real(curve, lbl)
                if lbl == b"shuffle challenge":
                    _seen["z"] = v
                return v
            tr.challenge_scalar = spy
            # the second-phase sizes are known from the circuit shape: k - 1 multipliers per side
            k = 2 if name == "shuffle_2" else 3
            n2 = 2 * (k - 1)
            rand = {"i_blinding1": rng.scalar(c), "o_blinding1": rng.scalar(c), "s_blinding1": rng.scalar(c),
                    "s_L1": [rng.scalar(c) for _ in range(n1)], "s_R1": [rng.scalar(c) for _ in range(n1)],
                    "i_blinding2": rng.scalar(c), "o_blinding2": rng.scalar(c), "s_blinding2": rng.scalar(c),
                    "s_L2": [rng.scalar(c) for _ in range(n2)], "s_R2": [rng.scalar(c) for _ in range(n2)]}
            for kk in (1, 3, 4, 5, 6):
                rand["t_%d_blinding" % kk] = rng.scalar(c)
            proof = R.r1cs_prove(prover, G, H, rand)
            n = len(prover.aL)
            assert n - n1 == n2 and "z" in seen
            assert all(prover.eval(lc) == 0 for lc in prover.constraints), "fixture circuit is not satisfied"
            rv = rng.scalar(c)
            ver = R.R1CSVerifier(c, R.Transcript(label))
            build(ver, list(V))
            sc, pts = R.r1cs_verifier_msm(ver, proof, g, hh, G, H, rv)
            assert ver.constraints == prover.constraints and ver.num_vars == n
            assert c.msm(sc, pts) is None, "fixture proof does not verify"
            # a y that is not a permutation of x must not verify: swap in a proof made for other commitments
            bad = dict(proof, t_x=(proof["t_x"] + 1) % c.r)
            ver2 = R.R1CSVerifier(c, R.Transcript(label))
            build(ver2, list(V))
            assert not R.r1cs_verify(ver2, bad, g, hh, G, H, rv)
            terms = R.constraints_to_terms(prover.constraints)
            cases.append({
                "name": name, "label": h(label), "n1": n1, "n2": n2, "m": m, "n_constraints_phase1": q1, "n_constraints": len(prover.constraints),
                "n_generators": ngens, "challenge_label": h(b"shuffle challenge"), "challenge": fr_le(c, seen["z"]),
                "terms": [[q, kd, i, fr_le(c, cf)] for q, kd, i, cf in terms],
                "g": pt_le(c, g), "h": pt_le(c, hh), "G": [pt_le(c, P) for P in G], "H": [pt_le(c, P) for P in H],
                "V": [pt_le(c, P) for P in V],
                "a_L": [fr_le(c, x) for x in prover.aL], "a_R": [fr_le(c, x) for x in prover.aR], "a_O": [fr_le(c, x) for x in prover.aO],
                "v": [fr_le(c, x) for x in prover.v], "v_blinding": [fr_le(c, x) for x in prover.v_blinding],
                "s_L": [fr_le(c, x) for x in rand["s_L1"] + rand["s_L2"]], "s_R": [fr_le(c, x) for x in rand["s_R1"] + rand["s_R2"]],
                "blindings": [fr_le(c, rand[kk]) for kk in ("i_blinding1", "o_blinding1", "s_blinding1", "i_blinding2", "o_blinding2", "s_blinding2",
                                                            "t_1_blinding", "t_3_blinding", "t_4_blinding", "t_5_blinding", "t_6_blinding")],
                "proof": h(R.r1cs_proof_to_le(c, proof)),
                "verifier_r": fr_le(c, rv), "verifier_msm_scalars": [fr_le(c, x) for x in sc],
            })
            print("  r1cs2", c.name, name, "gates", n1, "+", n2, file=sys.stderr)
        out[c.name] = cases
    return out


def gen_compressed():
    """bp_g1vec_compress / _decompress (this build's tag || X form, pyref.g1_compress): multiples of the generator with both y
    parities, the identity, hashed points; and encodings that must be refused."""
    out = {}
    for c in (R.BLS12_381, R.BN254):
        rng = R.SplitMix64(SEED + 700 + c.curve_id)
        pts = [None, c.g, c.neg(c.g), c.mul(2, c.g), c.mul(c.r - 1, c.g)] + [c.mul(rng.scalar(c), c.g) for _ in range(8)]
        pts += [R.g1_from_msg_hash(c, m) for m in (b"g", b"h", b"Q")] + [None]
        cases = []
        for P in pts:
            enc = R.g1_compress(c, P)
            assert R.g1_decompress(c, enc) == P
            cases.append({"point": pt_le(c, P), "compressed": h(enc)})
        assert {bytes.fromhex(x["compressed"])[0] for x in cases} == {0, 2, 3}
        mb = c.modbytes
        x_off = next(x for x in range(1, 200) if pow((x ** 3 + c.b) % c.p, (c.p - 1) // 2, c.p) != 1)      # not an abscissa
        invalid = [{"why": "x is not an abscissa of the curve", "bytes": h(bytes([2]) + x_off.to_bytes(mb, "big"))},
                   {"why": "x >= p", "bytes": h(bytes([3]) + (c.p + 1 if (c.p + 1).bit_length() <= 8 * mb else c.p).to_bytes(mb, "big"))},
                   {"why": "x = p", "bytes": h(bytes([2]) + c.p.to_bytes(mb, "big"))},
                   {"why": "unknown tag", "bytes": h(bytes([4]) + c.g[0].to_bytes(mb, "big"))},
                   {"why": "identity tag with a non-zero body", "bytes": h(bytes([0]) + (1).to_bytes(mb, "big"))}]
        for bad in invalid:
            try:
                R.g1_decompress(c, bytes.fromhex(bad["bytes"]))
                raise AssertionError("accepted: " + bad["why"])
            except ValueError:
                pass
        out[c.name] = {"cases": cases, "invalid": invalid}
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in (("curves", gen_curves), ("field", gen_field), ("g1", gen_g1), ("merlin", gen_merlin),
                     ("msm", gen_msm), ("ipp", gen_ipp), ("hash_to_g1", gen_hash_to_g1), ("r1cs", gen_r1cs), ("r1cs2", gen_r1cs2), ("compressed", gen_compressed)):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue                                                   # python3 oracle/gen_golden.py r1cs  -> only that file
        print("generating", name, file=sys.stderr)
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fn(), f, indent=0, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
