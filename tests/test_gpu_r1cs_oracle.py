"""R1CS layer against the INDEPENDENT oracle: `bp_r1cs_prove` / `bp_r1cs_verify` (csrc/bp_capi_r1cs.hip) are compared with
whole proofs made outside the product --
  * tests/golden/r1cs.json: Python-int proofs of the reference's own test shapes (oracle/pyref.py r1cs_prove, restated from
    /root/reference src/r1cs/prover.rs:322-593 / verifier.rs:267-457), byte for byte;
  * BASELINE config 3 at full size: 1024 chained 32-bit bound checks (65 536 gates, 136 192 constraints, m = 3 072) built by the
    oracle's restatement of the gadgets (src/r1cs/gadgets/bound_check.rs:13-39, helper_constraints/positive_no.rs:8-40), proven
    by the library and by the C oracle (oracle/orc_r1cs_tmpl.h): same bytes, and each verifier accepts the other's proof;
  * an inner-product argument at n = 2^16 against the C oracle's create_ipp / verify_ipp;
  * tests/golden/r1cs2.json: TWO-PHASE (randomised) systems -- a shuffle gadget whose constraints depend on a challenge drawn after the
    first-phase commitments (prover.rs:298-319,383-431; verifier.rs:245-263) -- through bp_r1cs_prove_begin / _finish and
    bp_r1cs_verify_begin / _finish, byte for byte against oracle/pyref.py.
PARITY UNPINNED w.r.t. the reference itself: the oracle is a restatement."""
import os
import sys

import pytest

import __graft_entry__ as G
import _oracle as O
from test_oracle_golden import r1cs_case_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu
CURVES = ["bls12_381", "bn254"]


@pytest.fixture(scope="module")
def bp():
    return G.load_package()


def host_threads():
    try:
        k = len(os.sched_getaffinity(0))
    except AttributeError:
        k = os.cpu_count() or 1
    return max(1, min(k, 32))


def start_transcript(bp, ctx, label, V):
    t = bp.Transcript(label)
    t.append_message(b"dom-sep", b"r1cs v1")
    for Vj in V:
        t.commit_point(ctx.curve, b"V", Vj)
    return t


@pytest.mark.parametrize("name", CURVES)
def test_r1cs_golden_fixture(bp, golden, name):
    ctx = bp.Context(bp.CURVE_IDS[name], 0)
    pb = ctx.point_bytes
    for c in golden("r1cs")[name]:
        a = r1cs_case_inputs(c)
        n, m, ng = c["n"], c["m"], c["n_generators"]
        plan = bp.R1CSPlan(ctx, a["terms"], c["n_constraints"], n, m)
        Gv, Hv = bp.G1Vector.from_bytes(ctx, a["G"], ng), bp.G1Vector.from_bytes(ctx, a["H"], ng)
        fe = lambda b, k: bp.FieldElementVector.from_bytes(ctx, b, k)
        proof = bp.r1cs_prove(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], fe(a["aL"], n), fe(a["aR"], n),
                              fe(a["aO"], n), fe(a["vb"], m) if m else None, fe(a["sL"], n), fe(a["sR"], n), a["blind"])
        assert proof == a["proof"], (name, c["name"])
        Vb = b"".join(a["V"])
        bp.r1cs_verify(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], Vb, n, a["proof"], a["r"])
        # the G / H scalars of the verifier's single MSM (verifier.rs:368-390) against the fixture's
        lg = max(0, (n - 1).bit_length())
        pn = 1 << lg
        L, R = a["proof"][11 * pb + 96:11 * pb + 96 + lg * pb], a["proof"][11 * pb + 96 + lg * pb:11 * pb + 96 + 2 * lg * pb]
        pa, pbb = a["proof"][-64:-32], a["proof"][-32:]
        t = start_transcript(bp, ctx, a["label"], a["V"])
        t.append_u64(b"m", m)
        ident = bytes(pb)
        for label, P in ((b"A_I1", a["proof"][:pb]), (b"A_O1", a["proof"][pb:2 * pb]), (b"S1", a["proof"][2 * pb:3 * pb])):
            t.commit_point(ctx.curve, label, P)
        t.append_message(b"dom-sep", b"r1cs-1phase")
        for label in (b"A_I2", b"A_O2", b"S2"):
            t.commit_point(ctx.curve, label, ident)
        y = int.from_bytes(t.challenge_scalar(ctx.curve, b"y"), "little")
        z = t.challenge_scalar(ctx.curve, b"z")
        for k, label in enumerate((b"T_1", b"T_3", b"T_4", b"T_5", b"T_6")):
            t.commit_point(ctx.curve, label, a["proof"][(6 + k) * pb:(7 + k) * pb])
        u = t.challenge_scalar(ctx.curve, b"u")
        x = t.challenge_scalar(ctx.curve, b"x")
        for k, label in enumerate((b"t_x", b"t_x_blinding", b"e_blinding")):
            t.commit_scalar(ctx.curve, label, a["proof"][11 * pb + 32 * k:11 * pb + 32 * k + 32])
        t.challenge_scalar(ctx.curve, b"w")
        wL, wR, wO, wV, wc = plan.flattened_constraints(z)
        y_inv = pow(y, -1, ctx.r).to_bytes(32, "little")
        _, _, g_sc, h_sc = bp.r1cs_verifier_scalars(ctx, t, L, R, pn, n, wL, wR, wO, y_inv, x, u, pa, pbb)
        want = [bytes.fromhex(s) for s in c["verifier_msm_scalars"]]
        off = 6 + m + 5 + 2
        assert g_sc.to_bytes() == b"".join(want[off:off + pn]) and h_sc.to_bytes() == b"".join(want[off + pn:off + 2 * pn]), (name, c["name"])
        # rejected: a changed scalar, a changed statement, too few generators
        bad = bytearray(a["proof"])
        bad[11 * pb] ^= 1
        with pytest.raises(bp.VerificationError):
            bp.r1cs_verify(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], Vb, n, bytes(bad), a["r"])
        if m:
            with pytest.raises(bp.VerificationError):
                bp.r1cs_verify(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], O.generator(ctx.curve) + Vb[pb:], n,
                               a["proof"], a["r"])
        # the same verdicts with the [G | H] terms over the generators' window tables (verifier.rs:431-451; the knob brings the
        # size limit down to the fixtures'), the proof's and the statement's points as the small MSM beside them
        if pn >= 2:
            ctx.set_tuning(bp.TUNE_VERIFY_TABLES, 2)
            Gv.precompute(8)
            Hv.precompute(8)
            bp.r1cs_verify(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], Vb, n, a["proof"], a["r"])
            assert ctx.verify_table_info()[0] == pn
            with pytest.raises(bp.VerificationError):
                bp.r1cs_verify(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], Vb, n, bytes(bad), a["r"])
            if m:
                with pytest.raises(bp.VerificationError):
                    bp.r1cs_verify(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], O.generator(ctx.curve) + Vb[pb:], n,
                                   a["proof"], a["r"])
            ctx.set_tuning(bp.TUNE_VERIFY_TABLES, 0)
        plan.free()
    ctx.close()


@pytest.mark.parametrize("name", CURVES)
def test_r1cs_two_phase_fixture(bp, golden, name):
    """Deferred (randomised) constraints through the split API: begin commits the first phase, the test plays the callback (draws the
    circuit's challenge from the SAME transcript and checks it against the oracle's), finish commits the second phase over
    G[n1..n), H[n1..n) and completes the proof; both verifier halves likewise.  n1 = 0 (no first-phase multiplier) is one of the cases."""
    ctx = bp.Context(bp.CURVE_IDS[name], 0)
    pb = ctx.point_bytes
    for c in golden("r1cs2")[name]:
        a = r1cs_case_inputs(c)
        n1, n2, m, ng = c["n1"], c["n2"], c["m"], c["n_generators"]
        n = n1 + n2
        Gv, Hv = bp.G1Vector.from_bytes(ctx, a["G"], ng), bp.G1Vector.from_bytes(ctx, a["H"], ng)
        fe = lambda b, k: bp.FieldElementVector.from_bytes(ctx, b, k) if k else None
        blind = a["blind"]                                   # i1 o1 s1 | i2 o2 s2 | t1 t3 t4 t5 t6
        t = start_transcript(bp, ctx, a["label"], a["V"])
        phase1 = bp.r1cs_prove_begin(ctx, t, Gv, Hv, a["h"], m, fe(a["aL"][:32 * n1], n1), fe(a["aR"][:32 * n1], n1), fe(a["aO"][:32 * n1], n1),
                                     fe(a["sL"][:32 * n1], n1), fe(a["sR"][:32 * n1], n1), blind[:96])
        z = t.challenge_scalar(ctx.curve, bytes.fromhex(c["challenge_label"]))          # the callback's challenge_scalar
        assert z == bytes.fromhex(c["challenge"]), (name, c["name"], "the circuit's challenge differs from the oracle's")
        plan = bp.R1CSPlan(ctx, a["terms"], c["n_constraints"], n, m)                    # the complete system, second-phase terms included
        proof = bp.r1cs_prove_finish(ctx, t, plan, Gv, Hv, a["g"], a["h"], phase1, fe(a["aL"], n), fe(a["aR"], n), fe(a["aO"], n),
                                     fe(a["vb"], m), fe(a["sL"], n), fe(a["sR"], n), blind[96:])
        assert proof == a["proof"], (name, c["name"])
        assert proof[3 * pb:6 * pb] != bytes(3 * pb)                                     # A_I2, A_O2, S2 are real commitments here
        Vb = b"".join(a["V"])

        def verify(pr, V=Vb, n1_=n1):
            tv = start_transcript(bp, ctx, a["label"], [V[k * pb:(k + 1) * pb] for k in range(m)])
            bp.r1cs_verify_begin(ctx, tv, m, pr)
            assert tv.challenge_scalar(ctx.curve, bytes.fromhex(c["challenge_label"])) is not None
            bp.r1cs_verify_finish(ctx, tv, plan, Gv, Hv, a["g"], a["h"], V, n1_, n, pr, a["r"])

        verify(a["proof"])
        for pos in (0, 3 * pb, 4 * pb + 1, 11 * pb, len(a["proof"]) - 1):                # A_I1, A_I2, A_O2, t_x, b
            bad = bytearray(a["proof"])
            bad[pos] ^= 1
            with pytest.raises(bp.VerificationError):
                verify(bytes(bad))
        with pytest.raises(bp.VerificationError):                                        # the phase boundary is part of the statement (G_factors)
            verify(a["proof"], n1_=n1 + 1)
        with pytest.raises(bp.VerificationError):                                        # another commitment: y is no longer a permutation of x
            verify(a["proof"], V=O.generator(ctx.curve) + Vb[pb:])
        # a single-phase call on the same data must not produce or accept this proof (domain separator, G_factors)
        with pytest.raises(bp.VerificationError):
            bp.r1cs_verify(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], Vb, n, a["proof"], a["r"])
        with pytest.raises(bp.ArgError):
            bp.r1cs_prove_finish(ctx, t, plan, Gv, Hv, a["g"], a["h"], bytes(len(phase1)), fe(a["aL"], n), fe(a["aR"], n), fe(a["aO"], n),
                                 fe(a["vb"], m), fe(a["sL"], n), fe(a["sR"], n), blind[96:])
        plan.free()
    ctx.close()


def oracle_bound_check_chain(R, c, checks, bits, seed):
    """The config-3 circuit from the ORACLE's gadgets (pyref.prove_bounded_num); the commitments themselves are made by the caller."""
    rng = R.SplitMix64(seed)

    class NullTranscript:
        def append_message(self, *a):
            pass

        def commit_point(self, *a):
            pass

    class CircuitOnly(R.R1CSProver):
        def commit(self, v, vb):                       # Prover::commit without the group arithmetic (done in bulk by the caller)
            self.v.append(v % c.r)
            self.v_blinding.append(vb % c.r)
            return None, (R.V_COMMITTED, len(self.v) - 1)

    cs = CircuitOnly(c, None, None, NullTranscript())
    triples = []
    for _ in range(checks):
        lo = rng.next() % (1 << 20)
        hi = lo + (1 << bits) - 1 - rng.next() % (1 << 10)
        val = lo + rng.next() % (hi - lo + 1)
        triples.append((val, lo, hi))
        R.prove_bounded_num(cs, val, rng.scalar(c), lo, hi, bits, rng.scalar(c), rng.scalar(c))
    return cs, triples


def test_cfg3_full_chain_vs_oracle(bp):
    """BASELINE config 3: R1CS prover + verifier at 2^16 multiplication gates (r1cs/gadgets bound-check chain)."""
    import pyref as R
    import bench_configs as BC
    c, cid = R.BLS12_381, 0
    checks, bits = 1024, 32
    cs, triples = oracle_bound_check_chain(R, c, checks, bits, 2024)
    n, m, nq = len(cs.aL), len(cs.v), len(cs.constraints)
    assert (n, m, nq) == (65536, 3072, 136192)
    terms = R.constraints_to_terms(cs.constraints)
    assert len(terms) == 338944
    # the product-side generator (bench_configs.py, used by the cfg3 benchmark) builds the same circuit and witness
    pterms, pnq, paL, paR, paO, pv = BC.bound_check_chain(c.r, checks, bits, None, triples=triples)
    norm = lambda ts: sorted((q, k, i if k != 4 else 0, cf % c.r) for q, k, i, cf in ts)
    assert pnq == nq and norm(pterms) == norm(terms) and paL == cs.aL and paR == cs.aR and paO == cs.aO and pv == cs.v

    thr = host_threads()
    O.set_threads(thr)
    try:
        le = lambda xs: b"".join(int(x).to_bytes(32, "little") for x in xs)
        Gb, Hb = O.get_generators(cid, "G", n, nthreads=thr), O.get_generators(cid, "H", n, nthreads=thr)
        g, h = O.g1_from_msg_hash(cid, b"g"), O.g1_from_msg_hash(cid, b"h")
        V = [O.binary_scalar_mul(cid, g, h, v.to_bytes(32, "little"), b.to_bytes(32, "little")) for v, b in zip(cs.v, cs.v_blinding)]
        Vb = b"".join(V)
        aL, aR, aO, vb = le(cs.aL), le(cs.aR), le(cs.aO), le(cs.v_blinding)
        sL, sR, bl = O.random_scalars(cid, 11, n), O.random_scalars(cid, 12, n), O.random_scalars(cid, 13, 8)
        T = O.R1CSTerms(terms, nq, n, m)

        ctx = bp.Context(cid, 0)
        # the library's hashed generators and batched commitments equal the oracle's
        Gv, Hv = bp.get_generators(ctx, "G", n), bp.get_generators(ctx, "H", n)
        assert Gv.to_bytes() == Gb and Hv.to_bytes() == Hb
        fe = lambda b, k: bp.FieldElementVector.from_bytes(ctx, b, k)
        assert bp.G1Vector.commit_pairs(ctx, g, h, fe(le(cs.v), m), fe(vb, m)).to_bytes() == Vb
        plan = bp.R1CSPlan(ctx, terms, nq, n, m)
        proof = bp.r1cs_prove(ctx, start_transcript(bp, ctx, b"cfg3", V), plan, Gv, Hv, g, h, fe(aL, n), fe(aR, n), fe(aO, n), fe(vb, m), fe(sL, n),
                              fe(sR, n), bl)
        assert len(proof) == O.r1cs_proof_bytes(cid, n) == 4288
        rnd = O.random_scalars(cid, 14, 1)
        # 1. the oracle's verifier (one 134 189-term MSM on the host) accepts the library's proof
        assert O.r1cs_verify(cid, O.r1cs_start_transcript(cid, b"cfg3", V), T, Vb, proof, g, h, Gb, Hb, n, rnd) == 0
        # 2. the oracle's prover (reference-shaped: it folds the generators every round) produces the same bytes
        rc, want = O.r1cs_prove(cid, O.r1cs_start_transcript(cid, b"cfg3", V), T, g, h, Gb, Hb, n, aL, aR, aO, vb, sL, sR, bl)
        assert rc == 0 and proof == want
        # 2b. with window-multiples tables on the generators (bp_g1vec_precompute: merged-window MSMs for the commitments and every
        # IPP round) the prover must produce the same bytes
        Gv.precompute(16)
        Hv.precompute(16)
        assert Gv.table_info()[:2] == (16, 16)
        proof_t = bp.r1cs_prove(ctx, start_transcript(bp, ctx, b"cfg3", V), plan, Gv, Hv, g, h, fe(aL, n), fe(aR, n), fe(aO, n), fe(vb, m), fe(sL, n),
                                fe(sR, n), bl)
        assert proof_t == want
        # 3. the library's verifier accepts it, and both reject a changed proof / statement.  G and H carry tables here: with the knob on
        # their terms run over them, the 3 072 commitments and the proof's points as a second MSM beside; then the plain path
        ctx.set_tuning(bp.TUNE_VERIFY_TABLES, 4096)
        bp.r1cs_verify(ctx, start_transcript(bp, ctx, b"cfg3", V), plan, Gv, Hv, g, h, Vb, n, proof, rnd)
        assert ctx.verify_table_info()[0] == n
        pb = ctx.point_bytes
        for off in (11 * pb + 5, 7 * pb + 1, len(proof) - 40):            # t_x, T_3, a
            bad = bytearray(proof)
            bad[off] ^= 4
            with pytest.raises((bp.VerificationError, bp.ArgError)):
                bp.r1cs_verify(ctx, start_transcript(bp, ctx, b"cfg3", V), plan, Gv, Hv, g, h, Vb, n, bytes(bad), rnd)
        bad = bytearray(proof)
        bad[11 * pb + 5] ^= 4
        assert O.r1cs_verify(cid, O.r1cs_start_transcript(cid, b"cfg3", V), T, Vb, bytes(bad), g, h, Gb, Hb, n, rnd) == 3
        V2 = [V[1], V[0]] + V[2:]
        with pytest.raises(bp.VerificationError):
            bp.r1cs_verify(ctx, start_transcript(bp, ctx, b"cfg3", V2), plan, Gv, Hv, g, h, b"".join(V2), n, proof, rnd)
        ctx.set_tuning(bp.TUNE_VERIFY_TABLES, 0)
        bp.r1cs_verify(ctx, start_transcript(bp, ctx, b"cfg3", V), plan, Gv, Hv, g, h, Vb, n, proof, rnd)
        with pytest.raises(bp.VerificationError):
            bp.r1cs_verify(ctx, start_transcript(bp, ctx, b"cfg3", V2), plan, Gv, Hv, g, h, b"".join(V2), n, proof, rnd)
        plan.free()
        ctx.close()
    finally:
        O.set_threads(1)


def test_ipp_2p16_vs_oracle(bp):
    """IPP::create_ipp / verify_ipp at n = 2^16 (the inner-product argument inside config 3) with random G_factors / H_factors:
    the library's proof equals the C oracle's (which folds the generators as src/ipp.rs:115-130,181-188 does)."""
    cid, n = 0, 1 << 16
    thr = host_threads()
    O.set_threads(thr)
    try:
        ctx = bp.Context(cid, 0)
        ks = O.random_scalars(cid, 501, 2 * n + 1)
        pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, 2 * n + 1)).to_bytes()
        pb = ctx.point_bytes
        Gb, Hb, Q = pts[:n * pb], pts[n * pb:2 * n * pb], pts[2 * n * pb:]
        a, b, Gf, Hf = (O.random_scalars(cid, 502 + i, n) for i in range(4))
        fe = lambda x: bp.FieldElementVector.from_bytes(ctx, x, n)
        Gv, Hv = bp.G1Vector.from_bytes(ctx, Gb, n), bp.G1Vector.from_bytes(ctx, Hb, n)
        proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"ipp 2^16"), Q, fe(Gf), fe(Hf), Gv, Hv, fe(a), fe(b))
        rc, want = O.ipp_create(cid, O.Transcript(b"ipp 2^16"), Q, Gf, Hf, Gb, Hb, a, b, n)
        assert rc == 0 and (proof.L, proof.R, proof.a, proof.b) == want
        # the same proof with precomputed generators (every round a merged-window MSM over the tables), two table widths
        for c in (16, 13):
            Gv.precompute(c)
            Hv.precompute(c)
            pt = bp.IPP.create_ipp(ctx, bp.Transcript(b"ipp 2^16"), Q, fe(Gf), fe(Hf), Gv, Hv, fe(a), fe(b))
            assert (pt.L, pt.R, pt.a, pt.b) == want, c
        # P = <a, Gf o G> + <b, Hf o H> + <a, b> Q (src/ipp.rs:353-372) and both verifiers (the library's over the tables, then without)
        sc = fe(a).hadamard_product(fe(Gf)).to_bytes() + fe(b).hadamard_product(fe(Hf)).to_bytes() + fe(a).inner_product(fe(b))
        P = bp.G1Vector.from_bytes(ctx, pts, 2 * n + 1).multi_scalar_mul_var_time(bp.FieldElementVector.from_bytes(ctx, sc, 2 * n + 1))
        assert O.ipp_verify(cid, O.Transcript(b"ipp 2^16"), n, Gf, Hf, P, Q, Gb, Hb, proof.a, proof.b, proof.L, proof.R, proof.lg_n) == 0
        ctx.set_tuning(bp.TUNE_VERIFY_TABLES, 2)
        bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"ipp 2^16"), fe(Gf), fe(Hf), P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
        assert ctx.verify_table_info()[0] == n
        with pytest.raises(bp.VerificationError):
            bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"ipp 2^16"), fe(Gf), fe(Hf), P, Q, Gv, Hv, proof.b, proof.a, proof.L, proof.R)
        Gv.drop_table()
        Hv.drop_table()
        ctx.drop_verify_table()
        bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"ipp 2^16"), fe(Gf), fe(Hf), P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
        assert ctx.verify_table_info() == (0, 0)
        ctx.close()
    finally:
        O.set_threads(1)
