"""End-to-end composition of the hot-path pieces in the shape of BASELINE config 3: an R1CS proof is created and verified with
every vector / group operation going through the C ABI (commitment MSMs, flattened constraints, l/r polynomials, t(x),
IPP, the verifier's single MSM), the host doing only what the reference's host does: the transcript and a handful of scalars.

The orchestration below restates `Prover::prove` (/root/reference src/r1cs/prover.rs:323-560) and `Verifier::verify`
(src/r1cs/verifier.rs:265-452) for a single-phase constraint system (no randomised second phase: n2 = 0, A_I2 = A_O2 = S2 = O).
There are no reference vectors for it (the reference cannot run here), so this is a CONSISTENCY test: honest proofs verify,
and a proof / statement changed anywhere does not."""
import random

import pytest

import __graft_entry__ as G
import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bp():
    return G.load_package()


def le(x):
    return int(x).to_bytes(32, "little")


def ints(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def random_satisfiable_system(rng, r, n, m, nq):
    """Witness (a_L, a_R, a_O = a_L * a_R, v) and nq linear constraints  sum coeff * var + c * One = 0  that it satisfies."""
    aL = [rng.randrange(r) for _ in range(n)]
    aR = [rng.randrange(r) for _ in range(n)]
    aO = [x * y % r for x, y in zip(aL, aR)]
    v = [rng.randrange(r) for _ in range(m)]
    val = {0: aL, 1: aR, 2: aO, 3: v}
    cons = []
    for _ in range(nq):
        terms, acc = [], 0
        for _ in range(rng.randrange(1, 6)):
            kind = rng.choice([0, 1, 2, 3]) if m else rng.choice([0, 1, 2])
            idx = rng.randrange(m if kind == 3 else n)
            coeff = rng.choice([1, r - 1, rng.randrange(r)])
            terms.append((kind, idx, coeff))
            acc = (acc + coeff * val[kind][idx]) % r
        if acc:
            terms.append((4, 0, (-acc) % r))          # the constant that makes the combination vanish
        cons.append(terms)
    return aL, aR, aO, v, cons


class Params:
    def __init__(self, bp, ctx, n_max):
        self.g = bp.G1Vector.from_msg_hash(ctx, [b"g"]).to_bytes()        # as in the reference's tests (bound_check.rs:202-203)
        self.h = bp.G1Vector.from_msg_hash(ctx, [b"h"]).to_bytes()
        self.G = bp.get_generators(ctx, "G", n_max).to_bytes()
        self.H = bp.get_generators(ctx, "H", n_max).to_bytes()


def msm(bp, ctx, points, scalars):
    n = len(scalars)
    pv = bp.G1Vector.from_bytes(ctx, points, n)
    sv = bp.FieldElementVector.from_ints(ctx, scalars)
    return pv.multi_scalar_mul_var_time(sv)


def start_transcript(bp, ctx, V):
    t = bp.Transcript(b"R1CS e2e")
    t.append_message(b"dom-sep", b"r1cs v1")                               # r1cs_domain_sep, src/transcript.rs:35-37
    for Vj in V:
        t.commit_point(ctx.curve, b"V", Vj)                                # Prover::commit, prover.rs:118-127
    return t


def prove(bp, ctx, pp, cons, aL, aR, aO, v, v_blinding, rng):
    r, pb, cv = ctx.r, ctx.point_bytes, ctx.curve
    n, m = len(aL), len(v)
    padded_n = 1 << max(0, (n - 1).bit_length())
    V = [msm(bp, ctx, pp.g + pp.h, [v[j], v_blinding[j]]) for j in range(m)]
    t = start_transcript(bp, ctx, V)
    t.append_u64(b"m", m)                                                  # prover.rs:328
    Gn, Hn = pp.G[: n * pb], pp.H[: n * pb]
    i_bl, o_bl, s_bl = (rng.randrange(r) for _ in range(3))
    sL = [rng.randrange(r) for _ in range(n)]
    sR = [rng.randrange(r) for _ in range(n)]
    A_I1 = msm(bp, ctx, Gn + Hn + pp.h, aL + aR + [i_bl])                  # :346-354
    A_O1 = msm(bp, ctx, Gn + pp.h, aO + [o_bl])                            # :357
    S1 = msm(bp, ctx, Gn + Hn + pp.h, sL + sR + [s_bl])                    # :360-361
    O_ = bytes(pb)
    for label, P in ((b"A_I1", A_I1), (b"A_O1", A_O1), (b"S1", S1)):
        t.commit_point(cv, label, P)
    t.append_message(b"dom-sep", b"r1cs-1phase")                           # create_randomized_constraints, :304-306
    for label in (b"A_I2", b"A_O2", b"S2"):
        t.commit_point(cv, label, O_)                                      # :429-431 (identity: no second phase)
    y = int.from_bytes(t.challenge_scalar(cv, b"y"), "little")
    z = int.from_bytes(t.challenge_scalar(cv, b"z"), "little")
    terms = [(q, k, i, c) for q, ts in enumerate(cons) for k, i, c in ts]
    plan = bp.R1CSPlan(ctx, terms, len(cons), n, m)
    wL, wR, wO, wV, _ = plan.flattened_constraints(le(z), want_constant=False)   # :438
    dev = lambda xs: bp.FieldElementVector.from_ints(ctx, xs)
    l_poly, r_poly = bp.r1cs_prover_polys(ctx, dev(aL), dev(aR), dev(aO), dev(sL), dev(sR), wL, wR, wO, le(y))   # :465-486
    tc = [int.from_bytes(c, "little") for c in bp.VecPoly3.special_inner_product(l_poly, r_poly)]           # t1..t6, :488
    tb = {k: rng.randrange(r) for k in (1, 3, 4, 5, 6)}
    T = {k: msm(bp, ctx, pp.g + pp.h, [tc[k - 1], tb[k]]) for k in (1, 3, 4, 5, 6)}                         # :496-500
    for k in (1, 3, 4, 5, 6):
        t.commit_point(cv, b"T_%d" % k, T[k])
    u = int.from_bytes(t.challenge_scalar(cv, b"u"), "little")
    x = int.from_bytes(t.challenge_scalar(cv, b"x"), "little")
    tb[2] = sum(a * b for a, b in zip(ints(wV.to_bytes()), v_blinding)) % r                                  # :513
    t_x = sum(tc[k - 1] * pow(x, k, r) for k in range(1, 7)) % r
    t_x_blinding = sum(tb[k] * pow(x, k, r) for k in range(1, 7)) % r
    l_vec, r_vec, Gf, Hf = bp.r1cs_ipp_inputs(ctx, l_poly.eval(le(x)), r_poly.eval(le(x)), le(y), le(u), n, padded_n)   # :526-563
    e_blinding = x * (i_bl + x * (o_bl + x * s_bl)) % r                                                     # :539-543 with the *2 terms = 0
    for label, s in ((b"t_x", t_x), (b"t_x_blinding", t_x_blinding), (b"e_blinding", e_blinding)):
        t.commit_scalar(cv, label, le(s))
    w = int.from_bytes(t.challenge_scalar(cv, b"w"), "little")
    Q = msm(bp, ctx, pp.g, [w])                                                                               # :552
    Gp = bp.G1Vector.from_bytes(ctx, pp.G[: padded_n * pb], padded_n)
    Hp = bp.G1Vector.from_bytes(ctx, pp.H[: padded_n * pb], padded_n)
    ipp = bp.IPP.create_ipp(ctx, t, Q, Gf, Hf, Gp, Hp, l_vec, r_vec)                                          # :567-576
    plan.free()
    return V, {"A_I1": A_I1, "A_O1": A_O1, "S1": S1, "T": T, "t_x": t_x, "t_x_blinding": t_x_blinding, "e_blinding": e_blinding, "ipp": ipp}


def verify(bp, ctx, pp, cons, V, proof, n, rng):
    """True iff the single verification MSM is the identity (verifier.rs:448-451)."""
    r, pb, cv = ctx.r, ctx.point_bytes, ctx.curve
    m = len(V)
    padded_n = 1 << max(0, (n - 1).bit_length())
    O_ = bytes(pb)
    t = start_transcript(bp, ctx, V)
    t.append_u64(b"m", m)                                                  # verifier.rs:278
    for label in (b"A_I1", b"A_O1", b"S1"):
        t.commit_point(cv, label, proof[label.decode()])
    t.append_message(b"dom-sep", b"r1cs-1phase")
    for label in (b"A_I2", b"A_O2", b"S2"):
        t.commit_point(cv, label, O_)
    y = int.from_bytes(t.challenge_scalar(cv, b"y"), "little")
    z = int.from_bytes(t.challenge_scalar(cv, b"z"), "little")
    for k in (1, 3, 4, 5, 6):
        t.commit_point(cv, b"T_%d" % k, proof["T"][k])
    u = int.from_bytes(t.challenge_scalar(cv, b"u"), "little")
    x = int.from_bytes(t.challenge_scalar(cv, b"x"), "little")
    for label in (b"t_x", b"t_x_blinding", b"e_blinding"):
        t.commit_scalar(cv, label, le(proof[label.decode()]))
    w = int.from_bytes(t.challenge_scalar(cv, b"w"), "little")
    terms = [(q, k, i, c) for q, ts in enumerate(cons) for k, i, c in ts]
    plan = bp.R1CSPlan(ctx, terms, len(cons), n, m)
    wL, wR, wO, wV, wc = plan.flattened_constraints(le(z))                 # :329
    wc = int.from_bytes(wc, "little")
    ipp = proof["ipp"]
    a, b = int.from_bytes(ipp.a, "little"), int.from_bytes(ipp.b, "little")
    y_inv = pow(y, -1, r)
    delta = sum(pow(y_inv, i, r) * wr % r * wl for i, (wr, wl) in enumerate(zip(ints(wR.to_bytes()), ints(wL.to_bytes())))) % r   # :350-352
    try:
        u_sq, u_inv_sq, g_sc, h_sc = bp.r1cs_verifier_scalars(ctx, t, ipp.L, ipp.R, padded_n, n, wL, wR, wO, le(y_inv), le(x), le(u), ipp.a, ipp.b)
    except bp.VerificationError:
        return False
    rr = rng.randrange(r)                                                  # :392
    x2, x3 = x * x % r, pow(x, 3, r)
    wV_s = wV.scaled_by(le(rr * x2 % r)) if m else None                    # :416
    tx, txb, eb = proof["t_x"], proof["t_x_blinding"], proof["e_blinding"]
    scalars = [x, x2, x3, u * x % r, u * x2 % r, u * x3 % r] + (ints(wV_s.to_bytes()) if m else [])
    scalars += [rr * x % r, rr * x3 % r, rr * pow(x, 4, r) % r, rr * pow(x, 5, r) % r, rr * pow(x, 6, r) % r]   # :398-408
    scalars.append((w * (tx - a * b) + rr * (x2 * (wc + delta) - tx)) % r)                                       # :422
    scalars.append((-(eb + rr * txb)) % r)                                                                       # :425
    head = bp.FieldElementVector.from_ints(ctx, scalars)
    tail = bp.FieldElementVector.from_bytes(ctx, u_sq + u_inv_sq, len(u_sq) // 16) if u_sq else None
    sc_bytes = head.to_bytes() + g_sc.to_bytes() + h_sc.to_bytes() + (tail.to_bytes() if tail else b"")
    points = proof["A_I1"] + proof["A_O1"] + proof["S1"] + O_ * 3 + b"".join(V) + b"".join(proof["T"][k] for k in (1, 3, 4, 5, 6))
    points += pp.g + pp.h + pp.G[: padded_n * pb] + pp.H[: padded_n * pb] + ipp.L + ipp.R                       # :431-446
    total = len(sc_bytes) // 32
    assert total * pb == len(points)
    res = bp.G1Vector.from_bytes(ctx, points, total).multi_scalar_mul_var_time(bp.FieldElementVector.from_bytes(ctx, sc_bytes, total))
    plan.free()
    return res == O_


@pytest.mark.parametrize("name,n,m,nq", [("bls12_381", 13, 3, 20), ("bls12_381", 64, 4, 150), ("bn254", 5, 0, 7), ("bls12_381", 300, 2, 400)])
def test_r1cs_prove_and_verify(bp, name, n, m, nq):
    ctx = bp.Context(bp.CURVE_IDS[name], 0)
    r, pb = ctx.r, ctx.point_bytes
    rng = random.Random(31 * n + m)
    padded_n = 1 << max(0, (n - 1).bit_length())
    pp = Params(bp, ctx, padded_n)
    aL, aR, aO, v, cons = random_satisfiable_system(rng, r, n, m, nq)
    v_blinding = [rng.randrange(r) for _ in range(m)]
    V, proof = prove(bp, ctx, pp, cons, aL, aR, aO, v, v_blinding, rng)
    assert verify(bp, ctx, pp, cons, V, proof, n, rng)
    assert verify(bp, ctx, pp, cons, V, proof, n, random.Random(5))           # the verifier's own randomness does not matter

    # anything changed -> rejected
    bad = dict(proof, t_x=(proof["t_x"] + 1) % r)
    assert not verify(bp, ctx, pp, cons, V, bad, n, rng)
    bad = dict(proof, e_blinding=(proof["e_blinding"] + 1) % r)
    assert not verify(bp, ctx, pp, cons, V, bad, n, rng)
    bad = dict(proof, S1=proof["A_O1"])
    assert not verify(bp, ctx, pp, cons, V, bad, n, rng)
    swapped = dict(proof["T"])
    swapped[4] = proof["T"][5]
    bad = dict(proof, T=swapped)
    assert not verify(bp, ctx, pp, cons, V, bad, n, rng)
    if m:
        assert not verify(bp, ctx, pp, cons, [O.generator(ctx.curve)] + V[1:], proof, n, rng)      # another statement
    # a different circuit: one coefficient of one constraint
    q = next(i for i, ts in enumerate(cons) if ts)
    k, i, c = cons[q][0]
    cons2 = [list(ts) for ts in cons]
    cons2[q][0] = (k, i, (c + 1) % r)
    assert not verify(bp, ctx, pp, cons2, V, proof, n, rng)
    # a witness that violates a multiplication gate cannot be proven
    aO_bad = list(aO)
    aO_bad[0] = (aO_bad[0] + 1) % r
    V2, proof2 = prove(bp, ctx, pp, cons, aL, aR, aO_bad, v, v_blinding, rng)
    assert not verify(bp, ctx, pp, cons, V2, proof2, n, rng)
    ctx.close()
