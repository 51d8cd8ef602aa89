"""End-to-end composition of the hot-path pieces in the shape of BASELINE config 3: an R1CS proof is created and verified with
every vector / group operation going through the C ABI (commitment MSMs, flattened constraints, l/r polynomials, t(x),
IPP, the verifier's single MSM), the host doing only what the reference's host does: the transcript and a handful of scalars.

The orchestration (tests/r1cs_twin.py) restates `Prover::prove` (/root/reference src/r1cs/prover.rs:323-560) and
`Verifier::verify` (src/r1cs/verifier.rs:265-452) for a single-phase constraint system (n2 = 0, A_I2 = A_O2 = S2 = O).
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


def run_prove(bp, R1, ctx, gens, cons, n, m, aL, aR, aO, v, v_blinding, rng):
    r = ctx.r
    V = gens.commit_many(v, v_blinding)                       # batched commit_to_field_element ...
    assert V == [gens.commit(v[j], v_blinding[j]) for j in range(m)]      # ... equals the 2-term MSMs
    terms = [(q, k, i, c) for q, ts in enumerate(cons) for k, i, c in ts]
    plan = bp.R1CSPlan(ctx, terms, len(cons), n, m)
    dev = lambda xs: bp.FieldElementVector.from_ints(ctx, xs)
    blindings = {k: rng.randrange(r) for k in ("i", "o", "s", "t1", "t3", "t4", "t5", "t6")}
    sL, sR = [rng.randrange(r) for _ in range(n)], [rng.randrange(r) for _ in range(n)]
    proof = R1.prove(ctx, gens, plan, R1.start_transcript(ctx, b"R1CS e2e", V), dev(aL), dev(aR), dev(aO), dev(v_blinding), dev(sL), dev(sR), blindings)
    # the same orchestration as ONE library call (bp_r1cs_prove, C++ inside libbpmsm.so): identical proof bytes
    order = ("i", "o", "s", "t1", "t3", "t4", "t5", "t6")
    raw = bp.r1cs_prove(ctx, R1.start_transcript(ctx, b"R1CS e2e", V), plan, gens.G, gens.H, gens.g, gens.h, dev(aL), dev(aR), dev(aO),
                        dev(v_blinding) if m else None, dev(sL), dev(sR), b"".join(le(blindings[k]) for k in order))
    assert raw == flatten_proof(ctx, proof)
    plan.free()
    return V, proof


def flatten_proof(ctx, proof):
    """the byte layout of bp_r1cs_prove (include/bpmsm.h)"""
    ident = bytes(ctx.point_bytes)
    ipp = proof["ipp"]
    return (proof["A_I1"] + proof["A_O1"] + proof["S1"] + ident * 3 + b"".join(proof["T"][k] for k in (1, 3, 4, 5, 6))
            + le(proof["t_x"]) + le(proof["t_x_blinding"]) + le(proof["e_blinding"]) + ipp.L + ipp.R + ipp.a + ipp.b)


def run_verify(bp, R1, ctx, gens, cons, n, V, proof, rng):
    terms = [(q, k, i, c) for q, ts in enumerate(cons) for k, i, c in ts]
    plan = bp.R1CSPlan(ctx, terms, len(cons), n, len(V))
    ok = R1.verify(ctx, gens, plan, R1.start_transcript(ctx, b"R1CS e2e", V), V, proof, r_weight=rng.randrange(ctx.r))
    # ... and the library's verifier (bp_r1cs_verify) must reach the same verdict on the same proof
    try:
        bp.r1cs_verify(ctx, R1.start_transcript(ctx, b"R1CS e2e", V), plan, gens.G, gens.H, gens.g, gens.h, b"".join(V), n, flatten_proof(ctx, proof),
                       le(rng.randrange(ctx.r)))
        ok_lib = True
    except bp.VerificationError:
        ok_lib = False
    assert ok_lib == ok
    plan.free()
    return ok


@pytest.mark.parametrize("name,n,m,nq", [("bls12_381", 13, 3, 20), ("bls12_381", 64, 4, 150), ("bn254", 5, 0, 7), ("bls12_381", 300, 2, 400)])
def test_r1cs_prove_and_verify(bp, name, n, m, nq):
    import r1cs_twin as R1
    ctx = bp.Context(bp.CURVE_IDS[name], 0)
    r = ctx.r
    rng = random.Random(31 * n + m)
    gens = R1.Generators(ctx, R1.padded(n) * (2 if n == 13 else 1))       # more generators than needed is fine (prover.rs:333)
    aL, aR, aO, v, cons = random_satisfiable_system(rng, r, n, m, nq)
    v_blinding = [rng.randrange(r) for _ in range(m)]
    V, proof = run_prove(bp, R1, ctx, gens, cons, n, m, aL, aR, aO, v, v_blinding, rng)
    check = lambda cons_, V_, proof_: run_verify(bp, R1, ctx, gens, cons_, n, V_, proof_, rng)
    assert check(cons, V, proof)
    assert run_verify(bp, R1, ctx, gens, cons, n, V, proof, random.Random(5))      # the verifier's own randomness does not matter

    # anything changed -> rejected
    assert not check(cons, V, dict(proof, t_x=(proof["t_x"] + 1) % r))
    assert not check(cons, V, dict(proof, e_blinding=(proof["e_blinding"] + 1) % r))
    assert not check(cons, V, dict(proof, S1=proof["A_O1"]))
    swapped = dict(proof["T"])
    swapped[4] = proof["T"][5]
    assert not check(cons, V, dict(proof, T=swapped))
    if m:
        assert not check(cons, [O.generator(ctx.curve)] + V[1:], proof)          # another statement
    # a different circuit: one coefficient of one constraint
    q = next(i for i, ts in enumerate(cons) if ts)
    k, i, c = cons[q][0]
    cons2 = [list(ts) for ts in cons]
    cons2[q][0] = (k, i, (c + 1) % r)
    assert not check(cons2, V, proof)
    # a witness that violates a multiplication gate cannot be proven
    aO_bad = list(aO)
    aO_bad[0] = (aO_bad[0] + 1) % r
    V2, proof2 = run_prove(bp, R1, ctx, gens, cons, n, m, aL, aR, aO_bad, v, v_blinding, rng)
    assert not check(cons, V2, proof2)
    ctx.close()
