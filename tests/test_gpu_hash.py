"""GPU parity tests for hash-to-G1 (`bp_g1vec_from_msg_hash`, `bp_get_generators`): the HIP kernel through the C ABI
against the committed golden vectors and the C oracle (reference: get_generators, src/utils/mod.rs:16-23)."""
import random

import pytest

import __graft_entry__ as G
import _oracle as O

pytestmark = pytest.mark.gpu
CURVES = ["bls12_381", "bn254"]


@pytest.fixture(scope="module")
def bp():
    return G.load_package()


@pytest.fixture(scope="module")
def ctxs(bp):
    c = {name: bp.Context(cid, 0) for name, cid in bp.CURVE_IDS.items()}
    yield c
    for x in c.values():
        x.close()


@pytest.mark.parametrize("name", CURVES)
def test_from_msg_hash_golden(bp, ctxs, golden, name):
    ctx = ctxs[name]
    cases = golden("hash_to_g1")["curves"][name]["from_msg_hash"]      # empty, 1-byte, 135/136/137-byte, 300-byte messages
    v = bp.G1Vector.from_msg_hash(ctx, [bytes.fromhex(c["msg"]) for c in cases])
    assert v.to_bytes().hex() == "".join(c["point"] for c in cases)


@pytest.mark.parametrize("name", CURVES)
def test_get_generators_golden(bp, ctxs, golden, name):
    ctx = ctxs[name]
    for prefix, pts in golden("hash_to_g1")["curves"][name]["get_generators"].items():
        assert bp.get_generators(ctx, prefix, len(pts)).to_bytes().hex() == "".join(pts)
    pts = golden("hash_to_g1")["curves"][name]["get_generators"]["G"]
    assert bp.get_generators(ctx, "G", 4, first=5).to_bytes().hex() == "".join(pts[4:8])
    assert len(bp.get_generators(ctx, "G", 0)) == 0
    assert len(bp.G1Vector.from_msg_hash(ctx, [])) == 0


@pytest.mark.parametrize("name", CURVES)
def test_get_generators_vs_oracle(bp, ctxs, name):
    """Several hundred points per prefix, counters with 1..20 digits, prefixes that straddle the 136-byte rate."""
    ctx, cid = ctxs[name], O.CURVE_IDS[name]
    for prefix, n, first in (("G", 300, 1), ("H", 300, 1), ("", 40, 0), ("p" * 126, 40, 9_999_990),
                             ("q" * 136, 20, 1), ("r" * 400, 20, 18_446_744_073_709_551_000)):
        got = bp.get_generators(ctx, prefix, n, first=first).to_bytes()
        assert got == O.get_generators(cid, prefix, n, first=first, nthreads=8), (prefix[:4], n, first)


@pytest.mark.parametrize("name", CURVES)
def test_from_msg_hash_ragged_vs_oracle(bp, ctxs, name):
    ctx, cid = ctxs[name], O.CURVE_IDS[name]
    rng = random.Random(5)
    msgs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 2, 7, 8, 9, 31, 64, 134, 135, 136, 137, 200, 272, 273, 500])))
            for _ in range(257)]
    got = bp.G1Vector.from_msg_hash(ctx, msgs).to_bytes()
    pb = ctx.point_bytes
    for i, m in enumerate(msgs):
        assert got[i * pb:(i + 1) * pb] == O.g1_from_msg_hash(cid, m), i


@pytest.mark.parametrize("name", CURVES)
def test_generators_feed_the_msm(bp, ctxs, name):
    """Hashed generators are ordinary resident points: an MSM over them matches the oracle's MSM over the oracle's points,
    and r * P = O for each (scalar r - 1 plus the point itself sums to the identity)."""
    ctx, cid = ctxs[name], O.CURVE_IDS[name]
    n = 1 << 10
    gens = bp.get_generators(ctx, "G", n)
    sc = O.random_scalars(cid, 77, n)
    sv = bp.FieldElementVector.from_bytes(ctx, sc, n)
    want = O.msm(cid, gens.to_bytes(), sc, n, algo=O.PIPPENGER, nthreads=8)
    assert gens.multi_scalar_mul_var_time(sv) == want
    r = O.group_order(cid)
    rm1 = bp.FieldElementVector.from_bytes(ctx, (r - 1).to_bytes(32, "little") * n, n)
    ones = bp.FieldElementVector.from_bytes(ctx, (1).to_bytes(32, "little") * n, n)
    a = gens.multi_scalar_mul_var_time(rm1)
    b = gens.multi_scalar_mul_var_time(ones)
    assert O.g1_add(cid, a, b) == bytes(ctx.point_bytes)


def test_large_batch_sampled(bp, ctxs):
    """2^16 generators in one launch; a random sample is checked against the oracle, all of them for being on the curve."""
    ctx, cid = ctxs["bls12_381"], 0
    n = 1 << 16
    got = bp.get_generators(ctx, "G", n).to_bytes()
    pb = ctx.point_bytes
    rng = random.Random(11)
    for i in [0, 1, n - 1] + [rng.randrange(n) for _ in range(40)]:
        assert got[i * pb:(i + 1) * pb] == O.g1_from_msg_hash(cid, b"G" + str(i + 1).encode()), i
    for i in range(0, n, 97):
        assert O.on_curve(cid, got[i * pb:(i + 1) * pb])


def test_argument_errors(bp, ctxs):
    import ctypes
    ctx = ctxs["bls12_381"]
    L = bp.lib()
    h = ctypes.c_void_p()
    offs = (ctypes.c_uint64 * 3)(0, 5, 3)            # decreasing offsets
    assert L.bp_g1vec_from_msg_hash(ctx.h, b"abcde", ctypes.cast(offs, ctypes.c_void_p), 2, ctypes.byref(h)) == bp.BP_ERR_ARG
    offs = (ctypes.c_uint64 * 2)(1, 2)               # must start at 0
    assert L.bp_g1vec_from_msg_hash(ctx.h, b"ab", ctypes.cast(offs, ctypes.c_void_p), 1, ctypes.byref(h)) == bp.BP_ERR_ARG
    assert L.bp_get_generators(ctx.h, b"G", 1, ctypes.c_uint64(2**64 - 2), 5, ctypes.byref(h)) == bp.BP_ERR_ARG   # counter would wrap


@pytest.mark.parametrize("name", CURVES)
def test_reference_test_ipp_end_to_end(bp, ctxs, golden, name):
    """The reference's `test_ipp` (src/ipp.rs:325-390) with every input built the way the test builds it -- on the device:
    G = get_generators("g", 4), H = get_generators("h", 4), Q = G1::from_msg_hash("Q") -- then create_ipp / verify_ipp."""
    ctx = ctxs[name]
    c = [x for x in golden("ipp")[name] if x["name"] == "test_ipp_n4_hashed_generators"][0]
    n = c["n"]
    Gv, Hv = bp.get_generators(ctx, "g", n), bp.get_generators(ctx, "h", n)
    Q = bp.G1Vector.from_msg_hash(ctx, [b"Q"]).to_bytes()
    assert Gv.to_bytes().hex() == "".join(c["G"]) and Hv.to_bytes().hex() == "".join(c["H"]) and Q.hex() == c["Q"]
    a = bp.FieldElementVector.from_ints(ctx, [1, 2, 3, 4])
    b = bp.FieldElementVector.from_ints(ctx, [5, 6, 7, 8])
    Gf = bp.FieldElementVector.from_ints(ctx, [1] * n)
    Hf = bp.FieldElementVector.from_bytes(ctx, b"".join(bytes.fromhex(x) for x in c["H_factors"]), n)
    proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
    assert proof.L.hex() == "".join(c["L"]) and proof.R.hex() == "".join(c["R"])
    assert proof.a.hex() == c["a_out"] and proof.b.hex() == c["b_out"]
    # P = G^a * H^(b .* y^-i) * Q^<a,b>   (src/ipp.rs:353-372)
    pts = bp.G1Vector.from_bytes(ctx, Gv.to_bytes() + Hv.to_bytes() + Q, 2 * n + 1)
    sc = bp.FieldElementVector.from_bytes(ctx, a.to_bytes() + b.hadamard_product(Hf).to_bytes() + a.inner_product(b), 2 * n + 1)
    P = pts.multi_scalar_mul_var_time(sc)
    assert P.hex() == c["P"]
    bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
