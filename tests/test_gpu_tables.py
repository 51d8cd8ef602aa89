"""GPU parity tests for the window-multiples tables (bp_g1vec_precompute): an MSM over a vector WITH a table must give the bytes
of the same MSM without one and of the CPU oracle -- golden vectors (identity points, P and -P, duplicates), every table width,
random sizes, structured scalars, the paired form, begin / end, and linearity at 2^18."""
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


@pytest.mark.parametrize("name", CURVES)
def test_golden_with_tables(bp, ctxs, golden, name):
    ctx = ctxs[name]
    for c in golden("msm")[name]:
        n = c["n"]
        if n == 0:
            continue
        pts = bp.G1Vector.from_bytes(ctx, b"".join(hx(p) for p in c["points"]), n)
        sc = bp.FieldElementVector.from_bytes(ctx, b"".join(hx(s) for s in c["scalars"]), n)
        for w in (0, 2, 5, 8, 13, 16):
            pts.precompute(w)
            cw, W, nbytes = pts.table_info()
            ctab = 32 * n * ctx.point_bytes if 64 % cw == 0 else 0      # + the compaction table (8 digit multiples of the 4 rows 2^(64 k) P) when the width divides 64
            assert cw == (w or 8) and W == -(-(ctx.fr_bits + 1) // cw) and nbytes == W * n * ctx.point_bytes + ctab
            assert pts.multi_scalar_mul_var_time(sc) == hx(c["out"]), (c["name"], w)
        pts.drop_table()
        assert pts.table_info() == (0, 0, 0)
        assert pts.multi_scalar_mul_var_time(sc) == hx(c["out"]), c["name"]


@pytest.mark.parametrize("name", CURVES)
def test_every_table_width_vs_oracle(bp, ctxs, name):
    ctx = ctxs[name]
    n = 3000
    ks = O.random_scalars(ctx.curve, 11, n)
    ss = O.random_scalars(ctx.curve, 12, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    want = O.msm(ctx.curve, pts.to_bytes(), ss, n, algo=O.PIPPENGER, nthreads=4)
    assert pts.multi_scalar_mul_var_time(sv) == want
    for c in range(2, 17):
        pts.precompute(c)
        assert pts.multi_scalar_mul_var_time(sv) == want, c
        # a range MSM does not use the table (its rows are indexed for the whole vector) and must still be right
        assert pts.msm_range(0, sv, 0, n - 1) == O.msm(ctx.curve, pts.to_bytes(0, n - 1), ss[:32 * (n - 1)], n - 1, algo=O.PIPPENGER, nthreads=4) if c == 9 else True


@pytest.mark.parametrize("name", CURVES)
def test_random_sizes_and_structured_scalars(bp, ctxs, name):
    ctx = ctxs[name]
    r = ctx.r
    for seed, n in enumerate((1, 2, 63, 257, 513, 1000, 4097, 70001)):
        ks = O.random_scalars(ctx.curve, 100 + seed, n)
        pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
        host_pts = pts.to_bytes()
        pts.precompute(0)
        kinds = {
            "uniform": O.random_scalars(ctx.curve, 200 + seed, n),
            "bits": b"".join(((i * 7 + seed) & 1).to_bytes(32, "little") for i in range(n)),
            "zeros": bytes(32 * n),
            "ones": (1).to_bytes(32, "little") * n,
            "near_r": b"".join((r - 1 - (i % 3)).to_bytes(32, "little") for i in range(n)),
        }
        for kind, ss in kinds.items():
            sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
            assert pts.multi_scalar_mul_var_time(sv) == O.msm(ctx.curve, host_pts, ss, n, algo=O.PIPPENGER, nthreads=8), (n, kind)


@pytest.mark.parametrize("name", CURVES)
def test_pair_and_begin_end_with_tables(bp, ctxs, name):
    ctx = ctxs[name]
    n = 5001
    ks = O.random_scalars(ctx.curve, 31, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    host_pts = pts.to_bytes()
    s1 = bytearray(O.random_scalars(ctx.curve, 32, n))
    s2 = bytearray(O.random_scalars(ctx.curve, 33, n))
    s1[32 * (n // 2):] = bytes(32 * (n - n // 2))          # the inner-product round shape: each set is zero on one half
    s2[:32 * (n // 2)] = bytes(32 * (n // 2))
    v1 = bp.FieldElementVector.from_bytes(ctx, bytes(s1), n)
    v2 = bp.FieldElementVector.from_bytes(ctx, bytes(s2), n)
    w1 = O.msm(ctx.curve, host_pts, bytes(s1), n, algo=O.PIPPENGER, nthreads=8)
    w2 = O.msm(ctx.curve, host_pts, bytes(s2), n, algo=O.PIPPENGER, nthreads=8)
    assert pts.multi_scalar_mul_pair(v1, v2) == (w1, w2)
    for c in (0, 7, 16):
        pts.precompute(c)
        assert pts.multi_scalar_mul_pair(v1, v2) == (w1, w2), c
        pts.msm_begin(v2)
        assert pts.msm_end() == w2


def test_linearity_2p18_with_table(bp, ctxs):
    """MSM(s, k.G) == (<s, k> mod r).G at 2^18 points over a table of 16-bit windows (the IPP / R1CS generator shape)"""
    ctx = ctxs["bls12_381"]
    n = 1 << 18
    ks = O.random_scalars(ctx.curve, 41, n)
    ss = O.random_scalars(ctx.curve, 42, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    plain = pts.multi_scalar_mul_var_time(sv)
    pts.precompute(16)
    want = O.g1_mul(ctx.curve, O.fr_inner(ctx.curve, ks, ss, n), O.generator(ctx.curve))
    assert plain == want
    assert pts.multi_scalar_mul_var_time(sv) == want
