"""GPU parity tests for the MSM path: the HIP library through its C ABI against the CPU oracle, the committed
golden vectors, and -- at full size -- the linearity property  MSM(s, k.G) == (<s, k> mod r).G ."""
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
def test_roundtrip_formats(bp, ctxs, golden, name):
    ctx = ctxs[name]
    cases = golden("g1")[name]["add"]
    pts = b"".join(hx(c["sum"]) for c in cases)     # includes the identity
    n = len(cases)
    v = bp.G1Vector.from_bytes(ctx, pts, n)
    assert len(v) == n
    assert v.to_bytes() == pts
    amcl = v.to_bytes(fmt=bp.FMT_AMCL)
    per = ctx.point_bytes + 1
    for i in range(n):
        assert amcl[i * per:(i + 1) * per] == O.g1_to_amcl(ctx.curve, pts[i * ctx.point_bytes:(i + 1) * ctx.point_bytes])
    v2 = bp.G1Vector.from_bytes(ctx, amcl, n, fmt=bp.FMT_AMCL)
    assert v2.to_bytes() == pts
    assert v.to_bytes(offset=3, n=2) == pts[3 * ctx.point_bytes:5 * ctx.point_bytes]
    with pytest.raises(bp.ValueError_):
        v.to_bytes(offset=n, n=1)
    s = O.random_scalars(ctx.curve, 5, 10)
    f = bp.FieldElementVector.from_bytes(ctx, s, 10)
    assert f.to_bytes() == s and len(f) == 10


@pytest.mark.parametrize("name", CURVES)
def test_msm_golden(bp, ctxs, golden, name):
    ctx = ctxs[name]
    for c in golden("msm")[name]:
        n = c["n"]
        pts = bp.G1Vector.from_bytes(ctx, b"".join(hx(p) for p in c["points"]), n)
        sc = bp.FieldElementVector.from_bytes(ctx, b"".join(hx(s) for s in c["scalars"]), n)
        assert pts.multi_scalar_mul_var_time(sc) == hx(c["out"]), c["name"]


@pytest.mark.parametrize("name", CURVES)
def test_msm_every_window_width(bp, ctxs, golden, name):
    ctx = ctxs[name]
    cases = [c for c in golden("msm")[name] if c["name"] in ("random_257", "window_edge_scalars", "duplicates_mixed", "r_minus_1")]
    try:
        for c in cases:
            n = c["n"]
            pts = bp.G1Vector.from_bytes(ctx, b"".join(hx(p) for p in c["points"]), n)
            sc = bp.FieldElementVector.from_bytes(ctx, b"".join(hx(s) for s in c["scalars"]), n)
            for bits in range(2, 17):
                ctx.set_window_bits(bits)
                assert pts.multi_scalar_mul_var_time(sc) == hx(c["out"]), (c["name"], bits)
    finally:
        ctx.set_window_bits(0)


@pytest.mark.parametrize("name", CURVES)
def test_device_tail_gives_the_same_bytes(bp, ctxs, golden, name):
    ctx = ctxs[name]
    try:
        ctx.set_device_tail(True)
        for c in golden("msm")[name]:
            n = c["n"]
            pts = bp.G1Vector.from_bytes(ctx, b"".join(hx(p) for p in c["points"]), n)
            sc = bp.FieldElementVector.from_bytes(ctx, b"".join(hx(s) for s in c["scalars"]), n)
            assert pts.multi_scalar_mul_var_time(sc) == hx(c["out"]), c["name"]
    finally:
        ctx.set_device_tail(False)


@pytest.mark.parametrize("name", CURVES)
def test_length_mismatch_is_value_error(bp, ctxs, name):
    ctx = ctxs[name]
    g = O.generator(ctx.curve)
    pts = bp.G1Vector.from_bytes(ctx, g * 4, 4)
    sc = bp.FieldElementVector.from_ints(ctx, [1, 2, 3])
    with pytest.raises(bp.ValueError_):           # amcl_wrapper ValueError (ipp.rs:91 .unwrap())
        pts.multi_scalar_mul_var_time(sc)
    with pytest.raises(bp.ValueError_):
        pts.msm_range(2, sc, 0, 3)
    assert pts.msm_range(1, sc, 0, 3) == O.g1_mul(ctx.curve, (6).to_bytes(32, "little"), g)
    assert pts.msm_range(0, sc, 0, 0) == bytes(ctx.point_bytes)


@pytest.mark.parametrize("name", CURVES)
def test_fixed_base_and_scalar_mul(bp, ctxs, name):
    ctx = ctxs[name]
    n = 300
    ks = O.random_scalars(ctx.curve, 21, n)
    kv = bp.FieldElementVector.from_bytes(ctx, ks, n)
    pts = bp.G1Vector.fixed_base(ctx, kv)
    host = pts.to_bytes()
    assert host == O.fixed_base_batch(ctx.curve, ks, n, nthreads=4)
    ss = (0).to_bytes(32, "little") + (1).to_bytes(32, "little") + (ctx.r - 1).to_bytes(32, "little") + O.random_scalars(ctx.curve, 22, n - 3)
    sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    got = pts.scaled_by(sv).to_bytes()
    pb = ctx.point_bytes
    for i in list(range(6)) + [n - 1]:
        assert got[i * pb:(i + 1) * pb] == O.g1_mul(ctx.curve, ss[i * 32:(i + 1) * 32], host[i * pb:(i + 1) * pb]), i


@pytest.mark.parametrize("name", CURVES)
# (512 / 513: one / two blocks per window of the single-launch path; 1536 / 1537: its limit, the bucket pipeline beyond)
@pytest.mark.parametrize("n", [1, 2, 5, 31, 32, 33, 512, 513, 1000, 1536, 1537, 4097, 70000])
def test_msm_random_vs_oracle(bp, ctxs, name, n):
    ctx = ctxs[name]
    ks = O.random_scalars(ctx.curve, 100 + n, n)
    ss = O.random_scalars(ctx.curve, 200 + n, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    got = pts.multi_scalar_mul_var_time(sv)
    want = O.msm(ctx.curve, pts.to_bytes(), ss, n, algo=O.PIPPENGER, nthreads=8)
    assert got == want


@pytest.mark.parametrize("name", CURVES)
@pytest.mark.parametrize("n", [1, 3, 129, 513, 1536, 1537, 5000, 70000])
def test_msm_pair_equals_two_msms(bp, ctxs, name, n):
    ctx = ctxs[name]
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, O.random_scalars(ctx.curve, 50 + n, n), n))
    s1b = O.random_scalars(ctx.curve, 60 + n, n)
    s2b = bytes(32) * (n // 2) + O.random_scalars(ctx.curve, 70 + n, n - n // 2)      # half zeros, like the IPP's L/R scalars
    s1, s2 = bp.FieldElementVector.from_bytes(ctx, s1b, n), bp.FieldElementVector.from_bytes(ctx, s2b, n)
    host = pts.to_bytes()
    got = pts.multi_scalar_mul_pair(s1, s2)
    assert got[0] == O.msm(ctx.curve, host, s1b, n, algo=O.PIPPENGER, nthreads=8)
    assert got[1] == O.msm(ctx.curve, host, s2b, n, algo=O.PIPPENGER, nthreads=8)
    assert got == (pts.multi_scalar_mul_var_time(s1), pts.multi_scalar_mul_var_time(s2))
    with pytest.raises(bp.ValueError_):
        pts.multi_scalar_mul_pair(s1, bp.FieldElementVector.from_ints(ctx, [1] * (n + 1)))


@pytest.mark.parametrize("name", CURVES)
def test_msm_skewed_scalars(bp, ctxs, name):
    """config-3-like inputs: bit scalars, all-equal scalars, 1% zeros -- heavy single buckets."""
    ctx = ctxs[name]
    n = 20000
    ks = O.random_scalars(ctx.curve, 31, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    host = pts.to_bytes()
    import random
    rnd = random.Random(7)
    for label, vals in (
        ("bits", [rnd.getrandbits(1) for _ in range(n)]),
        ("all_equal", [0x1234567] * n),
        ("zeros_1pct", [0 if rnd.random() < 0.01 else rnd.getrandbits(250) for _ in range(n)]),
        # small negative scalars are recoded as r - k with the point negated (k_digits_bin, round 4): a_R = a_L - 1 of a bit vector, small
        # negative weights, and the rule's boundary r - 2^128 (the last scalar NOT negated) with its neighbours, among ordinary scalars
        ("zero_or_minus_one", [(ctx.r - 1) * rnd.getrandbits(1) for _ in range(n)]),
        ("small_negatives", [(ctx.r - rnd.getrandbits(rnd.choice((1, 8, 64, 127, 128)))) % ctx.r if rnd.getrandbits(1) else rnd.getrandbits(250) for _ in range(n)]),
        ("negation_boundary", [ctx.r - (1 << 128) + rnd.choice((-2, -1, 0, 1, 2)) if rnd.getrandbits(2) else rnd.getrandbits(250) for _ in range(n)]),
    ):
        ss = b"".join(v.to_bytes(32, "little") for v in vals)
        sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
        got = pts.multi_scalar_mul_var_time(sv)
        want = O.msm(ctx.curve, host, ss, n, algo=O.PIPPENGER, nthreads=8)
        assert got == want, label
        if label in ("small_negatives", "negation_boundary"):          # ... as the second scalar set of a paired MSM, and ragged (a tile of 64 x k + 17 scalars)
            other = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(ctx.curve, 32, n), n)
            assert pts.multi_scalar_mul_pair(other, sv)[1] == want, label
            m = n - 2031
            assert pts.msm_range(0, sv, 0, m) == O.msm(ctx.curve, host[:m * ctx.point_bytes], ss[:32 * m], m, algo=O.PIPPENGER, nthreads=8), label


@pytest.mark.parametrize("name,lg", [("bls12_381", 20), ("bn254", 20), ("bls12_381", 22)])
def test_msm_full_size_linearity(bp, ctxs, name, lg):
    """BASELINE sizes (2^20, 2^22): MSM(s, k.G) must equal (<s,k> mod r).G  -- checked with the oracle's Fr inner
    product and one oracle scalar multiplication; a sample of the device-generated points is checked directly."""
    ctx = ctxs[name]
    n = 1 << lg
    ks = O.random_scalars(ctx.curve, 1000 + lg, n)
    ss = O.random_scalars(ctx.curve, 2000 + lg, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    pb = ctx.point_bytes
    gen = O.generator(ctx.curve)
    for i in (0, 1, n // 2, n - 1):
        assert pts.to_bytes(offset=i, n=1) == O.g1_mul(ctx.curve, ks[i * 32:(i + 1) * 32], gen)
    got = pts.multi_scalar_mul_var_time(bp.FieldElementVector.from_bytes(ctx, ss, n))
    want = O.g1_mul(ctx.curve, O.fr_inner(ctx.curve, ks, ss, n), gen)
    assert got == want
    pts.free()


@pytest.mark.parametrize("n", [513, 131071, 131073, 262145, 300001, 524287, 524289, 1048577])
def test_msm_tile_boundaries_linearity(bp, ctxs, n):
    """Sizes around the binning-tile switches (tile = 256 .. 2048 scalars per block, chosen so that ~512 blocks remain; ragged last
    tiles; one element past a power of two): MSM(s, k.G) = (<s, k> mod r).G with the oracle's inner product, single and paired."""
    ctx = ctxs["bls12_381"]
    ks = O.random_scalars(ctx.curve, 31 + n, n)
    s1 = O.random_scalars(ctx.curve, 32 + n, n)
    s2 = O.random_scalars(ctx.curve, 33 + n, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    gen = O.generator(ctx.curve)
    v1, v2 = bp.FieldElementVector.from_bytes(ctx, s1, n), bp.FieldElementVector.from_bytes(ctx, s2, n)
    want1 = O.g1_mul(ctx.curve, O.fr_inner(ctx.curve, ks, s1, n), gen)
    want2 = O.g1_mul(ctx.curve, O.fr_inner(ctx.curve, ks, s2, n), gen)
    assert pts.multi_scalar_mul_var_time(v1) == want1
    assert pts.multi_scalar_mul_pair(v1, v2) == (want1, want2)
    pts.free()


@pytest.mark.parametrize("n", [6000, 600, 2])     # 600 -> two shards of 300 terms: the single-launch small-MSM path
@pytest.mark.parametrize("name", CURVES)
def test_two_stage_sharded_msm(bp, ctxs, name, n):
    """The multi-GPU decomposition on one GPU: two index-range shards -> window records -> one finish."""
    import torch
    ctx = ctxs[name]
    ks = O.random_scalars(ctx.curve, 41, n)
    ss = O.random_scalars(ctx.curve, 42, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    half = n // 2
    W = bp.msm_window_records(ctx, half)
    rb = bp.msm_record_bytes(ctx.curve)
    buf = torch.zeros(2 * W * rb, dtype=torch.uint8, device="cuda:0")
    bp.msm_windows(ctx, pts, 0, sv, 0, half, buf.data_ptr())
    bp.msm_windows(ctx, pts, half, sv, half, half, buf.data_ptr() + W * rb)
    got = bp.msm_finish(ctx, buf.data_ptr(), 2, half)
    assert got == pts.multi_scalar_mul_var_time(sv)
    assert got == O.msm(ctx.curve, pts.to_bytes(), ss, n, algo=O.PIPPENGER, nthreads=8)


def test_two_contexts_in_concurrent_threads(bp):
    """One bp_ctx per host thread (include/bpmsm.h): the reference's tests run on parallel threads."""
    import threading
    results, errors = {}, []

    def worker(tid, curve):
        try:
            ctx = bp.Context(curve, 0)
            n = 3000 + 500 * tid
            ks = O.random_scalars(curve, 900 + tid, n)
            ss = O.random_scalars(curve, 950 + tid, n)
            pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
            sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
            outs = [pts.multi_scalar_mul_var_time(sv) for _ in range(5)]
            results[tid] = (curve, pts.to_bytes(), ss, n, outs)
            ctx.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t, t % 2)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for tid, (curve, host, ss, n, outs) in results.items():
        want = O.msm(curve, host, ss, n, algo=O.PIPPENGER, nthreads=4)
        assert all(o == want for o in outs), tid


def test_caller_owned_stream(bp):
    """bp_ctx_set_stream: kernels run on a torch side stream the caller owns."""
    import torch
    ctx = bp.Context(bp.BLS12_381, 0)
    side = torch.cuda.Stream(device="cuda:0")
    ctx.set_stream(side.cuda_stream)
    n = 5000
    ks, ss = O.random_scalars(0, 70, n), O.random_scalars(0, 71, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    got = pts.multi_scalar_mul_var_time(bp.FieldElementVector.from_bytes(ctx, ss, n))
    assert got == O.g1_mul(0, O.fr_inner(0, ks, ss, n), O.generator(0))
    ctx.set_stream(0)
    assert pts.multi_scalar_mul_var_time(bp.FieldElementVector.from_bytes(ctx, ss, n)) == got
    ctx.close()


@pytest.mark.parametrize("name", CURVES)
def test_msm_heavy_skew_2e18(bp, ctxs, name):
    """Structured scalars at 2^18: one value repeated (every window has a single bucket with all the points), bit
    vectors, and 16 distinct values -- the heavy-bucket / combine paths at scale; checked by linearity."""
    import time
    ctx = ctxs[name]
    n = 1 << 18
    ks = O.random_scalars(ctx.curve, 333, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    gen = O.generator(ctx.curve)
    import random
    rnd = random.Random(5)
    vals16 = [rnd.getrandbits(250) for _ in range(16)]
    # (round 4: at this size a coarse bin of more than 32 768 records is cut into slices of 8 192 for k_fine_huge_count / _place --
    # every case below has such bins: one per window, one in window 0, the carry bucket of window 1, two per window, ...)
    cases = (("all_equal", [0x1234567890ABCDEF1234567] * n), ("bits", [rnd.getrandbits(1) for _ in range(n)]),
             ("sixteen_values", [vals16[rnd.randrange(16)] for _ in range(n)]),
             ("zero_or_minus_one", [(ctx.r - 1) * rnd.getrandbits(1) for _ in range(n)]),          # a_R = a_L - 1 of a bit vector (positive_no.rs:18-24)
             ("8_bit", [rnd.getrandbits(8) for _ in range(n)]), ("16_bit", [rnd.getrandbits(16) for _ in range(n)]),
             ("mostly_one_value", [vals16[0] if rnd.randrange(10) else rnd.getrandbits(250) for _ in range(n)]))
    for label, vals in cases:
        ss = b"".join(v.to_bytes(32, "little") for v in vals)
        sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
        t0 = time.time()
        got = pts.multi_scalar_mul_var_time(sv)
        dt = time.time() - t0
        want = O.g1_mul(ctx.curve, O.fr_inner(ctx.curve, ks, ss, n), gen)
        assert got == want, label
        assert dt < 2.0, (label, dt)      # no pathological serialisation
        if label in ("bits", "8_bit"):    # ... and ragged: the last slice of a bin is short, the last tile of the scalars too
            m = n - 12345
            assert pts.msm_range(0, sv, 0, m) == O.g1_mul(ctx.curve, O.fr_inner(ctx.curve, ks[:32 * m], ss[:32 * m], m), gen), label
    # the merged-window pipeline over a table has the same bins per scalar set
    pts.precompute(16)
    for label, vals in cases[1:4]:
        ss = b"".join(v.to_bytes(32, "little") for v in vals)
        sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
        assert pts.multi_scalar_mul_var_time(sv) == O.g1_mul(ctx.curve, O.fr_inner(ctx.curve, ks, ss, n), gen), ("table", label)
    pts.free()


@pytest.mark.parametrize("name", CURVES)
def test_msm_many_medium_buckets(bp, ctxs, name):
    """Thousands of buckets of tens of task sums each -- the grouped k_combine_chunks (4 .. 256 lanes per chunk, chosen on the device)
    and the task length the device derives from the sort's own counts: scalars with 256 / 4096 distinct values, 8-bit values, and
    uniform scalars over narrow window-multiples tables (c = 12, 14: every bucket fat); also with the task target raised so that
    every bucket is cut into many more tasks.  Checked by linearity."""
    import random
    ctx = bp.Context(bp.CURVE_IDS[name], 0)
    try:
        n = 1 << 17
        ks = O.random_scalars(ctx.curve, 777, n)
        pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
        gen = O.generator(ctx.curve)
        rnd = random.Random(11)
        pool = [rnd.getrandbits(250) for _ in range(4096)]
        kinds = {
            "256_values": [pool[rnd.randrange(256)] for _ in range(n)],
            "4096_values": [pool[rnd.randrange(4096)] for _ in range(n)],
            "8_bit": [rnd.getrandbits(8) for _ in range(n)],
            "uniform": None,
        }
        for target in (0, 1 << 22):
            ctx.set_tuning(bp.TUNE_TASK_TARGET, target)
            for label, vals in kinds.items():
                ss = O.random_scalars(ctx.curve, 778, n) if vals is None else b"".join(v.to_bytes(32, "little") for v in vals)
                sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
                want = O.g1_mul(ctx.curve, O.fr_inner(ctx.curve, ks, ss, n), gen)
                assert pts.multi_scalar_mul_var_time(sv) == want, (label, target)
                if vals is None:
                    for c in (14, 12):
                        pts.precompute(c)
                        assert pts.multi_scalar_mul_var_time(sv) == want, (label, target, c)
                    pts.drop_table()
    finally:
        ctx.close()


def test_begin_end_two_contexts_one_thread(bp):
    """bp_msm_g1_begin / _end: two MSMs in flight from one host thread (two contexts, two streams)."""
    ca, cb = bp.Context(bp.BLS12_381, 0), bp.Context(bp.BLS12_381, 0)
    n = 40000
    ks, s1, s2 = O.random_scalars(0, 81, n), O.random_scalars(0, 82, n), O.random_scalars(0, 83, n)
    pa = bp.G1Vector.fixed_base(ca, bp.FieldElementVector.from_bytes(ca, ks, n))
    ca.synchronize()
    pb = bp.G1Vector.wrap_device(cb, pa.device_ptr(), n)
    va, vb = bp.FieldElementVector.from_bytes(ca, s1, n), bp.FieldElementVector.from_bytes(cb, s2, n)
    gen = O.generator(0)
    w1, w2 = O.g1_mul(0, O.fr_inner(0, ks, s1, n), gen), O.g1_mul(0, O.fr_inner(0, ks, s2, n), gen)
    for _ in range(3):
        pa.msm_begin(va)
        pb.msm_begin(vb)
        assert pa.msm_end() == w1
        assert pb.msm_end() == w2
    with pytest.raises(bp.ArgError):
        pa.msm_end()                       # nothing pending
    empty = bp.G1Vector.new(ca, 0)
    empty.msm_begin(bp.FieldElementVector.new(ca, 0))
    assert empty.msm_end() == bytes(ca.point_bytes)
    ca.close(); cb.close()


def test_short_differential_fuzz():
    """A few seconds of scripts/fuzz_msm.py and scripts/fuzz_ipp.py (random sizes, window widths, scalar structure,
    identity / duplicate / negated points, window tables, tuning knobs, both prover modes) against the oracle.  Longer runs of the
    same scripts after every kernel change: totals per round in DESIGN.md section 2."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, secs in (("fuzz_msm.py", "6"), ("fuzz_ipp.py", "6")):
        p = subprocess.run([sys.executable, os.path.join(root, "scripts", script), secs, "99"], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "fails 0" in p.stdout.splitlines()[-1], p.stdout[-2000:] + p.stderr[-2000:]


def test_one_bucket_holds_everything():
    """All scalars equal to 1 at n = 2^20 with the task length forced down to 8 (bp_ctx_set_tuning, BP_TUNE_TASK_TARGET): ONE bucket with 131072 task
    sums = 512 chunks, i.e. the strided second stage of the heavy-bucket combine (> 256 chunk sums per bucket).  The script
    checks sum_i (k_i G) == (sum_i k_i) G on both curves, for the default and a narrow window width."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_huge_bucket.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.count(" ok") == 4 and "MISMATCH" not in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
