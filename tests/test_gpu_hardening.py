"""Input validation, allocator and sharding-geometry behaviour of the C ABI (round-2 hardening), and BASELINE config 4's
decomposition on one GPU.
  * points from outside are checked like amcl's G1::from_bytes (on the curve, canonical coordinates);
  * scalars must be < r; the verifier's weight r / batch weights must be non-zero;
  * record blocks carry their window geometry: sets that do not fit together are refused;
  * 2^22 points as 8 index-range shards of 2^19 (the cfg4 split) == one 2^22 MSM, through bp_msm_g1_multi and through
    bp_msm_g1_windows / bp_msm_g1_finish."""
import pytest

import __graft_entry__ as G
import _oracle as O

pytestmark = pytest.mark.gpu
CURVES = ["bls12_381", "bn254"]


@pytest.fixture(scope="module")
def bp():
    return G.load_package()


def le(x, nb=32):
    return int(x).to_bytes(nb, "little")


@pytest.mark.parametrize("name", CURVES)
def test_points_are_validated(bp, name):
    cid = bp.CURVE_IDS[name]
    ctx = bp.Context(cid, 0)
    info = bp.curve_info(cid)
    fb, p = info.fp_bytes, int.from_bytes(bytes(info.p_le)[:info.fp_bytes], "little")
    gen = O.generator(cid)
    x, y = int.from_bytes(gen[:fb], "little"), int.from_bytes(gen[fb:], "little")
    good = gen + bytes(2 * fb) + O.g1_add(cid, gen, gen)                      # G, identity, 2G
    assert bp.G1Vector.from_bytes(ctx, good, 3).to_bytes() == good
    off_curve = le(x, fb) + le((y + 1) % p, fb)
    y_zero = le(x, fb) + le(0, fb)                                             # (x, 0): the case the doubling formula excludes
    noncanon_x = le(x + p, fb) + le(y, fb) if (x + p).bit_length() <= 8 * fb else None
    noncanon_y = le(x, fb) + le(y + p, fb) if (y + p).bit_length() <= 8 * fb else None
    for bad in (off_curve, y_zero, noncanon_x, noncanon_y):
        if bad is None:
            continue
        with pytest.raises(bp.ArgError):
            bp.G1Vector.from_bytes(ctx, gen + bad + gen, 3)
        amcl = b"\x04" + bad[:fb][::-1] + bad[fb:][::-1]
        with pytest.raises(bp.ArgError):
            bp.G1Vector.from_bytes(ctx, amcl, 1, fmt=bp.FMT_AMCL)
    # a proof whose L / R / Q is not a curve point fails verification (the reference fails in G1::from_bytes)
    n = 8
    Gv, Hv = bp.get_generators(ctx, "g", n), bp.get_generators(ctx, "h", n)
    Q = bp.G1Vector.from_msg_hash(ctx, [b"Q"]).to_bytes()
    fe = lambda seed: bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, seed, n), n)
    a, b, Gf, Hf = fe(1), fe(2), bp.FieldElementVector.from_ints(ctx, [1] * n), fe(3)
    pr = bp.IPP.create_ipp(ctx, bp.Transcript(b"v"), Q, Gf, Hf, Gv, Hv, a, b)
    pts = bp.G1Vector.from_bytes(ctx, Gv.to_bytes() + Hv.to_bytes() + Q, 2 * n + 1)
    sc = bp.FieldElementVector.from_bytes(ctx, a.hadamard_product(Gf).to_bytes() + b.hadamard_product(Hf).to_bytes() + a.inner_product(b), 2 * n + 1)
    P = pts.multi_scalar_mul_var_time(sc)
    bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"v"), Gf, Hf, P, Q, Gv, Hv, pr.a, pr.b, pr.L, pr.R)
    pb = ctx.point_bytes
    for badL, badQ in ((off_curve + pr.L[pb:], Q), (pr.L, off_curve)):
        with pytest.raises(bp.VerificationError):
            bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"v"), Gf, Hf, P, badQ, Gv, Hv, pr.a, pr.b, badL, pr.R)
        with pytest.raises(bp.VerificationError):
            bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, [(bp.Transcript(b"v"), P, badQ, pr.a, pr.b, badL, pr.R)])
    with pytest.raises(bp.ArgError):
        bp.IPP.create_ipp(ctx, bp.Transcript(b"v"), off_curve, Gf, Hf, Gv, Hv, a, b)
    # batch weights: library-drawn by default; a zero or non-canonical weight is refused
    item = (lambda: (bp.Transcript(b"v"), P, Q, pr.a, pr.b, pr.L, pr.R))
    bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, [item(), item()])
    with pytest.raises(bp.ArgError):
        bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, [item(), item()], weights=le(5) + le(0))
    with pytest.raises(bp.ArgError):
        bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, [item()], weights=le(ctx.r))
    ctx.close()


@pytest.mark.parametrize("name", CURVES)
def test_scalars_must_be_canonical(bp, name):
    cid = bp.CURVE_IDS[name]
    ctx = bp.Context(cid, 0)
    r = ctx.r
    ok = le(0) + le(1) + le(r - 1)
    assert bp.FieldElementVector.from_bytes(ctx, ok, 3).to_bytes() == ok
    for bad in (r, r + 1, (1 << 256) - 1):
        with pytest.raises(bp.ArgError):
            bp.FieldElementVector.from_bytes(ctx, le(5) + le(bad) + le(7), 3)
    assert len(bp.fr_random(cid, 4)) == 128 and all(0 < int.from_bytes(bp.fr_random(cid), "little") < r for _ in range(8))
    ctx.close()


def test_r1cs_verifier_weight(bp, golden):
    """bp_r1cs_verify draws r itself (NULL) as the reference does; an explicit r = 0 or r >= order is refused."""
    from test_oracle_golden import r1cs_case_inputs
    from test_gpu_r1cs_oracle import start_transcript
    ctx = bp.Context(0, 0)
    c = golden("r1cs")["bls12_381"][1]
    a = r1cs_case_inputs(c)
    n, m, ng = c["n"], c["m"], c["n_generators"]
    plan = bp.R1CSPlan(ctx, a["terms"], c["n_constraints"], n, m)
    Gv, Hv = bp.G1Vector.from_bytes(ctx, a["G"], ng), bp.G1Vector.from_bytes(ctx, a["H"], ng)
    Vb = b"".join(a["V"])
    run = lambda r, proof=a["proof"]: bp.r1cs_verify(ctx, start_transcript(bp, ctx, a["label"], a["V"]), plan, Gv, Hv, a["g"], a["h"], Vb, n, proof, r)
    run(None)
    run(le(1))
    for bad in (0, ctx.r, ctx.r + 5):
        with pytest.raises(bp.ArgError):
            run(le(bad))
    bad = bytearray(a["proof"])
    bad[11 * ctx.point_bytes + 40] ^= 1                       # t_x_blinding
    with pytest.raises(bp.VerificationError):
        run(None, bytes(bad))
    plan.free()
    ctx.close()


def test_handles_outlive_their_context_and_blocks_are_recycled(bp):
    ctx = bp.Context(0, 0)
    v = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(0, 3, 1000), 1000)
    p1 = v.device_ptr()
    v.free()
    w = bp.FieldElementVector.new(ctx, 1000)
    assert w.device_ptr() == p1                                # same size class: the block comes back from the pool
    assert w.to_bytes() == bytes(32000)                        # ... zeroed
    keep = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(0, 4, 10), 10)
    ctx.trim()
    ctx.close()
    keep.free()                                                # after its context: releases the pool's last reference
    w.free()


@pytest.mark.parametrize("name", CURVES)
def test_unequal_shards_need_a_common_width(bp, name):
    import torch
    from bulletproofs_amcl_amd import sharding
    cid = bp.CURVE_IDS[name]
    ctx = bp.Context(cid, 0)
    n, world = 32768 + 32767, 2                              # shards 32768 / 32767 straddle 2^15: c = 14 vs 13 when left to n
    ks, ss = O.random_scalars(cid, 21, n), O.random_scalars(cid, 22, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    want = O.g1_mul(cid, O.fr_inner(cid, ks, ss, n), O.generator(cid))
    spans = [sharding.shard_range(n, world, r) for r in range(world)]
    nmax = sharding.largest_shard(n, world)
    rb = bp.msm_record_bytes(cid)
    assert bp.msm_geometry(cid, spans[0][1] - spans[0][0])[0] != bp.msm_geometry(cid, spans[1][1] - spans[1][0])[0]
    # left to themselves the two ranks choose different geometries: refused, not mis-folded
    W0 = bp.msm_window_records(ctx, spans[0][1] - spans[0][0])
    W1 = bp.msm_window_records(ctx, spans[1][1] - spans[1][0])
    buf = torch.zeros((W0 + W1 + 2) * rb, dtype=torch.uint8, device="cuda:0")
    bp.msm_windows(ctx, pts, spans[0][0], sv, spans[0][0], spans[0][1] - spans[0][0], buf.data_ptr())
    bp.msm_windows(ctx, pts, spans[1][0], sv, spans[1][0], spans[1][1] - spans[1][0], buf.data_ptr() + W0 * rb)
    with pytest.raises(bp.ArgError):
        bp.msm_finish(ctx, buf.data_ptr(), 2, nmax)
    # with the common width every rank fixes first
    ctx.set_window_bits(sharding.common_window_bits(bp, cid, n, world))
    W = bp.msm_window_records(ctx, nmax)
    buf = torch.zeros(2 * W * rb, dtype=torch.uint8, device="cuda:0")
    for r, (lo, hi) in enumerate(spans):
        bp.msm_windows(ctx, pts, lo, sv, lo, hi - lo, buf.data_ptr() + r * W * rb)
    assert bp.msm_finish(ctx, buf.data_ptr(), 2, nmax) == want
    assert bp.msm_finish_host(cid, bytes(buf.cpu().tolist()), 2, nmax, sharding.common_window_bits(bp, cid, n, world)) == want
    ctx.set_window_bits(0)
    # begin / end keep the geometry of begin even if the width is changed in between
    pts.msm_begin(sv)
    ctx.set_window_bits(7)
    assert pts.msm_end() == want
    ctx.close()


@pytest.mark.parametrize("name", ["bls12_381", "bn254"])
def test_index_x_window_groups_with_structured_scalars(bp, name):
    """The 2-D split (index range x window group, bp_msm_g1_windows_subset + bp_msm_g1_finish_blocks) over scalars that the recoding treats
    specially: bit vectors, their complements 0 / r - 1 and other small negatives (recoded as r - k with the point negated -- every
    window group of a rank must negate the same scalars), values at the rule's boundary, all mixed with uniform ones.  Every split of 4
    'ranks' gives the single MSM's and the oracle's point."""
    import random
    import torch
    from bulletproofs_amcl_amd import sharding
    cid = bp.CURVE_IDS[name]
    ctx = bp.Context(cid, 0)
    n, world = 40000, 4
    rnd = random.Random(99)
    ks = O.random_scalars(cid, 5101, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    kinds = (lambda: rnd.getrandbits(1), lambda: (ctx.r - 1) * rnd.getrandbits(1), lambda: ctx.r - 1 - rnd.getrandbits(rnd.choice((3, 60, 127))),
             lambda: ctx.r - (1 << 128) + rnd.choice((-1, 0, 1)), lambda: rnd.randrange(ctx.r))
    ss = b"".join(rnd.choice(kinds)().to_bytes(32, "little") for _ in range(n))
    sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    want = O.g1_mul(cid, O.fr_inner(cid, ks, ss, n), O.generator(cid))
    assert pts.multi_scalar_mul_var_time(sv) == want
    rb = bp.msm_record_bytes(cid)
    c, cw, _, _ = bp.msm_geometry(cid, n // world)
    W = len(cw)
    ctx.set_window_bits(c)
    for wg in (1, 2, 4):
        if W % wg:
            continue
        n_set = n // (world // wg)
        stride = max(bp.msm_window_records_subset(ctx, n_set, g * (W // wg), W // wg) for g in range(wg))
        blocks = torch.zeros(world * stride * rb, dtype=torch.uint8, device="cuda:0")
        for r in range(world):
            lo, hi, w0, wn = sharding.shard_2d(n, world, r, W, wg)
            bp.msm_windows_subset(ctx, pts, lo, sv, lo, hi - lo, w0, wn, stride, blocks.data_ptr() + r * stride * rb)
        assert bp.msm_finish_blocks(ctx, blocks.data_ptr(), world, stride, n_set) == want, wg
        assert bp.msm_finish_blocks_host(cid, bytes(blocks.cpu().tolist()), world, stride, n_set, c) == want, wg
    ctx.set_window_bits(0)
    ctx.close()


def test_cfg4_split_of_2p22_over_8_shards(bp):
    """BASELINE config 4 on one GPU: 2^22 points as 8 contiguous shards of 2^19 (one context each, as 8 ranks / devices would
    hold them) through bp_msm_g1_multi, and through bp_msm_g1_windows + bp_msm_g1_finish(sets = 8); both equal the single
    2^22 MSM and the linearity value from the oracle."""
    import torch
    cid, lg, shards = 0, 22, 8
    n = 1 << lg
    per = n // shards
    ks = O.random_scalars(cid, 4001, n)
    ss = O.random_scalars(cid, 4002, n)
    want = O.g1_mul(cid, O.fr_inner(cid, ks, ss, n), O.generator(cid))
    main = bp.Context(cid, 0)
    pts = bp.G1Vector.fixed_base(main, bp.FieldElementVector.from_bytes(main, ks, n))
    sv = bp.FieldElementVector.from_bytes(main, ss, n)
    assert pts.multi_scalar_mul_var_time(sv) == want
    main.synchronize()
    # windows + finish (what bench.py --gpus 8 --strong does per rank, with the all-gather in between)
    W = bp.msm_window_records(main, per)
    rb = bp.msm_record_bytes(cid)
    buf = torch.zeros(shards * W * rb, dtype=torch.uint8, device="cuda:0")
    for s in range(shards):
        bp.msm_windows(main, pts, s * per, sv, s * per, per, buf.data_ptr() + s * W * rb)
    assert bp.msm_finish(main, buf.data_ptr(), shards, per) == want
    # bp_msm_g1_multi: one context per shard (here all on device 0), views into the resident arrays
    ctxs = [bp.Context(cid, 0) for _ in range(shards)]
    pb = main.point_bytes
    pviews = [bp.G1Vector.wrap_device(c, pts.device_ptr() + s * per * pb, per) for s, c in enumerate(ctxs)]
    sviews = [bp.FieldElementVector.wrap_device(c, sv.device_ptr() + s * per * 32, per) for s, c in enumerate(ctxs)]
    assert bp.msm_multi(ctxs, pviews, sviews) == want
    assert bp.msm_multi(ctxs[:1], [pts_v := bp.G1Vector.wrap_device(ctxs[0], pts.device_ptr(), n)],
                        [bp.FieldElementVector.wrap_device(ctxs[0], sv.device_ptr(), n)]) == want
    # ragged shards and an empty one
    cuts = [0, 5, 5, 300000, 1 << 20, n]
    rag_p = [bp.G1Vector.wrap_device(ctxs[i], pts.device_ptr() + cuts[i] * pb, cuts[i + 1] - cuts[i]) for i in range(5)]
    rag_s = [bp.FieldElementVector.wrap_device(ctxs[i], sv.device_ptr() + cuts[i] * 32, cuts[i + 1] - cuts[i]) for i in range(5)]
    assert bp.msm_multi(ctxs[:5], rag_p, rag_s) == want
    with pytest.raises(bp.ArgError):
        bp.msm_multi([ctxs[0], ctxs[0]], pviews[:2], sviews[:2])          # one shard per context
    for c in ctxs:
        c.close()
    main.close()


@pytest.mark.parametrize("name", CURVES)
def test_compressed_points(bp, golden, name):
    """tag || X wire form (this build's; see include/bpmsm.h) against the Python-int restatement in
    tests/golden/compressed.json -- both y parities, the identity, generator multiples -- and the decode errors."""
    cid = bp.CURVE_IDS[name]
    ctx = bp.Context(cid, 0)
    g = golden("compressed")[name]
    pts = b"".join(bytes.fromhex(c["point"]) for c in g["cases"])
    want = b"".join(bytes.fromhex(c["compressed"]) for c in g["cases"])
    n = len(g["cases"])
    v = bp.G1Vector.from_bytes(ctx, pts, n)
    assert v.to_compressed() == want
    assert bp.G1Vector.from_compressed(ctx, want, n).to_bytes() == pts
    per = len(want) // n
    assert {want[i * per] for i in range(n)} == {0, 2, 3}                 # identity and both parities are covered
    for bad in g["invalid"]:
        with pytest.raises(bp.ArgError):
            bp.G1Vector.from_compressed(ctx, want[:per] + bytes.fromhex(bad["bytes"]), 2)
    # random points at scale, round trip
    k = 5000
    big = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, 77, k), k))
    assert bp.G1Vector.from_compressed(ctx, big.to_compressed(), k).to_bytes() == big.to_bytes()
    ctx.close()


def test_r1cs_proof_compressed_form(bp, golden):
    from test_oracle_golden import r1cs_case_inputs
    from test_gpu_r1cs_oracle import start_transcript
    for name in CURVES:
        ctx = bp.Context(bp.CURVE_IDS[name], 0)
        for c in golden("r1cs")[name]:
            a = r1cs_case_inputs(c)
            n = c["n"]
            comp = bp.r1cs_proof_compress(ctx, n, a["proof"])
            assert len(comp) == bp.lib().bp_r1cs_proof_compressed_bytes(ctx.curve, n) < len(a["proof"])
            assert bp.r1cs_proof_decompress(ctx, n, comp) == a["proof"]
            bad = bytearray(comp)
            bad[0] = 7                                               # unknown tag on A_I1
            with pytest.raises(bp.VerificationError):
                bp.r1cs_proof_decompress(ctx, n, bytes(bad))
        ctx.close()
    assert bp.lib().bp_r1cs_proof_compressed_bytes(0, 1 << 16) == 2267 and bp.lib().bp_r1cs_proof_bytes(0, 1 << 16) == 4288


def test_tuning_knobs_are_validated_and_the_environment_changes_nothing(bp):
    """until round 2 BP_TILE / BP_REDUCE_M / BP_TASK_TARGET / ... were environment variables used as they were
    (BP_TILE=1000 silently dropped scalars).  Now: the library reads no environment variable that can change a result, and the
    knobs are per-context setters that refuse bad values.  Every accepted setting must give the oracle's bytes."""
    import os, subprocess, sys
    cid = 0
    ctx = bp.Context(cid, 0)
    n = 20000
    ks, ss = O.random_scalars(cid, 61, n), O.random_scalars(cid, 62, n)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
    sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    want = O.msm(cid, pts.to_bytes(), ss, n, algo=O.PIPPENGER, nthreads=8)
    assert pts.multi_scalar_mul_var_time(sv) == want
    for knob, bad in ((bp.TUNE_TILE, (1000, 255, 300, 16384 + 256, -1)), (bp.TUNE_REDUCE_M, (3, 6, 1 << 15, -2)), (bp.TUNE_TASK_TARGET, (5, 1 << 29)),
                      (bp.TUNE_SMALL_MSM, (2, 7)), (99, (1,))):
        for v in bad:
            with pytest.raises(bp.ArgError):
                ctx.set_tuning(knob, v)
        assert pts.multi_scalar_mul_var_time(sv) == want
    for knob, good in ((bp.TUNE_TILE, (256, 1024, 16384, 0)), (bp.TUNE_REDUCE_M, (1, 2, 16, 64, 0)), (bp.TUNE_TASK_TARGET, (1024, 1 << 20, 0)),
                       (bp.TUNE_SMALL_MSM, (0, 1))):
        for v in good:
            ctx.set_tuning(knob, v)
            assert pts.multi_scalar_mul_var_time(sv) == want, (knob, v)
            small = pts.msm_range(0, sv, 0, 300)
            assert small == O.msm(cid, pts.to_bytes(0, 300), ss[:300 * 32], 300, algo=O.PIPPENGER), (knob, v)
    ctx.close()
    # the variables of rounds 1-2, set to values that used to corrupt results, are ignored (fresh process: they were read once)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BP_TILE="1000", BP_REDUCE_M="3", BP_GROUPS="5", BP_TASK_TARGET="7", BP_ACC_WPS="9", BP_SMALL_MSM="0")
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_msm.py"), "3", "77"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "fails 0" in p.stdout.splitlines()[-1], p.stdout[-2000:] + p.stderr[-2000:]


def test_exported_vector_is_safe_to_free_while_a_view_computes(bp):
    """freed blocks go back to the context's pool without synchronising, which is only sound while every use is
    ordered on the owner's stream.  A vector whose raw pointer was handed out (device_ptr: torch / RCCL / a view on ANOTHER context)
    now waits for the device when it is freed: free the owner while the view's MSM is in flight, recycle the block at once with
    new contents, and the view's result must still be the old vector's."""
    cid = 0
    owner, other = bp.Context(cid, 0), bp.Context(cid, 0)
    n = 1 << 18
    ks, ss = O.random_scalars(cid, 71, n), O.random_scalars(cid, 72, n)
    pts = bp.G1Vector.fixed_base(owner, bp.FieldElementVector.from_bytes(owner, ks, n))
    sv = bp.FieldElementVector.from_bytes(owner, ss, n)
    want = O.g1_mul(cid, O.fr_inner(cid, ks, ss, n), O.generator(cid))
    owner.synchronize()
    pview = bp.G1Vector.wrap_device(other, pts.device_ptr(), n)
    sview = bp.FieldElementVector.wrap_device(other, sv.device_ptr(), n)
    pview.msm_begin(sview)                         # in flight on the other context's stream
    pts.free()                                     # owner frees: must not hand the blocks out again before the device is idle
    sv.free()
    junk = bp.FieldElementVector.from_bytes(owner, O.random_scalars(cid, 73, n), n)      # same size class: would reuse sv's block
    junk_pts = bp.G1Vector.fixed_base(owner, junk)                                       # ... and pts' block
    assert pview.msm_end() == want
    owner.synchronize()
    junk.free(); junk_pts.free()
    owner.close(); other.close()


def test_two_gpus_rccl_and_multi_device_contexts(bp):
    """Runs only where two GPUs are visible (the driver's multi-GPU box; skipped on the one-GPU pool): bench.py --gpus 2 --strong over
    the real nccl (= RCCL) backend, and bp_msm_g1_multi over contexts on two DIFFERENT device ordinals, both against the oracle."""
    import json, os, subprocess, sys
    if bp.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--strong", "--lg-n", "18", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["verified"] is True and "rehearsal" not in d
    cid = 0
    n = 1 << 16
    ks, ss = O.random_scalars(cid, 81, n), O.random_scalars(cid, 82, n)
    want = O.g1_mul(cid, O.fr_inner(cid, ks, ss, n), O.generator(cid))
    ctxs = [bp.Context(cid, 0), bp.Context(cid, 1)]
    half = n // 2
    pv, svs = [], []
    for i, c in enumerate(ctxs):
        kb, sb = ks[i * half * 32:(i + 1) * half * 32], ss[i * half * 32:(i + 1) * half * 32]
        pv.append(bp.G1Vector.fixed_base(c, bp.FieldElementVector.from_bytes(c, kb, half)))
        svs.append(bp.FieldElementVector.from_bytes(c, sb, half))
    assert bp.msm_multi(ctxs, pv, svs) == want
    # the inner-product argument with its generators sharded over the two devices: the single-device proof, byte for byte
    m = 1024
    gk, hk = O.random_scalars(cid, 83, m), O.random_scalars(cid, 84, m)
    ab, bb, gfb, hfb = (O.random_scalars(cid, 85 + i, m) for i in range(4))
    Q = O.g1_mul(cid, O.random_scalars(cid, 89, 1), O.generator(cid))
    fe = lambda c, b, k: bp.FieldElementVector.from_bytes(c, b, k)
    Gv, Hv = bp.G1Vector.fixed_base(ctxs[0], fe(ctxs[0], gk, m)), bp.G1Vector.fixed_base(ctxs[0], fe(ctxs[0], hk, m))
    single = bp.IPP.create_ipp(ctxs[0], bp.Transcript(b"2gpu"), Q, fe(ctxs[0], gfb, m), fe(ctxs[0], hfb, m), Gv, Hv, fe(ctxs[0], ab, m), fe(ctxs[0], bb, m))
    gb, hb, pb = Gv.to_bytes(), Hv.to_bytes(), ctxs[0].point_bytes
    cut = 400
    spans = ((0, cut), (cut, m))
    Gs = [bp.G1Vector.from_bytes(c, gb[lo * pb:hi * pb], hi - lo) for c, (lo, hi) in zip(ctxs, spans)]
    Hs = [bp.G1Vector.from_bytes(c, hb[lo * pb:hi * pb], hi - lo) for c, (lo, hi) in zip(ctxs, spans)]
    Gfs = [fe(c, gfb[lo * 32:hi * 32], hi - lo) for c, (lo, hi) in zip(ctxs, spans)]
    Hfs = [fe(c, hfb[lo * 32:hi * 32], hi - lo) for c, (lo, hi) in zip(ctxs, spans)]
    multi = bp.IPP.create_ipp_multi(ctxs, bp.Transcript(b"2gpu"), Q, Gfs, Hfs, Gs, Hs, ab, bb)
    assert (multi.L, multi.R, multi.a, multi.b) == (single.L, single.R, single.a, single.b)
    for c in ctxs:
        c.close()
