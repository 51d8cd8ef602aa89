"""Differential fuzz of IPP create / verify (both prover modes, both curves) against the CPU oracle (development aid).
usage: python scripts/fuzz_ipp.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
import _oracle as O
bp = G.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctxs = {0: bp.Context(0, 0), 1: bp.Context(1, 0)}
t_end, cases, fails = time.time() + budget, 0, 0
while time.time() < t_end:
    cid = rnd.randrange(2); ctx = ctxs[cid]; r = ctx.r
    n = rnd.choice([1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 64, 256, 1024, 2048])      # >= 256: tables can engage; 64 .. 512: split rounds over XYZZ rows; >= 1024: over affine rows
    ctx.set_ipp_fold_generators(rnd.random() < 0.3)
    seed = rnd.randrange(1 << 30)
    Gv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, seed, n), n))
    Hv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, seed + 1, n), n))
    Q = O.g1_mul(cid, O.random_scalars(cid, seed + 2, 1), O.generator(cid))
    def vec(kind, s):
        if kind == "rand": return O.random_scalars(cid, s, n)
        if kind == "ones": return (1).to_bytes(32, "little") * n
        if kind == "small": return b"".join(rnd.randrange(10).to_bytes(32, "little") for _ in range(n))
        return b"".join((0 if rnd.random() < 0.5 else rnd.randrange(r)).to_bytes(32, "little") for _ in range(n))
    ab, bb = vec(rnd.choice(["rand", "small", "sparse"]), seed + 3), vec(rnd.choice(["rand", "small", "sparse"]), seed + 4)
    gfb, hfb = vec(rnd.choice(["ones", "rand"]), seed + 5), vec(rnd.choice(["ones", "rand"]), seed + 6)
    dev = lambda b: bp.FieldElementVector.from_bytes(ctx, b, n)
    a, b, Gf, Hf = dev(ab), dev(bb), dev(gfb), dev(hfb)
    if n >= 256 and rnd.random() < 0.5:                              # round 3: precomputed generators (same or different widths)
        cG = rnd.choice([0, 16, rnd.randrange(6, 17)])
        Gv.precompute(cG); Hv.precompute(cG if rnd.random() < 0.8 else rnd.randrange(6, 17))
    proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"fuzz"), Q, Gf, Hf, Gv, Hv, a, b)
    rc, want = O.ipp_create(cid, O.Transcript(b"fuzz"), Q, gfb, hfb, Gv.to_bytes(), Hv.to_bytes(), ab, bb, n)
    ok = rc == 0 and (proof.L, proof.R, proof.a, proof.b) == want
    # P and verification on both sides
    pts = bp.G1Vector.from_bytes(ctx, Gv.to_bytes() + Hv.to_bytes() + Q, 2 * n + 1)
    sc = bp.FieldElementVector.from_bytes(ctx, a.hadamard_product(Gf).to_bytes() + b.hadamard_product(Hf).to_bytes() + a.inner_product(b), 2 * n + 1)
    P = pts.multi_scalar_mul_var_time(sc)
    try:
        bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"fuzz"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
    except bp.VerificationError:
        ok = False
    ok = ok and O.ipp_verify(cid, O.Transcript(b"fuzz"), n, gfb, hfb, P, Q, Gv.to_bytes(), Hv.to_bytes(), proof.a, proof.b, proof.L, proof.R, proof.lg_n) == 0
    bad = bytearray(proof.b); bad[rnd.randrange(31)] ^= 1 << rnd.randrange(8)
    try:
        bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"fuzz"), Gf, Hf, P, Q, Gv, Hv, proof.a, bytes(bad), proof.L, proof.R)
        ok = False
    except bp.VerificationError:
        pass
    cases += 1
    if not ok:
        fails += 1; print("FAIL", cid, n, seed, flush=True)
    if cases % 20 == 0: print("cases", cases, "fails", fails, flush=True)
print("done: cases", cases, "fails", fails, flush=True)
sys.exit(1 if fails else 0)
