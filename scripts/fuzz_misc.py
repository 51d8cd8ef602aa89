"""Differential fuzz of the round's later additions against the CPU oracle (development aid):
  * hash to G1: random ragged message batches and get_generators(prefix, first, n) vs the oracle's restatement,
  * batch verification: random batches over shared generators; accepted iff nothing was tampered with, and the
    verdict always agrees with the per-proof verifier,
  * R1CS prove / verify: random satisfiable circuits; the library's proof equals the Python mirror's byte for byte, both
    verifiers accept it, a flipped bit is rejected.
usage: python scripts/fuzz_misc.py [seconds] [seed]"""
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
    cid = rnd.randrange(2); ctx = ctxs[cid]; pb = ctx.point_bytes
    kind = rnd.choice(["hash", "gens", "batch", "r1cs", "r1cs"])
    ok = True
    if kind == "hash":
        msgs = [bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 3, 8, 31, 64, 135, 136, 137, 200, 272, 300, 409])))
                for _ in range(rnd.choice([1, 2, 17, 64, 65, 130]))]
        got = bp.G1Vector.from_msg_hash(ctx, msgs).to_bytes()
        ok = all(got[i * pb:(i + 1) * pb] == O.g1_from_msg_hash(cid, m) for i, m in enumerate(msgs))
    elif kind == "gens":
        prefix = bytes(rnd.randrange(32, 127) for _ in range(rnd.choice([0, 1, 5, 40, 120, 130, 136, 200])))
        first = rnd.choice([0, 1, 9, 99, 12345, 10**9 - 3, 2**63, 2**64 - 800])   # first + n - 1 must stay below 2^64
        n = rnd.choice([1, 3, 64, 257, 700])
        ok = bp.get_generators(ctx, prefix, n, first=first).to_bytes() == O.get_generators(cid, prefix, n, first=first, nthreads=16)
    elif kind == "r1cs":
        # random satisfiable single-phase circuit: bp_r1cs_prove (C++ in the library) == the Python mirror byte for byte,
        # both verifiers accept, and a flipped byte anywhere in the proof is rejected
        import r1cs_twin as R1
        r = ctx.r
        n = rnd.choice([1, 2, 3, 5, 8, 13, 31, 64, 100]); m = rnd.choice([0, 1, 2, 5]); nq = rnd.randrange(0, 3 * n + 2)
        aL = [rnd.randrange(r) for _ in range(n)]; aR = [rnd.randrange(r) for _ in range(n)]; aO = [x * y % r for x, y in zip(aL, aR)]
        vals = [rnd.randrange(r) for _ in range(m)]; vbl = [rnd.randrange(r) for _ in range(m)]
        val = {0: aL, 1: aR, 2: aO, 3: vals}
        terms = []
        for q in range(nq):
            acc = 0
            for _ in range(rnd.randrange(0, 5)):
                k = rnd.choice([0, 1, 2, 3]) if m else rnd.choice([0, 1, 2])
                i = rnd.randrange(m if k == 3 else n); c = rnd.choice([1, r - 1, rnd.randrange(r)])
                terms.append((q, k, i, c)); acc = (acc + c * val[k][i]) % r
            if acc:
                terms.append((q, 4, 0, (-acc) % r))
        gens = R1.Generators(ctx, R1.padded(n) * rnd.choice([1, 2]))
        V = gens.commit_many(vals, vbl)
        plan = bp.R1CSPlan(ctx, terms, nq, n, m)
        dev = lambda xs: bp.FieldElementVector.from_ints(ctx, xs)
        names = ("i", "o", "s", "t1", "t3", "t4", "t5", "t6")
        bl = {k: rnd.randrange(r) for k in names}
        sL = [rnd.randrange(r) for _ in range(n)]; sR = [rnd.randrange(r) for _ in range(n)]
        le = lambda x: int(x).to_bytes(32, "little")
        py = R1.prove(ctx, gens, plan, R1.start_transcript(ctx, b"fz", V), dev(aL), dev(aR), dev(aO), dev(vbl), dev(sL), dev(sR), bl)
        raw = bp.r1cs_prove(ctx, R1.start_transcript(ctx, b"fz", V), plan, gens.G, gens.H, gens.g, gens.h, dev(aL), dev(aR), dev(aO),
                            dev(vbl) if m else None, dev(sL), dev(sR), b"".join(le(bl[k]) for k in names))
        ipp = py["ipp"]
        flat = (py["A_I1"] + py["A_O1"] + py["S1"] + bytes(3 * pb) + b"".join(py["T"][k] for k in (1, 3, 4, 5, 6))
                + le(py["t_x"]) + le(py["t_x_blinding"]) + le(py["e_blinding"]) + ipp.L + ipp.R + ipp.a + ipp.b)
        def lib_ok(proof):
            try:
                bp.r1cs_verify(ctx, R1.start_transcript(ctx, b"fz", V), plan, gens.G, gens.H, gens.g, gens.h, b"".join(V), n, proof, le(rnd.randrange(r)))
                return True
            except bp.VerificationError:
                return False
        ok = raw == flat and lib_ok(raw) and R1.verify(ctx, gens, plan, R1.start_transcript(ctx, b"fz", V), V, py)
        bad = bytearray(raw)
        pos = rnd.choice([rnd.randrange(3 * pb), 6 * pb + rnd.randrange(5 * pb), 11 * pb + rnd.randrange(96), len(raw) - 1 - rnd.randrange(64)])
        bad[pos] ^= 1 << rnd.randrange(8)
        try:
            ok = ok and not lib_ok(bytes(bad))
        except bp.BpError:
            pass                                           # a flipped coordinate may no longer be a curve point: any error is a rejection
        plan.free()
    else:
        n = rnd.choice([1, 2, 8, 32, 64]); m = rnd.choice([1, 2, 3, 7, 16]); seed = rnd.randrange(1 << 30); r = ctx.r
        Gv = bp.get_generators(ctx, "g%d" % seed, n); Hv = bp.get_generators(ctx, "h%d" % seed, n)
        Gf = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, seed, n), n) if rnd.random() < 0.5 else bp.FieldElementVector.from_ints(ctx, [1] * n)
        Hf = bp.FieldElementVector.new_vandermonde_vector(ctx, O.random_scalars(cid, seed + 1, 1), n)
        items = []
        for j in range(m):
            Q = bp.G1Vector.from_msg_hash(ctx, [b"Q%d-%d" % (seed, j)]).to_bytes()
            a = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, seed + 10 + j, n), n)
            b = bp.FieldElementVector.from_bytes(ctx, O.random_scalars(cid, seed + 50 + j, n), n)
            pr = bp.IPP.create_ipp(ctx, bp.Transcript(b"fz%d" % j), Q, Gf, Hf, Gv, Hv, a, b)
            pts = bp.G1Vector.from_bytes(ctx, Gv.to_bytes() + Hv.to_bytes() + Q, 2 * n + 1)
            sc = bp.FieldElementVector.from_bytes(ctx, a.hadamard_product(Gf).to_bytes() + b.hadamard_product(Hf).to_bytes() + a.inner_product(b), 2 * n + 1)
            items.append([b"fz%d" % j, pts.multi_scalar_mul_var_time(sc), Q, pr.a, pr.b, pr.L, pr.R])
        tamper = rnd.random() < 0.5
        if tamper:
            it = items[rnd.randrange(m)]
            f = rnd.choice([3, 4] + ([5, 6] if n > 1 else []) + [1])
            if f in (3, 4):
                it[f] = ((int.from_bytes(it[f], "little") + 1 + rnd.randrange(1000)) % r).to_bytes(32, "little")
            else:                                       # replace one point by another valid point
                k = rnd.randrange(len(it[f]) // pb)
                it[f] = it[f][:k * pb] + O.g1_mul(cid, O.random_scalars(cid, seed + 99, 1), O.generator(cid)) + it[f][(k + 1) * pb:]
        def single_all():
            for lab, P, Q, a_, b_, L, R in items:
                try:
                    bp.IPP.verify_ipp(ctx, n, bp.Transcript(lab), Gf, Hf, P, Q, Gv, Hv, a_, b_, L, R)
                except bp.VerificationError:
                    return False
            return True
        try:
            bp.IPP.verify_batch(ctx, n, Gf, Hf, Gv, Hv, [(bp.Transcript(it[0]),) + tuple(it[1:]) for it in items])
            batch_ok = True
        except bp.VerificationError:
            batch_ok = False
        ok = batch_ok == (not tamper) and single_all() == (not tamper)
    cases += 1
    if not ok:
        fails += 1; print("FAIL", kind, cid, flush=True)
    if cases % 20 == 0: print("cases", cases, "fails", fails, flush=True)
print("done: cases", cases, "fails", fails, flush=True)
sys.exit(1 if fails else 0)
