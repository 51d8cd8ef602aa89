"""Differential fuzz of the GPU MSM / pair-MSM / begin-end paths against the CPU oracle (development aid).
usage: python scripts/fuzz_msm.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
import _oracle as O
bp = G.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctxs = {0: bp.Context(0, 0), 1: bp.Context(1, 0)}
pools = {}
for cid, ctx in ctxs.items():
    ks = O.random_scalars(cid, 4242 + cid, 4096)
    pools[cid] = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, 4096)).to_bytes()
t_end, cases, fails = time.time() + budget, 0, 0
while time.time() < t_end:
    cid = rnd.randrange(2); ctx = ctxs[cid]; pb = ctx.point_bytes; r = ctx.r
    n = rnd.choice([1, 2, 3, 7, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 5000, rnd.randrange(1, 20000), rnd.randrange(1, 70000)])
    kind = rnd.choice(["uniform", "small", "bits", "few_values", "one_value", "mostly_zero", "top_heavy", "r_minus"])
    def scalar():
        if kind == "uniform": return rnd.randrange(r)
        if kind == "small": return rnd.randrange(1 << rnd.choice([1, 8, 16, 17, 31, 33, 64]))
        if kind == "bits": return rnd.getrandbits(1)
        if kind == "few_values": return vals[rnd.randrange(len(vals))]
        if kind == "one_value": return vals[0]
        if kind == "mostly_zero": return 0 if rnd.random() < 0.9 else rnd.randrange(r)
        if kind == "top_heavy": return (r - 1 - rnd.randrange(1 << 40)) % r
        return r - 1 - rnd.randrange(3)
    vals = [rnd.randrange(r) for _ in range(rnd.choice([1, 2, 5, 16]))]
    ss = b"".join(scalar().to_bytes(32, "little") for _ in range(n))
    ptmode = rnd.choice(["pool", "pool", "dups", "with_identity", "neg_pairs"])
    idxs = [rnd.randrange(4096) for _ in range(n)]
    if ptmode == "dups": idxs = [idxs[i % max(1, rnd.choice([1, 2, 7]))] for i in range(n)]
    pts = bytearray(b"".join(pools[cid][i * pb:(i + 1) * pb] for i in idxs))
    if ptmode == "with_identity":
        for i in range(0, n, rnd.choice([2, 3, 10])): pts[i * pb:(i + 1) * pb] = bytes(pb)
    if ptmode == "neg_pairs" and n >= 2:
        p = int.from_bytes(bytes(bp.curve_info(cid).p_le)[: pb // 2], "little")
        for i in range(0, n - 1, 2):
            x = pts[i * pb:i * pb + pb // 2]; y = int.from_bytes(pts[i * pb + pb // 2:(i + 1) * pb], "little")
            pts[(i + 1) * pb:(i + 2) * pb] = x + ((p - y) % p).to_bytes(pb // 2, "little")
    pts = bytes(pts)
    c = rnd.choice([0, 0, 0, rnd.randrange(2, 17)])
    ctx.set_window_bits(c)
    ctx.set_device_tail(rnd.random() < 0.05)
    pv = bp.G1Vector.from_bytes(ctx, pts, n); sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
    tabled = rnd.random() < 0.3                                   # round 3: window-multiples table (merged-window pipeline), any width
    if tabled: pv.precompute(rnd.choice([0, 16, rnd.randrange(2, 17)]))
    knobs = rnd.random() < 0.2                                    # round 3: validated tuning knobs must never change a result
    if knobs:
        ctx.set_tuning(bp.TUNE_TILE, rnd.choice([0, 256, 512, 4096, 16384])); ctx.set_tuning(bp.TUNE_REDUCE_M, rnd.choice([0, 1, 2, 4, 16, 64]))
        ctx.set_tuning(bp.TUNE_TASK_TARGET, rnd.choice([0, 1024, 1 << 16, 1 << 22])); ctx.set_tuning(bp.TUNE_TAIL_CHAINS, rnd.choice([0, 1, 3, 8]))
        ctx.set_tuning(bp.TUNE_SMALL_MSM, rnd.choice([0, 1]))
    want = O.msm(cid, pts, ss, n, algo=O.PIPPENGER, nthreads=8)
    mode = rnd.choice(["msm", "pair", "beginend"])
    if mode == "msm": got = pv.multi_scalar_mul_var_time(sv)
    elif mode == "beginend": pv.msm_begin(sv); got = pv.msm_end()
    else:
        s2 = b"".join((0 if rnd.random() < 0.5 else rnd.randrange(r)).to_bytes(32, "little") for _ in range(n))
        got, got2 = pv.multi_scalar_mul_pair(sv, bp.FieldElementVector.from_bytes(ctx, s2, n))
        if got2 != O.msm(cid, pts, s2, n, algo=O.PIPPENGER, nthreads=8):
            fails += 1; print("FAIL pair second", cid, n, kind, ptmode, c, flush=True)
    cases += 1
    if got != want:
        fails += 1; print("FAIL", mode, cid, n, kind, ptmode, c, "tabled" if tabled else "", "knobs" if knobs else "", flush=True)
    ctx.set_window_bits(0); ctx.set_device_tail(False)
    if knobs:
        for k in (bp.TUNE_TILE, bp.TUNE_REDUCE_M, bp.TUNE_TASK_TARGET, bp.TUNE_TAIL_CHAINS): ctx.set_tuning(k, 0)
        ctx.set_tuning(bp.TUNE_SMALL_MSM, 1)
    if cases % 50 == 0: print("cases", cases, "fails", fails, flush=True)
print("done: cases", cases, "fails", fails, flush=True)
sys.exit(1 if fails else 0)
