"""Where the time of an R1CS proof at BASELINE config 3 goes: wall-clock of each C-ABI call made by r1cs.prove (monkey-patched
timers; development aid)."""
import os
import sys
import time
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G  # noqa: E402

bp = G.load_package()
import r1cs_twin as R1  # noqa: E402
from bench import random_scalars  # noqa: E402
from bench_configs import bound_check_chain  # noqa: E402

ctx = bp.Context(0, 0)
info = bp.curve_info(0)
r = ctx.r
terms, nq, aL, aR, aO, v = bound_check_chain(r, 1024, 32, np.random.default_rng(2024))
n, m = len(aL), len(v)
gens = R1.Generators(ctx, n)
plan = bp.R1CSPlan(ctx, terms, nq, n, m)
small = lambda xs: bp.FieldElementVector.from_bytes(ctx, b"".join(int(x).to_bytes(32, "little") for x in xs), len(xs))
dAL, dAR, dAO = small(aL), small(aR), small(aO)
vb = [int.from_bytes(random_scalars(r, info.fr_bits, 1, 9000 + j), "little") for j in range(m)]
dVB = small(vb)
V = gens.commit_many(v, vb)
sL = bp.FieldElementVector.from_bytes(ctx, random_scalars(r, info.fr_bits, n, 9100), n)
sR = bp.FieldElementVector.from_bytes(ctx, random_scalars(r, info.fr_bits, n, 9101), n)
bl = {k: 12345 + i for i, k in enumerate(("i", "o", "s", "t1", "t3", "t4", "t5", "t6"))}

acc = defaultdict(float)


def timed(obj, name, label=None):
    f = getattr(obj, name)

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        out = f(*a, **k)
        ctx.synchronize()
        acc[label or name] += time.perf_counter() - t0
        return out
    setattr(obj, name, wrapper)


timed(bp.G1Vector, "multi_scalar_mul_var_time", "msm (3 commitments over 2n+1, 5 T, Q)")
timed(bp.R1CSPlan, "flattened_constraints")
timed(bp, "r1cs_prover_polys")
timed(bp.VecPoly3, "special_inner_product")
timed(bp.VecPoly3, "eval")
timed(bp, "r1cs_ipp_inputs")
timed(bp.IPP, "create_ipp")
timed(R1, "_cat")
timed(R1, "start_transcript")
for rep in range(2):
    acc.clear()
    t0 = time.perf_counter()
    R1.prove(ctx, gens, plan, R1.start_transcript(ctx, b"cfg3", V), dAL, dAR, dAO, dVB, sL, sR, bl)
    total = time.perf_counter() - t0
print("total %.2f ms" % (total * 1e3))
for k, t in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-46s %7.2f ms" % (k, t * 1e3))
print("  %-46s %7.2f ms" % ("(other: transcript, host scalars)", (total - sum(acc.values())) * 1e3))
