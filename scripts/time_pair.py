"""Per-stage timing of the PAIRED MSM at the shape of a default-prover IPP round: 2n+1 resident points, two scalar sets
with n non-zero scalars each (the other half zero).  usage: time_pair.py [lgn,...] [c,...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

bp = G.load_package()
from scripts.time_msm import rand_scalars  # noqa: E402

lgs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "6,12,16").split(",")]
cs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
ctx = bp.Context(0, 0)
ctx.enable_timing(True)
for lg in lgs:
    n = 1 << lg
    N = 2 * n + 1
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, N, 1), N))
    a = np.frombuffer(rand_scalars(ctx, N, 2), dtype=np.uint8).reshape(N, 32).copy()
    b = np.frombuffer(rand_scalars(ctx, N, 3), dtype=np.uint8).reshape(N, 32).copy()
    # L uses [G_R | H_L | Q], R uses [G_L | H_R | Q]: half of each generator vector per set
    h = n // 2
    a[:h] = 0
    a[n + h:2 * n] = 0
    b[h:n] = 0
    b[n:n + h] = 0
    sa = bp.FieldElementVector.from_bytes(ctx, a.tobytes(), N)
    sb = bp.FieldElementVector.from_bytes(ctx, b.tobytes(), N)
    for c in cs:
        ctx.set_window_bits(c)
        pts.multi_scalar_mul_pair(sa, sb)
        best = None
        for _ in range(5):
            t0 = time.time()
            pts.multi_scalar_mul_pair(sa, sb)
            wall = (time.time() - t0) * 1e3
            tm = ctx.last_timing()
            if best is None or wall < best[0]:
                best = (wall, tm)
        wall, tm = best
        print("pair n=2^%d (N=%d) c=%d wall=%.3fms device=%.3fms [count %.3f scan %.3f scatter %.3f tasks %.3f accumulate %.3f reduce %.3f] host~%.3fms"
              % (lg, N, c, wall, tm[0], tm[1], tm[2], tm[3], tm[4], tm[5], tm[6], wall - tm[0]), flush=True)
