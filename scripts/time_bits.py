"""Per-stage timing of an MSM whose scalars are all 0 or 1 (the a_L / a_R commitments).  usage: time_bits.py [lgn]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402
from scripts.time_msm import rand_scalars  # noqa: E402

bp = G.load_package()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << lg
ctx = bp.Context(0, 0)
ctx.enable_timing(True)
pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 1), n))
bits = np.zeros((n, 32), dtype=np.uint8)
bits[:, 0] = np.random.default_rng(7).integers(0, 2, size=n)
sv = bp.FieldElementVector.from_bytes(ctx, bits.tobytes(), n)
for _ in range(3):
    pts.multi_scalar_mul_var_time(sv)
for _ in range(3):
    t0 = time.time()
    pts.multi_scalar_mul_var_time(sv)
    wall = (time.time() - t0) * 1e3
    tm = ctx.last_timing()
    print("bits n=2^%d wall=%.3fms device=%.3fms [count %.3f scan %.3f scatter %.3f tasks %.3f accumulate %.3f reduce %.3f]" % (lg, wall, *tm[:7]),
          flush=True)
