"""One MSM shape, repeated, for rocprofv3 --kernel-trace --stats.  usage: prof_one.py lgn c [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
from scripts.time_msm import rand_scalars
bp = G.load_package()
lg, c = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = bp.Context(0, 0)
n = 1 << lg
pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 1), n))
sv = bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 2), n)
ctx.set_window_bits(c)
for _ in range(reps):
    pts.multi_scalar_mul_var_time(sv)
