"""Development aid: wall time of each of N consecutive 2^20 MSMs right after input generation (is there a clock / cache ramp?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
from scripts.time_msm import rand_scalars
bp = G.load_package()
ctx = bp.Context(0, 0)
n = 1 << 20
pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 1), n))
sv = bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 2), n)
ctx.synchronize()
if len(sys.argv) > 1:
    time.sleep(float(sys.argv[1]))
ts = []
for i in range(40):
    t0 = time.perf_counter()
    pts.multi_scalar_mul_var_time(sv)
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.2f" % t for t in ts))
