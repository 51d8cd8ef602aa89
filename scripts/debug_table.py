"""Development aid: one small MSM over a table, stage by stage (BP_TRACE=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
import _oracle as O

bp = G.load_package()
curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
c = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = bp.Context(curve, 0)
ks = O.random_scalars(curve, 1, n)
ss = O.random_scalars(curve, 2, n)
pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, ks, n))
sv = bp.FieldElementVector.from_bytes(ctx, ss, n)
want = pts.multi_scalar_mul_var_time(sv)
print("plain ok", flush=True)
pts.precompute(c)
ctx.synchronize()
print("table built", pts.table_info(), flush=True)
got = pts.multi_scalar_mul_var_time(sv)
print("table msm", "ok" if got == want else "MISMATCH", flush=True)
