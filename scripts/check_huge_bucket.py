"""One-off check of the heavy-bucket combine with more than 256 chunks (nt > 65536 task sums in ONE bucket): all scalars 1
at n = 2^20 with a tiny task length forced through bp_ctx_set_tuning(BP_TUNE_TASK_TARGET).  Verified by linearity: sum_i (k_i G) = (sum k_i) G."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G  # noqa: E402
import _oracle as O  # noqa: E402
from scripts.time_msm import rand_scalars  # noqa: E402

bp = G.load_package()
for curve in (0, 1):
    ctx = bp.Context(curve, 0)
    ctx.set_tuning(bp.TUNE_TASK_TARGET, 1 << 26)
    n = 1 << 20
    kb = rand_scalars(ctx, n, 5)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, kb, n))
    ones = bp.FieldElementVector.from_bytes(ctx, (1).to_bytes(32, "little") * n, n)
    for c in (0, 12):
        ctx.set_window_bits(c)
        got = pts.multi_scalar_mul_var_time(ones)
        want = O.g1_mul(curve, O.fr_inner(curve, kb, (1).to_bytes(32, "little") * n, n), O.generator(curve))
        print("curve", curve, "c", c, "ok" if got == want else "MISMATCH", flush=True)
        assert got == want
