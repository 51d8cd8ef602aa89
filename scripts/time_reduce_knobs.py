"""Stage times of one MSM shape under the tuning knobs that move work between the accumulate and the bucket reduce (development aid).
usage: time_reduce_knobs.py <lg n> [curve]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
from scripts.time_msm import rand_scalars
bp = G.load_package()
lg = int(sys.argv[1])
curve = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1 << lg
ctx = bp.Context(curve, 0)
pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 1), n))
sv = bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 2), n)
ctx.enable_timing(True)
want = None
for c in (0, 12, 13, 14, 15):
    for tt in (0, 65536, 131072, 393216):
        for rm in (0, 1, 2, 4):
            ctx.set_window_bits(c)
            ctx.set_tuning(bp.TUNE_TASK_TARGET, tt)
            try:
                ctx.set_tuning(bp.TUNE_REDUCE_M, rm)
                r = pts.multi_scalar_mul_var_time(sv)
            except Exception as e:
                print("c=%d tt=%d m=%d: %s" % (c, tt, rm, e)); continue
            want = want or r
            assert r == want
            wall, acc, red, tot = [], [], [], []
            for _ in range(7):
                t0 = time.perf_counter(); pts.multi_scalar_mul_var_time(sv); wall.append(time.perf_counter() - t0)
                ms = ctx.last_timing()
                acc.append(ms[5]); red.append(ms[6]); tot.append(ms[0])
            med = lambda v: sorted(v)[len(v) // 2]
            print("c=%2d task_target=%6d reduce_m=%d: wall %.3f ms  device %.3f  accumulate %.3f  reduce+combine %.3f" % (c, tt, rm, med(wall) * 1e3, med(tot), med(acc), med(red)), flush=True)
