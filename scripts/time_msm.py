"""Quick per-stage timing of the MSM device pipeline (development aid; bench.py is the contract)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

bp = G.load_package()


def rand_scalars(ctx, n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x3F if ctx.curve == 0 else 0x1F     # < 2^254 < r (BLS12-381) / < 2^253 < r (BN254): uploads reject scalars >= r
    return a.tobytes()


def main():
    lgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 18, 20]
    curve = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    ctx = bp.Context(curve, 0)
    notime = bool(os.environ.get("TIME_MSM_NOTIMING"))      # wall clock only: the per-stage events cost ~0.1-0.2 ms per MSM
    ctx.enable_timing(not notime)
    if os.environ.get("BP_DEVICE_TAIL"):
        ctx.set_device_tail(True)
    # sweep knobs of THIS script (the library itself reads no tuning variable): TM_TILE / TM_REDUCE_M / TM_TASK_TARGET / TM_SMALL_MSM
    for var, knob in (("TM_TILE", bp.TUNE_TILE), ("TM_REDUCE_M", bp.TUNE_REDUCE_M), ("TM_TASK_TARGET", bp.TUNE_TASK_TARGET), ("TM_SMALL_MSM", bp.TUNE_SMALL_MSM), ("TM_TAIL_CHAINS", bp.TUNE_TAIL_CHAINS)):
        if os.environ.get(var):
            ctx.set_tuning(knob, int(os.environ[var]))
    for lg in lgs:
        n = 1 << lg
        kv = bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 1), n)
        t0 = time.time()
        pts = bp.G1Vector.fixed_base(ctx, kv)
        ctx.synchronize()
        tgen = time.time() - t0
        sv = bp.FieldElementVector.from_bytes(ctx, rand_scalars(ctx, n, 2), n)
        for c in cs:
            ctx.set_window_bits(c)
            pts.multi_scalar_mul_var_time(sv)
            best = None
            for _ in range(5):
                t0 = time.time()
                pts.multi_scalar_mul_var_time(sv)
                wall = (time.time() - t0) * 1e3
                tm = ctx.last_timing() if not notime else [0.0] * 7
                if best is None or wall < best[0]:
                    best = (wall, tm)
            wall, tm = best
            print("n=2^%d c=%d gen=%.1fms wall=%.3fms device=%.3fms [count %.3f scan %.3f scatter %.3f tasks %.3f accumulate %.3f reduce %.3f] host_tail~%.3fms  %.3e muls/s" % (
                lg, c, tgen * 1e3, wall, tm[0], tm[1], tm[2], tm[3], tm[4], tm[5], tm[6], wall - tm[0], n / (wall * 1e-3)), flush=True)
        pts.free()


if __name__ == "__main__":
    main()
