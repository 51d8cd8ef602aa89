"""Wall-clock (and, under rocprofv3, the kernels) of bp_ipp_verify alone: plain, or over the generators' window tables (development aid).
usage: time_verify.py <curve> <lg n> [table width | none] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
bp = importlib.import_module("bulletproofs-amcl_amd")
from bench_configs import ipp_instance

curve, lg = int(sys.argv[1]), int(sys.argv[2])
tw = sys.argv[3] if len(sys.argv) > 3 else "none"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
n = 1 << lg
ctx = bp.Context(curve, 0)
Gv, Hv, Q, Gf, Hf, a, b, P = ipp_instance(ctx, n, 5)
proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
if tw != "none":
    Gv.precompute(int(tw)); Hv.precompute(int(tw)); ctx.synchronize()
    ctx.set_tuning(bp.TUNE_VERIFY_TABLES, 2)
ver = lambda: bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
ver(); ver()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); ver(); ts.append(time.perf_counter() - t0)
ts.sort()
print("verify curve=%d n=2^%d tables=%s: median %.3f ms, best %.3f ms (kept table: %s)" % (curve, lg, tw, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, ctx.verify_table_info()), flush=True)
