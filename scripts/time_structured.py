"""Development aid: MSM wall time for structured scalar distributions (buckets of very uneven weight) at one size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
import _oracle as O
from scripts.time_msm import rand_scalars
bp = G.load_package()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ctx = bp.Context(0, 0)
ctx.enable_timing(True)          # HIP events around the pipeline's stages (a few microseconds per MSM)
n = 1 << lg
kb = rand_scalars(ctx, n, 1)
pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, kb, n))
rng = np.random.default_rng(3)
def small(bits):
    a = np.zeros((n, 32), dtype=np.uint8)
    v = rng.integers(0, 1 << bits, size=n, dtype=np.uint64)
    for i in range(8):
        a[:, i] = (v >> (8 * i)) & 0xff
    return a.tobytes()
def few(k):
    vals = np.frombuffer(rand_scalars(ctx, k, 9), dtype=np.uint8).reshape(k, 32)
    return vals[rng.integers(0, k, size=n)].tobytes()
def zero_or_minus_one():      # a_R = a_L - 1 of a bit decomposition (positive_no.rs:18-24): 0 or r - 1
    a = np.zeros((n, 32), dtype=np.uint8)
    a[rng.integers(0, 2, size=n) == 1] = np.frombuffer((ctx.r - 1).to_bytes(32, "little"), dtype=np.uint8)
    return a.tobytes()
kinds = {"uniform": rand_scalars(ctx, n, 2), "bits": small(1), "0 or r-1": zero_or_minus_one(), "8-bit": small(8), "16-bit": small(16), "20-bit": small(20), "32-bit": small(32), "2 values": few(2), "16 values": few(16),
         "256 values": few(256), "4096 values": few(4096)}
only = [k.strip() for k in os.environ.get("TS_KINDS", "").split(",") if k.strip()]      # e.g. TS_KINDS="bits,256 values"
for name, sb in kinds.items():
    if only and name not in only:
        continue
    sv = bp.FieldElementVector.from_bytes(ctx, sb, n)
    got = pts.multi_scalar_mul_var_time(sv)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); got = pts.multi_scalar_mul_var_time(sv); best = min(best, time.perf_counter() - t0)
    ok = got == O.g1_mul(0, O.fr_inner(0, kb, sb, n), O.generator(0))
    tm = ctx.last_timing()          # whole pipeline, digits, scan, scatter, tasks, accumulate, combine + reduce
    print("n=2^%d %-12s %8.3f ms  %s   device %.2f = sort %.2f + tasks %.2f + accumulate %.2f + combine/reduce %.2f" % (
        lg, name, best * 1e3, "ok" if ok else "MISMATCH", tm[0], tm[1] + tm[2] + tm[3], tm[4], tm[5], tm[6]), flush=True)
