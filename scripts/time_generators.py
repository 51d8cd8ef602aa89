"""Time get_generators (hash-to-G1) on the GPU against the CPU oracle.  usage: time_generators.py [lgn,lgn,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as G  # noqa: E402
import _oracle as O  # noqa: E402

bp = G.load_package()
lgs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,16,20").split(",")]
ncpu = os.cpu_count() or 1
for name, cid in bp.CURVE_IDS.items():
    ctx = bp.Context(cid, 0)
    bp.get_generators(ctx, "warm", 64)
    sample = 512
    t = time.perf_counter()
    ref = O.get_generators(cid, "G", sample, nthreads=1)
    cpu1 = (time.perf_counter() - t) / sample
    t = time.perf_counter()
    O.get_generators(cid, "G", sample * 8, nthreads=ncpu)
    cpun = (time.perf_counter() - t) / (sample * 8)
    for lg in lgs:
        n = 1 << lg
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            v = bp.get_generators(ctx, "G", n)
            best = min(best, time.perf_counter() - t)
        assert v.to_bytes(0, sample) == ref[: sample * ctx.point_bytes][: min(n, sample) * ctx.point_bytes] or n < sample
        print("%s n=2^%d gpu=%.2fms (%.3e points/s)  cpu oracle: %.3f ms/point 1 thread, %.4f ms/point %d threads -> x%.0f / x%.0f"
              % (name, lg, best * 1e3, n / best, cpu1 * 1e3, cpun * 1e3, ncpu, cpu1 * n / best, cpun * n / best), flush=True)
    ctx.close()
