"""Throughput of back-to-back MSMs issued from two host threads, each with its own context/stream (experiment)."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
bp = G.load_package()
def rs(n, seed):
    rng = np.random.default_rng(seed); a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x1F; return a.tobytes()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
K = 24
ctx0 = bp.Context(0, 0)
pts = bp.G1Vector.fixed_base(ctx0, bp.FieldElementVector.from_bytes(ctx0, rs(n, 1), n))
sb = rs(n, 2)
sv0 = bp.FieldElementVector.from_bytes(ctx0, sb, n)
ref = pts.multi_scalar_mul_var_time(sv0)
t0 = time.perf_counter()
for _ in range(K): pts.multi_scalar_mul_var_time(sv0)
t1 = time.perf_counter() - t0
print("1 context : %.3f ms/MSM  %.3e muls/s" % (t1 / K * 1e3, n * K / t1), flush=True)
for nthreads in (2, 3):
    ctxs = [bp.Context(0, 0) for _ in range(nthreads)]
    views = [(bp.G1Vector.wrap_device(c, pts.device_ptr(), n), bp.FieldElementVector.wrap_device(c, sv0.device_ptr(), n)) for c in ctxs]
    for p, s in views: assert p.multi_scalar_mul_var_time(s) == ref
    def work(i):
        p, s = views[i]
        for _ in range(K // nthreads): assert p.multi_scalar_mul_var_time(s) == ref
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t2 = time.perf_counter() - t0
    done = (K // nthreads) * nthreads
    print("%d contexts: %.3f ms/MSM  %.3e muls/s" % (nthreads, t2 / done * 1e3, n * done / t2), flush=True)
