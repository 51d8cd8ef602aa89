import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
bp = G.load_package()
def rs(n, seed):
    rng = np.random.default_rng(seed); a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x1F; return a.tobytes()
ctx = bp.Context(0, 0)
for lg, cs in ((16, [12, 13, 14, 15, 16]), (12, [8, 9, 10, 11, 12]), (6, [3, 4, 5, 6])):
    n = 1 << lg
    Gv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rs(n, 1), n))
    Hv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rs(n, 2), n))
    Q = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rs(1, 3), 1)).to_bytes()
    a = bp.FieldElementVector.from_bytes(ctx, rs(n, 4), n); b = bp.FieldElementVector.from_bytes(ctx, rs(n, 5), n)
    Gf = bp.FieldElementVector.from_ints(ctx, [1] * n); Hf = bp.FieldElementVector.new_vandermonde_vector(ctx, rs(1, 6), n)
    for c in cs:
        ctx.set_window_bits(c)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b); best = min(best, time.perf_counter() - t0)
        print("n=2^%d c=%d create=%.2fms (%.2f ms/round)" % (lg, c, best * 1e3, best * 1e3 / lg), flush=True)
    ctx.set_window_bits(0)
