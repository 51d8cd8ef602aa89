"""Wall-clock of IPP create / verify through the C ABI (development aid)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

bp = G.load_package()


def rs(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x1F
    return a.tobytes()


def main():
    curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    lgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6, 12]
    ctx = bp.Context(curve, 0)
    if os.environ.get("TIME_IPP_C"):
        ctx.set_window_bits(int(os.environ["TIME_IPP_C"]))       # window width for every MSM of the run (0 = the library's rule)
    if os.environ.get("TIME_IPP_CHAINS"):
        ctx.set_tuning(bp.TUNE_TAIL_CHAINS, int(os.environ["TIME_IPP_CHAINS"]))
    if os.environ.get("TIME_IPP_TASK_TARGET"):
        ctx.set_tuning(bp.TUNE_TASK_TARGET, int(os.environ["TIME_IPP_TASK_TARGET"]))
    if os.environ.get("TIME_IPP_NO_GLV"):
        ctx.set_tuning(bp.TUNE_GLV, 1)
    if os.environ.get("TIME_IPP_COMPACT_AT"):
        ctx.set_tuning(bp.TUNE_COMPACT_AT, int(os.environ["TIME_IPP_COMPACT_AT"]))     # 1 = never compact the generators
    for lg in lgs:
        n = 1 << lg
        Gv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rs(n, 1), n))
        Hv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rs(n, 2), n))
        Qv = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, rs(1, 3), 1))
        Q = Qv.to_bytes()
        a = bp.FieldElementVector.from_bytes(ctx, rs(n, 4), n)
        b = bp.FieldElementVector.from_bytes(ctx, rs(n, 5), n)
        Gf = bp.FieldElementVector.from_ints(ctx, [1] * n)
        Hf = bp.FieldElementVector.new_vandermonde_vector(ctx, rs(1, 6), n)
        # P = <a.Gf, G> + <b.Hf, H> + <a,b> Q
        pts = bp.G1Vector.from_bytes(ctx, Gv.to_bytes() + Hv.to_bytes() + Q, 2 * n + 1)
        sc = bp.FieldElementVector.from_bytes(ctx, a.to_bytes() + b.hadamard_product(Hf).to_bytes() + a.inner_product(b), 2 * n + 1)
        P = pts.multi_scalar_mul_var_time(sc)
        if os.environ.get("TIME_IPP_TABLES"):                    # precomputed generators: TIME_IPP_TABLES = table width (0 = automatic)
            t0 = time.perf_counter()
            Gv.precompute(int(os.environ["TIME_IPP_TABLES"])); Hv.precompute(int(os.environ["TIME_IPP_TABLES"])); ctx.synchronize()
            print("tables: c=%d W=%d, %.1f MB per vector, built in %.1f ms" % (Gv.table_info()[0], Gv.table_info()[1], Gv.table_info()[2] / 1e6, (time.perf_counter() - t0) * 1e3))
        best_c, best_v = 1e9, 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            proof = bp.IPP.create_ipp(ctx, bp.Transcript(b"innerproduct"), Q, Gf, Hf, Gv, Hv, a, b)
            t1 = time.perf_counter()
            bp.IPP.verify_ipp(ctx, n, bp.Transcript(b"innerproduct"), Gf, Hf, P, Q, Gv, Hv, proof.a, proof.b, proof.L, proof.R)
            t2 = time.perf_counter()
            best_c, best_v = min(best_c, t1 - t0), min(best_v, t2 - t1)
        if os.environ.get("TIME_IPP_SHARDS"):                    # the sharded prover over k contexts (here: all on device 0 -- measures its overhead, not a speed-up)
            k = int(os.environ["TIME_IPP_SHARDS"])
            cs = [bp.Context(curve, 0) for _ in range(k)]
            pb, per = ctx.point_bytes, n // k
            gb, hb, gfb, hfb = Gv.to_bytes(), Hv.to_bytes(), Gf.to_bytes(), Hf.to_bytes()
            Gs = [bp.G1Vector.from_bytes(c, gb[i * per * pb:(i + 1) * per * pb], per) for i, c in enumerate(cs)]
            Hs = [bp.G1Vector.from_bytes(c, hb[i * per * pb:(i + 1) * per * pb], per) for i, c in enumerate(cs)]
            Gfs = [bp.FieldElementVector.from_bytes(c, gfb[i * per * 32:(i + 1) * per * 32], per) for i, c in enumerate(cs)]
            Hfs = [bp.FieldElementVector.from_bytes(c, hfb[i * per * 32:(i + 1) * per * 32], per) for i, c in enumerate(cs)]
            if os.environ.get("TIME_IPP_TABLES"):
                for v in Gs + Hs:
                    v.precompute(int(os.environ["TIME_IPP_TABLES"]))
            ab, bb = a.to_bytes(), b.to_bytes()
            best_m = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                pm = bp.IPP.create_ipp_multi(cs, bp.Transcript(b"innerproduct"), Q, Gfs, Hfs, Gs, Hs, ab, bb)
                best_m = min(best_m, time.perf_counter() - t0)
            print("curve=%d n=2^%d sharded over %d contexts of ONE device: create=%.2fms same_proof=%s" % (curve, lg, k, best_m * 1e3, (pm.L, pm.R, pm.a, pm.b) == (proof.L, proof.R, proof.a, proof.b)), flush=True)
            for c in cs:
                c.close()
        # per-round split with the state API
        st = bp.IPPState(ctx, Gv, Hv, Q, Gf, Hf, a, b)
        tr = bp.Transcript(b"innerproduct")
        t_round = t_fold = 0.0
        while len(st) > 1:
            t0 = time.perf_counter()
            L, R = st.round()
            t1 = time.perf_counter()
            tr.commit_point(curve, b"L", L); tr.commit_point(curve, b"R", R)
            u = tr.challenge_scalar(curve, b"u")
            ui = bp.fr_inverse(curve, u)
            t2 = time.perf_counter()
            st.fold(u, ui); ctx.synchronize()
            t3 = time.perf_counter()
            t_round += t1 - t0; t_fold += t3 - t2
        print("curve=%d n=2^%d create=%.2fms verify=%.2fms | rounds(L,R MSMs)=%.2fms folds=%.2fms" % (curve, lg, best_c * 1e3, best_v * 1e3, t_round * 1e3, t_fold * 1e3), flush=True)


if __name__ == "__main__":
    main()
