"""One rank's workload of each way to split BASELINE config 4 (a 2^22-point BLS12-381 MSM) over 8 GPUs, timed on ONE GPU
(VERDICT r3 #6): index range only (2^19 points x all 16 windows) against index range x window group (2^20 x 8, 2^21 x 4, 2^22 x 2).
Per shape: device stage into a record block (bp_msm_g1_windows / _windows_subset, synchronised) + the host fold of the 8 ranks'
blocks (bp_msm_g1_finish_blocks), best of `reps`; the whole 2^22 MSM on one GPU for the implied speed-up.  Every shape's full result
(all 8 ranks' blocks computed one after the other on this GPU) is checked against the one-GPU MSM."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
from bench import random_scalars

bp = G.load_package()
from bulletproofs_amcl_amd import sharding


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = 7
    n = 1 << lg
    ctx = bp.Context(bp.BLS12_381, 0)
    info = bp.curve_info(ctx.curve)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, random_scalars(ctx.r, info.fr_bits, n, 11), n))
    sv = bp.FieldElementVector.from_bytes(ctx, random_scalars(ctx.r, info.fr_bits, n, 12), n)
    ctx.synchronize()
    want = pts.multi_scalar_mul_var_time(sv)
    t1 = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); pts.multi_scalar_mul_var_time(sv); t1 = min(t1, time.perf_counter() - t0)
    print("one GPU, whole 2^%d MSM: %.3f ms" % (lg, t1 * 1e3), flush=True)
    rb = bp.msm_record_bytes(ctx.curve)
    c, cw, off, bias = bp.msm_geometry(ctx.curve, n // world)
    W = len(cw)
    for wg in (1, 2, 4, 8):
        if world % wg or W % wg:
            continue
        ig = world // wg
        n_set = n // ig
        ctx.set_window_bits(c)                      # one recoding for every rank (the width of the index-only shards: 16 at 2^19)
        stride = max(bp.msm_window_records_subset(ctx, n_set, g * (W // wg), W // wg) for g in range(wg))
        blocks = torch.zeros(world * stride * rb, dtype=torch.uint8, device=dev)
        for r in range(world):                      # all ranks' blocks (for the check)
            lo, hi, w0, wn = sharding.shard_2d(n, world, r, W, wg)
            bp.msm_windows_subset(ctx, pts, lo, sv, lo, hi - lo, w0, wn, stride, blocks.data_ptr() + r * stride * rb)
        got = bp.msm_finish_blocks(ctx, blocks.data_ptr(), world, stride, n_set)
        best_dev, best_fin = 1e9, 1e9
        lo, hi, w0, wn = sharding.shard_2d(n, world, world - 1, W, wg)
        for _ in range(reps):                       # rank (world - 1)'s device stage, then the fold of all blocks
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bp.msm_windows_subset(ctx, pts, lo, sv, lo, hi - lo, w0, wn, stride, blocks.data_ptr() + (world - 1) * stride * rb)
            t1a = time.perf_counter()
            bp.msm_finish_blocks(ctx, blocks.data_ptr(), world, stride, n_set)
            t2 = time.perf_counter()
            best_dev, best_fin = min(best_dev, t1a - t0), min(best_fin, t2 - t1a)
        ctx.set_window_bits(0)
        per_rank = best_dev + best_fin
        print("%d index groups x %d window groups: 2^%d points x %2d windows per rank | device stage %.3f ms + fold of %d blocks (%d records each) %.3f ms = %.3f ms"
              " | implied speed-up over one GPU %.2fx (+ the all-gather, ~0.05 ms: %.2fx) | result %s" %
              (ig, wg, (n_set).bit_length() - 1, W // wg, best_dev * 1e3, world, stride - 1, best_fin * 1e3, per_rank * 1e3, t1 / per_rank, t1 / (per_rank + 5e-5),
               "ok" if got == want else "WRONG"), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
