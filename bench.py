#!/usr/bin/env python3
"""bench.py -- BLS12-381 G1 multi-scalar-multiplication throughput on MI355X (BASELINE.json metric).

A "step" is one MSM  sum_i s_i P_i  over synthetic inputs already resident in HBM: n_per_gpu = 2^20 random points
(k_i * G, generated on the device) and 2^20 uniformly random scalars per GPU.  With N > 1 ranks (one process per
GPU, launched by torch.distributed.run) the index range is sharded: every rank runs the bucket pipeline on its own
2^20-point slice, the per-window bucket sums (16 records x 192 B per rank) are all-gathered over RCCL, and the MSM
over all N * 2^20 points is finished on every rank ("scaling": "weak").  Output: ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--lg-n 20] [--curve bls12_381] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as G  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def random_scalars(r, bits, n, seed):
    """n uniform scalars in [0, r) as 32-byte little-endian rows (rejection sampling of `bits`-bit draws)."""
    rng = np.random.default_rng(seed)
    rw = np.array([(r >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.empty((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    while todo.size:
        w = rng.integers(0, 1 << 64, size=(todo.size, 4), dtype=np.uint64, endpoint=False)
        w[:, 3] &= top_mask
        lt = w[:, 0] < rw[0]
        for i in (1, 2, 3):
            lt = (w[:, i] < rw[i]) | ((w[:, i] == rw[i]) & lt)
        out[todo[lt]] = w[lt]
        todo = todo[~lt]
    return out.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lg-n", type=int, default=20, help="log2 of the points per GPU")
    ap.add_argument("--curve", default="bls12_381", choices=["bls12_381", "bn254"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="also time the same MSMs with two in flight (extra field; off by default so that rocprofv3 averages of the default "
                         "command are not mixed with concurrently running kernels)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Launched by torch.distributed.run (even with one rank): go through RCCL so that the sharded path is the one
    # that runs.  A plain `python bench.py` is the single-GPU path with no process group.
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner on stdout when the communicator is created; stdout must carry the one JSON line only,
        # so fd 1 points at stderr until the first collective has run.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            torch.cuda.set_device(local_rank)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    bp = G.load_package()
    from bulletproofs_amcl_amd import sharding
    curve = bp.CURVE_IDS[args.curve]
    ctx = bp.Context(curve, local_rank)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    info = bp.curve_info(curve)
    n = 1 << args.lg_n
    unit_bytes = 2 * info.fp_bytes + 32          # algorithmic bytes per scalar-mul: affine point + scalar

    # ---- synthetic inputs, resident in HBM before the timed region ------------------------------------------------
    k_bytes = random_scalars(ctx.r, info.fr_bits, n, 0xB0117E7 + 2 * rank)
    s_bytes = random_scalars(ctx.r, info.fr_bits, n, 0xB0117E7 + 2 * rank + 1)
    kv = bp.FieldElementVector.from_bytes(ctx, k_bytes, n)
    pts = bp.G1Vector.fixed_base(ctx, kv)             # P_i = k_i * G, generated on the device
    sv = bp.FieldElementVector.from_bytes(ctx, s_bytes, n)
    ctx.synchronize()

    W = bp.msm_window_records(ctx, n)
    rb = bp.msm_record_bytes(curve)
    mine = torch.zeros(W * rb, dtype=torch.uint8, device=dev)

    def step():
        if not use_dist:
            return pts.multi_scalar_mul_var_time(sv)
        bp.msm_windows(ctx, pts, 0, sv, 0, n, mine.data_ptr())
        allrec = sharding.all_gather_records(mine, world)          # RCCL all_gather_into_tensor
        torch.cuda.current_stream(dev).synchronize()               # the gather must be complete before the D2H of finish
        return bp.msm_finish(ctx, allrec.data_ptr(), world, n)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        result = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- extra (not `value`): the same K MSMs with two in flight (two contexts / HIP streams, one host thread).
    # The bucket reduce is latency-bound (one wave per SIMD), so a second MSM in flight fills the idle lanes.
    overlapped = None
    if not use_dist and args.overlap:
        ctxs2 = [bp.Context(curve, local_rank) for _ in range(2)]
        views = [(bp.G1Vector.wrap_device(c, pts.device_ptr(), n), bp.FieldElementVector.wrap_device(c, sv.device_ptr(), n)) for c in ctxs2]
        for p2, s2 in views:
            p2.multi_scalar_mul_var_time(s2)
        steps2 = max(2, args.steps)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        views[0][0].msm_begin(views[0][1])
        same = True
        for i in range(1, steps2):                       # MSM i is queued before MSM i-1 is finished on the host
            views[i & 1][0].msm_begin(views[i & 1][1])
            same &= views[(i - 1) & 1][0].msm_end() == result
        same &= views[(steps2 - 1) & 1][0].msm_end() == result
        torch.cuda.synchronize(dev)
        el2 = time.perf_counter() - t0
        overlapped = {"streams": 2, "steps": steps2, "ms_per_step": el2 / steps2 * 1e3, "value": n * steps2 / el2, "unit": "scalar-muls/s",
                      "same_result": bool(same), "how": "one host thread, bp_msm_g1_begin/_end alternating over two contexts"}
        for c in ctxs2:
            c.close()

    # ---- dominant-kernel roofline: HIP events on the kernel's own stream, separate passes ------------------------
    ctx.enable_timing(True)
    acc_ms, dev_ms = [], []
    for _ in range(max(3, min(args.steps, 10))):
        if not use_dist:
            pts.multi_scalar_mul_var_time(sv)
        else:
            bp.msm_windows(ctx, pts, 0, sv, 0, n, mine.data_ptr())
        tm = ctx.last_timing()
        if len(tm) >= 7:
            dev_ms.append(tm[0])
            acc_ms.append(tm[5])
    ctx.enable_timing(False)
    stages = {}
    if acc_ms:
        names = ["device_total", "digits_count", "scan", "scatter", "tasks", "accumulate", "reduce"]
        stages = {k: round(float(v), 4) for k, v in zip(names, tm)}

    out = None
    if rank == 0:
        total_units = world * n
        value = total_units * args.steps / elapsed
        roofline = None
        traffic, traffic_note = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_bench_n1_pmc_hbm.json")
        if args.curve == "bls12_381" and args.lg_n == 20 and os.path.exists(pmc):
            # HBM bytes per k_accumulate launch from the committed rocprofv3 --pmc passes of this same command
            # (FETCH_SIZE / WRITE_SIZE in KiB; read side doubled per MI355X_MICROARCH.md, uncalibrated for gathers)
            ks = json.load(open(pmc))["kernels"]
            k = next((v for name, v in ks.items() if name.startswith("bp::k_accumulate<bp::Bls381")), {})
            if "FETCH_SIZE_KiB_avg" in k and "WRITE_SIZE_KiB_avg" in k:
                traffic = int((2 * k["FETCH_SIZE_KiB_avg"] + k["WRITE_SIZE_KiB_avg"]) * 1024)
                traffic_note = "profiles/r01_bench_n1_pmc_hbm.json: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch"
        if acc_ms:
            avg = float(np.mean(acc_ms)) * 1e-3
            achieved = n * unit_bytes / avg / 1e9
            roofline = {"bound": "hbm", "kernel": "k_accumulate", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_note,
                        "kernel_ms": round(avg * 1e3, 4), "algorithmic_bytes_per_launch": n * unit_bytes,
                        "pipeline_GBs": round(n * unit_bytes / (float(np.mean(dev_ms)) * 1e-3) / 1e9, 2)}
            # The kernel is ALU-bound (VALU ~91 % busy, profiles/r01_bench_n1_pmc_sq.json), so the HBM fraction above says
            # little about it; beside it, the field-multiplication rate against the microbenchmarked ceiling of the same
            # multiplier (microbench/fpmul_rate.hip, formulation C at 8 waves/SIMD).  One mixed addition = 8M + 2S.
            if args.curve == "bls12_381":
                fpmul = n * int(W) * 10 / avg
                roofline["alu"] = {"achieved": round(fpmul, 0), "peak": 6.6e10, "unit": "381-bit Fp-mul/s", "frac": round(fpmul / 6.6e10, 4),
                                   "how": "n * windows mixed additions * 10 products each / kernel time; peak = microbench/fpmul_rate.hip"}
        out = {
            "metric": "BLS12-381 G1 scalar-muls/s at n=2^20 MSM" if (args.curve == "bls12_381" and args.lg_n == 20) else
                      "%s G1 scalar-muls/s at n=2^%d MSM" % (args.curve, args.lg_n),
            "value": value, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "2^%d-point %s G1 Pippenger MSM per GPU, uniform random scalars, points k_i*G, inputs resident in HBM"
                                   % (args.lg_n, args.curve),
                       "curve": args.curve, "n_per_gpu": n, "n_total": total_units, "windows": int(W),
                       "sharding": "index range per rank, all_gather of %d window records/rank over RCCL" % W if use_dist else "single GPU"},
            "stages_ms": stages,
            "roofline": roofline,
            "overlapped_2_streams": overlapped,
        }

    # ---- correctness of the timed result: MSM(s, k.G) == (<s,k> mod r).G  (oracle = checker only) ------------------
    if not args.no_verify:
        import _oracle as O
        if rank == 0:
            acc = 0
            for rk in range(world):
                kb = k_bytes if rk == 0 else random_scalars(ctx.r, info.fr_bits, n, 0xB0117E7 + 2 * rk)
                sb = s_bytes if rk == 0 else random_scalars(ctx.r, info.fr_bits, n, 0xB0117E7 + 2 * rk + 1)
                acc = (acc + int.from_bytes(O.fr_inner(curve, kb, sb, n), "little")) % ctx.r
            want = O.g1_mul(curve, acc.to_bytes(32, "little"), O.generator(curve))
            out["verified"] = bool(result == want)
            if not out["verified"]:
                print("ERROR: MSM result does not match the oracle", file=sys.stderr)

    # ---- CPU baseline on this box's host cores (rank 0, N = 1 only; bounded sample) --------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import _oracle as O
        cores = os.cpu_count() or 1
        ns = min(n, 1 << 17)                      # Strauss/wNAF-5, one thread: the reference's algorithm class
        host_pts = pts.to_bytes(0, ns)
        r1, sec1 = O.msm_timed(curve, host_pts, s_bytes[: ns * 32], ns, O.STRAUSS, 1)
        out["cpu_baseline"] = {"value": ns / sec1, "unit": "scalar-muls/s", "cores": 1, "kind": "port",
                               "sample": "first 2^%d terms of the same inputs, single-thread Strauss/wNAF-5 restatement of amcl_wrapper's "
                                         "multi_scalar_mul_var_time (oracle/orc_curve_tmpl.h), %.2f s" % (ns.bit_length() - 1, sec1)}
        nb = n
        host_pts = pts.to_bytes(0, nb)
        r2, sec2 = O.msm_timed(curve, host_pts, s_bytes[: nb * 32], nb, O.PIPPENGER, cores)
        out["cpu_baseline_best"] = {"value": nb / sec2, "unit": "scalar-muls/s", "cores": cores, "kind": "port",
                                    "sample": "first 2^%d terms, oracle Pippenger on all host cores, %.2f s" % (nb.bit_length() - 1, sec2)}
        # the two CPU results must agree with the GPU on their prefixes
        chk = pts.msm_range(0, sv, 0, ns)
        out["cpu_baseline"]["matches_gpu"] = bool(chk == r1)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
