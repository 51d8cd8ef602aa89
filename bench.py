#!/usr/bin/env python3
"""bench.py -- BLS12-381 G1 multi-scalar-multiplication throughput on MI355X (BASELINE.json metric).

A "step" is one MSM  sum_i s_i P_i  over synthetic inputs already resident in HBM: random points (k_i * G, generated on
the device) and uniformly random scalars.  With N > 1 ranks (one process per GPU) the index range is sharded: every rank
runs the bucket pipeline on its own slice, its tail records (208 x 192 B at 2^20 points: per window one weighted sum + one record per bit plane; + a geometry header) are
all-gathered over RCCL, and the MSM over all points is finished on every rank.
  default ("scaling": "weak")   2^lg-n points PER GPU (2^20: the headline configuration, BASELINE config 2)
  --strong ("scaling": "strong") 2^lg-n points IN TOTAL, split by index range (default 2^22: BASELINE config 4, 2^19 per GPU at 8)
`python bench.py --gpus N` with no RANK in the environment starts the N ranks itself (torch.distributed.run, before anything
touches a GPU) and relays rank 0's line; under an external torchrun it joins the group it is given.  Output: ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--strong] [--lg-n L] [--curve bls12_381] [--no-cpu-baseline]
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
ACC_KERNEL_PREFIX = "k_accumulate<bp::Bls381"   # dominant kernel as rocprofv3 names it (void bp::k_accumulate<bp::Bls381, 2>(...))


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


def visible_gpu_count():
    """GPUs this process could use, WITHOUT touching the HIP runtime (the launcher must never initialise it: it only ever starts
    child processes).  KFD topology nodes with SIMDs are GPUs; a *_VISIBLE_DEVICES list narrows them.  None = unknown."""
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """--gpus N > 1 without a process group in the environment: start N ranks (one per GPU) as CHILD processes and relay rank 0's
    JSON line.  This process never imports torch and never calls into HIP; it must only ever spawn children (an exec from a process
    that has initialised the GPU takes the machine down on this pool).  To profile the ranks, put rocprofv3 around an external
    `python -m torch.distributed.run ... bench.py`, not around this launcher (scripts/README.md)."""
    import socket
    import subprocess
    have = visible_gpu_count()
    if have is not None and have < args.gpus and not args.rehearse_one_device:
        print("bench.py: --gpus %d but only %d GPU(s) are visible" % (args.gpus, have), file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    return p.returncode if p.returncode else (0 if lines else 1)


def size_sweep(bp, ctx, curve, info, unit_bytes, lgs=(16, 17, 18, 19, 20, 21, 22), reps=5):
    """MSM over the first 2^lg of one resident 2^22-point input set, best of `reps` wall-clock runs per size (inputs resident in
    HBM, result on the host), each result checked by linearity:  MSM(s, k.G) == (<s, k> mod r).G  with the oracle as the checker."""
    import _oracle as O
    nmax = 1 << max(lgs)
    kb = random_scalars(ctx.r, info.fr_bits, nmax, 0x5EE9)
    sb = random_scalars(ctx.r, info.fr_bits, nmax, 0x5EEA)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, kb, nmax))
    sv = bp.FieldElementVector.from_bytes(ctx, sb, nmax)
    ctx.synchronize()
    out = {}
    for lg in lgs:
        n = 1 << lg
        got = pts.msm_range(0, sv, 0, n)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            got = pts.msm_range(0, sv, 0, n)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        want = O.g1_mul(curve, O.fr_inner(curve, kb[: 32 * n], sb[: 32 * n], n), O.generator(curve))
        out["2^%d" % lg] = {"ms": round(best * 1e3, 4), "scalar_muls_per_s": round(n / best, 1), "algorithmic_GBs": round(n * unit_bytes / best / 1e9, 2),
                            "verified": bool(got == want)}
    pts.free()
    sv.free()
    ctx.trim()
    return out


def h2d_headline(bp, ctx, torch, pts, s_bytes, n, expect, steps):
    """The headline MSM with its scalars arriving from the HOST every step: pageable memory (what a Rust Vec is) and page-locked
    memory (what a binding that owns its staging buffer can offer).  Upload = bp_frvec_upload: H2D copy + the canonicity check."""
    import numpy as np

    def run(src):
        sv2 = bp.FieldElementVector.from_bytes(ctx, src, n)
        got = pts.multi_scalar_mul_var_time(sv2)
        sv2.free()
        return got

    res = {"workload": "the 2^%d-point MSM of `value` with its %d MiB of scalars uploaded inside every step" % (n.bit_length() - 1, 32 * n >> 20)}
    pinned = torch.from_numpy(np.frombuffer(s_bytes, dtype=np.uint8).copy()).pin_memory()
    for name, src in (("pageable", s_bytes), ("pinned", bp.HostPtr(pinned.data_ptr(), 32 * n))):
        ok = run(src) == expect
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            got = run(src)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[name] = {"ms_per_step": dt * 1e3, "value": n / dt if (ok and got == expect) else None, "unit": "scalar-muls/s", "verified": bool(ok and got == expect)}
    res["ms_per_step"] = res["pinned"]["ms_per_step"]
    res["value"] = res["pinned"]["value"]
    return res


def strong_extra(bp, sharding, ctx, curve, info, dev, world, rank, use_dist, dist, torch, args, lg_total=22, steps=5, warmup=1, window_groups=1):
    """BASELINE config 4 inside the weak run: 2^22 points IN TOTAL split over the ranks, timed like the main value (barrier +
    synchronize on both sides, max over ranks) and verified against the oracle on rank 0.  window_groups = 1: by index range (what
    north_star names); > 1: index range x window group (sharding.shard_2d) -- world / window_groups index groups, every rank
    1 / window_groups of the windows of a window_groups times longer slice."""
    import _oracle as O
    n_total = 1 << lg_total
    igroups = world // window_groups
    grp = rank // window_groups                        # index group of this rank (ranks of one group hold the same points)
    lo, hi = sharding.shard_range(n_total, igroups, grp)
    n = hi - lo
    n_set = sharding.largest_shard(n_total, igroups)
    if n_total % igroups or window_groups > 1:
        ctx.set_window_bits(sharding.common_window_bits(bp, curve, n_total, igroups))
    seed_of = lambda g: 0xC0F164 + 2 * g
    kb = random_scalars(ctx.r, info.fr_bits, n, seed_of(grp))
    sb = random_scalars(ctx.r, info.fr_bits, n, seed_of(grp) + 1)
    pts = bp.G1Vector.fixed_base(ctx, bp.FieldElementVector.from_bytes(ctx, kb, n))
    sv = bp.FieldElementVector.from_bytes(ctx, sb, n)
    ctx.synchronize()
    if window_groups > 1:
        nwin = len(bp.msm_geometry(curve, n_set, sharding.common_window_bits(bp, curve, n_total, igroups))[1])
        _, _, w0, wn = sharding.shard_2d(n_total, world, rank, nwin, window_groups)
        W = max(bp.msm_window_records_subset(ctx, n_set, g * wn, wn) for g in range(window_groups))
    else:
        W = bp.msm_window_records(ctx, n_set)
    mine = torch.zeros(W * bp.msm_record_bytes(curve), dtype=torch.uint8, device=dev)

    def step():
        if window_groups > 1:
            bp.msm_windows_subset(ctx, pts, 0, sv, 0, n, w0, wn, W, mine.data_ptr())
        else:
            bp.msm_windows(ctx, pts, 0, sv, 0, n, mine.data_ptr())
        allrec = sharding.all_gather_records(mine, world)
        torch.cuda.current_stream(dev).synchronize()
        if window_groups > 1:
            return bp.msm_finish_blocks(ctx, allrec.data_ptr(), world, W, n_set)
        return bp.msm_finish(ctx, allrec.data_ptr(), world, n_set)

    def fence():
        dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        result = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        result = step()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_one_device else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ctx.set_window_bits(0)
    out = None
    if rank == 0:
        acc = 0
        for g in range(igroups):
            a, b = sharding.shard_range(n_total, igroups, g)
            k2 = kb if g == 0 else random_scalars(ctx.r, info.fr_bits, b - a, seed_of(g))
            s2 = sb if g == 0 else random_scalars(ctx.r, info.fr_bits, b - a, seed_of(g) + 1)
            acc = (acc + int.from_bytes(O.fr_inner(curve, k2, s2, b - a), "little")) % ctx.r
        want = O.g1_mul(curve, acc.to_bytes(32, "little"), O.generator(curve))
        ok = bool(result == want)
        how = "index range split over %d GPUs" % world if window_groups == 1 else \
              "%d index groups x %d window groups over %d GPUs (every rank 1/%d of the windows of a 2^%d-point slice)" % (igroups, window_groups, world, window_groups, n_set.bit_length() - 1)
        out = {"workload": "2^%d-point bls12_381 G1 MSM in total, %s (BASELINE config 4 shape)" % (lg_total, how),
               "shard_mode": "index" if window_groups == 1 else "index_x_windows", "index_groups": igroups, "window_groups": window_groups,
               "scaling": "strong", "n_total": n_total, "n_per_gpu": n_set, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
               "value": n_total * steps / elapsed if ok else None, "unit": "scalar-muls/s", "verified": ok,
               "speedup_vs_1gpu_same_n": None,       # needs the N = 1 run's sweep["2^22"]: the driver has both lines, this one has not
               "one_gpu_reference": "sweep[\"2^%d\"] of the --gpus 1 line" % lg_total}
    pts.free()
    sv.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lg-n", type=int, default=None, help="log2 of the points per GPU (weak, default 20) or in total (--strong, default 22)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the total size is fixed and split over the ranks (BASELINE config 4)")
    ap.add_argument("--curve", default="bls12_381", choices=["bls12_381", "bn254"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra keys of the default line (N = 1: \"sweep\" over 2^16..2^22; N > 1: \"strong_2p22\")")
    ap.add_argument("--extras", action="store_true", help="emit the extra keys although --lg-n is not the headline size (tests)")
    ap.add_argument("--sweep-max-lg", type=int, default=22, help="largest size of the N = 1 sweep (tests shrink it)")
    ap.add_argument("--strong-lg", type=int, default=22, help="log2 of the total size of the N > 1 strong-scaling extra (tests shrink it)")
    ap.add_argument("--window-groups", type=int, default=0, help="window groups of the 2-D strong-scaling extra (0: 4 if the world size allows, else 2)")
    ap.add_argument("--configs-small", action="store_true", help="shrunk sizes for the \"configs\" extra (cfg1 / cfg3_e2e / cfg5; tests)")
    ap.add_argument("--overlap", action="store_true",
                    help="also time the same MSMs with two in flight (extra field; off by default so that rocprofv3 averages of the default "
                         "command are not mixed with concurrently running kernels)")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="N ranks that all use GPU 0 and exchange their records over gloo: rehearses the N > 1 code path (self-launch, "
                         "sharding, gather, finish) on a one-GPU box; the line is marked \"rehearsal\": true and is not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.lg_n is None:
        args.lg_n = 22 if args.strong else 20
    in_group = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_group:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1")) if in_group else 1
    rank = int(os.environ.get("RANK", "0")) if in_group else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if in_group else 0
    if args.rehearse_one_device:
        local_rank = 0
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d does not match WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # Launched by torch.distributed.run (even with one rank): go through RCCL so that the sharded path is the one
    # that runs.  A plain `python bench.py` is the single-GPU path with no process group.
    use_dist = in_group
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner on stdout when the communicator is created; stdout must carry the one JSON line only,
        # so fd 1 points at stderr until the first collective has run.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            import datetime
            tmo = datetime.timedelta(seconds=120)
            if args.rehearse_one_device:
                dist.init_process_group(backend="gloo", timeout=tmo)
            else:
                if local_rank >= torch.cuda.device_count():
                    raise RuntimeError("rank %d: local rank %d has no GPU (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
            torch.cuda.set_device(local_rank)
            dist.barrier()
            torch.cuda.synchronize()
        except Exception as e:      # never a hang, never a retry: one line and a non-zero exit
            print("bench.py: rank %d could not join the %d-rank process group (%s backend): %s" % (rank, world, "gloo" if args.rehearse_one_device else "nccl/RCCL", e),
                  file=sys.stderr, flush=True)
            os._exit(3)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    bp = G.load_package()
    from bulletproofs_amcl_amd import sharding
    curve = bp.CURVE_IDS[args.curve]
    ctx = bp.Context(curve, local_rank)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    info = bp.curve_info(curve)
    if args.strong:
        n_total = 1 << args.lg_n
        lo, hi = sharding.shard_range(n_total, world, rank)
        n = hi - lo
        n_set = sharding.largest_shard(n_total, world)
        if world > 1 and n_total % world:
            ctx.set_window_bits(sharding.common_window_bits(bp, curve, n_total, world))    # unequal shards: one width for all ranks
    else:
        n = 1 << args.lg_n
        n_total = world * n
        n_set = n
    shard_sizes = [n] if not args.strong else [b - a for a, b in (sharding.shard_range(n_total, world, r) for r in range(world))]
    unit_bytes = 2 * info.fp_bytes + 32          # algorithmic bytes per scalar-mul: affine point + scalar

    # ---- synthetic inputs, resident in HBM before the timed region ------------------------------------------------
    seed_of = lambda rk: 0xB0117E7 + 2 * rk
    k_bytes = random_scalars(ctx.r, info.fr_bits, n, seed_of(rank))
    s_bytes = random_scalars(ctx.r, info.fr_bits, n, seed_of(rank) + 1)
    kv = bp.FieldElementVector.from_bytes(ctx, k_bytes, n)
    pts = bp.G1Vector.fixed_base(ctx, kv)             # P_i = k_i * G, generated on the device
    sv = bp.FieldElementVector.from_bytes(ctx, s_bytes, n)
    ctx.synchronize()

    W = bp.msm_window_records(ctx, n_set)             # window records + geometry header per rank
    rb = bp.msm_record_bytes(curve)
    mine = torch.zeros(W * rb, dtype=torch.uint8, device=dev)

    def step():
        if not use_dist:
            return pts.multi_scalar_mul_var_time(sv)
        bp.msm_windows(ctx, pts, 0, sv, 0, n, mine.data_ptr())
        allrec = sharding.all_gather_records(mine, world)          # RCCL all_gather_into_tensor
        torch.cuda.current_stream(dev).synchronize()               # the gather must be complete before the D2H of finish
        return bp.msm_finish(ctx, allrec.data_ptr(), world, n_set)

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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_one_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- extra (not `value`): the same K MSMs with two in flight (two contexts / HIP streams, one host thread).
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
    n_windows = len(bp.msm_geometry(curve, n_set, 0)[1])

    out = None
    if rank == 0:
        value = n_total * args.steps / elapsed
        roofline = None
        # HBM bytes per k_accumulate launch: from a committed rocprofv3 --pmc run of THIS command whose kernel names match the
        # library being benchmarked (profiles/README lists how it was taken); raw FETCH_SIZE + WRITE_SIZE (the guide's x2 read
        # correction is calibrated for wide streaming reads, not for 96-byte row gathers, so it is not applied).
        traffic, traffic_note = None, None
        pmc = os.path.join(ROOT, "profiles", "r04_bench_n1_pmc_hbm.json")
        if args.curve == "bls12_381" and args.lg_n == 20 and not args.strong and os.path.exists(pmc):
            pj = json.load(open(pmc))
            k = next((v for name, v in pj.get("kernels", {}).items() if ACC_KERNEL_PREFIX in name), {})
            if "FETCH_SIZE_KiB_avg" in k and "WRITE_SIZE_KiB_avg" in k:
                traffic = int((k["FETCH_SIZE_KiB_avg"] + k["WRITE_SIZE_KiB_avg"]) * 1024)
                traffic_note = "stored profile profiles/r04_bench_n1_pmc_hbm.json (%s): raw FETCH_SIZE + WRITE_SIZE per launch" % pj.get("taken", "?")
        if acc_ms:
            avg = float(np.mean(acc_ms)) * 1e-3
            achieved = n * unit_bytes / avg / 1e9
            roofline = {"bound": "hbm", "kernel": "k_accumulate", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_note,
                        "kernel_ms": round(avg * 1e3, 4), "algorithmic_bytes_per_launch": n * unit_bytes,
                        "pipeline_GBs": round(n * unit_bytes / (float(np.mean(dev_ms)) * 1e-3) / 1e9, 2)}
            # The kernel is integer-ALU bound, so the HBM fraction says little about it.  Beside it: the rate of mixed additions
            # against (a) the instruction-ISSUE ceiling -- the kernel's mixed addition is 6 products + 2 squares + one two-product /
            # one-reduction form (bp_curve.cuh: xyzz_lazy_add_aff_fast) with the fused Montgomery core of round 3
            # (bp_field.cuh: mont_core; per product 169 + 169 + 3 mads + 13 v_mul_lo) = 6 x 354 + 2 x 276 + 549 = 3225 multiply-class instructions
            # (v_mad_u64_u32 / v_mul_lo) at the measured 4.5 cycles per wave64 instruction per SIMD, 1024 SIMDs at 2.4 GHz, nothing
            # else counted -- and (b) the multiply-instruction rate this library's own multiplier loop sustains
            # (microbench/fpmul_rate.hip: 6.6e10 products/s x 351).
            if args.curve == "bls12_381":
                adds = n * n_windows / avg
                mul_instr = 6 * 354 + 2 * 276 + 549
                issue_peak = 1024 * 64 * 2.4e9 / (mul_instr * 4.5)
                roofline["alu"] = {"achieved": round(adds, 0), "unit": "mixed additions/s", "peak": round(issue_peak, 0), "frac": round(adds / issue_peak, 4),
                                   "how": "n * windows additions / kernel time; peak = issue limit of the multiply instructions alone (6*354 + 2*276 + 549 = 3225 per addition, "
                                          "4.5 cyc per wave64 instruction, 1024 SIMDs, 2.4 GHz)",
                                   "vs_own_multiplier_microbench": round(adds * mul_instr / (6.6e10 * 351), 4)}
        out = {
            "metric": "BLS12-381 G1 scalar-muls/s at n=2^20 MSM" if (args.curve == "bls12_381" and args.lg_n == 20 and not args.strong) else
                      "%s G1 scalar-muls/s at n=2^%d MSM%s" % (args.curve, args.lg_n, " (total, strong scaling)" if args.strong else ""),
            "value": value, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("2^%d-point %s G1 Pippenger MSM in total, index range split over the GPUs" if args.strong else
                                    "2^%d-point %s G1 Pippenger MSM per GPU") % (args.lg_n, args.curve) +
                                   ", uniform random scalars, points k_i*G, inputs resident in HBM",
                       "curve": args.curve, "n_per_gpu": max(shard_sizes), "n_total": n_total, "windows": n_windows,
                       "sharding": "index range per rank, all_gather of %d records/rank (windows + geometry header) over RCCL" % W if use_dist else "single GPU"},
            "stages_ms": stages,
            "roofline": roofline,
            "overlapped_2_streams": overlapped,
        }
        if args.rehearse_one_device:
            out["rehearsal"] = True

    # ---- correctness of the timed result: MSM(s, k.G) == (<s,k> mod r).G  (oracle = checker only) ------------------
    failed = False
    if not args.no_verify:
        import _oracle as O
        if rank == 0:
            acc = 0
            for rk in range(world):
                nk = shard_sizes[rk] if args.strong else n
                kb = k_bytes if rk == 0 else random_scalars(ctx.r, info.fr_bits, nk, seed_of(rk))
                sb = s_bytes if rk == 0 else random_scalars(ctx.r, info.fr_bits, nk, seed_of(rk) + 1)
                acc = (acc + int.from_bytes(O.fr_inner(curve, kb, sb, nk), "little")) % ctx.r
            want = O.g1_mul(curve, acc.to_bytes(32, "little"), O.generator(curve))
            out["verified"] = bool(result == want)
            if not out["verified"]:
                failed = True
                print("ERROR: MSM result does not match the oracle -- no value is reported", file=sys.stderr)
                out["value"] = None                      # a wrong result is not a measurement

    # ---- CPU baseline on this box's host cores (rank 0, N = 1 only; bounded sample) --------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not failed:
        import _oracle as O
        cores = os.cpu_count() or 1
        ns = min(n, 1 << 17)                      # Strauss/wNAF-5, one thread: the reference's algorithm class
        host_pts = pts.to_bytes(0, ns)
        r1, sec1 = O.msm_timed(curve, host_pts, s_bytes[: ns * 32], ns, O.STRAUSS, 1)
        out["cpu_baseline"] = {"value": ns / sec1, "unit": "scalar-muls/s", "cores": 1, "kind": "port",
                               "sample": "first 2^%d terms of the same inputs, single-thread Strauss/wNAF-5 restatement of amcl_wrapper's "
                                         "multi_scalar_mul_var_time (oracle/orc_curve_tmpl.h), %.2f s" % (ns.bit_length() - 1, sec1)}
        nb = min(n, 1 << 20)
        host_pts = pts.to_bytes(0, nb)
        r2, sec2 = O.msm_timed(curve, host_pts, s_bytes[: nb * 32], nb, O.PIPPENGER, cores)
        out["cpu_baseline_best"] = {"value": nb / sec2, "unit": "scalar-muls/s", "cores": cores, "kind": "port",
                                    "sample": "first 2^%d terms, oracle Pippenger on all host cores, %.2f s" % (nb.bit_length() - 1, sec2)}
        # the two CPU results must agree with the GPU on their prefixes
        chk = pts.msm_range(0, sv, 0, ns)
        out["cpu_baseline"]["matches_gpu"] = bool(chk == r1)

    # ---- extras in the SAME line (never `value`): what the driver's one command would otherwise not measure -------
    #   N = 1: the n = 2^16 .. 2^22 sweep of north_star (best of 5 each, every size verified by linearity)
    #   N > 1: BASELINE config 4 -- a 2^22-point MSM split by index range over the N ranks (strong scaling)
    # (the decision must be the same on every rank: it gates collectives.  `failed` is only known to rank 0, so it gates the N = 1 sweep alone.)
    extras = not args.no_extras and not args.strong and args.curve == "bls12_381" and (args.lg_n == 20 or args.extras)
    if extras and world == 1 and not failed:
        out["sweep"] = size_sweep(bp, ctx, curve, info, unit_bytes, lgs=tuple(range(min(16, args.sweep_max_lg), args.sweep_max_lg + 1)))
        # The reference's call site hands over HOST scalars with every call (multi_scalar_mul_var_time(&scalars), src/ipp.rs:251-253):
        # the same MSM with the 32 n bytes of scalars uploaded inside the step.  Never `value`.
        out["with_scalar_h2d"] = h2d_headline(bp, ctx, torch, pts, s_bytes, n, result, max(3, min(args.steps, 10)))
        pts.free(); sv.free(); kv.free()
        ctx.trim()
        import bench_configs as BC
        out["configs"] = BC.driver_configs(small=args.configs_small)
    if extras and world > 1:
        st = strong_extra(bp, sharding, ctx, curve, info, dev, world, rank, use_dist, dist, torch, args, lg_total=args.strong_lg)
        if rank == 0:
            out["strong_2p%d" % args.strong_lg] = st
        # the same total with the WINDOWS split as well (round 4; one-GPU timings of a rank's share, profiles/r04_shard_shapes.log:
        # 8 x 1 -> 1.96 ms, 2 x 4 -> 1.77 ms per rank): 4 window groups where the world size allows, else 2
        wg = args.window_groups if args.window_groups else (4 if world % 4 == 0 else 2 if world % 2 == 0 else 1)
        if wg > 1 and world % wg == 0:
            st2 = strong_extra(bp, sharding, ctx, curve, info, dev, world, rank, use_dist, dist, torch, args, lg_total=args.strong_lg, window_groups=wg)
            if rank == 0:
                out["strong_2p%d_index_x_windows" % args.strong_lg] = st2
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
