"""CPU, world_size 2, gloo: the N > 1 path of the sharded MSM without a GPU.  Each rank plays the DEVICE stage with the oracle
(the W window sums of its index range, recoded with the library's own window table, bp_msm_geometry) and packs them as the
library's record block; the exchange step (sharding.all_gather_records, RCCL on GPUs / gloo here) and stage 2 -- header
validation + the host fold, bp_msm_g1_finish_host, the same code bp_msm_g1_finish runs after its D2H copy -- are the
PRODUCT's.  Ragged shards (501 + 500) with the common window width of sharding.common_window_bits."""
import os
import subprocess
import sys

import pytest

import __graft_entry__ as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
import __graft_entry__ as G
import _oracle as O
bp = G.load_package()
from bulletproofs_amcl_amd import sharding
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
curve, n = 0, 1001                                   # ragged: 501 + 500
r = O.group_order(curve)
ks = O.random_scalars(curve, 5, n); ss = O.random_scalars(curve, 6, n)
pts = O.fixed_base_batch(curve, ks, n, 2)
lo, hi = sharding.shard_range(n, world, rank)
c = sharding.common_window_bits(bp, curve, n, world)
nmax = sharding.largest_shard(n, world)
_, cw, off, bias = bp.msm_geometry(curve, nmax, c)
# device stage, played by the oracle: S_w = sum_i digit_w(k_i) P_i over this rank's index range, placed in the record of weight 2^off_w
pos = bp.msm_record_positions(curve, nmax, c)
zero = bytes(bp.msm_record_bytes(curve))
recs = [zero] * len(pos)
for w in range(len(cw)):
    digs = b""
    for i in range(lo, hi):
        k = int.from_bytes(ss[32 * i:32 * i + 32], "little") + bias
        d = ((k >> off[w]) & ((1 << cw[w]) - 1)) - ((1 << (cw[w] - 1)) - 1)
        digs += (d %% r).to_bytes(32, "little")
    S = O.msm(curve, pts[lo * 96:hi * 96], digs, hi - lo, algo=O.PIPPENGER)
    slot = next(j for j in range(len(pos)) if pos[j] == off[w] and recs[j] is zero)
    recs[slot] = bp.msm_record_from_affine(curve, S)
block = b"".join(recs)
block += bp.msm_record_header(curve, nmax, c)
mine = torch.frombuffer(bytearray(block), dtype=torch.uint8)
allrec = sharding.all_gather_records(mine, world)
got = bp.msm_finish_host(curve, bytes(allrec.tolist()), world, nmax, c)
want = O.msm(curve, pts, ss, n, algo=O.PIPPENGER)
assert got == want, "sharded sum differs"
# a rank that picked another width is refused, not mis-folded
bad = bytearray(bytes(allrec.tolist())); bad[len(block) - 192 + 4] ^= 1
try:
    bp.msm_finish_host(curve, bytes(bad), world, nmax, c); raise SystemExit("geometry mismatch was not detected")
except bp.ArgError:
    pass
# ---- 2-D mode (round 4): ONE index group, the WINDOWS split over the ranks -- every rank all points, half of the windows; blocks carry
# their window group in the header and bp_msm_g1_finish_blocks_host adds what it is given
ig, wg = sharding.plan_2d(world, len(cw))
assert (ig, wg) == (1, world)
lo2, hi2, w0, wn = sharding.shard_2d(n, world, rank, len(cw))
assert (lo2, hi2) == (0, n) and wn == len(cw) // world and w0 == rank * wn
c2, cw2, off2, bias2 = bp.msm_geometry(curve, n, 0)
stride = 1 + max(len(bp.msm_record_positions_subset(curve, n, c2, g * wn, wn)) for g in range(world))
pos2 = bp.msm_record_positions_subset(curve, n, c2, w0, wn)
recs2 = [zero] * (stride - 1)
for w in range(w0, w0 + wn):
    digs = b""
    for i in range(n):
        k = int.from_bytes(ss[32 * i:32 * i + 32], "little") + bias2
        d = ((k >> off2[w]) & ((1 << cw2[w]) - 1)) - ((1 << (cw2[w] - 1)) - 1)
        digs += (d %% r).to_bytes(32, "little")
    S = O.msm(curve, pts, digs, n, algo=O.PIPPENGER)
    slot = next(j for j in range(len(pos2)) if pos2[j] == off2[w] and recs2[j] is zero)
    recs2[slot] = bp.msm_record_from_affine(curve, S)
block2 = b"".join(recs2) + bp.msm_record_header_subset(curve, n, c2, w0, wn)
all2 = sharding.all_gather_records(torch.frombuffer(bytearray(block2), dtype=torch.uint8), world)
assert bp.msm_finish_blocks_host(curve, bytes(all2.tolist()), world, stride, n, c2) == want, "window-sharded sum differs"
bad = bytearray(bytes(all2.tolist())); bad[len(block2) - 192 + 40] ^= 1          # the header's window group
try:
    bp.msm_finish_blocks_host(curve, bytes(bad), world, stride, n, c2); raise SystemExit("a damaged window group was not detected")
except bp.ArgError:
    pass
t = torch.tensor([float(rank + 1)]); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == world
dist.barrier(); dist.destroy_process_group()
sys.stdout.write("rank-%%d-ok\n" %% rank); sys.stdout.flush()
'''


def test_shard_range_tiles():
    bp = G.load_package()
    from bulletproofs_amcl_amd import sharding
    for n in (0, 1, 7, 8, 1001, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def test_two_rank_gloo_sharded_msm(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rank-0-ok" in p.stdout and "rank-1-ok" in p.stdout
