"""CPU, world_size 2, gloo: the N > 1 plumbing (index-range shards, one all_gather of per-rank records, finish).
The per-rank device stage is played by the oracle here (there is no GPU); what is under test is the sharding and
the exchange step that bench.py --gpus N uses with RCCL."""
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
ks = O.random_scalars(curve, 5, n); ss = O.random_scalars(curve, 6, n)
pts = O.fixed_base_batch(curve, ks, n, 2)
lo, hi = sharding.shard_range(n, world, rank)
part = O.msm(curve, pts[lo * 96:hi * 96], ss[lo * 32:hi * 32], hi - lo, algo=O.PIPPENGER)
mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
allrec = sharding.all_gather_records(mine, world)
acc = bytes(96)
for r in range(world):
    acc = O.g1_add(curve, acc, bytes(allrec[r * 96:(r + 1) * 96].tolist()))
want = O.msm(curve, pts, ss, n, algo=O.PIPPENGER)
assert acc == want, "sharded sum differs"
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
