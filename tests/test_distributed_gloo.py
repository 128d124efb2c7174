"""CPU tier: the N > 1 path (batch sharding + all-gather of trajectories + max-over-ranks timing)
with the gloo backend, world_size 2 (the production backend is RCCL)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "dojo.jl_amd", "host"))
import torch
from dojo_amd import distributed as D
rank, world, local = D.init_from_env(backend="gloo")
B = 10
lo, hi = D.shard_slice(B, rank, world)
glob = torch.arange(B * 3, dtype=torch.float64).reshape(B, 3)
mine = glob[lo:hi] * 2.0                      # "simulate" the shard
out = D.all_gather_states(mine, world)
assert torch.equal(out, glob * 2.0), (rank, out)
t = D.max_over_ranks(1.0 + rank, world)
assert t == float(world)
if rank == 0:
    print("GLOO_OK", lo, hi)
''' % ROOT


def test_shard_slices_cover_batch():
    sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
    from dojo_amd.distributed import shard_slice
    for B in (1, 7, 8, 4096, 8192):
        for W in (1, 2, 3, 8):
            s = [shard_slice(B, r, W) for r in range(W)]
            assert s[0][0] == 0 and s[-1][1] == B
            assert all(s[i][1] == s[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in s]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_allgather(tmp_path):
    f = tmp_path / "worker.py"
    f.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29513", str(f)], capture_output=True, text=True, timeout=300, env=env)
    assert "GLOO_OK" in r.stdout, r.stdout + r.stderr
