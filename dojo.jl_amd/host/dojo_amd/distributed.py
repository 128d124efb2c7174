"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

Environments are independent: the batch is sharded in contiguous slices, there is
no exchange inside the solver, and the only collective is an all-gather of per-rollout-chunk
outputs (final states / status).  The same code runs on CPU tensors with the gloo backend in the
world_size-2 tests.
"""
import os
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract)."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend)
    return rank, world, local


def shard_slice(batch, rank, world):
    """Contiguous slice [lo, hi) of a global batch owned by `rank` (remainder goes to the first ranks)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_states(local, world):
    """All-gather equally sized per-rank tensors [B_local, ...] -> [world * B_local, ...] (rank-major,
    i.e. the original global order for equal shards)."""
    if world == 1:
        return local
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local.contiguous())
    return torch.cat(out, dim=0)


def max_over_ranks(value, world, device="cpu"):
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
