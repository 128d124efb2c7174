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


def connect_handle(gm, rank, world):
    """Joins a BatchedMechanism's device into an RCCL communicator of the LIBRARY (dojo_comm_init): rank 0 creates the
    128-byte id, torch.distributed (whatever backend is up) carries it to the other ranks.  The same three calls are what a
    Julia host makes with Distributed / MPI as the carrier."""
    if world == 1:
        return
    box = [gm.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    gm.comm_init(rank, world, box[0])


def all_gather_states_rccl(gm, local, world):
    """All-gather of equally sized per-rank device tensors through the library's own RCCL communicator (dojo_allgather_dev):
    [B_local, ...] -> [world * B_local, ...] in rank order, on torch's current stream."""
    if world == 1:
        return local
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    gm.allgather_dev(local.data_ptr(), out.data_ptr(), local.numel(), as_int32=(local.dtype == torch.int32),
                     stream=torch.cuda.current_stream().cuda_stream)
    return out


def sharded_step(step_fn, Z, U, rank, world):
    """The N > 1 data path in one place: every rank steps its contiguous slice of the global batch (no exchange inside the
    solver), the per-rank results are all-gathered in rank order.  step_fn(Z_local, U_local) -> tuple of arrays with the batch
    as leading axis; returns the gathered tuple as torch tensors (equal on every rank).  Shards must be equally sized."""
    lo, hi = shard_slice(len(Z), rank, world)
    outs = step_fn(Z[lo:hi], U[lo:hi] if U is not None else None)
    return tuple(all_gather_states(torch.as_tensor(o), world) for o in outs)


def max_over_ranks(value, world, device="cpu"):
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
