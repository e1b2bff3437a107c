"""Multi-GPU driver: the batch shards embarrassingly, one process per GPU.

Every rank solves a contiguous slice of the parameter batch on its own MI355X with its own
replica of the (few-kB) constant pack; the only exchange is one gather of the per-shard primal
solutions and exit flags (`torch.distributed` -- backend "nccl" is RCCL over xGMI on ROCm, "gloo"
in the CPU tests).  Nothing of this exists in the reference (single-threaded, one solve at a
time: /root/reference/src/simulation.jl:106); it is the batched counterpart of calling
compute_control once per scenario.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous [lo, hi) slice of rank; the first n_total % world ranks get one extra problem."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(n_total: int, world: int):
    return [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]


def gather_shards(local: torch.Tensor, n_total: int, group=None, dst: int | None = None):
    """Concatenate per-rank result slices (dim 0) in rank order.

    dst=None: every rank receives the full tensor (all-gather); dst=k: only rank k does (gather),
    others get None.  Equal shards go through one all_gather_into_tensor / gather call; ragged
    shards are padded to the largest shard for the exchange and trimmed afterwards.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = shard_counts(n_total, world)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank}: local shard has {local.shape[0]} rows, expected {counts[rank]}")
    if world == 1:
        return local
    cmax = max(counts)
    tail = tuple(local.shape[1:])
    if local.shape[0] < cmax:
        pad = torch.zeros((cmax - local.shape[0],) + tail, dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], 0)
    else:
        send = local.contiguous()
    if dst is None:
        out = torch.empty((world * cmax,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, send, group=group)
        if cmax * world == n_total:
            return out
        return torch.cat([out[r * cmax:r * cmax + counts[r]] for r in range(world)], 0)
    bufs = None
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:counts[r]] for r in range(world)], 0)


def solve_sharded(qp, theta_local: torch.Tensor, n_total: int, group=None, dst: int | None = None):
    """Solve this rank's slice on its GPU and gather X* and exit flags.

    `qp` is a BatchedQP bound to this rank's device; theta_local is the (count_r, nth) CUDA slice.
    Returns (X, exitflag) as gather_shards does."""
    x, ef = qp.solve_device(theta_local)
    return gather_shards(x, n_total, group, dst), gather_shards(ef, n_total, group, dst)
