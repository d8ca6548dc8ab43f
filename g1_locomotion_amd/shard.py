"""Multi-GPU sharding of independent QP batches.

QPs are fully independent, so the batch is cut into contiguous per-rank ranges with no data-path collective; the
only exchange is the all-gather of the first-step contact forces u_opt0 (12 scalars per QP -- what the consumer of
the MPC reads, g1_mujoco_sim/src/ros_run_simulation.py:214-215), over RCCL/xGMI on GPUs (backend "nccl") or gloo
in the CPU tests.  torch.distributed is plumbing here; no compute happens in this module.
"""
from __future__ import annotations


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous range [lo, hi) of QPs owned by `rank`; the first (total % world) ranks get one extra."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_u0(u_local, group=None):
    """All-gather of u_opt0.  u_local: (B_local, 12) tensor, same B_local on every rank (weak scaling).
    Returns (world * B_local, 12), rank-major.  One collective, no staging copies beyond making the slice contiguous."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    u_local = u_local.contiguous()
    out = torch.empty((world * u_local.shape[0],) + tuple(u_local.shape[1:]), dtype=u_local.dtype, device=u_local.device)
    dist.all_gather_into_tensor(out, u_local, group=group)
    return out
