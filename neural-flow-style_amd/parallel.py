"""Multi-GPU plumbing of the one exchange step the path has: views (or
frames) shard over ranks, every rank holds a full replica of the field, and ONE all-reduce(sum) of
the field gradient per iteration makes the replicas take the identical Adam step.

The functions are backend-agnostic (`nccl` = RCCL over xGMI on the GPU box, `gloo` in the CPU
tests): they only touch `torch.distributed`, never the HIP kernels.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def rank_world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard(items, rank=None, world=None, group=None):
    """round-robin shard of a list of work units (views / frames): unit i goes to rank i % world.
    With world | len(items) every rank gets the same count (the bench asserts it)."""
    if rank is None or world is None:
        rank, world = rank_world(group)
    return list(items[rank::world])


def all_reduce_sum_(tensors, group=None):
    """in-place sum over ranks of a list of tensors (field gradient, loss); no-op for world 1"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors


def replicas_identical(t, group=None, atol=0.0):
    """debug check: max |t - t_rank0| over ranks is <= atol"""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    ref = t.clone()
    dist.broadcast(ref, src=0, group=group)
    diff = (t - ref).abs().max()
    dist.all_reduce(diff, op=dist.ReduceOp.MAX, group=group)
    return float(diff) <= atol
