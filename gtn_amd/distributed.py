"""Utterance sharding across ranks (one process per GPU).

A batch of independent utterance graphs shards embarrassingly: rank r owns a
contiguous block of utterances, runs the whole loss locally, and the only
exchange is an all_gather of the per-utterance scalar losses (RCCL over xGMI on
GPUs; gloo in the CPU tests).  No gradient collective is needed for CTC: the
emission gradients stay on the rank that owns the utterances.  The ASG variant has one
real exchange: the transitions graph is shared by every utterance, so its gradient is
a sum over the WHOLE batch -- `all_reduce_shared_grad` (C*C + C floats).
"""
import os

import torch
import torch.distributed as dist


def _single_rank_skip():
    """With ONE rank the collectives have nothing to exchange and are skipped -- unless
    GTN_AMD_FORCE_COLLECTIVES=1 asks for them anyway (tests/test_distributed_gpu.py, bench.py with
    GTN_BENCH_FORCE_DIST=1: the RCCL calls of the multi-GPU path, executed on the one GPU a test box has)."""
    return dist.get_world_size() == 1 and os.environ.get("GTN_AMD_FORCE_COLLECTIVES") != "1"


def shard_range(n_items, rank, world):
    """contiguous block [lo, hi) of rank `rank`; blocks differ by at most one item"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_losses(local_losses, n_items=None):
    """all_gather of the ranks' loss vectors -> one tensor in utterance order.
    Ragged shards (n_items not divisible by world) are padded to the longest."""
    if not (dist.is_available() and dist.is_initialized()) or _single_rank_skip():
        return local_losses
    world = dist.get_world_size()
    if n_items is None:
        n_items = local_losses.numel() * world
    longest = -(-n_items // world)
    pad = local_losses.new_zeros(longest)
    pad[: local_losses.numel()] = local_losses
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out)


def all_reduce_shared_grad(grad):
    """Sum over ranks of the gradient of a graph every utterance shares (ASG transitions:
    criterion_test.cpp:289-305 accumulates it over the batch; `asg_loss` / `gtn_asg_loss_n` return
    the rank's partial sum as a tensor).  In place; returns `grad`.  1.05 MB at C = 512: ring
    all-reduce over xGMI is ~12 us of wire time, so it is issued once per step, not bucketed."""
    if dist.is_available() and dist.is_initialized() and not _single_rank_skip():
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
    return grad


def max_over_ranks(seconds, device=None):
    """the step time the contract reports: slowest rank"""
    if not (dist.is_available() and dist.is_initialized()) or _single_rank_skip():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
