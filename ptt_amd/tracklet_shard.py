"""Multi-GPU sharding of the hot path: tracklets / frames are independent units.

One process per GPU. There is NO collective on the data path: each rank runs the hot path on its own
share; only results (small boxes / metrics) are merged through torch.distributed's object gather.
The partition mirrors the reference's evaluation sampler and merge
(ptt/datasets/__init__.py:18-39 — pad by wrapping, then rank::world; ptt/utils/common_utils.py:309-330 —
interleave the parts, cut to the true size) so a maintainer can swap it in without re-ordering results.
"""
import math

import torch
import torch.distributed as dist


def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_items, rank=None, world=None):
    """Indices of the items this rank processes: arange(n) padded by wrap-around to a multiple of
    `world`, then every world-th starting at `rank` (reference DistributedSampler.__iter__, shuffle=False)."""
    if rank is None or world is None:
        rank, world = dist_info()
    if n_items == 0:
        return []
    per_rank = int(math.ceil(n_items / float(world)))
    total = per_rank * world
    idx = list(range(n_items))
    while len(idx) < total:
        idx += idx[:total - len(idx)]
    return idx[rank:total:world]


def merge_results(parts, size):
    """Inverse of shard_indices for per-item results: interleave the per-rank lists and drop the padding
    (reference merge_results_dist, common_utils.py:324-329)."""
    ordered = []
    for group in zip(*parts):
        ordered.extend(group)
    return ordered[:size]


def gather_results(local_results, size):
    """Collect every rank's per-item results on all ranks, in item order."""
    rank, world = dist_info()
    if world == 1:
        return list(local_results)[:size]
    parts = [None] * world
    dist.all_gather_object(parts, list(local_results))
    return merge_results(parts, size)


def max_over_ranks(seconds, device=None):
    """Elapsed-time reduction used by bench.py (control plane only)."""
    rank, world = dist_info()
    if world == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
