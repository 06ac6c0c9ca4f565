"""Multi-GPU evaluation of one tree collection: trees sharded by rank, results gathered.

The reference's only parallel axis is "independent trees, private result slot per tree"
(`FatBeagleParallelize`, reference src/fat_beagle.hpp:151-184).  Here each rank (one
process per GPU) owns a contiguous block of trees; the compressed alignment is
replicated; the only exchange is handing results back: an all-gather of the per-tree
log-likelihoods / gradients and an all-reduce of the summed log-likelihood
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np


def shard_bounds(tree_count: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of trees of one rank (same split as Workload.shard)."""
    return (tree_count * rank) // world, (tree_count * (rank + 1)) // world


def _pad_rows(a: np.ndarray, rows: int) -> np.ndarray:
    if a.shape[0] == rows:
        return a
    out = np.zeros((rows,) + a.shape[1:], dtype=a.dtype)
    out[: a.shape[0]] = a
    return out


def sharded_evaluate(evaluate: Callable[[int, int], Dict[str, np.ndarray]], tree_count: int, node_count: int,
                     want_gradient: bool = True, device: Optional[str] = None) -> Dict[str, np.ndarray]:
    """Run ``evaluate(lo, hi)`` on this rank's block of trees and assemble the whole
    collection's results on every rank.

    ``evaluate`` returns {"log_likelihood": [hi-lo], "branch_lengths": [hi-lo][node_count]}
    for trees lo..hi-1 (e.g. a closure over ``bito_amd.Engine.gradients``).  Returns the
    same keys for all ``tree_count`` trees plus "sum_log_likelihood" (all-reduced).
    """
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        out = evaluate(0, tree_count)
        out = dict(out)
        out["sum_log_likelihood"] = float(np.sum(out["log_likelihood"]))
        return out
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(tree_count, rank, world)
    local = evaluate(lo, hi) if hi > lo else {"log_likelihood": np.zeros(0), "branch_lengths": np.zeros((0, node_count))}
    rows = max(shard_bounds(tree_count, r, world)[1] - shard_bounds(tree_count, r, world)[0] for r in range(world))
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    ll = torch.from_numpy(_pad_rows(np.ascontiguousarray(local["log_likelihood"], dtype=np.float64), rows)).to(dev)
    ll_all = torch.empty(world * rows, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(ll_all, ll)
    total = ll[: hi - lo].sum().reshape(1).clone()
    dist.all_reduce(total)
    out: Dict[str, np.ndarray] = {}
    keep = np.concatenate([np.arange(r * rows, r * rows + (shard_bounds(tree_count, r, world)[1] -
                                                           shard_bounds(tree_count, r, world)[0]))
                           for r in range(world)])
    out["log_likelihood"] = ll_all.cpu().numpy()[keep]
    out["sum_log_likelihood"] = float(total.item())
    if want_gradient:
        gr = torch.from_numpy(_pad_rows(np.ascontiguousarray(local["branch_lengths"], dtype=np.float64), rows)).to(dev)
        gr_all = torch.empty(world * rows, node_count, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(gr_all, gr)
        out["branch_lengths"] = gr_all.cpu().numpy()[keep]
    return out
