"""Multi-GPU evaluation of one tree collection with one process per GPU: trees sharded by rank, results gathered.

The reference's only parallel axis is "independent trees, private result slot per tree"
(`FatBeagleParallelize`, reference src/fat_beagle.hpp:151-184).  Inside one process that axis is the engine's own
(`Engine(devices=[...])`, include/bito_amd.h: one engine over several GPUs).  This module is the other arrangement --
one process per GPU, as `bench.py --gpus N` runs under `torch.distributed.run`: each rank owns a contiguous block of
trees; the compressed alignment is replicated; the only exchange is handing results back -- an all-gather of the
per-tree log-likelihoods / gradients and an all-reduce of the summed log-likelihood (torch.distributed: backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

  * ``ShardedEngine``      blocking calls: host arrays in, the whole collection's results out on every rank.
  * ``ResidentSumReducer`` a batch that stays in HBM: per pass one asynchronous all-reduce of the summed
                           log-likelihood, taken from where the engine left the values (no copy, no host wait).
  * ``sharded_evaluate``   the same gather around any per-block evaluator (kept for callers that bring their own).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np


def shard_bounds(tree_count: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of trees of one rank (same split as Workload.shard and as the engine's device slots)."""
    return (tree_count * rank) // world, (tree_count * (rank + 1)) // world


def _pad_rows(a: np.ndarray, rows: int) -> np.ndarray:
    if a.shape[0] == rows:
        return a
    out = np.zeros((rows,) + a.shape[1:], dtype=a.dtype)
    out[: a.shape[0]] = a
    return out


def _group_info():
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return None, 0, 1
    return dist, dist.get_rank(), dist.get_world_size()


def host_threads_for_rank(cap: int = 8) -> int:
    """``host_threads`` for this rank's ``bito_amd.Engine``: the CPUs this process may use (affinity mask, cgroup
    quota) shared among the ranks of the node (``LOCAL_WORLD_SIZE``, as torch.distributed.run exports it), at most
    ``cap``.  An engine's own default -- min(8, usable CPUs) -- assumes it has the node to itself."""
    import os

    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
            if quota != "max":
                cpus = min(cpus, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    # (a rank's calling thread polls for its results: it keeps a CPU of the rank's share to itself when the share is small)
    share = cpus // local_world
    return max(1, min(cap, share if share > cap else share - 1))


class ShardedEngine:
    """A rank's engine behind the whole collection's interface.

    ``engine`` needs ``gradients_into(parent_ids, branch_lengths, params, out_ll, out_branch, rescaling=...)`` and
    ``log_likelihoods_into(parent_ids, branch_lengths, params, out, rescaling=...)`` over C-contiguous arrays --
    ``bito_amd.Engine`` has them.  Every rank passes the SAME whole-collection arrays; each evaluates its own block
    straight into its rows of the gather buffers (no intermediate copies), then one all-gather per result array and
    one all-reduce of the summed log-likelihood."""

    def __init__(self, engine, device: Optional[str] = None):
        self.engine = engine
        self.device = device
        self._buffers: Dict[Tuple[int, int], tuple] = {}

    def _buffer(self, rows: int, width: int):
        """(numpy view, torch tensor) of ``rows x width`` doubles, pinned when the group runs on GPUs"""
        import torch

        key = (rows, width)
        if key not in self._buffers:
            pin = self._device() != "cpu"
            t = torch.zeros((rows, width) if width else (rows,), dtype=torch.float64, pin_memory=pin)
            self._buffers[key] = (t.numpy(), t)
        return self._buffers[key]

    def _device(self) -> str:
        dist, _, _ = _group_info()
        if self.device:
            return self.device
        return "cuda" if dist is not None and dist.get_backend() == "nccl" else "cpu"

    def evaluate(self, parent_ids: np.ndarray, branch_lengths: np.ndarray, params: np.ndarray,
                 want_gradient: bool = True, rescaling: bool = False) -> Dict[str, np.ndarray]:
        """-> {"log_likelihood": [T], "branch_lengths": [T][2n-1] (gradients only), "sum_log_likelihood": float},
        the same on every rank."""
        import torch

        dist, rank, world = _group_info()
        T = int(parent_ids.shape[0])
        N = 2 * int(self.engine.taxon_count) - 1
        lo, hi = shard_bounds(T, rank, world)
        rows = max(shard_bounds(T, r, world)[1] - shard_bounds(T, r, world)[0] for r in range(world))
        ll_np, ll_t = self._buffer(rows, 0)
        gr_np, gr_t = self._buffer(rows, N) if want_gradient else (None, None)
        if hi > lo:
            pid = np.ascontiguousarray(parent_ids[lo:hi], dtype=np.int32)
            bl = np.ascontiguousarray(branch_lengths[lo:hi], dtype=np.float64)
            par = np.ascontiguousarray(params[lo:hi], dtype=np.float64)
            if want_gradient:
                self.engine.gradients_into(pid, bl, par, ll_np[: hi - lo], gr_np[: hi - lo], rescaling=rescaling)
            else:
                self.engine.log_likelihoods_into(pid, bl, par, ll_np[: hi - lo], rescaling=rescaling)
        ll_np[hi - lo:] = 0.0
        if dist is None:
            out = {"log_likelihood": ll_np[:T].copy(), "sum_log_likelihood": float(ll_np[:T].sum())}
            if want_gradient:
                out["branch_lengths"] = gr_np[:T].copy()
            return out
        dev = self._device()
        keep = np.concatenate([np.arange(r * rows, r * rows + (shard_bounds(T, r, world)[1] - shard_bounds(T, r, world)[0]))
                               for r in range(world)])
        ll_dev = ll_t.to(dev, non_blocking=True)
        ll_all = torch.empty(world * rows, dtype=torch.float64, device=dev)
        work = [dist.all_gather_into_tensor(ll_all, ll_dev, async_op=True)]
        total = ll_dev.sum().reshape(1)
        work.append(dist.all_reduce(total, async_op=True))
        gr_all = None
        if want_gradient:
            gr_np[hi - lo:] = 0.0
            gr_dev = gr_t.to(dev, non_blocking=True)
            gr_all = torch.empty(world * rows, N, dtype=torch.float64, device=dev)
            work.append(dist.all_gather_into_tensor(gr_all, gr_dev, async_op=True))
        for w in work:
            w.wait()
        out = {"log_likelihood": ll_all.cpu().numpy()[keep], "sum_log_likelihood": float(total.item())}
        if want_gradient:
            out["branch_lengths"] = gr_all.cpu().numpy()[keep]
        return out

    def gradients(self, parent_ids, branch_lengths, params, rescaling: bool = False):
        return self.evaluate(parent_ids, branch_lengths, params, True, rescaling)

    def log_likelihoods(self, parent_ids, branch_lengths, params, rescaling: bool = False):
        return self.evaluate(parent_ids, branch_lengths, params, False, rescaling)


class _DeviceVector:
    """zero-copy view of `count` doubles at a device address, for torch.as_tensor"""

    def __init__(self, address: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (address, False), "version": 2}


class ResidentSumReducer:
    """Passes over a batch that is resident in HBM (``engine.upload`` / ``update`` / ``run``) with the summed
    log-likelihood of the whole collection reduced over the ranks after every pass -- RCCL, 8 bytes.

    Nothing waits on the host and nothing is added to the engine's stream: torch's current stream waits for the pass
    through the event the engine records behind it anyway (``bito_amd_engine_results_async``), sums the per-tree
    values where the engine left them -- a ring of four buffers, so the next passes do not touch them -- and hands the
    scalar to the collective; the next pass and its set-up are submitted meanwhile.  The engine's stream waits for the
    sum that last read a ring slot before the pass that rewrites it, four passes later.  ``finish()`` waits for
    everything and returns the reduced sums in pass order.  Works without a process group (the sums are then local)."""

    RING = 4

    def __init__(self, engine):
        import torch

        self.engine = engine
        self.torch = torch
        self.dist, self.rank, self.world = _group_info()
        self.stream = torch.cuda.ExternalStream(engine.stream_handle())
        self.sum_done: List[Optional["torch.cuda.Event"]] = [None] * self.RING
        self.pending = []
        self.passes = 0

    def run(self, want_gradient: bool, rescaling: bool = False):
        torch = self.torch
        slot = self.passes % self.RING
        self.passes += 1
        if self.sum_done[slot] is not None:
            self.stream.wait_event(self.sum_done[slot])
        self.engine.run(want_gradient, rescaling)
        here = torch.cuda.current_stream()
        ll_address, _ = self.engine.results_async(here.cuda_stream)
        values = torch.as_tensor(_DeviceVector(ll_address, self.engine.tree_count), device="cuda")
        total = values.sum().reshape(1)
        self.sum_done[slot] = torch.cuda.Event()
        self.sum_done[slot].record(here)
        work = self.dist.all_reduce(total, async_op=True) if self.dist is not None else None
        self.pending.append((work, total))

    def finish(self) -> List[float]:
        self.engine.sync()
        for work, _ in self.pending:
            if work is not None:
                work.wait()
        self.torch.cuda.synchronize()
        sums = [float(total.item()) for _, total in self.pending]
        self.pending.clear()
        return sums


def sharded_evaluate(evaluate: Callable[[int, int], Dict[str, np.ndarray]], tree_count: int, node_count: int,
                     want_gradient: bool = True, device: Optional[str] = None) -> Dict[str, np.ndarray]:
    """Run ``evaluate(lo, hi)`` on this rank's block of trees and assemble the whole
    collection's results on every rank.

    ``evaluate`` returns {"log_likelihood": [hi-lo], "branch_lengths": [hi-lo][node_count]}
    for trees lo..hi-1 (e.g. a closure over ``bito_amd.Engine.gradients``).  Returns the
    same keys for all ``tree_count`` trees plus "sum_log_likelihood" (all-reduced).
    """
    import torch

    dist, rank, world = _group_info()
    if dist is None:
        out = evaluate(0, tree_count)
        out = dict(out)
        out["sum_log_likelihood"] = float(np.sum(out["log_likelihood"]))
        return out
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
