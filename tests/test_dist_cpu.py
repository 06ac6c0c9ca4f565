"""World-size-2 gloo tests of the process-per-GPU path (bito_amd/dist.py: sharding by trees + gather / reduce of
results).  The per-rank evaluator here is the CPU oracle behind the engine's two entry points; on the GPU box the same
ShardedEngine wraps bito_amd.Engine with backend nccl = RCCL (tests/test_dist_gpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bito_amd import dist as bdist
from bito_amd import workloads
from oracle import oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleAsEngine:
    """the CPU checker behind the two calls ShardedEngine needs of an engine (a rank's evaluator in these CPU tests;
    on the GPU box it is bito_amd.Engine: tests/test_dist_gpu.py)"""

    def __init__(self, w):
        self.eng = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 2)
        self.taxon_count = w.taxon_count
        self.calls = []

    def gradients_into(self, parent_ids, branch_lengths, params, out_ll, out_branch, rescaling=False):
        self.calls.append(parent_ids.shape[0])
        out = self.eng.gradients(parent_ids, branch_lengths, params, rescaling=rescaling)
        out_ll[:] = out["log_likelihood"]
        out_branch[:] = out["branch_lengths"]

    def log_likelihoods_into(self, parent_ids, branch_lengths, params, out, rescaling=False):
        self.calls.append(parent_ids.shape[0])
        out[:] = self.eng.log_likelihoods(parent_ids, branch_lengths, params, rescaling=rescaling)


def _sharded_engine_worker(rank, world, port, tree_count, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = workloads.ds1_gtr_weibull4(1).subset(tree_count)
    engine = _OracleAsEngine(w)
    sharded = bdist.ShardedEngine(engine)
    res = sharded.gradients(w.parent_ids, w.branch_lengths, w.params)
    ll_only = sharded.log_likelihoods(w.parent_ids, w.branch_lengths * 2.0, w.params)
    lo, hi = bdist.shard_bounds(tree_count, rank, world)
    assert engine.calls == [hi - lo, hi - lo]  # every rank evaluated its own block only
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ll2=ll_only["log_likelihood"], sum2=ll_only["sum_log_likelihood"],
             **{k: np.asarray(v) for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tree_count", [7, 2])  # uneven shards (3 + 4 trees); one tree per rank
def test_sharded_engine_two_ranks(tmp_path, tree_count):
    world = 2
    mp.spawn(_sharded_engine_worker, args=(world, _free_port(), tree_count, str(tmp_path)), nprocs=world, join=True)
    w = workloads.ds1_gtr_weibull4(1).subset(tree_count)
    eng = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 2)
    ref = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    ref2 = eng.log_likelihoods(w.parent_ids, w.branch_lengths * 2.0, w.params)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.array_equal(got["log_likelihood"], ref["log_likelihood"])
        assert np.array_equal(got["branch_lengths"], ref["branch_lengths"])
        assert abs(float(got["sum_log_likelihood"]) - ref["log_likelihood"].sum()) < 1e-9
        assert np.array_equal(got["ll2"], ref2) and abs(float(got["sum2"]) - ref2.sum()) < 1e-9


def test_sharded_engine_without_a_group():
    w = workloads.ds1_gtr_weibull4(1).subset(3)
    res = bdist.ShardedEngine(_OracleAsEngine(w)).gradients(w.parent_ids, w.branch_lengths, w.params)
    ref = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 2).gradients(
        w.parent_ids, w.branch_lengths, w.params)
    assert np.array_equal(res["log_likelihood"], ref["log_likelihood"])
    assert np.array_equal(res["branch_lengths"], ref["branch_lengths"])
    assert abs(res["sum_log_likelihood"] - ref["log_likelihood"].sum()) < 1e-9


def _worker(rank, world, port, tree_count, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = workloads.ds1_gtr_weibull4(1).subset(tree_count)
    eng = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 2)

    def evaluate(lo, hi):
        return eng.gradients(w.parent_ids[lo:hi], w.branch_lengths[lo:hi], w.params[lo:hi])

    res = bdist.sharded_evaluate(evaluate, tree_count, 2 * w.taxon_count - 1)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{k: np.asarray(v) for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tree_count", [7])  # uneven shards: 3 + 4 trees
def test_two_ranks_match_single_process(tmp_path, tree_count):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), tree_count, str(tmp_path)), nprocs=world, join=True)
    w = workloads.ds1_gtr_weibull4(1).subset(tree_count)
    eng = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 2)
    ref = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.array_equal(got["log_likelihood"], ref["log_likelihood"])
        assert np.array_equal(got["branch_lengths"], ref["branch_lengths"])
        assert abs(float(got["sum_log_likelihood"]) - ref["log_likelihood"].sum()) < 1e-9


def test_shard_bounds_cover_everything():
    for T in (1, 7, 100, 1600):
        for world in (1, 2, 3, 8):
            spans = [bdist.shard_bounds(T, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == T
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            w = workloads.ds1_gtr_weibull4(1) if T == 100 else None
            if w is not None:
                assert [s.tree_count for s in (w.shard(r, world) for r in range(world))] == [b - a for a, b in spans]


def test_single_process_passthrough():
    res = bdist.sharded_evaluate(lambda lo, hi: {"log_likelihood": np.arange(lo, hi, dtype=float),
                                                  "branch_lengths": np.ones((hi - lo, 5))}, 6, 5)
    assert res["sum_log_likelihood"] == 15.0 and res["branch_lengths"].shape == (6, 5)


def test_host_threads_are_shared_among_the_ranks_of_a_node(monkeypatch):
    """An engine's helper threads (bito_amd_engine_spec.host_threads) divide the CPUs the process may use by the
    ranks of the node, as torch.distributed.run exports them."""
    import os

    from bito_amd.dist import host_threads_for_rank

    cpus = len(os.sched_getaffinity(0))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    alone = host_threads_for_rank()
    assert 1 <= alone <= min(8, cpus)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", str(4 * cpus))
    assert host_threads_for_rank() == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    assert host_threads_for_rank(cap=3) <= 3
