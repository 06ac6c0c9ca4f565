"""World-size-2 gloo test of the multi-rank path (sharding by trees + gather/reduce of
results).  The per-rank evaluator here is the CPU oracle; on the GPU box the same
sharded_evaluate wraps bito_amd.Engine with backend nccl (RCCL)."""
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
