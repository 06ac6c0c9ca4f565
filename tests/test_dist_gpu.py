"""The process-per-GPU path on the real engine (bito_amd/dist.py): a one-rank RCCL group on the box's GPU -- the
code every rank of `torch.distributed.run --nproc-per-node N bench.py` runs, with the collectives going through
RCCL.  In a subprocess, so that the process group and torch's HIP context do not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
import bito_amd
from bito_amd import dist as bdist, workloads
from oracle import oracle

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = workloads.ds1_gtr_weibull4(2)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
# blocking calls, results gathered over the group
sharded = bdist.ShardedEngine(eng)
out = sharded.gradients(w.parent_ids, w.branch_lengths, w.params)
assert np.abs(out["log_likelihood"] - ref["log_likelihood"]).max() < 1e-10
assert np.abs(out["branch_lengths"] - ref["branch_lengths"]).max() < 1e-6
assert abs(out["sum_log_likelihood"] - ref["log_likelihood"].sum()) < 1e-8
ll = sharded.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
assert np.abs(ll["log_likelihood"] - ref["log_likelihood"]).max() < 1e-10 and "branch_lengths" not in ll
# a resident batch, the summed log-likelihood reduced after every pass without a host wait
eng.upload(w.parent_ids, w.branch_lengths, w.params)
reducer = bdist.ResidentSumReducer(eng)
scales = [1.0, 0.9, 1.1, 1.2, 0.8, 1.05, 0.95]   # more passes than the ring has slots
for s in scales:
    eng.update(w.branch_lengths * s)
    reducer.run(True)
sums = reducer.finish()
want = [cpu.log_likelihoods(w.parent_ids, w.branch_lengths * s, w.params).sum() for s in scales]
assert len(sums) == len(scales) and max(abs(a - b) for a, b in zip(sums, want)) < 1e-7, (sums, want)
ll_last, grad_last = eng.download()
assert np.abs(ll_last - cpu.log_likelihoods(w.parent_ids, w.branch_lengths * scales[-1], w.params)).max() < 1e-10
dist.barrier()
dist.destroy_process_group()
print("dist-gpu ok", eng.kernel_name())
"""


def test_sharded_engine_and_resident_reducer_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-c", _SCRIPT, ROOT], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          timeout=600, env=env)
    assert proc.returncode == 0 and "dist-gpu ok" in proc.stdout, proc.stdout[-3000:]
