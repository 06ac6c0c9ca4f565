"""BASELINE config 2 (DS1, JC69, one rate category, log-likelihood only, 1600 trees resident): every kernel
that takes it, against the CPU checker, with its time per pass.
usage: python scripts/gpu_config2.py [replicas of the 100 topologies per pass, default 16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import _capi, workloads
from oracle import oracle

replicas = int(sys.argv[1]) if len(sys.argv) > 1 else 16  # x100 topologies per pass
w = workloads.ds1_jc69(replicas)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
small = w.subset(10)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
ref = cpu.log_likelihoods(small.parent_ids, small.branch_lengths, small.params)
eng.upload(w.parent_ids, w.branch_lengths, w.params)
for kern in (_capi.KERNEL_LDS_PIPE, _capi.KERNEL_LDS, _capi.KERNEL_HBM_ARENA):
    eng.set_kernel(kern)
    ll = eng.log_likelihoods(small.parent_ids, small.branch_lengths, small.params)
    eng.upload(w.parent_ids, w.branch_lengths, w.params)
    for g in (False, True):
        eng.time_runs(g, False, 3)
        total, k, launches = eng.time_runs(g, False, 20)
        print(f"kernel={eng.kernel_name()} grad={g}: {total/20:.3f} ms per pass of {w.tree_count} trees "
              f"({w.tree_count / (total / 20) / 1e3:.2f} M trees/s), walk kernel {k/launches:.3f} ms; max|dLL| {np.abs(ll - ref).max():.2e}")

# the blocking call (host arrays in, host log-likelihoods out) on the same batch, AUTO
eng.set_kernel(_capi.KERNEL_AUTO)
import time

pid = np.ascontiguousarray(w.parent_ids, dtype=np.int32)
bls = [np.ascontiguousarray(w.branch_lengths), np.ascontiguousarray(w.branch_lengths * 1.03125)]
par = np.ascontiguousarray(w.params)
out = np.zeros(w.tree_count)
for k in range(3):
    eng.log_likelihoods_into(pid, bls[k & 1], par, out)
t0 = time.perf_counter()
for k in range(30):
    eng.log_likelihoods_into(pid, bls[k & 1], par, out)
ms = (time.perf_counter() - t0) / 30 * 1e3
print(f"blocking log_likelihoods call: {ms:.3f} ms per {w.tree_count} trees ({w.tree_count / ms / 1e3:.2f} M trees/s), kernel {eng.kernel_name()}")
