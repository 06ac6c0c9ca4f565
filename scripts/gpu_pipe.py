"""walk_pipe_kernel against the CPU checker on config-3 trees, then its timing next to walk_lds_kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import _capi, workloads
from oracle import oracle

grad = len(sys.argv) < 2 or sys.argv[1] != "ll"
w = workloads.ds1_gtr_weibull4(1).subset(12)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
eng.set_kernel(_capi.KERNEL_LDS_PIPE)
if grad:
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    dg = np.abs(out["branch_lengths"] - ref["branch_lengths"])
    print("kernel", eng.kernel_name(), "max|dLL|", np.abs(out["log_likelihood"] - ref["log_likelihood"]).max(),
          "max|dgrad|", dg.max(), "at", np.unravel_index(dg.argmax(), dg.shape))
    if dg.max() > 1e-6:
        t = int(np.unravel_index(dg.argmax(), dg.shape)[0])
        print("gpu ", out["branch_lengths"][t])
        print("cpu ", ref["branch_lengths"][t])
        print("parents", w.parent_ids[t])
else:
    ll = eng.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
    print("kernel", eng.kernel_name(), "max|dLL|", np.abs(ll - ref["log_likelihood"]).max())
    print(ll[:4], ref["log_likelihood"][:4])
big = workloads.ds1_gtr_weibull4(16)
eng.upload(big.parent_ids, big.branch_lengths, big.params)
for kern in (_capi.KERNEL_LDS_PIPE, _capi.KERNEL_LDS):
    eng.set_kernel(kern)
    for g in ((False, True) if grad else (False,)):
        eng.time_runs(g, False, 3)
        total, k, launches = eng.time_runs(g, False, 10)
        print(f"kernel={eng.kernel_name()} grad={g}: total {total/10:.3f} ms/step, walk kernel {k/launches:.3f} ms")
