"""A 100-tree blocking gradients call on DS1 under three models: which part of the set-up kernel is the eigensystem?
(run under rocprofv3 --kernel-trace --stats for the kernels' own durations)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bito_amd
from bito_amd import workloads

base = workloads.ds1_gtr_weibull4(64)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
only = sys.argv[2] if len(sys.argv) > 2 else None
for sub, site in (("GTR", "weibull+4"), ("JC69", "weibull+4"), ("GTR", "constant")):
    if only and only != sub + "+" + site:
        continue
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(sub, site, "none"), base.patterns, base.weights)
    w = base.subset(T)
    params = eng.default_params(T)
    if sub == "GTR":
        params[:, :10] = base.params[:T, :10]
    for _ in range(20):
        eng.gradients(w.parent_ids, w.branch_lengths, params)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.gradients(w.parent_ids, w.branch_lengths, params)
    dt = (time.perf_counter() - t0) / reps
    print(f"{sub}+{site}: {dt * 1e3:.4f} ms per {T}-tree call, kernel {eng.kernel_name()}", flush=True)
