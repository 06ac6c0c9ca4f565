import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bito_amd
from bito_amd import workloads
T = int(sys.argv[1])
w = workloads.ds1_gtr_weibull4(-(-T // 100)).subset(T)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
pid = np.ascontiguousarray(w.parent_ids, dtype=np.int32); par = np.ascontiguousarray(w.params)
bls = [np.ascontiguousarray(w.branch_lengths), np.ascontiguousarray(w.branch_lengths * 1.03125)]
ll, grad = np.zeros(T), np.zeros((T, 2 * w.taxon_count - 1))
for k in range(12):
    if k == 10: sys.stderr.write("---- call %d\n" % k)
    t0 = time.perf_counter()
    eng.gradients_into(pid, bls[k & 1], par, ll, grad)
    if k >= 10: sys.stderr.write("call %d: %.3f ms\n" % (k, (time.perf_counter() - t0) * 1e3))
