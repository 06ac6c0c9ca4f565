"""Config 5 on the GPU box: parity summary + timing of the general-state kernels (fluA codons, GY94).
usage: python scripts/gpu_codon.py [tree_count] [site]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bito_amd
from bito_amd import workloads
from oracle import gs

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
site = sys.argv[2] if len(sys.argv) > 2 else "constant"
distinct = len(sys.argv) > 3 and sys.argv[3] == "distinct"  # every tree its own kappa / omega: T eigensystems
w = workloads.flua_codon(T, site)
if distinct:
    rng = np.random.default_rng(3)
    w.params[:, 4] = rng.uniform(1.0, 4.0, T)
    w.params[:, 5] = rng.uniform(0.1, 1.0, T)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
k = min(T, 4)
cpu = gs.GsOracleEngine("GY94", site, w.patterns, w.weights, 8)
t0 = time.time()
ref = cpu.gradients(w.parent_ids[:k], w.branch_lengths[:k], w.params[:k])
cpu_s = (time.time() - t0) / k * min(8, k)
out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
print("max|dLL| %.3e  max|dgrad| %.3e  (LL %.6f)" % (np.abs(out["log_likelihood"][:k] - ref["log_likelihood"]).max(),
      np.abs(out["branch_lengths"][:k] - ref["branch_lengths"]).max(), ref["log_likelihood"][0]))
print("cpu oracle: %.3f s per tree per core (port, %d threads)" % (cpu_s, min(8, k)))
eng.upload(w.parent_ids, w.branch_lengths, w.params)
for grad in (True, False):
    eng.time_runs(grad, False, 2)
    total, kern, launches = eng.time_runs(grad, False, 5)
    print("grad=%d  T=%d  %.3f ms/step  walk kernel %.3f ms/launch  -> %.0f trees/s" % (grad, T, total / 5, kern / launches, T / (total / 5e3)))
