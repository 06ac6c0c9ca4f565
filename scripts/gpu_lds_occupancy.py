"""Experiment: the LDS walk kernel on an 18-taxon synthetic alignment (16 stored PLVs per group, what
cherry inlining would leave of DS1's 25), to compare one wave per SIMD with G groups against two
waves per SIMD with G/2 groups at the same LDS footprint.  usage: BITO_AMD_LIB=... python scripts/gpu_lds_occupancy.py [taxa]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bito_amd
from bito_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
P, T = 934, 1600
patterns = workloads.simulate_patterns(n, P, seed=1)
trees = [workloads.random_unrooted_tree(n, np.random.default_rng(2 + i), 0.1) for i in range(T)]
pid = np.stack([t.parent_ids for t in trees]).astype(np.int32)
bl = np.stack([t.branch_lengths for t in trees]); bl[:, -1] = 0.0
params = workloads.gtr_weibull_params(T)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification("GTR", "weibull+4", "none"), patterns, np.ones(P))
eng.set_kernel(2)
eng.upload(pid, bl, params)
for grad in (True, False):
    eng.time_runs(grad, False, 3)
    total, k, launches = eng.time_runs(grad, False, 20)
    print(f"n={n} grad={grad}: {total/20:.3f} ms/step, walk {k/launches:.3f} ms ({os.environ.get('BITO_AMD_LIB','default')})")
ll, g = eng.download()
print("checksum", float(ll.sum()), float(np.abs(g).sum()))
