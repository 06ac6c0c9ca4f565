"""BASELINE config 4 shape (1000 taxa x 10000 patterns, GTR+weibull4, rescaling on): timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bito_amd
from bito_amd import workloads
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = workloads.synthetic_gtr_weibull4(n=1000, P=10000, tree_count=T)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
eng.upload(w.parent_ids, w.branch_lengths, w.params)
for grad in (False, True):
    eng.time_runs(grad, True, 1)
    total, k, launches = eng.time_runs(grad, True, 3)
    print(f"config4 T={T} kernel={eng.kernel_name()} grad={grad}: {total/3:.2f} ms/step = {T/(total/3)*1e3:.1f} trees/s; walk kernel {k/3:.2f} ms/step in {launches//3} launches")
ll, g = eng.download()
print("finite:", np.isfinite(ll).all(), np.isfinite(g).all(), ll[:2])
w2 = workloads.ds1_jc69(16)
eng2 = bito_amd.Engine(bito_amd.PhyloModelSpecification(w2.substitution, w2.site, w2.clock), w2.patterns, w2.weights)
eng2.upload(w2.parent_ids, w2.branch_lengths, w2.params)
eng2.time_runs(False, False, 3)
total, k, launches = eng2.time_runs(False, False, 10)
print(f"config2 T=1600 kernel={eng2.kernel_name()} LL only: {total/10:.3f} ms/step = {1600/(total/10)*1e3:.0f} trees/s; walk {k/launches:.3f} ms")
