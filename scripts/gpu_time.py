"""Times the resident config-3 batch: LL-only and LL+gradient, forced kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bito_amd
from bito_amd import workloads
w = workloads.ds1_gtr_weibull4(16)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
eng.upload(w.parent_ids, w.branch_lengths, w.params)
for kern in (3, 2, 1):
    eng.set_kernel(kern)
    for grad in (False, True):
        eng.time_runs(grad, False, 3)
        total, k, launches = eng.time_runs(grad, False, 10)
        print(f"kernel={eng.kernel_name()} grad={grad}: total {total/10:.3f} ms/step, walk kernel {k/launches:.3f} ms")
