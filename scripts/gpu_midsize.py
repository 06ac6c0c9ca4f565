"""Trees of 30 to 38 taxa (DS3-sized): walk_pipe_kernel (two pattern groups per wave there) against walk_lds_kernel.
usage: python scripts/gpu_midsize.py [taxa ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import _capi, workloads

for n in [int(a) for a in sys.argv[1:]] or [27, 31, 36, 38]:
    w = workloads.synthetic_gtr_weibull4(n=n, P=1000, tree_count=1600)
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
    eng.upload(w.parent_ids, w.branch_lengths, w.params)
    results = {}
    for kern in (_capi.KERNEL_LDS_PIPE, _capi.KERNEL_LDS):
        eng.set_kernel(kern)
        eng.run(True, False)
        eng.sync()
        results[kern] = eng.download(True)
        eng.time_runs(True, False, 3)
        total, k, launches = eng.time_runs(True, False, 10)
        print(f"n={n} kernel={eng.kernel_name()}: step {total / 10:.3f} ms per 1600 trees ({1600 / (total / 10):.0f} k trees/s), walk kernel {k / launches:.3f} ms")
    a, b = results[_capi.KERNEL_LDS_PIPE], results[_capi.KERNEL_LDS]
    print(f"   max |dLL| between the two {np.abs(a[0] - b[0]).max():.2e}, max |dgrad| {np.abs(a[1] - b[1]).max():.2e}")
