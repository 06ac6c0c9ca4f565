"""Trees of 30 taxa and more (DS3- to DS8-sized): walk_pipe_kernel (two pattern groups per wave from 33 taxa, up to
38 taxa) against walk_lds_kernel and the HBM-arena walk.
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
    for kern in (_capi.KERNEL_LDS_PIPE, _capi.KERNEL_LDS, _capi.KERNEL_HBM_ARENA):
        eng.set_kernel(kern)
        try:
            eng.run(True, False)
        except bito_amd.BitoAmdError as err:
            print(f"n={n} kernel {kern}: {str(err)[:60]}")
            continue
        eng.sync()
        results[kern] = eng.download(True)
        eng.time_runs(True, False, 3)
        total, k, launches = eng.time_runs(True, False, 10)
        print(f"n={n} kernel={eng.kernel_name()}: step {total / 10:.3f} ms per 1600 trees ({1600 / (total / 10):.0f} k trees/s), walk kernel {k / launches:.3f} ms")
    a = results[_capi.KERNEL_HBM_ARENA]
    for kern, name in ((_capi.KERNEL_LDS, "walk_lds_kernel"), (_capi.KERNEL_LDS_PIPE, "walk_pipe_kernel")):
        if kern in results:
            b = results[kern]
            print(f"   HBM-arena walk vs {name}: max |dLL| {np.abs(a[0] - b[0]).max():.2e}, max |dgrad| {np.abs(a[1] - b[1]).max():.2e}")
