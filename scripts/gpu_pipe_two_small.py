"""One wave per SIMD x 4 groups against two waves per SIMD x 2 groups on SMALL trees, where the two-wave form has registers
to spare (build: scripts/build_pipe_variants.sh two16 "PIPE_TWO_TIPS=16 PIPE_TWO_VBASE=64" "" -- 60 image registers, the
compiler keeps v0..v63, nothing spills): does the form win once the code around the loops is no longer starved?
usage: BITO_AMD_LIB=bito_amd/variants/two16.so BITO_AMD_PIPE_MIN_BRANCH=0 python scripts/gpu_pipe_two_small.py [taxa=16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import _capi, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = workloads.synthetic_gtr_weibull4(n=n, P=934, tree_count=6400)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
res = {}
for kern, form in ((_capi.KERNEL_LDS_PIPE, "one wave/SIMD"), (_capi.KERNEL_LDS_PIPE2, "two waves/SIMD")):
    eng.set_kernel(kern)
    eng.upload(w.parent_ids, w.branch_lengths, w.params)
    eng.run(True)
    eng.sync()
    res[kern] = eng.download(True)
    eng.time_runs(True, False, 3)
    total, k, launches = eng.time_runs(True, False, 10)
    print(f"n={n} {form:<15} walk {k / 10:7.3f} ms per 6400 trees  [{eng.kernel_form()}]", flush=True)
a, b = res[_capi.KERNEL_LDS_PIPE], res[_capi.KERNEL_LDS_PIPE2]
print(f"   the two forms agree to max |dLL| {np.abs(a[0] - b[0]).max():.2e}, max |dgrad| {np.abs(a[1] - b[1]).max():.2e}")
