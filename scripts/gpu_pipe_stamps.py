"""Per-workgroup busy spans of walk_pipe_kernel on config 3 (needs a library built with -DPIPE_STAMPS:
scripts/build_pipe_variants.sh stamps "X=1" "-DPIPE_STAMPS", run with BITO_AMD_LIB=bito_amd/variants/stamps.so)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import _capi, workloads

big = workloads.ds1_gtr_weibull4(16)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(big.substitution, big.site, big.clock), big.patterns, big.weights)
eng.set_kernel(_capi.KERNEL_LDS_PIPE)
eng.upload(big.parent_ids, big.branch_lengths, big.params)
eng.time_runs(True, False, 5)
for _ in range(3):
    total, k, launches = eng.time_runs(True, False, 1)
    out = np.zeros(3 * 1024, dtype=np.int64)
    _capi.lib().bito_amd_debug_pipe_stamps(out.ctypes.data_as(C.c_void_p))
    a = out.reshape(1024, 3)
    a = a[a[:, 2] > 0]
    start = a[:, 0].min()
    end = (a[:, 1] - start) / 100.0
    print(f"kernel {k:.3f} ms; {len(a)} workgroups, last unit done after (us): min {end.min():.1f} median {np.median(end):.1f} "
          f"p90 {np.percentile(end, 90):.1f} p99 {np.percentile(end, 99):.1f} max {end.max():.1f}; mean idle behind it "
          f"{(end.max() - end).mean():.1f}; units per workgroup {np.bincount(a[:, 2])[1:]}")
