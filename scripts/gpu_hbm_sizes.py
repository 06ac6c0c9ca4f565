"""HBM-arena walk on trees beyond walk_pipe_kernel's range (1600 trees x 1000 patterns, GTR+weibull4, gradient).
usage: [BITO_AMD_LIB=variant.so] python scripts/gpu_hbm_sizes.py [taxa ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bito_amd
from bito_amd import _capi, workloads

for n in [int(a) for a in sys.argv[1:]] or [70, 100, 128]:
    w = workloads.synthetic_gtr_weibull4(n=n, P=1000, tree_count=1600)
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
    eng.set_kernel(_capi.KERNEL_HBM_ARENA)
    eng.upload(w.parent_ids, w.branch_lengths, w.params)
    eng.time_runs(True, False, 3)
    total, k, launches = eng.time_runs(True, False, 10)
    print(f"n={n} {eng.kernel_name()}: {total / 10:.3f} ms per 1600 trees ({1600 / (total / 10):.0f} k trees/s), walk kernel {k / launches:.3f} ms", flush=True)
