"""Debug: per-phase cycle stamps of walk_lds_kernel blocks (needs the instrumented library)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BITO_AMD_LIB"] = os.path.join(ROOT, "bito_amd", "libbito_amd_dbg.so")
import numpy as np
import bito_amd
from bito_amd import workloads, _capi
w = workloads.ds1_gtr_weibull4(16)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
eng.upload(w.parent_ids, w.branch_lengths, w.params)
for grad in (False, True):
    for _ in range(3):
        eng.run(grad); eng.sync()
    out = np.zeros(8 * 64, dtype=np.uint64)
    _capi.lib().bito_amd_debug_walk_stamps(out.ctypes.data_as(C.c_void_p))
    st = out.reshape(64, 8).astype(np.int64)
    d = np.diff(st[:, :6], axis=1)
    print("grad", grad, "median cycles per phase [preamble, post, root, pre, sums]:", np.median(d, axis=0), "total", np.median(st[:, 5] - st[:, 0]))
