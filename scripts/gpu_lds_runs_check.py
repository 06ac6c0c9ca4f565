"""Every tree of a 1600-tree batch, four passes: walk_lds_kernel (or the kernel given) against the HBM-arena walk.
How the intermittent fault of walk_lds_kernel's tile runs at 55 taxa and more was found and mapped.
usage: [BITO_AMD_LDS_TILE_RUN=k] python scripts/gpu_lds_runs_check.py <taxa> [kernel id, default 2 = walk_lds_kernel]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bito_amd
from bito_amd import _capi, workloads
from test_gpu_parity import engines
n, T = int(sys.argv[1]), 1600
w = workloads.synthetic_gtr_weibull4(n=n, P=1000, tree_count=T)
gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
gpu.set_kernel(_capi.KERNEL_HBM_ARENA)
good = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
again = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
print("HBM-arena walk, two passes bitwise equal:", np.array_equal(good["branch_lengths"], again["branch_lengths"]) and np.array_equal(good["log_likelihood"], again["log_likelihood"]))
gpu.set_kernel(int(sys.argv[2]) if len(sys.argv) > 2 else _capi.KERNEL_LDS)
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    o = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    dg = np.abs(o["branch_lengths"] - good["branch_lengths"])
    bad = np.unique(np.where(dg > 1e-6)[0])
    if len(bad) or rep < 2: print("n", n, gpu.kernel_name(), "run", os.environ.get("BITO_AMD_LDS_TILE_RUN"), "rep", rep, "max|dgrad|", dg.max(), "bad trees", bad[:10], "dLL", np.abs(o["log_likelihood"] - good["log_likelihood"]).max())
