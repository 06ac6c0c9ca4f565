"""walk_pipe_kernel's one-image-per-branch form (39 to 48 taxa) against the CPU checker as the branch lengths shrink:
how far the reversibility identity pi_i P_ij = pi_j P_ji holds for the COMPUTED transition matrices.  Run with
BITO_AMD_PIPE_MIN_BRANCH=0 (the engine otherwise sends batches with branches shorter than 1e-6 to the HBM-arena walk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bito_amd
from bito_amd import _capi, workloads
from test_gpu_parity import engines

rng = np.random.default_rng(41)
n, P, T = 41, 200, 8
patterns = rng.integers(0, 4, (n, P)).astype(np.int32)
weights = np.ones(P)
pid = np.stack([workloads.random_unrooted_tree(n, rng, 0.1).parent_ids for _ in range(T)]).astype(np.int32)
gpu, cpu = engines("GTR", "weibull+4", "none", patterns, weights, 8)
params = gpu.default_params(T)
params[:, :4] = rng.dirichlet([5, 5, 5, 5], T)
params[:, 4:10] = rng.dirichlet([3] * 6, T)
params[:, 10] = 0.5
for short in (1e-2, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-10, 0.0):
    bl = rng.exponential(0.1, (T, 2 * n - 2))
    bl[rng.random(bl.shape) < 0.2] = short  # a fifth of the branches are that short
    bl[:, -1] = 0.0
    ref = cpu.gradients(pid, bl, params)
    row = []
    for kern in (_capi.KERNEL_LDS_PIPE, _capi.KERNEL_HBM_ARENA):
        gpu.set_kernel(kern)
        out = gpu.gradients(pid, bl, params)
        g, r = out["branch_lengths"], ref["branch_lengths"]
        rel = np.max(np.abs(g - r) / (1e-6 + 1e-9 * np.abs(r)))
        row.append(f"{gpu.kernel_name()}: worst gradient error {rel:.3g} tolerances, max |dLL| {np.abs(out['log_likelihood'] - ref['log_likelihood']).max():.2e}")
    print(f"short branches {short:g}: " + "; ".join(row))
