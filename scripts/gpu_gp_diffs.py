"""Path B: how far the GPU executor is from the CPU checker on the cases of tests/test_gp.py (log-likelihoods,
marginals, per-edge derivatives, optimised branch lengths) -- the numbers the tests' tolerances are set from."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_gp as T
from bito_amd import gp, workloads
from oracle import gp as ogp

data_dir = os.path.join(ROOT, "tests", "golden", "data")
cases = []
sp, tree, dag = T.hello_instance(data_dir)
cases.append(("hello", sp, dag, dag.branch_lengths(tree.branch_lengths), 1e-40))
sp, tree, dag = T._flu(data_dir)
cases.append(("fluA thr 1e-40", sp, dag, dag.branch_lengths(np.full(tree.node_count, 0.01)), 1e-40))
cases.append(("fluA thr 1e-4", sp, dag, dag.branch_lengths(np.full(tree.node_count, 0.01)), 1e-4))
tc, sp2 = workloads.load_ds1("DS1.subsampled_10.t")
w = workloads.ds1_gtr_weibull4(1)
pid = list(w.parent_ids[0])
n = sp2.taxon_count
kids = [c for c, p in enumerate(pid) if p == 2 * n - 3]
pid = pid + [2 * n - 2]
pid[kids[0]] = 2 * n - 2
dag3 = gp.single_tree_dag(pid)
bl3 = np.append(w.branch_lengths[0, :2 * n - 2], 0.0)
bl3[2 * n - 3] = 0.05
cases.append(("DS1 rooted", sp2, dag3, dag3.branch_lengths(bl3), 1e-40))
for name, sp, dag, bl, thr in cases:
    gpu = gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count, thr)
    cpu = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count, thr)
    for eng in (gpu, cpu):
        eng.set_branch_lengths(bl)
        eng.process_operations(dag.populate_plvs())
        eng.process_operations(dag.compute_likelihoods())
    child = dag.children[dag.root][0]
    args = (dag.edge(child), dag.pv(gp.R_LEFT, dag.root), dag.pv(gp.P, child))
    a, b = gpu.log_likelihood_and_first_two_derivatives(*args), cpu.log_likelihood_and_first_two_derivatives(*args)
    print(f"{name}: marginal {cpu.get_log_marginal_likelihood():.6f} d={abs(gpu.get_log_marginal_likelihood() - cpu.get_log_marginal_likelihood()):.3e}; "
          f"per-edge LL d={np.abs(gpu.get_per_gpcsp_log_likelihoods() - cpu.get_per_gpcsp_log_likelihoods()).max():.3e}; "
          f"edge LL d={abs(a[0] - b[0]):.3e} d1 {b[1]:.4f} d={abs(a[1] - b[1]):.3e} d2 {b[2]:.4f} d={abs(a[2] - b[2]):.3e}")
newton, gpu = T._optimized_venus_length(T._gpu_factory, data_dir, gp.NEWTON)
newton_cpu, cpu = T._optimized_venus_length(T._oracle_factory, data_dir, gp.NEWTON)
print(f"optimised lengths d={np.abs(gpu.get_branch_lengths() - cpu.get_branch_lengths()).max():.3e}; marginal d="
      f"{abs(gpu.get_log_marginal_likelihood() - cpu.get_log_marginal_likelihood()):.3e}")
