"""Path B on the GPU box: the Brent optimiser's function evaluations, device against CPU checker, edge by edge
(tests/gp_trace.py) -- hello, fluA (two sweeps), the DS1 ten-tree DAG and the 970-edge seeded DAG; prints where the two
records part, with the rows around that place, and the distance of the optimised lengths.
usage: python scripts/gpu_gp_brent_trace.py [scheduled: 1 | 0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    os.environ["BITO_AMD_GP_SCHEDULE"] = sys.argv[1]

import gp_trace  # noqa: E402
from test_gp import _flu, _gpu_factory, _oracle_factory, _traced_sweep, hello_instance  # noqa: E402

from bito_amd import gp, workloads  # noqa: E402

D = os.path.join(ROOT, "tests", "golden", "data")
sp, tree, hello = hello_instance(D)
cases = [("hello", sp, hello, hello.branch_lengths(tree.branch_lengths), 1)]
sp, tree, flu = _flu(D)
cases.append(("fluA", sp, flu, flu.branch_lengths(np.full(tree.node_count, 0.01)), 2))
dag, sp2 = workloads.ds1_subsplit_dag(10)
cases.append(("DS1 ten-tree DAG", sp2, dag, np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count), 1))
dag3, sp3 = workloads.seeded_subsplit_dag(20)
cases.append(("seeded 970-edge DAG", sp3, dag3, np.random.default_rng(1).uniform(0.01, 0.2, dag3.gpcsp_count), 1))
for name, sp_, dag_, bl0, sweeps in cases:
    for method, label in ((gp.BRENT, "Brent"), (gp.BRENT_WITH_GRADIENTS, "Brent with gradients")):
        cpu, bl_cpu = _traced_sweep(_oracle_factory, sp_, dag_, bl0, method, sweeps=sweeps)
        gpu, bl_gpu = _traced_sweep(_gpu_factory, sp_, dag_, bl0, method, sweeps=sweeps)
        problems, stats = gp_trace.compare(cpu, gpu)
        print(f"{name}, {label}: {len(cpu)} evaluations on the CPU, {len(gpu)} on the device; compared {stats['compared']}, "
              f"max |dx| {stats['max_dx']:.2e} |dt| {stats['max_dt']:.2e} |df| {stats['max_df']:.2e}, near-ties {stats['ties']}, "
              f"smallest margins {stats['smallest_choice_margin']:.2e} (choice) {stats['smallest_value_margin']:.2e} (value); max |d length| {np.abs(bl_gpu - bl_cpu).max():.3e}"
              + (f"; explained by a near-tie: {stats['explained_at']}" if stats["explained_at"] else ""))
        for msg in problems[:3]:
            print("   PROBLEM", msg)
        if problems:
            e = int(problems[0].split()[1])
            a, b = gp_trace.by_edge(cpu)[e][0], gp_trace.by_edge(gpu).get(e, [np.zeros((0, 4))])[0]
            for j in range(max(len(a), len(b))):
                print("     ", j, a[j, 1:].tolist() if j < len(a) else None, b[j, 1:].tolist() if j < len(b) else None)
