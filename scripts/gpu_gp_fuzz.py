"""Randomised sweep of the GP executor against the CPU checker: random subsplit DAGs (1-8 random rooted
trees on 4-12 taxa), random patterns and branch lengths; schedules (levelled execution), multi-tree
branch-length optimisation, batched NNI proposals.  usage: python scripts/gpu_gp_fuzz.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bito_amd import gp
from bito_amd.gp_dag import SubsplitDAG
from bito_amd.nni import NNIEvalEngineViaGP
from oracle import gp as ogp
from test_gpu_parity import _random_rooted_parent_ids

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad, t0 = 0, time.time()
for case in range(cases):
    n = int(rng.integers(4, 13))
    K = int(rng.integers(1, 9))
    P = int(rng.choice([1, 7, 64, 65, 200]))
    pids = [_random_rooted_parent_ids(n, rng) for _ in range(K)]
    dag = SubsplitDAG(n, pids)
    if rng.random() < 0.3:
        dag = dag.fully_connected()
    patterns = rng.integers(0, 4, (n, P)).astype(np.int32)
    patterns[rng.random((n, P)) < 0.1] = 4
    weights = rng.integers(1, 6, P).astype(np.float64)
    bl = rng.uniform(0.01, 0.5, dag.gpcsp_count)
    q = dag.uniform_on_topological_support_prior()
    desc = f"case {case}: n={n} trees={K} P={P} nodes={dag.node_count} edges={dag.gpcsp_count}"
    try:
        res = []
        for make in (gp.GPEngine, ogp.OracleGPEngine):
            dag.set_clean()
            eng = make(patterns, weights, dag.node_count, dag.gpcsp_count)
            eng.set_branch_lengths(bl)
            eng.set_sbn_parameters(q)
            eng.process_operations(dag.populate_plvs())
            eng.process_operations(dag.compute_likelihoods())
            marg, per = eng.get_log_marginal_likelihood(), eng.get_per_gpcsp_log_likelihoods()
            ev = NNIEvalEngineViaGP(dag, eng)
            scores = ev.score_adjacent_nnis()
            eng.set_optimization_method(gp.NEWTON)
            sweeps = gp.estimate_branch_lengths(eng, dag, 1e-6, 3)
            res.append((marg, per, scores, eng.get_branch_lengths(), eng.get_log_marginal_likelihood(), sweeps))
        (m1, p1, s1, b1, a1, w1), (m2, p2, s2, b2, a2, w2) = res
        ok = abs(m1 - m2) < 1e-9 * max(1, abs(m2)) and np.allclose(p1, p2, rtol=1e-11, atol=1e-9)
        ok = ok and s1.keys() == s2.keys() and all(abs(s1[k] - s2[k]) < 1e-9 * max(1, abs(s2[k])) for k in s1)
        ok = ok and w1 == w2 and np.allclose(b1, b2, rtol=1e-6, atol=1e-8) and abs(a1 - a2) < 1e-7 * max(1, abs(a2))
        if not ok:
            bad += 1
            print("MISMATCH", desc, abs(m1 - m2), np.abs(p1 - p2).max(), np.abs(b1 - b2).max(), abs(a1 - a2), w1, w2)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("ERROR", desc, repr(e)[:300])
print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s")
