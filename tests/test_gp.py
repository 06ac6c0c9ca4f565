"""GPEngine path: the CPU oracle is pinned to the reference's
known answers (CPU), and the GPU executor is checked against the oracle (-m gpu)."""
import os

import numpy as np
import pytest

from bito_amd import gp, treeio, workloads
from bito_amd.site_pattern import SitePattern
from oracle import gp as ogp
from oracle import oracle


def hello_instance(data_dir, fasta="hello.fasta"):
    """MakeHelloGPInstance (reference src/gp_doctest.cpp:59-76):
    (jupiter:0.113,(mars:0.15,saturn:0.1)venus:0.22):0 with hello.fasta."""
    coll = treeio.parse_newick_strings(["(jupiter:0.113,(mars:0.15,saturn:0.1):0.22):0;"])
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, fasta)), coll.taxon_names)
    tree = coll.trees[0]
    dag = gp.single_tree_dag(tree.parent_ids)
    return sp, tree, dag


def run(engine, dag, tree):
    engine.set_branch_lengths(dag.branch_lengths(tree.branch_lengths))
    engine.process_operations(dag.populate_plvs())
    engine.process_operations(dag.compute_likelihoods())
    return engine


def test_gp_oracle_classical_likelihood(data_dir):
    """src/gp_doctest.cpp:119-131: every per-GPCSP log-likelihood and the marginal are -84.77961943."""
    sp, tree, dag = hello_instance(data_dir)
    eng = run(ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count), dag, tree)
    assert np.abs(eng.get_per_gpcsp_log_likelihoods() - -84.77961943).max() < 1e-6
    assert abs(eng.get_log_marginal_likelihood() - -84.77961943) < 1e-6


def test_gp_oracle_derivatives(data_dir):
    """src/gp_doctest.cpp:257-308: log-likelihood and derivatives on the rootsplit -> jupiter edge."""
    for fasta, expect in (("hello_single_nucleotide.fasta", (-4.806671945, -0.6109379521, None)),
                          ("hello.fasta", (-84.77961943, -18.22479569, -5.4460787413))):
        sp, tree, dag = hello_instance(data_dir, fasta)
        eng = run(ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count), dag, tree)
        jupiter = 0
        parent, is_left = dag.parent[jupiter]
        assert parent == dag.root
        got = eng.log_likelihood_and_first_two_derivatives(
            dag.edge(jupiter), dag.pv(gp.R_LEFT if is_left else gp.R_RIGHT, parent), dag.pv(gp.P, jupiter))
        assert abs(got[0] - expect[0]) < 1e-6 and abs(got[1] - expect[1]) < 1e-6
        if expect[2] is not None:
            assert abs(got[2] - expect[2]) < 1e-6
    P = ogp.transition_matrix(0.75)  # src/gp_engine.hpp:382-393
    assert abs(P[0, 0] - 0.52590958087) < 1e-10 and abs(P[0, 1] - 0.1580301397) < 1e-10


def _flu(data_dir):
    tc = treeio.read_newick_file(os.path.join(data_dir, "fluA.tree"))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, "fluA.fa")), tc.taxon_names)
    tree = tc.trees[0]
    return sp, tree, gp.single_tree_dag(tree.parent_ids)


def test_gp_oracle_equals_per_tree_engine_and_is_rescaling_invariant(data_dir):
    """For a one-tree DAG the GP marginal is the tree's rooted JC69 log-likelihood (the reference's
    GP-vs-FatBeagle consistency check, src/gp_doctest.cpp:140-233), and it does not depend on the
    rescaling threshold (src/gp_doctest.cpp:348-360, fluA with all branches 0.01)."""
    sp, tree, dag = _flu(data_dir)
    bl = np.full(tree.node_count, 0.01)
    results = []
    for thr in (1e-40, 1e-4):
        eng = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count, thr)
        eng.set_branch_lengths(dag.branch_lengths(bl))
        eng.process_operations(dag.populate_plvs())
        eng.process_operations(dag.compute_likelihoods())
        results.append((eng.get_log_marginal_likelihood(), eng.get_per_gpcsp_log_likelihoods()))
    assert abs(results[0][0] - results[1][0]) < 1e-10
    assert np.abs(results[0][1] - results[1][1]).max() < 1e-9
    per_tree = oracle.OracleEngine("JC69", "constant", "none", sp.patterns, sp.weights, 1)
    ll = per_tree.log_likelihoods(tree.parent_ids[None, :], bl[None, :])[0]
    assert abs(results[0][0] - ll) < 1e-8
    assert np.abs(results[0][1] - ll).max() < 1e-8  # every edge of a one-tree DAG carries the tree likelihood


@pytest.mark.gpu
def test_gp_executor_matches_oracle(data_dir):
    cases = []
    sp, tree, dag = hello_instance(data_dir)
    cases.append((sp, dag, dag.branch_lengths(tree.branch_lengths), 1e-40))
    sp, tree, dag = _flu(data_dir)
    cases.append((sp, dag, dag.branch_lengths(np.full(tree.node_count, 0.01)), 1e-40))
    cases.append((sp, dag, dag.branch_lengths(np.full(tree.node_count, 0.01)), 1e-4))
    tc, sp2 = workloads.load_ds1("DS1.subsampled_10.t")
    # a rooted version of a DS1 tree: split the trifurcation
    w = workloads.ds1_gtr_weibull4(1)
    pid = list(w.parent_ids[0])
    n = sp2.taxon_count
    kids = [c for c, p in enumerate(pid) if p == 2 * n - 3]
    pid = pid + [2 * n - 2]
    pid[kids[0]] = 2 * n - 2
    dag3 = gp.single_tree_dag(pid)
    bl3 = np.append(w.branch_lengths[0, :2 * n - 2], 0.0)
    bl3[2 * n - 3] = 0.05
    cases.append((sp2, dag3, dag3.branch_lengths(bl3), 1e-40))
    for sp, dag, bl, thr in cases:
        gpu = gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count, thr)
        cpu = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count, thr)
        for eng in (gpu, cpu):
            eng.set_branch_lengths(bl)
            eng.process_operations(dag.populate_plvs())
            eng.process_operations(dag.compute_likelihoods())
        assert abs(gpu.get_log_marginal_likelihood() - cpu.get_log_marginal_likelihood()) < 1e-9
        assert np.abs(gpu.get_per_gpcsp_log_likelihoods() - cpu.get_per_gpcsp_log_likelihoods()).max() < 1e-9
        child = dag.children[dag.root][0]
        args = (dag.edge(child), dag.pv(gp.R_LEFT, dag.root), dag.pv(gp.P, child))
        a, b = gpu.log_likelihood_and_first_two_derivatives(*args), cpu.log_likelihood_and_first_two_derivatives(*args)
        assert abs(a[0] - b[0]) < 1e-9 and abs(a[1] - b[1]) < 1e-7 and abs(a[2] - b[2]) < 1e-6
        assert np.array_equal(gpu.get_branch_lengths(), bl)
    # the reference's hello goldens straight from the GPU
    sp, tree, dag = hello_instance(data_dir)
    gpu = run(gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count), dag, tree)
    assert np.abs(gpu.get_per_gpcsp_log_likelihoods() - -84.77961943).max() < 1e-6
    assert abs(gpu.get_log_marginal_likelihood() - -84.77961943) < 1e-6
    got = gpu.log_likelihood_and_first_two_derivatives(dag.edge(0), dag.pv(gp.R_LEFT, dag.root), dag.pv(gp.P, 0))
    assert abs(got[1] - -18.22479569) < 1e-6 and abs(got[2] - -5.4460787413) < 1e-6
    # SBN update across two sibling edges: softmax of weighted per-edge likelihoods + log prior
    s = gp.OpStream()
    s.add(gp.UPDATE_SBN_PROBABILITIES, 1, 3)
    q0 = np.ones(dag.gpcsp_count)
    q0[1], q0[2] = 0.3, 0.7
    cpu = run(ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count), dag, tree)
    for eng in (gpu, cpu):
        eng.set_sbn_parameters(q0)
        eng.process_operations(s)
    assert np.abs(gpu.get_sbn_parameters() - cpu.get_sbn_parameters()).max() < 1e-12
    with pytest.raises(Exception, match="OptimizeBranchLength"):
        bad = gp.OpStream()
        bad.add(gp.OPTIMIZE_BRANCH_LENGTH, 0, 0, 0)
        gpu.process_operations(bad)
