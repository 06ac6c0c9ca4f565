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
        # BASELINE.json's bars: 1e-10 on log-likelihoods (2e-14 relative where 1e-10 is below an ulp: these are
        # -4985 and -13815), 1e-6 on derivatives -- held here to 1e-9 (+ 1e-12 relative).  Measured on the MI355X
        # (scripts/gpu_gp_diffs.py, profiles/r3_gp_diffs.log): log-likelihoods 0 to 1.8e-11, first derivatives of
        # -388 and -1665 to 3.4e-12, second derivatives of -28078 to 8.5e-13.
        ll_bound = lambda v: 1e-10 + 2e-14 * abs(v)  # noqa: E731
        d_bound = lambda v: 1e-9 + 1e-12 * abs(v)  # noqa: E731
        want = cpu.get_log_marginal_likelihood()
        assert abs(gpu.get_log_marginal_likelihood() - want) < ll_bound(want)
        per_edge = cpu.get_per_gpcsp_log_likelihoods()
        assert np.all(np.abs(gpu.get_per_gpcsp_log_likelihoods() - per_edge) < 1e-10 + 2e-14 * np.abs(per_edge))
        child = dag.children[dag.root][0]
        args = (dag.edge(child), dag.pv(gp.R_LEFT, dag.root), dag.pv(gp.P, child))
        a, b = gpu.log_likelihood_and_first_two_derivatives(*args), cpu.log_likelihood_and_first_two_derivatives(*args)
        assert abs(a[0] - b[0]) < ll_bound(b[0]) and abs(a[1] - b[1]) < d_bound(b[1]) and abs(a[2] - b[2]) < d_bound(b[2])
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
    with pytest.raises(Exception, match="out of range"):
        bad = gp.OpStream()
        bad.add(gp.OPTIMIZE_BRANCH_LENGTH, 0, 0, dag.gpcsp_count)
        gpu.process_operations(bad)


def _optimized_venus_length(engine_factory, data_dir, method):
    """ObtainBranchLengthWithOptimization (reference src/gp_doctest.cpp:310-324): the branch of
    PCSP 100|011 -> 001, i.e. the edge above venus = (mars, saturn)."""
    sp, tree, dag = hello_instance(data_dir)
    eng = engine_factory(sp, dag)
    eng.set_branch_lengths(dag.branch_lengths(tree.branch_lengths))
    eng.set_optimization_method(method)
    gp.estimate_branch_lengths(eng, dag, 0.0001, 100)
    venus = dag.children[dag.root][1]
    assert venus == 3
    return eng.get_branch_lengths()[dag.edge(venus)], eng


def _oracle_factory(sp, dag):
    return ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)


def test_gp_oracle_branch_length_optimization(data_dir):
    """src/gp_doctest.cpp:326-346: Newton reaches 0.0694244266 to 1e-6 and beats Brent."""
    true_length = 0.0694244266
    brent, _ = _optimized_venus_length(_oracle_factory, data_dir, gp.BRENT)
    newton, eng = _optimized_venus_length(_oracle_factory, data_dir, gp.NEWTON)
    assert abs(newton - true_length) < 1e-6
    assert abs(newton - true_length) < abs(brent - true_length)
    # every method climbs to (nearly) the same optimum of the marginal likelihood
    best = eng.get_log_marginal_likelihood()
    for method in (gp.BRENT, gp.BRENT_WITH_GRADIENTS):
        value, e2 = _optimized_venus_length(_oracle_factory, data_dir, method)
        assert abs(value - true_length) < 2e-4, method
        assert best - 1e-5 < e2.get_log_marginal_likelihood() <= best + 1e-9
    # plain gradient ascent also reaches the optimum of the marginal; the two root edges are not
    # separately identifiable and the reference clamps at the minimum LOG length, so only their sum is checked
    _, e3 = _optimized_venus_length(_oracle_factory, data_dir, gp.GRADIENT_ASCENT)
    bl = e3.get_branch_lengths()
    assert abs(bl[1] + bl[4] - true_length) < 1e-4 and abs(e3.get_log_marginal_likelihood() - best) < 1e-6


def test_single_tree_branch_length_schedule_shape(data_dir):
    sp, tree, dag = hello_instance(data_dir)
    ops, side = dag.branch_length_optimization().arrays()
    opt = ops[ops["opcode"] == gp.OPTIMIZE_BRANCH_LENGTH]
    # one optimisation per edge of the tree, children of a node left then right, depth first
    assert [int(c) for c in opt["c"]] == [dag.edge(0), dag.edge(1), dag.edge(2), dag.edge(3)]
    assert int(opt["a"][0]) == dag.pv(gp.P, 0) and int(opt["b"][0]) == dag.pv(gp.R_LEFT, dag.root)
    only = dag.branch_length_optimization({dag.edge(3)}).arrays()[0]
    assert (only["opcode"] == gp.OPTIMIZE_BRANCH_LENGTH).sum() == 1


def _gpu_factory(sp, dag):
    return gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)


@pytest.mark.gpu
def test_gp_branch_length_optimization_on_device(data_dir):
    """src/gp_doctest.cpp:326-346 on the GPU executor, and the device optimisers against the oracle."""
    true_length = 0.0694244266
    brent, _ = _optimized_venus_length(_gpu_factory, data_dir, gp.BRENT)
    newton, gpu = _optimized_venus_length(_gpu_factory, data_dir, gp.NEWTON)
    assert abs(newton - true_length) < 1e-6
    assert abs(newton - true_length) < abs(brent - true_length)
    newton_cpu, cpu = _optimized_venus_length(_oracle_factory, data_dir, gp.NEWTON)
    # (measured: optimised lengths 1.8e-16 apart, marginals equal -- scripts/gpu_gp_diffs.py)
    assert np.abs(gpu.get_branch_lengths() - cpu.get_branch_lengths()).max() < 1e-12
    assert abs(gpu.get_log_marginal_likelihood() - cpu.get_log_marginal_likelihood()) < 1e-10
    # one sweep on fluA (69 taxa, 136 optimised edges), every method.  Brent is deterministic and none of its decisions on
    # this workload is near a tie (tests/gp_trace.py, test_brent_trace_comparison: rounding noise moves the optimised
    # lengths by 2e-12): the device follows the checker's iterates (test_device_brent_follows_the_checkers_iterates) and
    # ends at its lengths.  The gradient variant's last decisions on an edge are near-ties by construction (its second
    # trial point closes in on the best one as the derivative vanishes) and Brent stops at 10 significant BITS
    # (ldexp(1, 1 - digits), src/optimization.hpp:75): its argmin is compared loosely.
    sp, tree, dag = _flu(data_dir)
    bl0 = dag.branch_lengths(np.full(tree.node_count, 0.01))
    for method, bl_tol in ((gp.NEWTON, 1e-8), (gp.BRENT, 1e-8), (gp.BRENT_WITH_GRADIENTS, 2e-2)):
        results = []
        for factory in (_gpu_factory, _oracle_factory):
            eng = factory(sp, dag)
            eng.set_branch_lengths(bl0)
            eng.set_optimization_method(method)
            eng.reset_optimization_count()
            eng.process_operations(dag.populate_plvs())
            eng.process_operations(dag.branch_length_optimization())
            eng.process_operations(dag.populate_plvs())
            eng.process_operations(dag.marginal_likelihood())
            results.append((eng.get_branch_lengths(), eng.get_branch_length_differences(),
                            eng.get_log_marginal_likelihood()))
        (bg, dg, lg), (bc, dc, lc) = results
        assert np.abs(bg - bc).max() < bl_tol * max(1.0, np.abs(bc).max()), method
        assert np.abs(dg - dc).max() < bl_tol * max(1.0, np.abs(bc).max()), method
        assert abs(lg - lc) < (1e-8 if bl_tol < 1e-6 else 5e-2), method
        assert lg > -5000  # the sweep improved on the starting tree
    # gradient ascent (fixed step 5e-4) only behaves on small problems: hello, device against the oracle
    _, ga_gpu = _optimized_venus_length(_gpu_factory, data_dir, gp.GRADIENT_ASCENT)
    _, ga_cpu = _optimized_venus_length(_oracle_factory, data_dir, gp.GRADIENT_ASCENT)
    assert np.abs(ga_gpu.get_branch_lengths() - ga_cpu.get_branch_lengths()).max() < 1e-6
    _, lg_gpu = _optimized_venus_length(_gpu_factory, data_dir, gp.LOGSPACE_GRADIENT_ASCENT)
    assert np.all(np.isfinite(lg_gpu.get_branch_lengths()))
    # the second sweep skips converged edges (differences below 1e-15) only after the count is incremented
    eng = _gpu_factory(sp, dag)
    eng.set_branch_lengths(bl0)
    eng.set_optimization_method(gp.NEWTON)
    sweeps = gp.estimate_branch_lengths(eng, dag, 1e-6, 20)
    assert 1 < sweeps <= 20 and float(np.mean(eng.get_branch_length_differences())) < 1e-6


# ---- multi-tree subsplit DAGs: GP marginal == brute force over every tree of the DAG ------------
# (the reference's TestCompositeMarginal, src/gp_doctest.cpp:140-254; the identity holds for any
# per-PCSP branch lengths, so seeded ones are used instead of an optimisation run)

from bito_amd import gp_dag  # noqa: E402

COMPOSITE_CASES = [("hello.fasta", "hello_rooted_two_trees.nwk"), ("five_taxon.fasta", "five_taxon_rooted.nwk"),
                   ("ds1-reduced-5.fasta", "ds1-reduced-5.nwk"),
                   ("7-taxon-slice-of-ds1.fasta", "simplest-hybrid-marginal-all-trees.nwk")]


def _composite_case(data_dir, fasta, newick):
    tc = treeio.read_newick_file(os.path.join(data_dir, newick))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, fasta)), tc.taxon_names)
    dag = gp_dag.SubsplitDAG(sp.taxon_count, [t.parent_ids for t in tc.trees])
    rng = np.random.default_rng(11)
    bl = rng.uniform(0.01, 0.3, dag.gpcsp_count)
    bl[:len(dag.rootsplits)] = 0.0  # rootsplit edges carry no length
    return sp, dag, bl


def _exact_marginals(sp, dag, bl):
    """ComputeExactMarginal (src/gp_doctest.cpp:140-187) with site patterns in place of columns:
    the tree likelihoods come from the Path A oracle, one single-pattern engine per pattern."""
    trees = list(dag.all_trees())
    assert len(trees) == dag.topology_count
    n = sp.taxon_count
    pid = np.stack([t[0] for t in trees])
    lengths = np.array([[bl[e] for e in t[1]] for t in trees])
    log_prior = np.log(1.0 / len(trees))
    total, per_pcsp = 0.0, np.zeros(dag.gpcsp_count)
    for p in range(sp.patterns.shape[1]):
        eng = oracle.OracleEngine("JC69", "constant", "strict", sp.patterns[:, p:p + 1], np.ones(1), 1)
        ll = eng.log_likelihoods(pid, lengths)
        total += sp.weights[p] * (np.logaddexp.reduce(ll) + log_prior)
        for e in range(dag.gpcsp_count):
            members = [ll[k] for k, t in enumerate(trees) if e in t[1]]
            per_pcsp[e] += sp.weights[p] * (np.logaddexp.reduce(members) + log_prior)
    return total, per_pcsp


def _check_composite(engine, sp, dag, bl):
    engine.set_branch_lengths(bl)
    q = dag.uniform_on_topological_support_prior()
    engine.set_sbn_parameters(q)
    engine.process_operations(dag.populate_plvs())
    engine.process_operations(dag.compute_likelihoods())
    exact, exact_per_pcsp = _exact_marginals(sp, dag, bl)
    assert abs(engine.get_log_marginal_likelihood() - exact) < 1e-6
    # PerGPCSPComponentsOfFullLogMarginal (src/gp_engine.cpp:455-458)
    components = engine.get_per_gpcsp_log_likelihoods() + sp.weights.sum() * np.log(q)
    assert np.abs(components - exact_per_pcsp).max() < 1e-5


@pytest.mark.parametrize("fasta,newick", COMPOSITE_CASES)
def test_gp_oracle_composite_marginal(data_dir, fasta, newick):
    sp, dag, bl = _composite_case(data_dir, fasta, newick)
    _check_composite(ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count), sp, dag, bl)


def test_subsplit_dag_shape(data_dir):
    """DAGSummaryStatistics of the two-tree hello DAG (src/gp_doctest.cpp:105-109): 8 nodes and 10
    edges with the DAG root and its rootsplit edges counted."""
    sp, dag, _ = _composite_case(data_dir, "hello.fasta", "hello_rooted_two_trees.nwk")
    assert dag.node_count + 1 == 8 and dag.gpcsp_count == 10
    assert dag.topology_count == 2 and len(dag.rootsplits) == 2
    assert abs(dag.uniform_on_topological_support_prior()[:2].sum() - 1.0) < 1e-15


@pytest.mark.gpu
@pytest.mark.parametrize("fasta,newick", COMPOSITE_CASES)
def test_gp_executor_composite_marginal(data_dir, fasta, newick):
    sp, dag, bl = _composite_case(data_dir, fasta, newick)
    gpu = gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
    _check_composite(gpu, sp, dag, bl)
    # SBN parameter optimisation on the multi-parent DAG, device against the oracle
    cpu = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
    for eng in (gpu, cpu):
        eng.set_branch_lengths(bl)
        eng.set_sbn_parameters(dag.uniform_on_topological_support_prior())
        eng.process_operations(dag.populate_plvs())
        eng.process_operations(dag.compute_likelihoods())
        eng.process_operations(dag.optimize_sbn_parameters())
    assert np.abs(gpu.get_sbn_parameters() - cpu.get_sbn_parameters()).max() < 1e-10


# -- f1 on multi-tree DAGs: GPDAG::BranchLengthOptimization over the tidy traversal ------------------

def _motivating_dag():
    """TidySubsplitDAG::MotivatingExample (src/tidy_subsplit_dag.cpp:149-153): (0,(1,(2,3))) and ((0,(2,3)),1)."""
    tc = treeio.parse_newick_strings(["(x0,(x1,(x2,x3)));", "((x0,(x2,x3)),x1);"])
    return gp_dag.SubsplitDAG(4, [t.parent_ids for t in tc.trees])


def test_tidy_dag_slicing():
    """"TidySubsplitDAG: slicing" (src/tidy_subsplit_dag.hpp:204-240) restated by subsplit: the reference's
    node ids are 4 = 2|3, 5/6/7 = the nodes holding 2|3 in their right clade, 8 = 023|1, 9 = the DAG root
    (which has no id here)."""
    dag = _motivating_dag()
    name = lambda ids: sorted(dag.subsplits[i] for i in ids)
    n23, n0_23 = dag.node_id[(4, 8)], dag.node_id[(1, 12)]
    assert name(dag.above_node(0, n23)) == sorted([(4, 8), (1, 12), (2, 12), (1, 14)])  # [0,0,0,0,1,1,1,1,0,0]
    assert name(dag.above_node(1, n23)) == sorted([(4, 8), (13, 2)])  # [0,0,0,0,1,0,0,0,1,1] less the root
    assert name(dag.above_node(0, n0_23)) == [(1, 12)]
    assert name(dag.above_node(1, n0_23)) == sorted([(1, 12), (13, 2)])
    assert name(dag.below_node(0, n0_23)) == sorted([(0, 4), (0, 8), (4, 8), (1, 12)])  # taxa 2, 3, node 2|3, itself
    assert name(dag.below_node(1, n0_23)) == sorted([(0, 1), (1, 12)])
    dag.set_dirty_strictly_above(n23)
    assert name(dag.dirty_vector(1)) == [(13, 2)]
    assert name(dag.dirty_vector(0)) == sorted([(1, 12), (2, 12), (1, 14)])
    dag.set_clean()
    assert not dag.dirty_vector(0) and not dag.dirty_vector(1)


def test_tidy_traversal_modifies_every_edge_once_and_updates_where_needed():
    """The motivating example is the one where a clade goes stale: 2|3 hangs under two parents, so by the
    time the traversal reaches 0|23 its right p-hat was dirtied by the optimisation of 2|3's edges under
    1|23 -- the traversal must bring it up to date before it descends (src/tidy_subsplit_dag.hpp:93-96)."""
    dag = _motivating_dag()
    record = []
    stream = dag.branch_length_optimization(record=record)
    modified = [(r[1], r[2]) for r in record if r[0] == "modify"]
    non_root_edges = [k for k in dag.edge_id if k[0] >= 0]
    assert sorted(modified) == sorted(non_root_edges) and len(set(modified)) == len(modified)
    updates = [r for r in record if r[0] == "update"]
    assert [(dag.subsplits[r[1]], dag.subsplits[r[2]], r[3]) for r in updates] == [((1, 12), (4, 8), 0)]
    # the update comes before 0|23's first descent
    first_descent = next(i for i, r in enumerate(record) if r[0] == "descend" and dag.subsplits[r[1]] == (1, 12))
    assert record.index(updates[0]) < first_descent
    ops = stream.arrays()[0]
    assert (ops["opcode"] == gp.OPTIMIZE_BRANCH_LENGTH).sum() == len(non_root_edges)
    # one tree: never an update
    single = gp_dag.SubsplitDAG(4, [treeio.parse_newick_strings(["(x0,(x1,(x2,x3)));"]).trees[0].parent_ids])
    record = []
    single.branch_length_optimization(record=record)
    assert not [r for r in record if r[0] == "update"]


def _edge_key(dag, e):
    (p, c), = [k for k, v in dag.edge_id.items() if v == e]
    return (None if p < 0 else dag.subsplits[p], dag.subsplits[c])


def test_multi_tree_schedule_equals_the_single_tree_schedule_on_one_tree(data_dir):
    """The general traversal run on the DAG of ONE tree optimises to the same lengths as the single-tree
    schedule (ids differ; edges are matched by the clade below them)."""
    tc = treeio.read_newick_file(os.path.join(data_dir, "five_taxon_rooted.nwk"))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, "five_taxon.fasta")), tc.taxon_names)
    pid = tc.trees[0].parent_ids
    general = gp_dag.SubsplitDAG(sp.taxon_count, [pid])
    single = gp.single_tree_dag(pid)
    clade = [1 << i for i in range(sp.taxon_count)] + [0] * (len(pid) + 1 - sp.taxon_count)
    for c, p in enumerate(pid):
        clade[p] |= clade[c]
    results = []
    for dag in (general, single):
        eng = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
        eng.set_branch_lengths(np.full(dag.gpcsp_count, 0.1))
        eng.set_optimization_method(gp.NEWTON)
        sweeps = gp.estimate_branch_lengths(eng, dag, 1e-8, 50)
        results.append((eng.get_branch_lengths(), eng.get_log_marginal_likelihood(), sweeps))
    (gb, gl, gs), (sb, sl, ss) = results
    assert gs == ss and abs(gl - sl) < 1e-10
    for e in range(general.gpcsp_count):
        parent, child = _edge_key(general, e)
        if parent is None:
            continue
        node = clade.index(child[0] | child[1])
        assert abs(gb[e] - sb[single.edge(node)]) < 1e-10


def _worst_edge_gradient(eng, dag):
    """max over edges (away from the optimiser's bounds) of |d logL_e / d log t_e| with current PLVs."""
    bl, worst = eng.get_branch_lengths(), 0.0
    for (p, c), e in dag.edge_id.items():
        if p < 0 or not 2e-6 < bl[e] < 2.9:
            continue
        side = 1 if c in dag.children[p][1] else 0
        _, d1, _ = eng.log_likelihood_and_first_two_derivatives(e, dag.pv(gp.R_LEFT if side else gp.R_RIGHT, p), dag.pv(gp.P, c))
        worst = max(worst, abs(d1 * bl[e]))
    return worst


@pytest.mark.parametrize("zero_before_update", [False, True])
@pytest.mark.parametrize("fasta,newick", COMPOSITE_CASES)
def test_estimate_branch_lengths_on_multi_tree_dags(data_dir, fasta, newick, zero_before_update):
    """TestCompositeMarginal as the reference runs it (src/gp_doctest.cpp:210-232): EstimateBranchLengths
    first, then the composite-marginal identity with the optimised lengths.  On top: the marginal went up
    and the sweeps reach a fixed point.  With the p-hat cleared before an update (see
    SubsplitDAG.branch_length_optimization) that fixed point is a stationary point of every edge's own
    likelihood; with the reference's schedule it is not on ds1-reduced-5 (measured 2e-2 in d logL / d log t),
    the one fixture where an update step fires on a clade with several parents' worth of mass."""
    sp, dag, _ = _composite_case(data_dir, fasta, newick)
    eng = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
    eng.set_branch_lengths(np.full(dag.gpcsp_count, 0.1))  # init_default_branch_length_
    eng.set_sbn_parameters(dag.uniform_on_topological_support_prior())
    eng.process_operations(dag.populate_plvs())
    eng.process_operations(dag.marginal_likelihood())
    before = eng.get_log_marginal_likelihood()
    eng.set_optimization_method(gp.NEWTON)
    sweeps = gp.estimate_branch_lengths(eng, dag, 1e-9, 200, zero_before_update=zero_before_update)
    assert sweeps < 200
    assert eng.get_log_marginal_likelihood() > before
    bl = eng.get_branch_lengths()
    eng.reset_optimization_count()  # one more sweep, no edge skipped as converged: nothing moves
    eng.process_operations(dag.branch_length_optimization(zero_before_update=zero_before_update))
    assert np.abs(eng.get_branch_lengths() - bl).max() < 1e-7
    eng.process_operations(dag.populate_plvs())
    if zero_before_update:
        assert _worst_edge_gradient(eng, dag) < 1e-5
    bl = eng.get_branch_lengths()
    bl[:len(dag.rootsplits)] = 0.0
    _check_composite(eng, sp, dag, bl)


@pytest.mark.gpu
@pytest.mark.parametrize("fasta,newick", COMPOSITE_CASES)
def test_estimate_branch_lengths_on_multi_tree_dags_gpu(data_dir, fasta, newick):
    """The executor runs the multi-tree optimisation schedule to the same lengths as the CPU route."""
    sp, dag, _ = _composite_case(data_dir, fasta, newick)
    out = []
    for make in (gp.GPEngine, ogp.OracleGPEngine):
        dag.set_clean()
        eng = make(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
        eng.set_branch_lengths(np.full(dag.gpcsp_count, 0.1))
        eng.set_sbn_parameters(dag.uniform_on_topological_support_prior())
        eng.set_optimization_method(gp.NEWTON)
        sweeps = gp.estimate_branch_lengths(eng, dag, 1e-7, 60)
        out.append((eng.get_branch_lengths(), eng.get_log_marginal_likelihood(), sweeps))
    (gb, gl, gs), (cb, cl, cs) = out
    assert gs == cs
    assert np.abs(gb - cb).max() < 1e-7 and abs(gl - cl) < 1e-8


def _populated(factory_engine, sp, dag, bl, thr):
    eng = factory_engine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count, thr)
    eng.set_branch_lengths(bl)
    eng.process_operations(dag.populate_plvs())
    eng.process_operations(dag.compute_likelihoods())
    return eng


def test_oracle_rescaling_counts_grow_with_the_threshold(data_dir):
    """fluA with every branch 0.5: at threshold 1e-4 whole PLVs are rescaled up to six times, at 1e-40 (the default,
    src/gp_engine.hpp:285) never; with the short branches of the reference's own rescaling test (0.01,
    src/gp_doctest.cpp:348-360) some site pattern always keeps a PLV's maximum above 1e-4, so no whole-PLV count moves"""
    sp, tree, dag = _flu(data_dir)
    bl = dag.branch_lengths(np.full(tree.node_count, 0.5))
    loose = _populated(ogp.OracleGPEngine, sp, dag, bl, 1e-4).get_rescaling_counts(0, 6 * dag.node_count)
    tight = _populated(ogp.OracleGPEngine, sp, dag, bl, 1e-40).get_rescaling_counts(0, 6 * dag.node_count)
    assert loose.max() >= 5 and (loose > 0).sum() > 300 and tight.max() == 0
    short = dag.branch_lengths(np.full(tree.node_count, 0.01))
    assert _populated(ogp.OracleGPEngine, sp, dag, short, 1e-4).get_rescaling_counts(0, 6 * dag.node_count).max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("branch,thr", [(0.5, 1e-4), (0.1, 1e-2), (0.01, 1e-4)])
def test_gp_rescaling_counts_and_plvs_as_the_reference_holds_them(data_dir, branch, thr):
    """The executor rescales per pattern, the reference per whole PLV (RescalePLVIfNeeded, src/gp_engine.cpp:583-597).
    Through bito_amd_gp_rescaling_counts / _get_plv_as_reference a caller sees the reference's rescaling_counts_ and the
    values its GetPLV would return: every PLV of the fluA DAG, against the CPU checker, which keeps one count per PLV
    as the reference does -- with counts up to six (branches 0.5, threshold 1e-4) and with none moving (0.01, 1e-4:
    there the per-pattern counts do move, and the reference view must undo them)."""
    sp, tree, dag = _flu(data_dir)
    bl = dag.branch_lengths(np.full(tree.node_count, branch))
    gpu = _populated(gp.GPEngine, sp, dag, bl, thr)
    cpu = _populated(ogp.OracleGPEngine, sp, dag, bl, thr)
    count = 6 * dag.node_count
    want = cpu.get_rescaling_counts(0, count)
    got = gpu.get_rescaling_counts()
    assert np.array_equal(got, want)
    if branch >= 0.1:
        assert want.max() >= 3
    checked = differing = 0
    for plv in range(count):
        values, c = gpu.get_plv_as_reference(plv)
        ref = cpu.get_plv(plv)
        assert c == want[plv]
        assert np.allclose(values, ref, rtol=1e-12, atol=0.0), plv
        checked += int(ref.max() > 0)
        differing += int(not np.allclose(gpu.get_plv(plv), ref, rtol=1e-9))
    assert checked > count // 2
    # the raw per-pattern view differs from the reference's wherever a pattern was rescaled more often than its PLV
    assert differing > 0
    assert abs(gpu.get_log_marginal_likelihood() - cpu.get_log_marginal_likelihood()) < 1e-9


def _oracle_state(eng, dag):
    plvs = np.stack([eng.get_plv(k) for k in range(6 * dag.node_count)])
    return (eng.get_branch_lengths(), eng.get_branch_length_differences(), eng.get_per_gpcsp_log_likelihoods(),
            eng.get_sbn_parameters(), plvs, eng.get_rescaling_counts(0, 6 * dag.node_count))


def _run_streams_on_oracle(sp, dag, bl0, streams, method, threshold=1e-40):
    eng = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count, threshold)
    eng.set_sbn_parameters(dag.uniform_on_topological_support_prior())
    eng.set_optimization_method(method)
    eng.set_branch_lengths(bl0)
    eng.reset_optimization_count()
    for s in streams:
        eng.process_operations(s)
    return _oracle_state(eng, dag)


def _shuffled_within_levels(scheduled, launch, level, rng):
    """the scheduled stream with the operations of every (launch, level) in a random order: by the schedule's own claim
    they touch disjoint results, so any order of them computes the same bits"""
    out = gp.OpStream()
    out.side = list(scheduled.side)
    order = np.arange(len(scheduled.ops))
    keys = launch.astype(np.int64) * (1 << 32) + level
    for key in np.unique(keys):
        idx = np.nonzero(keys == key)[0]
        order[idx] = rng.permutation(idx)
    out.ops = [scheduled.ops[k] for k in order]
    return out


SCHEDULE_CASES = COMPOSITE_CASES + [("DS1.fasta", "ten-tree DAG")]


@pytest.mark.parametrize("fasta,newick", SCHEDULE_CASES)
def test_executor_schedule_is_the_sequential_arithmetic(data_dir, fasta, newick):
    """The order in which the executor runs a stream (bito_amd_gp_schedule_operations, bito_amd/csrc/gp_schedule.hpp:
    per-pattern operations by dependency level, the OptimizeBranchLength operations of equal optimiser depth as one
    launch of concurrent workgroups) must compute what the reference's one-after-the-other execution computes
    (src/gp_engine.cpp:213-339) -- bit for bit.  Checked on the CPU: the oracle, which executes strictly in stream order,
    runs the stream as given, the scheduled stream, and the scheduled stream with every (launch, level) shuffled; branch
    lengths, differences, log-likelihood rows, every PLV and every rescaling count must be identical.  Streams: the three
    schedules of GPInstance::EstimateBranchLengths (src/gp_instance.cpp:241-308) and the SBN-parameter update
    (src/gp_dag.cpp:123-137), whose UpdateSBNProbabilities operations are barriers."""
    if newick == "ten-tree DAG":
        dag, sp = workloads.ds1_subsplit_dag(10)
        bl0 = np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count)
    else:
        sp, dag, bl0 = _composite_case(data_dir, fasta, newick)
        bl0 = np.maximum(bl0, 0.01)
    streams = [dag.populate_plvs(), dag.compute_likelihoods(), dag.branch_length_optimization(), dag.populate_plvs(),
               dag.optimize_sbn_parameters(), dag.branch_length_optimization(zero_before_update=True), dag.marginal_likelihood()]
    rng = np.random.default_rng(23)
    for method in (gp.BRENT, gp.NEWTON):
        for threshold in (1e-40, 1e-4):  # (1e-4: the rescaling counts take part)
            want = _run_streams_on_oracle(sp, dag, bl0, streams, method, threshold)
            scheduled, shuffled = [], []
            for s in streams:
                t, launch, level, kinds = gp.schedule_operations(s)
                assert sorted(t.ops) == sorted(s.ops)  # a permutation of the stream
                scheduled.append(t)
                shuffled.append(_shuffled_within_levels(t, launch, level, rng))
            for variant in (scheduled, shuffled):
                got = _run_streams_on_oracle(sp, dag, bl0, variant, method, threshold)
                for x, y in zip(want, got):
                    assert np.array_equal(x, y, equal_nan=True)
            if newick != "ten-tree DAG":
                break  # (one threshold for the small cases' second method keeps the CPU suite short)
    # (the sweeps did move the lengths; on the ten-tree DAG Newton after the SBN update meets edges whose q underflowed to
    # zero and returns nan for them, as the reference's arithmetic does -- the nans are compared like every other value)
    assert np.nanmax(np.abs(want[0] - bl0)) > 1e-3


def test_executor_schedule_shape(data_dir):
    """What the schedule buys: the number of optimiser launches is the longest chain of optimisations in the stream's
    dependency graph -- 56 instead of 118 on the DS1 ten-tree DAG --, a single tree's sweep stays one chain (every edge's
    optimisation depends on the one before it through the partial vectors it refreshes), and without reordering every
    optimisation is a launch of its own in stream order."""
    dag, sp = workloads.ds1_subsplit_dag(10)
    sweep = dag.branch_length_optimization()
    optimisations = sum(1 for op in sweep.ops if op[0] == gp.OPTIMIZE_BRANCH_LENGTH)
    t, launch, level, kinds = gp.schedule_operations(sweep)
    assert optimisations == 118 and int((kinds == 1).sum()) == 56
    assert np.all(np.diff(launch) >= 0) and launch[-1] == len(kinds) - 1
    widths = np.bincount(launch[[k for k, op in enumerate(t.ops) if op[0] == gp.OPTIMIZE_BRANCH_LENGTH]])
    assert widths.max() >= 3
    # kinds alternate: never two per-pattern launches or two optimiser launches in a row
    assert np.all(kinds[1:] != kinds[:-1])
    plain, launch0, _, kinds0 = gp.schedule_operations(sweep, reorder=False)
    assert int((kinds0 == 1).sum()) == 118
    assert [op for op in plain.ops if op[0] == gp.OPTIMIZE_BRANCH_LENGTH] == [op for op in sweep.ops if op[0] == gp.OPTIMIZE_BRANCH_LENGTH]
    # schedules without an optimiser are one launch of levelled per-pattern operations
    t, launch, level, kinds = gp.schedule_operations(dag.populate_plvs())
    assert list(kinds) == [0] and level.max() + 1 < len(t.ops) // 8
    # a single tree (fluA, 69 taxa): one chain -- as many optimiser launches as optimisations
    _, _, flu = _flu(data_dir)
    _, _, _, kinds = gp.schedule_operations(flu.branch_length_optimization())
    assert int((kinds == 1).sum()) == 136


def _traced_sweep(factory, sp, dag, bl0, method, noise=0.0, sweeps=1):
    eng = factory(sp, dag)
    eng.set_sbn_parameters(dag.uniform_on_topological_support_prior()) if hasattr(dag, "uniform_on_topological_support_prior") else None
    eng.set_branch_lengths(bl0)
    eng.set_optimization_method(method)
    eng.reset_optimization_count()
    eng.start_optimizer_trace()
    if noise:
        eng.set_eval_noise(noise, 7)
    eng.process_operations(dag.populate_plvs())
    for _ in range(sweeps):
        eng.process_operations(dag.branch_length_optimization())
        eng.increment_optimization_count()
    trace = eng.optimizer_trace()
    if noise:
        eng.set_eval_noise(0.0)
    return trace, eng.get_branch_lengths()


def test_brent_trace_comparison(data_dir):
    """The instrument that holds the device's Brent to the checker's (tests/gp_trace.py), checked on the CPU: the checker
    against itself with every function value perturbed by 1e-15 relative -- what separates two correct implementations --
    follows the same iterates within the comparison's tolerances (same evaluation count on every edge, no near-tie on
    these workloads), while a perturbation of 1e-7 relative is told apart.  Reference: src/optimization.hpp:71-331."""
    import gp_trace

    sp, tree, flu = _flu(data_dir)
    cases = [(sp, flu, flu.branch_lengths(np.full(tree.node_count, 0.01)))]
    dag, sp2 = workloads.ds1_subsplit_dag(10)
    cases.append((sp2, dag, np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count)))
    for sp_, dag_, bl0 in cases:
        clean, bl_clean = _traced_sweep(_oracle_factory, sp_, dag_, bl0, gp.BRENT)
        noisy, bl_noisy = _traced_sweep(_oracle_factory, sp_, dag_, bl0, gp.BRENT, noise=1e-15)
        problems, stats = gp_trace.compare(clean, noisy)
        assert not problems, problems[:3]
        # Brent's decisions are far from ties (8e-7 at the least on DS1): nothing is left uncompared
        assert stats["explained_at"] is None and stats["compared"] == stats["rows"] == len(noisy)
        assert stats["ties"] == 0 and stats["smallest_value_margin"] > 1e-8 and stats["smallest_choice_margin"] > 1e-5
        assert np.abs(bl_clean - bl_noisy).max() < 1e-9
        rough, _ = _traced_sweep(_oracle_factory, sp_, dag_, bl0, gp.BRENT, noise=1e-7)
        problems, _ = gp_trace.compare(clean, rough)
        assert problems
        # The gradient variant is another matter: its second trial point (a step of 1.0005 x the derivative from the best
        # point) closes in on the best point as the derivative vanishes, so its last decisions on an edge are near-ties by
        # construction and rounding noise does flip some (fluA edge 74: 30 evaluations or 22).  The comparison says so.
        clean, _ = _traced_sweep(_oracle_factory, sp_, dag_, bl0, gp.BRENT_WITH_GRADIENTS)
        _, stats = gp_trace.compare(clean, clean)
        assert stats["ties"] > 0 and stats["compared"] == len(clean)


@pytest.mark.gpu
def test_device_brent_follows_the_checkers_iterates(data_dir):
    """Brent is the reference's DEFAULT optimiser (src/dag_branch_handler.hpp:262) and it is deterministic: with function
    values equal to rounding, the device's restatement of src/optimization.hpp:71-331 must visit the points the checker's
    visits -- the same number of evaluations on every edge, every trial point within 1e-6 + 1e-10 / length in the log
    length (tests/gp_trace.py; what rounding noise alone does is measured in test_brent_trace_comparison) -- and end
    at the same branch lengths.  hello, fluA (136 edges, two sweeps: the second skips converged edges), the DS1 ten-tree
    DAG (118 edges, optimised concurrently where the schedule allows).  (The gradient variant's last decisions on an edge
    are near-ties by construction -- test_brent_trace_comparison -- and stay with the looser comparison of
    test_gp_branch_length_optimization_on_device.)"""
    import gp_trace

    sp, tree, hello = hello_instance(data_dir)
    cases = [("hello", sp, hello, hello.branch_lengths(tree.branch_lengths), 1)]
    sp, tree, flu = _flu(data_dir)
    cases.append(("fluA", sp, flu, flu.branch_lengths(np.full(tree.node_count, 0.01)), 2))
    dag, sp2 = workloads.ds1_subsplit_dag(10)
    cases.append(("DS1 DAG", sp2, dag, np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count), 1))
    for name, sp_, dag_, bl0, sweeps in cases:
        cpu, bl_cpu = _traced_sweep(_oracle_factory, sp_, dag_, bl0, gp.BRENT, sweeps=sweeps)
        gpu, bl_gpu = _traced_sweep(_gpu_factory, sp_, dag_, bl0, gp.BRENT, sweeps=sweeps)
        problems, stats = gp_trace.compare(cpu, gpu)
        assert not problems, (name, problems[:3], stats)
        if name == "hello":
            # its edge 3 meets the tie Brent has by construction (tests/gp_trace.py: a rejected parabolic step becomes the
            # bound, the same parabola is fitted again and p is compared with q * (p / q)): from there on the two runs may
            # part, and end Brent's own tolerance apart (2^-9 relative in the log length and 2^-11)
            assert stats["ties"] == 1 and stats["compared"] >= 35, stats
            assert np.all(np.abs(np.log(bl_gpu[1:]) - np.log(bl_cpu[1:])) <= 4 * (2.0 ** -9 * np.abs(np.log(bl_cpu[1:])) + 2.0 ** -11))
        else:
            # no decision of the checker's run is near a tie on these workloads (the comparisons that choose a point are 1e-4
            # relative from one at the least, the values 7e-7): nothing may be left uncompared
            assert stats["ties"] == 0 and stats["compared"] == len(cpu) == len(gpu), (name, stats)
            assert np.abs(bl_gpu - bl_cpu).max() < 1e-8, name


def _scheduled_and_sequential_sweeps_agree(cases, methods):
    import os

    for (dag, sp), threshold in cases:
        bl0 = np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count)
        results = {}
        for scheduled in ("1", "0"):
            os.environ["BITO_AMD_GP_SCHEDULE"] = scheduled
            try:
                eng = gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count, threshold)
                eng.set_sbn_parameters(dag.uniform_on_topological_support_prior())
                out = []
                for method in methods:
                    eng.set_optimization_method(method)
                    eng.set_branch_lengths(bl0)
                    eng.reset_optimization_count()
                    eng.process_operations(dag.populate_plvs())
                    eng.process_operations(dag.branch_length_optimization())
                    eng.increment_optimization_count()  # (the second sweep skips converged edges: differences in play)
                    eng.process_operations(dag.branch_length_optimization())
                    eng.process_operations(dag.populate_plvs())
                    eng.process_operations(dag.compute_likelihoods())
                    out.append((eng.get_branch_lengths(), eng.get_branch_length_differences(), eng.get_per_gpcsp_log_likelihoods()))
                results[scheduled] = out
            finally:
                os.environ.pop("BITO_AMD_GP_SCHEDULE", None)
        for a, b in zip(results["1"], results["0"]):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
        assert np.abs(results["1"][0][0] - bl0).max() > 1e-3  # (the sweep did move the lengths)


@pytest.mark.gpu
def test_scheduled_sweep_is_bitwise_the_sequential_one():
    """The executor runs a branch-length optimisation sweep in the order of gp_schedule.hpp (the optimisations of equal
    optimiser depth as concurrent workgroups of one launch: 56 launches instead of 118 on the DS1 ten-tree DAG); the same
    arithmetic on the same inputs, so the branch lengths, their changes and the per-GPCSP log-likelihoods are those of
    the one-optimisation-per-launch route in stream order (BITO_AMD_GP_SCHEDULE=0) bit for bit -- Brent, Brent with
    gradients and Newton (reference schedule: src/gp_dag.cpp:78-121; optimisers: src/optimization.hpp:71-417), without
    and with rescaling in play, and on the 970-edge DAG of twenty seeded topologies."""
    from bito_amd import workloads

    _scheduled_and_sequential_sweeps_agree(((workloads.ds1_subsplit_dag(10), 1e-40), (workloads.ds1_subsplit_dag(10), 1e-4),
                                            (workloads.seeded_subsplit_dag(20), 1e-40)),
                                           (gp.BRENT, gp.BRENT_WITH_GRADIENTS, gp.NEWTON))


def test_schedule_entry_point_rejects_streams_it_cannot_index():
    """bito_amd_gp_schedule_operations indexes the side array by the stream's PrepForMarginalization records: a record
    that points outside it is refused (BAD_ARG) before anything is read; an empty stream is fine."""
    import ctypes as C

    from bito_amd import _capi

    L = gp._lib()
    ops = np.zeros(2, dtype=gp.OP_DTYPE)
    ops[0] = (gp.PREP_FOR_MARGINALIZATION, 3, 5, 1, 0)  # three sources from side[1 .. 4): side holds two entries
    ops[1] = (gp.ZERO_PLV, 0, 5, 0, 0)
    side = np.array([7, 8], dtype=np.uint64)
    out = np.zeros(2, dtype=gp.OP_DTYPE)
    count = C.c_int64(-1)
    rc = L.bito_amd_gp_schedule_operations(ops.ctypes.data, 2, side.ctypes.data, 2, 1, out.ctypes.data, None, None, None, C.byref(count))
    assert rc == _capi.ERR_BAD_ARG
    rc = L.bito_amd_gp_schedule_operations(ops.ctypes.data, 2, None, 0, 1, out.ctypes.data, None, None, None, C.byref(count))
    assert rc == _capi.ERR_BAD_ARG
    rc = L.bito_amd_gp_schedule_operations(None, 0, None, 0, 1, None, None, None, None, C.byref(count))
    assert rc == 0 and count.value == 0
    ops[0] = (gp.PREP_FOR_MARGINALIZATION, 2, 5, 0, 0)
    rc = L.bito_amd_gp_schedule_operations(ops.ctypes.data, 2, side.ctypes.data, 2, 1, out.ctypes.data, None, None, None, C.byref(count))
    assert rc == 0 and count.value == 1  # (the ZeroPLV of PLV 5 must follow the Prep that writes its count: two levels, one launch)
    assert [int(o["opcode"]) for o in out] == [gp.PREP_FOR_MARGINALIZATION, gp.ZERO_PLV]
    # a start near 2^64 must not wrap around into the side array (the range is compared without a sum) ...
    big = np.zeros(2, dtype=gp.OP_DTYPE)
    big[0]["opcode"], big[0]["count"], big[0]["a"], big[0]["b"] = gp.PREP_FOR_MARGINALIZATION, 2, 5, 2 ** 64 - 1
    big[1] = ops[1]
    rc = L.bito_amd_gp_schedule_operations(big.ctypes.data, 2, side.ctypes.data, 2, 1, out.ctypes.data, None, None, None, C.byref(count))
    assert rc == _capi.ERR_BAD_ARG
    # ... and an opcode the read / write sets do not know is refused
    big[0] = ops[1]
    big[1]["opcode"] = 77
    rc = L.bito_amd_gp_schedule_operations(big.ctypes.data, 2, side.ctypes.data, 2, 1, out.ctypes.data, None, None, None, C.byref(count))
    assert rc == _capi.ERR_BAD_ARG
