"""Top-pruning mirror (bito_amd/tp.py): tree sources, choice map and top trees on
the host, and -- on the GPU -- the reference's own check that an edge's top-tree score is the
likelihood of that tree (src/gp_doctest.cpp:2876-2931)."""
import os

import numpy as np
import pytest

import bito_amd
from bito_amd import treeio
from bito_amd.gp_dag import SubsplitDAG
from bito_amd.site_pattern import SitePattern
from bito_amd.tp import TPEngine


def _load(data_dir, fasta, newick):
    tc = treeio.read_newick_file(os.path.join(data_dir, newick))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, fasta)), tc.taxon_names)
    pids = [t.parent_ids for t in tc.trees]
    dag = SubsplitDAG(len(tc.taxon_names), pids)
    return tc, sp, pids, dag


@pytest.mark.parametrize("newick", ["six_taxon_rooted_simple.nwk", "six_taxon_rooted_single.nwk",
                                    "five_taxon_rooted.nwk", "hello_rooted_two_trees.nwk"])
def test_top_trees_exist_in_the_dag(data_dir, newick):
    """"TPEngine: Initialize TPEngine and ChoiceMap" (src/gp_doctest.cpp:2876-2905): the top tree through
    every edge is one of the DAG's trees -- and contains the edge."""
    fasta = {"six": "six_taxon.fasta", "fiv": "five_taxon.fasta", "hel": "hello.fasta"}[newick[:3]]
    tc, sp, pids, dag = _load(data_dir, fasta, newick)
    tp = TPEngine(dag, pids)
    spanned = {pid.tobytes() for pid, _ in dag.all_trees()}
    inputs = [np.asarray(p, dtype=np.int32).tobytes() for p in pids]
    assert tp.tree_source.min() == 1 and tp.tree_source.max() <= len(pids)
    for e in range(dag.gpcsp_count):
        pid, edge_of_node = tp.top_tree(e)
        assert pid.tobytes() in spanned
        assert e in edge_of_node
        assert len(set(edge_of_node)) == 2 * dag.taxon_count - 1
        # an edge first seen in input tree k has a top tree no later than k in priority: for edges of the
        # first tree the top tree IS the first tree (every adjacent choice has source 1)
        if tp.tree_source[e] == 1:
            assert pid.tobytes() == inputs[0]


def test_tree_source_takes_the_first_tree(data_dir):
    """TPEngine::SetTreeSourceByTakingFirst (src/tp_engine.cpp:658-694) on two six-taxon trees that share
    part of their subsplits."""
    tc, sp, pids, dag = _load(data_dir, "six_taxon.fasta", "six_taxon_rooted_simple.nwk")
    tp = TPEngine(dag, pids)
    first_only = TPEngine(SubsplitDAG(6, pids[:1]), pids[:1])
    assert (first_only.tree_source == 1).all()
    assert sorted(set(tp.tree_source.tolist())) == [1, 2]
    assert (tp.tree_source == 1).sum() == first_only.dag.gpcsp_count  # every edge of tree 1, and only those
    # choice map: a leaf edge has no children, a rootsplit edge no parent or sister
    for e in range(dag.gpcsp_count):
        leaf = tp.edge_child[e] < 6
        assert (tp.choice_left[e] == -1) == leaf and (tp.choice_right[e] == -1) == leaf
        root = tp.edge_parent[e] < 0
        assert (tp.choice_parent[e] == -1) == root and (tp.choice_sister[e] == -1) == root


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["six", "ds1"])
def test_top_tree_likelihoods_equal_tree_likelihoods(data_dir, case):
    """"TPEngine Likelihood scores vs BEAGLE Likelihood scores" (src/gp_doctest.cpp:2909-2931): per edge,
    the top-tree score equals the likelihood of the edge's top tree under the DAG's branch lengths --
    here against the CPU oracle, tree by tree."""
    from oracle import oracle

    if case == "six":
        tc, sp, pids, dag = _load(data_dir, "six_taxon.fasta", "six_taxon_rooted_simple.nwk")
    else:
        tc = treeio.read_nexus_file(os.path.join(data_dir, "DS1.subsampled_10.t"))
        sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, "DS1.fasta")), tc.taxon_names)
        # the golden trees are unrooted (trifurcating root): root them on their first root child
        pids = []
        for t in tc.trees:
            p = np.asarray(t.parent_ids).copy()
            M = len(p) + 1
            kids = [c for c in range(M - 1) if p[c] == M - 1]
            q = np.append(p, M)  # old root id M-1 now has parent M (the new root)
            q[kids[0]] = M
            pids.append(_renumber(q))
        dag = SubsplitDAG(len(tc.taxon_names), pids)
    tp = TPEngine(dag, pids)
    bl = np.random.default_rng(11).uniform(0.01, 0.3, dag.gpcsp_count)
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification("JC69", "constant", "none"), sp.patterns, sp.weights)
    scores = tp.top_tree_likelihoods(eng, bl)
    cpu = oracle.OracleEngine("JC69", "constant", "none", sp.patterns, sp.weights, 4)
    n = dag.taxon_count
    for e in range(0, dag.gpcsp_count, 1 if case == "six" else 7):
        pid, edge_of_node = tp.top_tree(e)
        tree_bl = np.zeros(2 * n - 1)
        tree_bl[: 2 * n - 2] = bl[edge_of_node[: 2 * n - 2]]
        ref = cpu.log_likelihoods(pid[None, :], tree_bl[None, :])[0]
        assert abs(scores[e] - ref) < 1e-10 + 2e-14 * abs(ref)
    assert np.all(np.isfinite(scores))


def _renumber(parents):
    """parent vector with arbitrary internal ids (root = the largest id) -> bito's post-order internal ids."""
    parents = [int(x) for x in parents]
    count = len(parents) + 1
    n = (count + 1) // 2
    kids = {}
    for c, p in enumerate(parents):
        kids.setdefault(p, []).append(c)
    new_id, order = {}, [n]

    def walk(v):
        if v < n:
            new_id[v] = v
            return
        for c in sorted(kids[v]):
            walk(c)
        new_id[v] = order[0]
        order[0] += 1

    walk(count - 1)
    out = [0] * (count - 1)
    for c, p in enumerate(parents):
        out[new_id[c]] = new_id[p]
    return np.array(out, dtype=np.int32)


# -- proposed NNIs: "TPEngine: Proposed NNI vs DAG NNI vs BEAGLE Likelihood" (src/gp_doctest.cpp:2973-3099) ----

PROPOSED_CASES = [("hello.fasta", "hello_rooted_diff_branches.nwk"),
                  ("five_taxon.fasta", "five_taxon_trees_3_4_diff_branches.nwk"),
                  ("six_taxon.fasta", "six_taxon_rooted_simple.nwk")]


def _subsplits_of(pid, n):
    clade = [1 << i for i in range(n)] + [0] * (len(pid) + 1 - n)
    kids = {}
    for c, p in enumerate(pid):
        clade[int(p)] |= clade[c]
        kids.setdefault(int(p), []).append(c)
    out = set()
    for v, (a, b) in kids.items():
        ca, cb = clade[a], clade[b]
        out.add((ca, cb) if (ca & -ca) < (cb & -cb) else (cb, ca))
    return out


@pytest.mark.parametrize("fasta,newick", PROPOSED_CASES)
def test_proposed_nni_top_tree_is_the_neighbours_top_tree_with_the_clades_exchanged(data_dir, fasta, newick):
    from bito_amd.nni import adjacent_nnis

    tc, sp, pids, dag = _load(data_dir, fasta, newick)
    n = dag.taxon_count
    tp = TPEngine(dag, pids)
    nnis = adjacent_nnis(dag)
    assert nnis
    for x in nnis:
        pre = tp.find_highest_priority_neighbor_nni(x)
        pid, edges = tp.proposed_nni_top_tree(x)
        before, _ = tp.top_tree(dag.edge(dag.node_id[pre.parent], dag.node_id[pre.child]))
        a, b = _subsplits_of(before, n), _subsplits_of(pid, n)
        assert a - b == {pre.parent, pre.child} - {x.parent} and b - a == {x.parent, x.child} - {pre.parent}
        assert len(edges) == 2 * n - 1 and all(0 <= e < dag.gpcsp_count for e in edges)
        # ids are bito's: every child below its parent, root last
        assert all(pid[c] > c for c in range(2 * n - 2)) and pid.max() == 2 * n - 2
        # "score_dag == score_proposed" in structure terms: once the proposed tree is part of the collection, the
        # top tree through the NNI's edge in the grown DAG is that very tree
        grown_pids = list(pids) + [pid]
        grown = SubsplitDAG(n, grown_pids)
        tp2 = TPEngine(grown, grown_pids)
        again, _ = tp2.top_tree(grown.edge(grown.node_id[x.parent], grown.node_id[x.child]))
        assert np.array_equal(again, pid)


def test_branch_lengths_by_taking_first(data_dir):
    """TPEngineSetBranchLengthsByTakingFirst: the two five-taxon trees carry lengths 1.x and 2.x."""
    tc, sp, pids, dag = _load(data_dir, "five_taxon.fasta", "five_taxon_trees_3_4_diff_branches.nwk")
    tp = TPEngine(dag, pids)
    bl = tp.branch_lengths_by_taking_first(pids, [t.branch_lengths for t in tc.trees])
    for e in range(dag.gpcsp_count):
        if tp.edge_parent[e] < 0:
            continue
        assert int(bl[e]) == tp.tree_source[e], (e, bl[e], tp.tree_source[e])  # 1.x from tree 1, 2.x from tree 2
    # the pendant branch of x0 is shared: tree 1 wins
    x0 = [e for e in range(dag.gpcsp_count) if tp.edge_child[e] == 0]
    assert all(abs(bl[e] - 1.1) < 1e-12 for e in x0 if tp.tree_source[e] == 1)


@pytest.mark.gpu
@pytest.mark.parametrize("fasta,newick", PROPOSED_CASES)
def test_proposed_nni_scores_equal_tree_likelihoods(data_dir, fasta, newick):
    """score_proposed == BEAGLE's likelihood of the proposal's top tree (the reference asks 1e-5), here the
    per-tree CPU oracle on the same tree and lengths, with the branch lengths taken from the first tree."""
    from oracle import oracle

    tc, sp, pids, dag = _load(data_dir, fasta, newick)
    n = dag.taxon_count
    tp = TPEngine(dag, pids)
    bl = tp.branch_lengths_by_taking_first(pids, [t.branch_lengths for t in tc.trees])
    if newick.startswith("six"):  # this fixture carries no branch lengths
        bl = np.random.default_rng(2).uniform(0.02, 0.3, dag.gpcsp_count)
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification("JC69", "constant", "none"), sp.patterns, sp.weights)
    scores = tp.proposed_nni_likelihoods(eng, bl)
    assert scores
    cpu = oracle.OracleEngine("JC69", "constant", "none", sp.patterns, sp.weights, 2)
    own = tp.top_tree_likelihoods(eng, bl)
    for x, score in scores.items():
        pid, edges = tp.proposed_nni_top_tree(x)
        tree_bl = np.zeros(2 * n - 1)
        tree_bl[: 2 * n - 2] = bl[edges[: 2 * n - 2]]
        ref = cpu.log_likelihoods(pid[None, :], tree_bl[None, :])[0]
        assert abs(score - ref) < 1e-10
        # a proposal differs from its neighbour's top tree by one NNI: same taxa, same lengths, another topology
        pre = tp.find_highest_priority_neighbor_nni(x)
        assert score != own[dag.edge(dag.node_id[pre.parent], dag.node_id[pre.child])]
