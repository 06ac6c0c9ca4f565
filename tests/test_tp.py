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
