"""scripts/sim_hbm_traffic.py restates, on the CPU, which vectors walk_hbm_cat_kernel moves through its HBM arena
(bito_amd/csrc/walk_hbm_cat.hip: the visiting order of hbm_order_kernel, the two thread-private columns, cherries and
pitchforks rebuilt instead of stored).  DESIGN.md and README.md quote its counts; this holds the restatement to the
properties the kernel relies on and the quoted counts to the script."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import sim_hbm_traffic as sim  # noqa: E402

from bito_amd import workloads  # noqa: E402


def _tree(n, seed):
    t = workloads.random_unrooted_tree(n, np.random.default_rng(seed), 0.1)
    return sim.children_of(np.asarray(t.parent_ids), n)


def test_unstored_nodes_are_cherries_and_pitchforks_never_the_root():
    for n, seed in ((9, 1), (64, 2), (333, 3)):
        ch = _tree(n, seed)
        N = n + len(ch)
        cherries = sim.unstored_nodes(ch, n, False)
        both = sim.unstored_nodes(ch, n, True)
        assert cherries <= both and N - 1 not in both
        for v in cherries:
            assert ch[v - n, 0] < n and ch[v - n, 1] < n
        for v in both - cherries:  # a tip and a cherry under one node
            a, b = ch[v - n]
            assert (a < n and b in cherries) or (b < n and a in cherries)


def test_heavier_subtree_first_order_is_a_post_order_of_the_stored_nodes():
    for fold in (False, True):
        ch = _tree(200, 7)
        n, N = 200, 200 + len(ch)
        order, need = sim.heavy_first_order(ch, n, fold=fold)
        unstored = sim.unstored_nodes(ch, n, fold)
        steps = [v for v in order if v not in unstored]
        assert sorted(order) == list(range(n, N)) and steps[-1] == N - 1
        seen = set()
        for v in steps:  # children that are steps come first
            for c in ch[v - n]:
                assert c < n or c in unstored or c in seen
            seen.add(v)
        assert need.max() <= int(np.ceil(np.log2(n))) + 1  # (Sethi-Ullman: logarithmic in the tree's size)


def test_transfer_counts_quoted_in_the_documents():
    """config 4's trees (1000 taxa, seeds 2..9 as the script draws them): 1478 vector transfers per tree with two columns
    in the post-order pass and one pending vector in the pre-order pass, 1104 with the pitchforks folded (round 4),
    1329 = one store and one load per stored vector without folding; 888 with the four-tip subtrees folded too (round 6)."""
    rows = {"d2/1": [], "fold d2/1": [], "fold2 d2/1": [], "d8": [], "stored": [], "stored folded": [], "stored fold 2": []}
    for t in range(8):
        ch = _tree(1000, 2 + t)
        n = 1000
        N = n + len(ch)
        order, _ = sim.heavy_first_order(ch, n)
        forder, _ = sim.heavy_first_order(ch, n, fold=True)
        rows["d2/1"].append(sum(sim.kernel_traffic(ch, n, order, 2, 1)))
        rows["fold d2/1"].append(sum(sim.kernel_traffic(ch, n, forder, 2, 1, fold=True)))
        f2order, _ = sim.heavy_first_order(ch, n, fold=2)
        rows["fold2 d2/1"].append(sum(sim.kernel_traffic(ch, n, f2order, 2, 1, fold=2)))
        rows["stored fold 2"].append(N - n - len(sim.unstored_nodes(ch, n, 2)) - 1)
        rows["d8"].append(sum(sim.kernel_traffic(ch, n, order, 8)))
        rows["stored"].append(N - n - len(sim.unstored_nodes(ch, n, False)) - 1)
        rows["stored folded"].append(N - n - len(sim.unstored_nodes(ch, n, True)) - 1)
    mean = {k: float(np.mean(v)) for k, v in rows.items()}
    assert round(mean["d2/1"]) == 1478 and round(mean["fold d2/1"]) == 1104 and round(mean["d8"]) == 1329
    assert round(mean["stored"]) == 664 and round(mean["stored folded"]) == 498
    # round 6: caterpillars (a tip and a pitchfork under one node) and twin cherries rebuilt as well
    assert round(mean["fold2 d2/1"]) == 888 and round(mean["stored fold 2"]) == 402
    assert 2 * round(mean["stored"]) + 1 >= round(mean["d8"])  # (a store and a load of every stored vector)


def test_four_tip_shapes_are_what_the_kernel_folds():
    """fold level 2: caterpillars (a tip and a pitchfork under one node) and twin cherries, never the root, and every
    one of them a subtree of exactly four tips"""
    for n, seed in ((12, 5), (64, 2), (333, 3)):
        ch = _tree(n, seed)
        N = n + len(ch)
        cherry, fork, cat, twin = sim.shapes(ch, n)
        four = cat | twin
        second = {max(ch[v - n]) for v in range(n, N) if ch[v - n, 0] in four and ch[v - n, 1] in four}  # (one per step)
        assert sim.unstored_nodes(ch, n, 2) == (cherry | fork | four) - second and N - 1 not in four
        assert sim.unstored_nodes(ch, n, 1) == cherry | fork == sim.unstored_nodes(ch, n, True)

        def tips(v):
            return 1 if v < n else tips(ch[v - n, 0]) + tips(ch[v - n, 1])

        for v in cat | twin:
            assert tips(v) == 4
        for v in range(n, N - 1):  # ... and every other four-tip subtree is one of the two
            if tips(v) == 4:
                assert v in cat or v in twin
