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
    1329 = one store and one load per stored vector without folding."""
    rows = {"d2/1": [], "fold d2/1": [], "d8": [], "stored": [], "stored folded": []}
    for t in range(8):
        ch = _tree(1000, 2 + t)
        n = 1000
        N = n + len(ch)
        order, _ = sim.heavy_first_order(ch, n)
        forder, _ = sim.heavy_first_order(ch, n, fold=True)
        rows["d2/1"].append(sum(sim.kernel_traffic(ch, n, order, 2, 1)))
        rows["fold d2/1"].append(sum(sim.kernel_traffic(ch, n, forder, 2, 1, fold=True)))
        rows["d8"].append(sum(sim.kernel_traffic(ch, n, order, 8)))
        rows["stored"].append(N - n - len(sim.unstored_nodes(ch, n, False)) - 1)
        rows["stored folded"].append(N - n - len(sim.unstored_nodes(ch, n, True)) - 1)
    mean = {k: float(np.mean(v)) for k, v in rows.items()}
    assert round(mean["d2/1"]) == 1478 and round(mean["fold d2/1"]) == 1104 and round(mean["d8"]) == 1329
    assert round(mean["stored"]) == 664 and round(mean["stored folded"]) == 498
    assert 2 * round(mean["stored"]) + 1 >= round(mean["d8"])  # (a store and a load of every stored vector)
