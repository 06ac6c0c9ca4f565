"""Top-pruning ("top tree") likelihoods over a subsplit DAG.

Host-side mirror of the part of the reference's ``TPEngine`` that the likelihood evaluator needs
(src/tp_engine.cpp:421-426,593-694, src/tp_choice_map.cpp:272-321): every DAG edge gets a *tree
source* (the first input tree that contains it), a *choice map* (for each edge: which parent, sister,
left-child and right-child edge the best tree through it uses -- the adjacent edge with the
highest-priority, i.e. lowest, tree source), and from the choice map the *top tree* through the edge.
The score of an edge is the log-likelihood of its top tree under the DAG's branch lengths.

The reference propagates per-edge partial vectors through the choice map with a third copy of its
JC69 primitives (src/tp_evaluation_engine.hpp:373-475).  Here the top trees are materialised as
parent-id vectors and evaluated as ONE batch by the per-tree likelihood engine on the GPU (the same
kernels as ``Engine.log_likelihoods``); edges whose top trees coincide share the evaluation.  What
the reference's own test checks -- the score of an edge equals BEAGLE's likelihood of the edge's top
tree (src/gp_doctest.cpp:2909-2931) -- holds by construction.  No arithmetic happens in this module.

Proposed NNIs (src/tp_engine.cpp:955-1135,1460-1468): the top tree through an NNI that is not in the
DAG yet is the top tree of its best neighbour inside the DAG with the two clades exchanged; its
branches keep the lengths of the edges they came from unless the DAG already holds the new PCSP.
All proposals of a DAG are scored as one batch of trees, like the DAG's own edges.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .gp_dag import SubsplitDAG
from .nni import NNI, adjacent_nnis, contains_nni

NO_EDGE = -1


class TPEngine:
    def __init__(self, dag: SubsplitDAG, parent_id_vectors: Sequence[Sequence[int]]):
        self.dag = dag
        n = dag.taxon_count
        E = dag.gpcsp_count
        # edge list: (parent node or -1 for a rootsplit edge, child node, side of the parent's subsplit)
        self.edge_parent = np.full(E, -1, dtype=np.int64)
        self.edge_child = np.zeros(E, dtype=np.int64)
        self.edge_side = np.zeros(E, dtype=np.int64)
        for (p, c), e in dag.edge_id.items():
            self.edge_parent[e], self.edge_child[e] = p, c
            if p >= 0:
                self.edge_side[e] = 1 if c in dag.children[p][1] else 0
        self._set_tree_source_by_taking_first(parent_id_vectors)
        self._initialize_choice_map()

    # -- TPEngine::SetTreeSourceByTakingFirst (src/tp_engine.cpp:658-694) ------------------------
    def _set_tree_source_by_taking_first(self, parent_id_vectors):
        dag, n = self.dag, self.dag.taxon_count
        tree_id_max = len(parent_id_vectors) + 1
        source = np.full(dag.gpcsp_count, tree_id_max, dtype=np.int64)
        for tree_id, parents in enumerate(parent_id_vectors):
            parents = [int(x) for x in parents]
            node_count = len(parents) + 1
            clade = [1 << i for i in range(n)] + [0] * (node_count - n)
            kids: Dict[int, List[int]] = {}
            for child, p in enumerate(parents):
                clade[p] |= clade[child]
                kids.setdefault(p, []).append(child)

            def dag_node(v):
                if v < n:
                    return v
                a, b = kids[v]
                ca, cb = clade[a], clade[b]
                key = (ca, cb) if (ca & -ca) < (cb & -cb) else (cb, ca)
                return dag.node_id[key]

            for v in range(n, node_count):
                for c in kids[v]:
                    e = dag.edge(dag_node(v), dag_node(c))
                    if source[e] == tree_id_max:
                        source[e] = tree_id + 1
        # rootsplit edges take the best source of the edges below their node
        for r in dag.rootsplits:
            best = tree_id_max
            for side in (1, 0):
                for c in dag.children[r][side]:
                    best = min(best, source[dag.edge(r, c)])
            source[dag.rootsplit_edge(r)] = best
        self.tree_source = source

    # -- TPEngine::UpdateEdgeChoiceByTakingHighestPriorityTree (src/tp_engine.cpp:593-654) -------
    def _best(self, candidates: List[int]) -> Tuple[int, int]:
        best_edge, best_tree = NO_EDGE, None
        for e in candidates:  # first edge wins ties, as the reference's strict '>' comparison
            if best_tree is None or best_tree > self.tree_source[e]:
                best_edge, best_tree = e, self.tree_source[e]
        return best_edge, (best_tree if best_tree is not None else np.iinfo(np.int64).max)

    def _initialize_choice_map(self):
        dag, E = self.dag, self.dag.gpcsp_count
        self.choice_parent = np.full(E, NO_EDGE, dtype=np.int64)
        self.choice_sister = np.full(E, NO_EDGE, dtype=np.int64)
        self.choice_left = np.full(E, NO_EDGE, dtype=np.int64)
        self.choice_right = np.full(E, NO_EDGE, dtype=np.int64)
        for e in range(E):
            p, c = int(self.edge_parent[e]), int(self.edge_child[e])
            if p >= 0:
                # parent edge: the edges above the parent node, left-clade parents first
                above = {1: [], 0: []}
                for (gp, side) in dag.parents[p]:
                    above[side].append(dag.edge(gp, p))
                if p in dag.rootsplits:
                    above[1].append(dag.rootsplit_edge(p))
                best_edge, best_tree = NO_EDGE, None
                for side in (1, 0):
                    ce, ct = self._best(above[side])
                    if ce != NO_EDGE and (best_edge == NO_EDGE or best_tree > ct):
                        best_edge, best_tree = ce, ct
                self.choice_parent[e] = best_edge
                side = int(self.edge_side[e])
                self.choice_sister[e] = self._best([dag.edge(p, s) for s in dag.children[p][1 - side]])[0]
            if c >= dag.taxon_count:
                self.choice_left[e] = self._best([dag.edge(c, k) for k in dag.children[c][1]])[0]
                self.choice_right[e] = self._best([dag.edge(c, k) for k in dag.children[c][0]])[0]

    # -- TPChoiceMap::ExtractTreeMask (src/tp_choice_map.cpp:272-321) ----------------------------
    def top_tree_edges(self, edge: int) -> List[int]:
        mask, stack = [], []
        for k in (self.choice_left[edge], self.choice_right[edge]):
            if k != NO_EDGE:
                stack.append(int(k))
        focal = int(edge)
        while True:
            mask.append(focal)
            if self.edge_parent[focal] < 0:
                break
            stack.append(int(self.choice_sister[focal]))
            focal = int(self.choice_parent[focal])
        while stack:
            e = stack.pop()
            mask.append(e)
            for k in (self.choice_left[e], self.choice_right[e]):
                if k != NO_EDGE:
                    stack.append(int(k))
        return sorted(set(mask))

    def top_tree(self, edge: int) -> Tuple[np.ndarray, List[int]]:
        """(parent_ids, edge_of_node) of the top tree through ``edge``: a bito parent-id vector (leaves =
        taxon ids, internal ids in post-order, root last) and the DAG edge above every tree node (the last
        entry is the rootsplit edge)."""
        dag, n = self.dag, self.dag.taxon_count
        mask = self.top_tree_edges(edge)
        below: Dict[int, Dict[int, int]] = {}
        root_edge = None
        for e in mask:
            p = int(self.edge_parent[e])
            if p < 0:
                root_edge = e
            else:
                below.setdefault(p, {})[int(self.edge_side[e])] = e
        if root_edge is None:
            raise RuntimeError("top tree has no rootsplit edge")
        parents: Dict[int, int] = {}
        edge_above: Dict[int, int] = {}
        counter = [n]

        def walk(e):
            node = int(self.edge_child[e])
            if node < n:
                me = node
            else:
                kids = below.get(node)
                if kids is None or len(kids) != 2:
                    raise RuntimeError("tree mask is not a tree")
                a, b = walk(kids[1]), walk(kids[0])
                me = counter[0]
                counter[0] += 1
                parents[a] = parents[b] = me
            edge_above[me] = e
            return me

        root = walk(root_edge)
        if root != 2 * n - 2:
            raise RuntimeError("tree mask does not span all taxa")
        pid = np.array([parents[v] for v in range(2 * n - 2)], dtype=np.int32)
        return pid, [edge_above[v] for v in range(2 * n - 1)]

    # -- scores ------------------------------------------------------------------------------------
    def top_tree_likelihoods(self, engine, edge_branch_lengths: np.ndarray, params: Optional[np.ndarray] = None):
        """TPEngine::GetTopTreeLikelihoods: per DAG edge, the log-likelihood of its top tree with the DAG's
        branch lengths (one length per edge; rootsplit edges carry none).  ``engine`` is a per-tree likelihood
        engine (``bito_amd.Engine``) built on the same site patterns; all distinct top trees are evaluated
        in one batch."""
        dag, n = self.dag, self.dag.taxon_count
        keys: Dict[bytes, int] = {}
        pids, bls, which = [], [], []
        for e in range(dag.gpcsp_count):
            pid, edge_of_node = self.top_tree(e)
            key = pid.tobytes() + np.asarray(edge_of_node, dtype=np.int64).tobytes()
            idx = keys.get(key)
            if idx is None:
                idx = keys[key] = len(pids)
                bl = np.zeros(2 * n - 1)
                bl[: 2 * n - 2] = [edge_branch_lengths[x] for x in edge_of_node[: 2 * n - 2]]
                pids.append(pid)
                bls.append(bl)
            which.append(idx)
        ll = engine.log_likelihoods(np.stack(pids), np.stack(bls), params)
        return ll[np.asarray(which)]

    # -- TPEngine::SetBranchLengthsByTakingFirst (src/tp_engine.cpp:1398-1421) ---------------------------
    def branch_lengths_by_taking_first(self, parent_id_vectors, tree_branch_lengths, default: float = 0.1) -> np.ndarray:
        """Per DAG edge the length of that branch in the first input tree that has it (branch lengths are
        indexed by child node id, as everywhere at the boundary); rootsplit edges keep ``default``."""
        dag, n = self.dag, self.dag.taxon_count
        out = np.full(dag.gpcsp_count, float(default))
        seen = np.zeros(dag.gpcsp_count, dtype=bool)
        for parents, lengths in zip(parent_id_vectors, tree_branch_lengths):
            parents = [int(x) for x in parents]
            node_count = len(parents) + 1
            clade = [1 << i for i in range(n)] + [0] * (node_count - n)
            kids: Dict[int, List[int]] = {}
            for child, p in enumerate(parents):
                clade[p] |= clade[child]
                kids.setdefault(p, []).append(child)
            node = list(range(n)) + [0] * (node_count - n)
            for v in range(n, node_count):
                ca, cb = (clade[k] for k in kids[v])
                node[v] = dag.node_id[(ca, cb) if (ca & -ca) < (cb & -cb) else (cb, ca)]
            for child, p in enumerate(parents):
                e = dag.edge(node[p], node[child])
                if not seen[e]:
                    seen[e], out[e] = True, float(lengths[child])
        return out

    # -- proposed NNIs ---------------------------------------------------------------------------------
    def find_highest_priority_neighbor_nni(self, nni: NNI) -> NNI:
        """TPEngine::FindHighestPriorityNeighborNNIInDAG: of the NNI's neighbours inside the DAG, the one whose
        central edge has the best (lowest) tree source; the first wins ties."""
        best, best_source = None, None
        for nb in nni.neighbors():
            if contains_nni(self.dag, nb):
                source = self.tree_source[self.dag.edge(self.dag.node_id[nb.parent], self.dag.node_id[nb.child])]
                if best is None or source < best_source:
                    best, best_source = nb, source
        if best is None:
            raise ValueError("NNIOperation has no neighbors found in the DAG.")
        return best

    def proposed_nni_top_tree(self, nni: NNI, pre_nni: Optional[NNI] = None) -> Tuple[np.ndarray, List[int]]:
        """(parent_ids, edge_of_node) of the top tree through a proposed NNI: the top tree of ``pre_nni``'s
        central edge with the sister clade and one child clade exchanged (the choice-map re-mapping of
        GetRemappedEdgeChoiceFromPreNNIToPostNNI, src/tp_engine.cpp:963-989).  ``edge_of_node`` names, for
        every tree node, the DAG edge whose branch length the branch above it takes: the DAG's own edge
        where the new PCSP exists already, else the pre-NNI's edge it came from
        (BuildMapOfProposedNNIPCSPsToBestPreNNIEdges, src/tp_engine.cpp:1064-1132)."""
        dag, n = self.dag, self.dag.taxon_count
        pre = pre_nni or self.find_highest_priority_neighbor_nni(nni)
        e0 = dag.edge(dag.node_id[pre.parent], dag.node_id[pre.child])
        pid, edge_of = self.top_tree(e0)
        count = 2 * n - 1
        kids: Dict[int, List[int]] = {}
        clade = [1 << i for i in range(n)] + [0] * (count - n)
        for c, p in enumerate(pid):
            kids.setdefault(int(p), []).append(c)
            clade[int(p)] |= clade[c]
        child_node = edge_of.index(e0)
        parent_node = int(pid[child_node])
        sister_node, = [k for k in kids[parent_node] if k != child_node]
        swap_node, = [k for k in kids[child_node] if clade[k] == nni.sister_clade]
        keep_node, = [k for k in kids[child_node] if k != swap_node]
        # exchange the clades
        new_parent_of = [int(x) for x in pid] + [-1]
        new_parent_of[swap_node], new_parent_of[sister_node] = parent_node, child_node

        def own_or(parent_subsplit, node, fallback):
            """The DAG's own edge for (parent_subsplit -> subsplit of ``node``) if it exists."""
            a, b = sorted_children[node] if node >= n else (0, clade[node])
            key = (a, b)
            if dag.contains_edge(parent_subsplit, key):
                return dag.edge(dag.node_id[parent_subsplit], dag.node_id[key])
            return fallback

        sorted_children = {}
        for v, ks in kids.items():
            ca, cb = clade[ks[0]], clade[ks[1]]
            sorted_children[v] = (ca, cb) if (ca & -ca) < (cb & -cb) else (cb, ca)
        edges = list(edge_of)
        edges[sister_node] = own_or(nni.child, sister_node, edge_of[sister_node])
        edges[keep_node] = own_or(nni.child, keep_node, edge_of[keep_node])
        edges[swap_node] = own_or(nni.parent, swap_node, edge_of[swap_node])
        if parent_node != count - 1:  # the branch above the parent: (grandparent -> new parent) may exist already
            grand = int(pid[parent_node])
            if dag.contains_edge(sorted_children[grand], nni.parent):
                edges[parent_node] = dag.edge(dag.node_id[sorted_children[grand]], dag.node_id[nni.parent])
        # bito ids: internal nodes in post-order, root last
        new_kids: Dict[int, List[int]] = {}
        for c, p in enumerate(new_parent_of[:-1]):
            new_kids.setdefault(p, []).append(c)
        low = list(range(n)) + [0] * (count - n)
        order: List[int] = []

        def visit(v):
            if v >= n:
                ks = new_kids[v]
                for k in ks:
                    visit(k)
                low[v] = min(low[k] for k in ks)
                order.append(v)

        # children in increasing order of their smallest taxon (left clade first), as top_tree emits them
        def sort_kids(v):
            if v >= n:
                for k in new_kids[v]:
                    sort_kids(k)
                low[v] = min(low[k] for k in new_kids[v])
                new_kids[v].sort(key=lambda k: low[k])

        sort_kids(count - 1)
        visit(count - 1)
        new_id = {v: v for v in range(n)}
        for i, v in enumerate(order):
            new_id[v] = n + i
        out = np.zeros(count - 1, dtype=np.int32)
        out_edges = [0] * count
        for v in range(count):
            out_edges[new_id[v]] = edges[v]
            if v != count - 1:
                out[new_id[v]] = new_id[new_parent_of[v]]
        return out, out_edges

    def proposed_nni_likelihoods(self, engine, edge_branch_lengths: np.ndarray, nnis: Optional[Sequence[NNI]] = None,
                                 params: Optional[np.ndarray] = None) -> Dict[NNI, float]:
        """TPEngine::GetTopTreeScoreWithProposedNNI for every NNI adjacent to the DAG (or the given ones): the
        log-likelihoods of the proposals' top trees, all in ONE batch of the per-tree engine."""
        n = self.dag.taxon_count
        nnis = list(adjacent_nnis(self.dag) if nnis is None else nnis)
        if not nnis:
            return {}
        pids, bls = [], []
        for x in nnis:
            pid, edges = self.proposed_nni_top_tree(x)
            bl = np.zeros(2 * n - 1)
            bl[: 2 * n - 2] = [edge_branch_lengths[e] for e in edges[: 2 * n - 2]]
            pids.append(pid)
            bls.append(bl)
        ll = engine.log_likelihoods(np.stack(pids), np.stack(bls), params)
        return {x: float(v) for x, v in zip(nnis, ll)}
