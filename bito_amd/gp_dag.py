"""Subsplit DAG of a set of rooted trees and the GPDAG schedules over it.

Host-side mirror of what the generalized-pruning executor consumes: the reference builds a
``SubsplitDAG`` from the loaded topologies (src/subsplit_dag.cpp:16-60,1085-1140) and ``GPDAG``
turns traversals of it into ``GPOperation`` streams (src/gp_dag.cpp:177-411).  No arithmetic
happens here -- the streams are executed on the GPU through ``bito_amd.gp.GPEngine``.

Conventions (chosen here; what the executor sees is only ids):
  * a clade is an int bit mask, bit t = taxon t; a subsplit is (left, right) with the left
    ("rotated") clade the one holding the smaller taxon id (Bitset::SubsplitFromUnorderedClades,
    reference src/bitset.cpp:268-272,326-331);
  * DAG node ids: leaves = taxon ids, then internal subsplits by increasing clade size, so every
    child id is below its parents' ids; the DAG root ("universal ancestor") has no id here;
  * edge (GPCSP) ids: rootsplit edges first, then the edges below each (parent, clade) as one
    contiguous range (what UpdateSBNProbabilities{start, stop} needs, src/gp_engine.cpp:297-321);
  * PV id = type * node_count + node (src/pv_handler.hpp:487-490).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from .gp import (INCREMENT_MARGINAL_LIKELIHOOD, INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, LIKELIHOOD, MULTIPLY,
                 OPTIMIZE_BRANCH_LENGTH, P, PHAT_LEFT, PHAT_RIGHT, RESET_MARGINAL_LIKELIHOOD, RHAT, R_LEFT, R_RIGHT, SET_TO_STATIONARY,
                 UPDATE_SBN_PROBABILITIES, ZERO_PLV, OpStream)


def _low_bit(mask: int) -> int:
    return mask & -mask


class SubsplitDAG:
    def __init__(self, taxon_count: int, parent_id_vectors: Sequence[Sequence[int]] = (), subsplits=None, pcsps=None):
        n = taxon_count
        self.taxon_count = n
        subsplits = set(subsplits or ())
        pcsps = set(pcsps or ())  # (parent subsplit, is_left, child subsplit-or-leaf mask)
        for parents in parent_id_vectors:
            parents = [int(x) for x in parents]
            node_count = len(parents) + 1
            if node_count != 2 * n - 1:
                raise ValueError("every tree must be a rooted bifurcating tree on the same taxa")
            clade = [1 << i for i in range(n)] + [0] * (node_count - n)
            kids: Dict[int, List[int]] = {}
            for child, p in enumerate(parents):  # post-order ids: children before parents
                clade[p] |= clade[child]
                kids.setdefault(p, []).append(child)
            key = {}
            for v in range(node_count):
                if v < n:
                    key[v] = (0, clade[v])
                else:
                    a, b = kids[v]
                    ca, cb = clade[a], clade[b]
                    key[v] = (ca, cb) if _low_bit(ca) < _low_bit(cb) else (cb, ca)
                    subsplits.add(key[v])
            for v in range(n, node_count):
                for c in kids[v]:
                    pcsps.add((key[v], clade[c] == key[v][0], key[c]))
        if not subsplits:
            raise ValueError("a subsplit DAG needs at least one tree or one subsplit")
        self._subsplit_set, self._pcsp_set = frozenset(subsplits), frozenset(pcsps)
        full = (1 << n) - 1
        internal = sorted(subsplits, key=lambda s: (bin(s[0] | s[1]).count("1"), s))
        self.subsplits: List[Tuple[int, int]] = [(0, 1 << i) for i in range(n)] + internal
        self.node_count = len(self.subsplits)  # without the DAG root
        self.node_id = {s: i for i, s in enumerate(self.subsplits)}
        self.rootsplits = [i for i, s in enumerate(self.subsplits) if (s[0] | s[1]) == full and i >= n]
        # children[node][side] with side 1 = left clade, 0 = right clade; parents[node] = [(parent, side)]
        self.children: List[List[List[int]]] = [[[], []] for _ in range(self.node_count)]
        self.parents: List[List[Tuple[int, int]]] = [[] for _ in range(self.node_count)]
        for ps, is_left, cs in sorted(pcsps):
            p, c = self.node_id[ps], self.node_id[cs]
            self.children[p][1 if is_left else 0].append(c)
            self.parents[c].append((p, 1 if is_left else 0))
        # edge ids
        self.edge_id: Dict[Tuple[int, int], int] = {}
        self.sibling_ranges: List[Tuple[int, int]] = [(0, len(self.rootsplits))]
        for r in self.rootsplits:
            self.edge_id[(-1, r)] = len(self.edge_id)
        for node in range(n, self.node_count):
            for side in (1, 0):
                start = len(self.edge_id)
                for c in self.children[node][side]:
                    self.edge_id[(node, c)] = len(self.edge_id)
                self.sibling_ranges.append((start, len(self.edge_id)))
        self.gpcsp_count = len(self.edge_id)
        # number of topologies below every node (SubsplitDAG::topology_count_below_)
        self.topology_count_below = [1.0] * self.node_count
        for node in range(n, self.node_count):
            self.topology_count_below[node] = (sum(self.topology_count_below[c] for c in self.children[node][1]) *
                                               sum(self.topology_count_below[c] for c in self.children[node][0]))
        self.topology_count = sum(self.topology_count_below[r] for r in self.rootsplits)

    # -- growing the DAG (src/subsplit_dag.cpp:1902-2085): a new DAG object, ids re-derived -----
    def contains_node(self, subsplit) -> bool:
        return tuple(subsplit) in self.node_id

    def contains_edge(self, parent_subsplit, child_subsplit) -> bool:
        p, c = self.node_id.get(tuple(parent_subsplit)), self.node_id.get(tuple(child_subsplit))
        return p is not None and c is not None and (p, c) in self.edge_id

    def _compatible_pcsps(self, subsplits, only=None):
        """Every (parent, side, child) with the child's taxon set equal to the parent's side clade;
        with ``only``, just the pairs that involve one of those subsplits."""
        n = self.taxon_count
        by_clade: Dict[int, List[Tuple[int, int]]] = {}
        for s in list(subsplits) + [(0, 1 << i) for i in range(n)]:
            by_clade.setdefault(s[0] | s[1], []).append(s)
        out = set()
        for ps in subsplits:
            for is_left, clade in ((True, ps[0]), (False, ps[1])):
                for cs in by_clade.get(clade, ()):
                    if only is None or ps in only or cs in only:
                        out.add((ps, is_left, cs))
        return out

    def fully_connected(self) -> "SubsplitDAG":
        """SubsplitDAG::FullyConnect: every compatible parent/child pair of existing nodes gets its edge."""
        return SubsplitDAG(self.taxon_count, subsplits=self._subsplit_set,
                           pcsps=self._pcsp_set | self._compatible_pcsps(self._subsplit_set))

    def with_node_pair(self, parent_subsplit, child_subsplit) -> "SubsplitDAG":
        """SubsplitDAG::AddNodePair (src/subsplit_dag.cpp:1965-2085): nodes that are new are connected to
        all their compatible parents and children; the pair's own edge is added."""
        parent_subsplit, child_subsplit = tuple(parent_subsplit), tuple(child_subsplit)
        focal = child_subsplit[0] | child_subsplit[1]
        if focal not in parent_subsplit:
            raise ValueError("the child subsplit does not descend from the parent subsplit")
        fresh = {x for x in (parent_subsplit, child_subsplit) if x not in self._subsplit_set}
        subsplits = self._subsplit_set | fresh
        pcsps = set(self._pcsp_set) | {(parent_subsplit, parent_subsplit[0] == focal, child_subsplit)}
        if fresh:
            pcsps |= self._compatible_pcsps(subsplits, only=fresh)
        return SubsplitDAG(self.taxon_count, subsplits=subsplits, pcsps=pcsps)

    def reindexers_to(self, grown: "SubsplitDAG"):
        """(node_reindexer, gpcsp_reindexer) from this DAG's ids to those of a DAG grown out of it: old index ->
        new index for every node / edge both hold (matched by subsplit), the ids that are new taking the
        remaining places in order -- the permutations GPEngine::GrowPLVs / GrowGPCSPs expect
        (SubsplitDAG::ModificationResult::{node,edge}_reindexer, src/subsplit_dag.hpp)."""
        def complete(known, new_count):
            taken = set(known)
            rest = iter(i for i in range(new_count) if i not in taken)
            return np.array(list(known) + [next(rest) for _ in range(new_count - len(known))], dtype=np.int64)

        nodes = [grown.node_id[s] for s in self.subsplits]
        edges = [0] * self.gpcsp_count
        for (p, c), e in self.edge_id.items():
            gp_, gc = (-1 if p < 0 else grown.node_id[self.subsplits[p]]), grown.node_id[self.subsplits[c]]
            edges[e] = grown.edge_id[(gp_, gc)]
        return complete(nodes, grown.node_count), complete(edges, grown.gpcsp_count)

    # DAGBranchHandler::BuildBranchLengthMap / ApplyBranchLengthMap (src/dag_branch_handler.hpp:214-219):
    # branch lengths keyed by the edge's (parent subsplit, child subsplit), None = the DAG root
    def branch_length_map(self, branch_lengths) -> Dict:
        out = {}
        for (p, c), e in self.edge_id.items():
            out[(None if p < 0 else self.subsplits[p], self.subsplits[c])] = float(branch_lengths[e])
        return out

    def apply_branch_length_map(self, length_map: Dict, default: float = 0.1) -> np.ndarray:
        out = np.full(self.gpcsp_count, float(default))
        for (p, c), e in self.edge_id.items():
            key = (None if p < 0 else self.subsplits[p], self.subsplits[c])
            if key in length_map:
                out[e] = length_map[key]
        return out

    # -- ids -------------------------------------------------------------------
    def pv(self, plv_type: int, node: int) -> int:
        return plv_type * self.node_count + node

    def edge(self, parent: int, child: int) -> int:
        return self.edge_id[(parent, child)]

    def rootsplit_edge(self, rootsplit: int) -> int:
        return self.edge_id[(-1, rootsplit)]

    def uniform_on_topological_support_prior(self) -> np.ndarray:
        """SubsplitDAG::BuildUniformOnTopologicalSupportPrior (src/subsplit_dag.cpp:644-664)."""
        q = np.ones(self.gpcsp_count)
        for r in self.rootsplits:
            q[self.rootsplit_edge(r)] = self.topology_count_below[r] / self.topology_count
        for node in range(self.taxon_count, self.node_count):
            for side in (0, 1):
                kids = self.children[node][side]
                total = sum(self.topology_count_below[c] for c in kids)
                for c in kids:
                    q[self.edge(node, c)] = self.topology_count_below[c] / total
        return q

    # -- schedules (src/gp_dag.cpp:177-347) -------------------------------------
    def _rootward_pass(self, s: OpStream):
        for node in range(self.taxon_count, self.node_count):
            for side, phat in ((0, PHAT_RIGHT), (1, PHAT_LEFT)):  # AddPhatOperations(node, false) then (node, true)
                kids = self.children[node][side]
                dest = self.pv(phat, node)
                s.prep_for_marginalization(dest, [self.pv(P, c) for c in kids])
                for c in kids:
                    s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, dest, self.edge(node, c), self.pv(P, c))
            s.add(MULTIPLY, self.pv(P, node), self.pv(PHAT_RIGHT, node), self.pv(PHAT_LEFT, node))

    def _leafward_pass(self, s: OpStream):
        for node in range(self.node_count - 1, -1, -1):  # parents before children
            if self.parents[node]:
                srcs = [(self.pv(R_LEFT if side else R_RIGHT, p), self.edge(p, node)) for p, side in self.parents[node]]
                s.prep_for_marginalization(self.pv(RHAT, node), [src for src, _ in srcs])
                for src, e in srcs:
                    s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, self.pv(RHAT, node), e, src)
            s.add(MULTIPLY, self.pv(R_RIGHT, node), self.pv(RHAT, node), self.pv(PHAT_LEFT, node))
            s.add(MULTIPLY, self.pv(R_LEFT, node), self.pv(RHAT, node), self.pv(PHAT_RIGHT, node))

    def populate_plvs(self) -> OpStream:
        s = OpStream()
        for node in range(self.taxon_count, self.node_count):  # SetRootwardZero
            for t in (P, PHAT_RIGHT, PHAT_LEFT):
                s.add(ZERO_PLV, self.pv(t, node))
        for node in range(self.node_count):  # SetLeafwardZero
            for t in (RHAT, R_RIGHT, R_LEFT):
                s.add(ZERO_PLV, self.pv(t, node))
        for r in self.rootsplits:  # SetRhatToStationary
            s.add(SET_TO_STATIONARY, self.pv(RHAT, r), self.rootsplit_edge(r))
        self._rootward_pass(s)
        self._leafward_pass(s)
        return s

    def marginal_likelihood(self) -> OpStream:
        s = OpStream()
        s.add(RESET_MARGINAL_LIKELIHOOD)
        for r in self.rootsplits:
            s.add(INCREMENT_MARGINAL_LIKELIHOOD, self.pv(RHAT, r), self.rootsplit_edge(r), self.pv(P, r))
        return s

    def compute_likelihoods(self) -> OpStream:
        s = OpStream()
        for node in range(self.taxon_count, self.node_count):
            for side in (1, 0):
                for c in self.children[node][side]:
                    s.add(LIKELIHOOD, self.edge(node, c), self.pv(R_LEFT if side else R_RIGHT, node), self.pv(P, c))
        s.extend(self.marginal_likelihood())
        return s

    def optimize_sbn_parameters(self) -> OpStream:
        """GPDAG::OptimizeSBNParameters (src/gp_dag.cpp:213-224): one softmax per (parent, clade)
        with more than one child, then the rootsplits."""
        s = OpStream()
        for start, stop in self.sibling_ranges[1:]:
            if stop - start > 1:
                s.add(UPDATE_SBN_PROBABILITIES, start, stop)
        s.add(UPDATE_SBN_PROBABILITIES, 0, len(self.rootsplits))
        return s

    # -- GPDAG::BranchLengthOptimization (src/gp_dag.cpp:126-175) over the tidy depth-first traversal
    #    (TidySubsplitDAG, src/tidy_subsplit_dag.hpp:66-173, src/tidy_subsplit_dag.cpp:49-99) ---------
    def _tidy_state(self):
        """below[side][v]: v and every node under its ``side`` clade (the columns of the reference's
        above_rotated_ / above_sorted_ matrices); dirty[side]: node-clades whose p-hat is stale.  Like the
        reference's, the dirty flags live as long as the DAG object."""
        if not hasattr(self, "_below"):
            below = [[{v} for v in range(self.node_count)] for _ in (0, 1)]
            for v in range(self.node_count):  # children have smaller ids: their sets are complete
                for side in (0, 1):
                    for c in self.children[v][side]:
                        below[side][v] |= below[0][c] | below[1][c]
            self._below = below
            self._above = [[set() for _ in range(self.node_count)] for _ in (0, 1)]  # (side, v) -> ancestors via side
            for side in (0, 1):
                for a in range(self.node_count):
                    for v in below[side][a]:
                        self._above[side][v].add(a)
            self._dirty = [set(), set()]
        return self._below, self._above, self._dirty

    def set_clean(self):
        """TidySubsplitDAG::SetClean."""
        self._tidy_state()
        self._dirty = [set(), set()]

    def above_node(self, side: int, node: int):
        """TidySubsplitDAG::AboveNode(is_edge_on_left, node) without the DAG root: the nodes whose ``side``
        clade holds ``node``, and ``node`` itself."""
        return set(self._tidy_state()[1][side][node])

    def below_node(self, side: int, node: int):
        return set(self._tidy_state()[0][side][node])

    def set_dirty_strictly_above(self, node: int):
        below, above, dirty = self._tidy_state()
        for side in (0, 1):
            dirty[side] |= above[side][node] - {node}

    def dirty_vector(self, side: int):
        return set(self._tidy_state()[2][side])

    def _update_rhat(self, s: OpStream, node: int):
        """GPDAG::UpdateRHat (src/gp_dag.cpp:349-363): parents through their right clade first."""
        s.add(ZERO_PLV, self.pv(RHAT, node))
        srcs = [(self.pv(R_LEFT if side else R_RIGHT, p), self.edge(p, node))
                for want in (0, 1) for p, side in self.parents[node] if side == want]
        s.prep_for_marginalization(self.pv(RHAT, node), [src for src, _ in srcs])
        for src, e in srcs:
            s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, self.pv(RHAT, node), e, src)

    def branch_length_optimization(self, edges_to_optimize=None, record=None, zero_before_update: bool = False) -> OpStream:
        """One sweep of branch-length optimisation over the whole DAG.  An edge is optimised when the
        traversal first comes down its parent's clade ("modify"); before a node's clade is entered, a
        sister clade made stale by modifications further down (possible when a node has several
        parents) is brought up to date without optimising ("update").  ``record`` collects the
        traversal as (kind, node, child, side) tuples, like TidySubsplitDAG::RecordTraversal.

        The reference's update step adds the refreshed children onto the p-hat without clearing it
        first (UpdateEdge = UpdatePHatComputeLikelihood, no ZeroPLV; its own comments point at #321), so
        while the sister clade is optimised that p-hat holds old + new.  That is what the default
        emits.  ``zero_before_update=True`` clears the p-hat first; only then is the fixed point of the
        sweeps a stationary point of every edge's likelihood (tests/test_gp.py)."""
        below, above, dirty = self._tidy_state()
        n = self.taxon_count
        s = OpStream()
        visited = set()
        updating = [None]

        def is_dirty_below(node, side):
            return bool(below[side][node] & dirty[side])

        def note(*event):
            if record is not None:
                record.append(event)

        def phat(node, side):
            return self.pv(PHAT_LEFT if side else PHAT_RIGHT, node)

        def increment_phat(node, child, side, with_likelihood):
            e = self.edge(node, child)
            s.prep_for_marginalization(phat(node, side), [self.pv(P, child)])
            s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, phat(node, side), e, self.pv(P, child))
            if with_likelihood:  # UpdatePHatComputeLikelihood (src/gp_dag.cpp:365-384)
                s.add(LIKELIHOOD, e, self.pv(R_LEFT if side else R_RIGHT, node), self.pv(P, child))

        def after_node(node):
            s.add(MULTIPLY, self.pv(P, node), self.pv(PHAT_RIGHT, node), self.pv(PHAT_LEFT, node))

        def for_node(node):
            if node not in self.rootsplits:  # BeforeNode
                self._update_rhat(s, node)
            for_node_clade(node, 1)
            for_node_clade(node, 0)
            after_node(node)

        def for_node_clade(node, side):
            if updating[0] is not None:
                update_clade(node, side)
            else:
                modify_clade(node, side)

        def update_clade(node, side):
            if is_dirty_below(node, side):
                if zero_before_update:
                    s.add(ZERO_PLV, phat(node, side))
                for child in self.children[node][side]:
                    if child >= n:
                        for_node_clade(child, 1)
                        for_node_clade(child, 0)
                        after_node(child)
                    note("update", node, child, side)
                    increment_phat(node, child, side, True)
                    dirty[side].discard(node)
            if updating[0] == (node, side):
                updating[0] = None

        def modify_clade(node, side):
            if is_dirty_below(node, 1 - side):
                updating[0] = (node, 1 - side)
                update_clade(node, 1 - side)
            # BeforeNodeClade: RUpdateOfRotated, then the p-hat of this clade is rebuilt edge by edge
            note("descend", node, -1, side)
            if side:
                s.add(MULTIPLY, self.pv(R_LEFT, node), self.pv(RHAT, node), self.pv(PHAT_RIGHT, node))
            else:
                s.add(MULTIPLY, self.pv(R_RIGHT, node), self.pv(RHAT, node), self.pv(PHAT_LEFT, node))
            s.add(ZERO_PLV, phat(node, side))
            for child in self.children[node][side]:
                if child not in visited:
                    visited.add(child)
                    if child >= n:
                        for_node(child)
                note("modify", node, child, side)
                e = self.edge(node, child)
                if edges_to_optimize is None or e in edges_to_optimize:  # OptimizeBranchLengthUpdatePHat
                    s.add(OPTIMIZE_BRANCH_LENGTH, self.pv(P, child), self.pv(R_LEFT if side else R_RIGHT, node), e)
                increment_phat(node, child, side, False)
                self.set_dirty_strictly_above(node)
                dirty[side].discard(node)

        for r in self.rootsplits:
            for_node(r)
        return s

    # -- every tree the DAG spans (SubsplitDAG::GenerateAllTopologies, src/subsplit_dag.cpp:666-715) --
    def all_trees(self):
        """Yields (parent_ids, edge_of_node): a bito parent-id vector (leaves = taxon ids, internal
        ids in post-order, root last) and, per non-root tree node, the GPCSP id of the edge above it;
        the last entry is the rootsplit edge."""
        n = self.taxon_count

        def below(node):
            if node < n:
                return [(node,)]
            out = []
            for lc in self.children[node][1]:
                for lt in below(lc):
                    for rc in self.children[node][0]:
                        for rt in below(rc):
                            out.append((node, lc, lt, rc, rt))
            return out

        for r in self.rootsplits:
            for shape in below(r):
                parents: Dict[int, int] = {}
                edge_above: Dict[int, int] = {}
                counter = [n]

                def walk(t):
                    if len(t) == 1:
                        return t[0]
                    node, lc, lt, rc, rt = t
                    a, b = walk(lt), walk(rt)
                    me = counter[0]
                    counter[0] += 1
                    parents[a], parents[b] = me, me
                    edge_above[a], edge_above[b] = self.edge(node, lc), self.edge(node, rc)
                    return me

                root = walk(shape)
                assert root == 2 * n - 2
                pid = np.array([parents[v] for v in range(2 * n - 2)], dtype=np.int32)
                edges = [edge_above[v] for v in range(2 * n - 2)] + [self.rootsplit_edge(r)]
                yield pid, edges
