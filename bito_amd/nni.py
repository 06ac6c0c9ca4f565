"""NNI proposals scored through the generalized-pruning executor.

Mirror of the part of the reference's NNI search that sits on top of GPEngine:
``NNIOperation`` (src/nni_operation.hpp), the set of NNIs adjacent to a DAG
(``NNIEngine::SyncAdjacentNNIsWithDAG``, src/nni_engine.cpp:766-875) and
``NNIEvalEngineViaGP`` (src/nni_evaluation_engine.cpp:49-461): every proposed NNI gets twelve
spare PLVs and a run of spare GPCSPs behind the DAG's own, its partial vectors are rebuilt from
the neighbours of the NNI it was derived from, and its score is the per-GPCSP log-likelihood of
its central edge.

The reference scores the proposals one at a time, each with its own ``ProcessOperations`` call
over the same temporaries.  Here every proposal owns its spare slots, the proposals' operation
lists are independent, and ``score_adjacent_nnis`` hands all of them to the executor as
side-by-side sub-streams: one launch, grid = pattern tiles x proposals.

Only ids and schedules live here; the arithmetic is the executor's.
"""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from .gp import (INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, LIKELIHOOD, MULTIPLY, OPTIMIZE_BRANCH_LENGTH, P, R_LEFT, R_RIGHT,
                 SET_TO_STATIONARY, ZERO_PLV, OpStream)
from .gp_dag import SubsplitDAG

Subsplit = Tuple[int, int]

# spare PLVs of one proposal, in the order of NNIEvalEngineViaGP::GetTempAdjPVIds
# (src/nni_evaluation_engine.cpp:619-645)
(PARENT_P, PARENT_PHAT_FOCAL, PARENT_PHAT_SISTER, PARENT_RHAT, PARENT_R_FOCAL, PARENT_R_SISTER, CHILD_P, CHILD_PHAT_LEFT,
 CHILD_PHAT_RIGHT, CHILD_RHAT, CHILD_R_LEFT, CHILD_R_RIGHT) = range(12)
SPARE_PLVS_PER_NNI = 12


def make_subsplit(a: int, b: int) -> Subsplit:
    """Left clade = the one holding the smaller taxon id (the convention of ``gp_dag``)."""
    return (a, b) if (a & -a) < (b & -b) else (b, a)


class NNI(NamedTuple):
    """NNIOperation: a parent subsplit and the child subsplit on one of its clades."""
    parent: Subsplit
    child: Subsplit

    @property
    def focal_clade(self) -> int:
        return self.child[0] | self.child[1]

    @property
    def sister_clade(self) -> int:
        return self.parent[1] if self.parent[0] == self.focal_clade else self.parent[0]

    def neighbors(self) -> List["NNI"]:
        """NNIOperation::GetNeighboringNNI for both child clades: the sister changes places with the
        child's left clade, then with its right clade."""
        sister = self.sister_clade
        out = []
        for swapped, kept in ((self.child[0], self.child[1]), (self.child[1], self.child[0])):
            out.append(NNI(make_subsplit(swapped, sister | kept), make_subsplit(sister, kept)))
        return out


def contains_nni(dag: SubsplitDAG, nni: NNI) -> bool:
    """SubsplitDAG::ContainsNNI (src/subsplit_dag.cpp:1554-1557)."""
    return dag.contains_edge(nni.parent, nni.child)


def adjacent_nnis(dag: SubsplitDAG, include_rootsplits: bool = True) -> List[NNI]:
    """NNIEngine::SyncAdjacentNNIsWithDAG: the neighbours of every internal edge that the DAG does not hold."""
    found = set()
    n = dag.taxon_count
    for (p, c) in dag.edge_id:
        if p < 0 or c < n:  # the parent is the DAG root, or the child is a leaf
            continue
        if not include_rootsplits and p in dag.rootsplits:
            continue
        for nb in NNI(dag.subsplits[p], dag.subsplits[c]).neighbors():
            if not contains_nni(dag, nb):
                found.add(nb)
    return sorted(found)


def find_nni_neighbor_in_dag(dag: SubsplitDAG, nni: NNI) -> NNI:
    """SubsplitDAG::FindNNINeighborInDAG (src/subsplit_dag.cpp:559-572)."""
    for nb in nni.neighbors():
        if contains_nni(dag, nb):
            return nb
    raise ValueError("NNIOperation has no neighbors found in the DAG.")


class _Adjacent(NamedTuple):
    nodes: List[int]  # DAG node ids; [-1] stands for the DAG root above a rootsplit
    edges: List[int]  # GPCSP ids, same order
    sides: List[int]  # for rootward neighbours: the side of the grandparent the edge hangs on


def _adjacent_by_clade(dag: SubsplitDAG, pre: NNI) -> Tuple[Dict[int, _Adjacent], int]:
    """The four neighbourhoods of an NNI inside the DAG (NNIEvalEngineViaGP::GetAdjNodeAndEdgeIds,
    src/nni_evaluation_engine.cpp:464-509), keyed by the clade they hang on: the whole clade of the
    parent (grandparents above), the sister clade, and the child's two clades."""
    parent, child = dag.node_id[pre.parent], dag.node_id[pre.child]
    out: Dict[int, _Adjacent] = {}
    if dag.parents[parent]:
        ups = dag.parents[parent]
        out[pre.parent[0] | pre.parent[1]] = _Adjacent([g for g, _ in ups], [dag.edge(g, parent) for g, _ in ups],
                                                       [side for _, side in ups])
    else:
        out[pre.parent[0] | pre.parent[1]] = _Adjacent([-1], [dag.rootsplit_edge(parent)], [0])
    sister_side = 1 if pre.parent[0] == pre.sister_clade else 0
    for node, side, clade in ((parent, sister_side, pre.sister_clade), (child, 1, pre.child[0]), (child, 0, pre.child[1])):
        kids = dag.children[node][side]
        out[clade] = _Adjacent(list(kids), [dag.edge(node, k) for k in kids], [0] * len(kids))
    return out, dag.edge(parent, child)


class NNIProposal(NamedTuple):
    nni: NNI
    pre_nni: NNI
    spare_plv_base: int  # offset into the spare PLVs (12 per proposal)
    central_edge: int  # GPCSP id of the proposal's central edge (a spare id)
    copy_src: List[int]  # pre-NNI GPCSP ids ...
    copy_dst: List[int]  # ... whose data go to these spare GPCSP ids
    stream: OpStream


def nni_edge_sources(dag: SubsplitDAG, pre: NNI, nni: NNI) -> Dict[Tuple[Optional[Subsplit], Subsplit], int]:
    """For every edge around ``nni`` -- (grandparent, parent), (parent, sister), (parent, child),
    (child, grandchild) -- the GPCSP of the DAG whose branch length it takes over
    (NNIEvalEngineViaGP::CopyGPEngineDataAfterAddingNNI, src/nni_evaluation_engine.cpp:140-192)."""
    adj, central = _adjacent_by_clade(dag, pre)
    out: Dict[Tuple[Optional[Subsplit], Subsplit], int] = {(nni.parent, nni.child): central}
    up = adj[nni.parent[0] | nni.parent[1]]
    for g, e in zip(up.nodes, up.edges):
        out[(None if g < 0 else dag.subsplits[g], nni.parent)] = e
    for owner, clade in ((nni.parent, nni.sister_clade), (nni.child, nni.child[0]), (nni.child, nni.child[1])):
        for k, e in zip(adj[clade].nodes, adj[clade].edges):
            out[(owner, dag.subsplits[k])] = e
    return out


def build_proposal(dag: SubsplitDAG, nni: NNI, spare_plv_base: int, spare_edge_base: int, plv_count: int,
                   pre_nni: Optional[NNI] = None, optimize_new_edges: bool = False,
                   optimization_max_iteration: int = 10) -> NNIProposal:
    """Operation list of NNIEvalEngineViaGP::ComputeAdjacentNNILikelihood (rootward pass, leafward
    pass, likelihood of the central edge; src/nni_evaluation_engine.cpp:206-461) on this proposal's
    own spare slots.  ``spare_edge_base`` is the first GPCSP id it may use, ``plv_count`` the
    number of PLVs of the DAG itself (6 * node_count).  With ``optimize_new_edges`` the list carries
    the reference's NNIBranchLengthOptimization round (left children, right children, sisters,
    central, parents; each followed by the refresh of the partial vector it feeds) and a leafward
    pass, ``optimization_max_iteration`` times, before the final passes."""
    pre = pre_nni or find_nni_neighbor_in_dag(dag, nni)
    adj, pre_central = _adjacent_by_clade(dag, pre)
    pv = lambda k: plv_count + spare_plv_base + k
    next_edge = [spare_edge_base]
    copy_src: List[int] = []
    copy_dst: List[int] = []

    def take(src_edges: Sequence[int]) -> List[int]:
        ids = list(range(next_edge[0], next_edge[0] + len(src_edges)))
        next_edge[0] += len(src_edges)
        copy_src.extend(src_edges)
        copy_dst.extend(ids)
        return ids

    # central edge first, then parents, sisters, left children, right children (GetTempAdjEdgeIds)
    central = take([pre_central])[0]
    up = adj[nni.parent[0] | nni.parent[1]]
    sis, left, right = adj[nni.sister_clade], adj[nni.child[0]], adj[nni.child[1]]
    e_up, e_sis, e_left, e_right = take(up.edges), take(sis.edges), take(left.edges), take(right.edges)

    s = OpStream()

    def gather(dest: int, sources: Sequence[int], edges: Sequence[int]):
        s.add(ZERO_PLV, dest)
        s.prep_for_marginalization(dest, list(sources))
        for src, e in zip(sources, edges):
            s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, dest, e, src)

    left_p, right_p, sis_p = ([dag.pv(P, k) for k in adj.nodes] for adj in (left, right, sis))
    at_root = up.nodes == [-1]
    up_r = [] if at_root else [dag.pv(R_LEFT if side else R_RIGHT, g) for g, side in zip(up.nodes, up.sides)]

    def optimize(rootward: int, leafward: int, edge: int):
        s.add(OPTIMIZE_BRANCH_LENGTH, leafward, rootward, edge)

    # the ten partial-vector updates of an NNI (Update*Rootward / Update*Leafward lambdas of the reference)
    def left_rootward():
        gather(pv(CHILD_PHAT_LEFT), left_p, e_left)

    def right_rootward():
        gather(pv(CHILD_PHAT_RIGHT), right_p, e_right)

    def central_rootward():
        s.add(MULTIPLY, pv(CHILD_P), pv(CHILD_PHAT_LEFT), pv(CHILD_PHAT_RIGHT))
        gather(pv(PARENT_PHAT_FOCAL), [pv(CHILD_P)], [central])

    def sister_rootward():
        gather(pv(PARENT_PHAT_SISTER), sis_p, e_sis)

    def parent_rootward():
        s.add(MULTIPLY, pv(PARENT_P), pv(PARENT_PHAT_FOCAL), pv(PARENT_PHAT_SISTER))

    def parent_leafward():
        if at_root:
            s.add(ZERO_PLV, pv(PARENT_RHAT))
            s.add(SET_TO_STATIONARY, pv(PARENT_RHAT), e_up[0])
        else:
            gather(pv(PARENT_RHAT), up_r, e_up)

    def central_leafward():
        s.add(MULTIPLY, pv(PARENT_R_FOCAL), pv(PARENT_RHAT), pv(PARENT_PHAT_SISTER))
        gather(pv(CHILD_RHAT), [pv(PARENT_R_FOCAL)], [central])

    def sister_leafward():
        s.add(MULTIPLY, pv(PARENT_R_SISTER), pv(PARENT_RHAT), pv(PARENT_PHAT_FOCAL))

    def left_leafward():
        s.add(MULTIPLY, pv(CHILD_R_LEFT), pv(CHILD_RHAT), pv(CHILD_PHAT_RIGHT))

    def right_leafward():
        s.add(MULTIPLY, pv(CHILD_R_RIGHT), pv(CHILD_RHAT), pv(CHILD_PHAT_LEFT))

    def rootward_pass():
        left_rootward(), right_rootward(), central_rootward(), sister_rootward(), parent_rootward()

    def leafward_pass():
        parent_leafward(), central_leafward(), sister_leafward(), left_leafward(), right_leafward()

    if optimize_new_edges:
        rootward_pass()
        leafward_pass()
        for _ in range(optimization_max_iteration):
            for src, e in zip(left_p, e_left):  # OptimizeLeftChild
                optimize(pv(CHILD_R_LEFT), src, e)
            left_rootward()
            for src, e in zip(right_p, e_right):  # OptimizeRightChild
                optimize(pv(CHILD_R_RIGHT), src, e)
            right_rootward()
            sister_leafward()  # OptimizeSister
            for src, e in zip(sis_p, e_sis):
                optimize(pv(PARENT_R_SISTER), src, e)
            sister_rootward()
            central_leafward()  # OptimizeCentral
            optimize(pv(PARENT_R_FOCAL), pv(CHILD_P), central)
            central_rootward()
            parent_leafward()  # OptimizeParent: nothing to optimise above a rootsplit
            if at_root:
                pass
            else:
                for src, e in zip(up_r, e_up):
                    optimize(src, pv(PARENT_P), e)
                parent_rootward()
            leafward_pass()
    rootward_pass()
    leafward_pass()
    s.add(LIKELIHOOD, central, pv(PARENT_R_FOCAL), pv(CHILD_P))
    return NNIProposal(nni, pre, spare_plv_base, central, copy_src, copy_dst, s)


class NNIEvalEngineViaGP:
    """``NNIEvalEngineViaGP`` over a GPEngine mirror (anything with its methods: the GPU executor in
    production, the CPU checker in the tests)."""

    def __init__(self, dag: SubsplitDAG, engine, include_rootsplit_nnis: bool = True,
                 optimize_new_edges: bool = False, optimization_max_iteration: int = 10):
        self.dag, self.engine = dag, engine
        self.include_rootsplit_nnis = include_rootsplit_nnis
        # SetOptimizeNewEdges / SetOptimizationMaxIteration (src/nni_evaluation_engine.hpp:127-138)
        self.optimize_new_edges = optimize_new_edges
        self.optimization_max_iteration = optimization_max_iteration
        self.scored_nnis: Dict[NNI, float] = {}
        self.proposals: List[NNIProposal] = []

    def prep(self):
        """NNIEvalEngineViaGP::Prep (src/nni_evaluation_engine.cpp:58-61)."""
        self.engine.process_operations(self.dag.populate_plvs())
        self.engine.process_operations(self.dag.compute_likelihoods())

    def adjacent_nnis(self) -> List[NNI]:
        return adjacent_nnis(self.dag, self.include_rootsplit_nnis)

    def score_internal_nni(self, nni: NNI) -> float:
        """ScoreInternalNNIByNNI (src/nni_evaluation_engine.cpp:198-206): an NNI the DAG already holds."""
        if not contains_nni(self.dag, nni):
            raise ValueError("DAG does not contain NNI.")
        e = self.dag.edge(self.dag.node_id[nni.parent], self.dag.node_id[nni.child])
        return float(self.engine.get_per_gpcsp_log_likelihoods_range(e, 1)[0])

    def score_adjacent_nnis(self, nnis: Optional[Sequence[NNI]] = None) -> Dict[NNI, float]:
        """ScoreAdjacentNNIs / ComputeAdjacentNNILikelihoods with unique temporaries per proposal
        (GrowEngineForAdjacentNNIs(via_reference = true, use_unique_temps = true)): all proposals in
        ONE batched launch.  The DAG's own PLVs must be current (``prep``)."""
        nnis = list(self.adjacent_nnis() if nnis is None else nnis)
        plv_count = 6 * self.dag.node_count
        self.proposals = []
        edge_base = self.dag.gpcsp_count
        for i, nni in enumerate(nnis):
            prop = build_proposal(self.dag, nni, SPARE_PLVS_PER_NNI * i, edge_base, plv_count,
                                  optimize_new_edges=self.optimize_new_edges,
                                  optimization_max_iteration=self.optimization_max_iteration)
            edge_base += len(prop.copy_dst)
            self.proposals.append(prop)
        self.engine.grow_spare(SPARE_PLVS_PER_NNI * len(nnis), edge_base - self.dag.gpcsp_count)
        if not nnis:
            return {}
        self.engine.copy_gpcsp_data([x for p in self.proposals for x in p.copy_src],
                                    [x for p in self.proposals for x in p.copy_dst])
        self.engine.process_operation_batches([p.stream for p in self.proposals])
        first = self.dag.gpcsp_count
        scores = self.engine.get_per_gpcsp_log_likelihoods_range(first, edge_base - first)
        out = {p.nni: float(scores[p.central_edge - first]) for p in self.proposals}
        self.scored_nnis.update(out)
        return out
