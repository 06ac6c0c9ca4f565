"""Generalized-pruning (GPEngine) seam: op-stream encoding, a schedule generator for
single-tree DAGs, and the ctypes mirror of the GPU executor (SURVEY.md section 8b seam 3).

The reference turns subsplit-DAG traversals into a flat ``GPOperationVector``
(src/gp_dag.cpp:177-411, src/gp_operation.hpp:24-170) that ``GPEngine::ProcessOperations``
executes one op at a time (src/gp_engine.cpp:213-339).  The subsplit DAG itself is out of
scope; what crosses the seam is the op stream, encoded here as POD records
``{uint32 opcode, uint32 count, uint64 a, b, c}`` plus a side array of PLV ids for
``PrepForMarginalization``.  ``single_tree_schedule`` emits the stream GPDAG would emit for
the DAG of ONE rooted tree (every SBN probability 1), which is what the reference's
``hello`` / ``fluA`` GP tests use.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _capi
from .engine import BitoAmdError

# opcodes = alternative index in the reference's std::variant GPOperation (gp_operation.hpp:162-167)
ZERO_PLV, SET_TO_STATIONARY, INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, MULTIPLY, LIKELIHOOD = 0, 1, 2, 3, 4
OPTIMIZE_BRANCH_LENGTH, UPDATE_SBN_PROBABILITIES, RESET_MARGINAL_LIKELIHOOD = 5, 6, 7
INCREMENT_MARGINAL_LIKELIHOOD, PREP_FOR_MARGINALIZATION = 8, 9

# PLV types in the order of PLVTypeEnum (reference src/pv_handler.hpp:26-34)
P, PHAT_RIGHT, PHAT_LEFT, RHAT, R_RIGHT, R_LEFT = range(6)

OP_DTYPE = np.dtype([("opcode", np.uint32), ("count", np.uint32), ("a", np.uint64), ("b", np.uint64),
                     ("c", np.uint64)])


class OpStream:
    def __init__(self):
        self.ops: List[Tuple[int, int, int, int, int]] = []
        self.side: List[int] = []
        self._arrays = None  # (op count, side count, ops array, side array) of the last conversion

    def add(self, opcode, a=0, b=0, c=0):
        self.ops.append((opcode, 0, a, b, c))

    def prep_for_marginalization(self, dest: int, sources: Sequence[int]):
        self.ops.append((PREP_FOR_MARGINALIZATION, len(sources), dest, len(self.side), 0))
        self.side.extend(int(x) for x in sources)

    def arrays(self):
        """The POD image handed to the executor; cached, a schedule is usually replayed many times."""
        if self._arrays is None or self._arrays[0] != len(self.ops) or self._arrays[1] != len(self.side):
            ops = np.array(self.ops, dtype=OP_DTYPE) if self.ops else np.zeros(0, dtype=OP_DTYPE)
            side = np.array(self.side if self.side else [0], dtype=np.uint64)
            self._arrays = (len(self.ops), len(self.side), ops, side)
        return self._arrays[2], self._arrays[3]

    def extend(self, other: "OpStream"):
        base = len(self.side)
        for opcode, count, a, b, c in other.ops:
            if opcode == PREP_FOR_MARGINALIZATION:
                b += base
            self.ops.append((opcode, count, a, b, c))
        self.side.extend(other.side)


@dataclass
class SingleTreeDAG:
    """The subsplit DAG of one rooted tree: DAG node ids = tree node ids, GPCSP 0 is the
    rootsplit edge, GPCSP 1+c the edge above tree node c."""
    taxon_count: int
    node_count: int  # without the DAG root
    gpcsp_count: int
    children: Dict[int, Tuple[int, int]]  # (left, right)
    parent: Dict[int, Tuple[int, bool]]  # node -> (parent, node is the left child)
    root: int

    def pv(self, plv_type: int, node: int) -> int:
        """PVId = type * node_count + node (reference src/pv_handler.hpp:487-490)."""
        return plv_type * self.node_count + node

    def edge(self, child: int) -> int:
        return 0 if child == self.root else 1 + child

    def branch_lengths(self, tree_branch_lengths: np.ndarray) -> np.ndarray:
        out = np.zeros(self.gpcsp_count)
        out[1:] = tree_branch_lengths[: self.node_count - 1]
        return out

    # -- GPDAG::PopulatePLVs (reference src/gp_dag.cpp:296-304) ---------------------
    def populate_plvs(self) -> OpStream:
        s = OpStream()
        n = self.taxon_count
        for node in range(n, self.node_count):  # SetRootwardZero
            for t in (P, PHAT_RIGHT, PHAT_LEFT):
                s.add(ZERO_PLV, self.pv(t, node))
        for node in range(self.node_count):  # SetLeafwardZero
            for t in (RHAT, R_RIGHT, R_LEFT):
                s.add(ZERO_PLV, self.pv(t, node))
        s.add(SET_TO_STATIONARY, self.pv(RHAT, self.root), 0)  # SetRhatToStationary
        for node in range(n, self.node_count):  # RootwardPass: ids are a post-order
            left, right = self.children[node]
            for t, child in ((PHAT_RIGHT, right), (PHAT_LEFT, left)):
                s.prep_for_marginalization(self.pv(t, node), [self.pv(P, child)])
                s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, self.pv(t, node), self.edge(child), self.pv(P, child))
            s.add(MULTIPLY, self.pv(P, node), self.pv(PHAT_RIGHT, node), self.pv(PHAT_LEFT, node))
        order, stack = [], [self.root]  # LeafwardPass: parents before children
        while stack:
            node = stack.pop()
            order.append(node)
            if node in self.children:
                stack.extend(reversed(self.children[node]))
        for node in order:
            if node != self.root:
                parent, is_left = self.parent[node]
                src = self.pv(R_LEFT if is_left else R_RIGHT, parent)
                s.prep_for_marginalization(self.pv(RHAT, node), [src])
                s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, self.pv(RHAT, node), self.edge(node), src)
            s.add(MULTIPLY, self.pv(R_RIGHT, node), self.pv(RHAT, node), self.pv(PHAT_LEFT, node))
            s.add(MULTIPLY, self.pv(R_LEFT, node), self.pv(RHAT, node), self.pv(PHAT_RIGHT, node))
        return s

    # -- GPDAG::ComputeLikelihoods + MarginalLikelihood (src/gp_dag.cpp:177-211) -------
    def compute_likelihoods(self) -> OpStream:
        s = OpStream()
        for node in range(self.taxon_count, self.node_count):
            left, right = self.children[node]
            for child, is_left in ((left, True), (right, False)):
                s.add(LIKELIHOOD, self.edge(child), self.pv(R_LEFT if is_left else R_RIGHT, node), self.pv(P, child))
        s.extend(self.marginal_likelihood())
        return s

    def marginal_likelihood(self) -> OpStream:
        """GPDAG::MarginalLikelihood (src/gp_dag.cpp:201-211)."""
        s = OpStream()
        s.add(RESET_MARGINAL_LIKELIHOOD)
        s.add(INCREMENT_MARGINAL_LIKELIHOOD, self.pv(RHAT, self.root), 0, self.pv(P, self.root))
        return s

    # -- GPDAG::BranchLengthOptimization (src/gp_dag.cpp:126-175) over the tidy depth-first
    #    traversal (src/tidy_subsplit_dag.hpp:82-173).  In the DAG of one tree a modification
    #    below one clade never dirties the sister clade, so the traversal never enters its
    #    "updating" mode and reduces to the recursion below.
    def branch_length_optimization(self, edges_to_optimize=None) -> OpStream:
        s = OpStream()

        def visit(node: int):
            if node != self.root:  # BeforeNode: UpdateRHat (src/gp_dag.cpp:349-363)
                parent, is_left = self.parent[node]
                src = self.pv(R_LEFT if is_left else R_RIGHT, parent)
                s.add(ZERO_PLV, self.pv(RHAT, node))
                s.prep_for_marginalization(self.pv(RHAT, node), [src])
                s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, self.pv(RHAT, node), self.edge(node), src)
            left, right = self.children[node]
            for child, is_left in ((left, True), (right, False)):
                # BeforeNodeClade: RUpdateOfRotated (src/gp_dag.cpp:17-24), then zero the p-hat
                if is_left:
                    s.add(MULTIPLY, self.pv(R_LEFT, node), self.pv(RHAT, node), self.pv(PHAT_RIGHT, node))
                else:
                    s.add(MULTIPLY, self.pv(R_RIGHT, node), self.pv(RHAT, node), self.pv(PHAT_LEFT, node))
                phat = self.pv(PHAT_LEFT if is_left else PHAT_RIGHT, node)
                s.add(ZERO_PLV, phat)
                if child in self.children:
                    visit(child)
                # ModifyEdge: OptimizeBranchLengthUpdatePHat (src/gp_dag.cpp:386-405)
                if edges_to_optimize is None or self.edge(child) in edges_to_optimize:
                    s.add(OPTIMIZE_BRANCH_LENGTH, self.pv(P, child), self.pv(R_LEFT if is_left else R_RIGHT, node),
                          self.edge(child))
                s.prep_for_marginalization(phat, [self.pv(P, child)])
                s.add(INCREMENT_WITH_WEIGHTED_EVOLVED_PLV, phat, self.edge(child), self.pv(P, child))
            # AfterNode
            s.add(MULTIPLY, self.pv(P, node), self.pv(PHAT_RIGHT, node), self.pv(PHAT_LEFT, node))

        visit(self.root)
        return s


# OptimizationMethod (reference src/optimization.hpp:28-34)
BRENT, BRENT_WITH_GRADIENTS, GRADIENT_ASCENT, LOGSPACE_GRADIENT_ASCENT, NEWTON = range(5)


def estimate_branch_lengths(engine, dag, tol: float, max_iter: int, method=None, **schedule_options) -> int:
    """``GPInstance::EstimateBranchLengths`` (reference src/gp_instance.cpp:241-300): sweeps of
    BranchLengthOptimization + PopulatePLVs + MarginalLikelihood until the mean absolute change
    of the branch lengths drops below ``tol``.  Works with any engine exposing the GPEngine
    mirror's methods and with either DAG (``SingleTreeDAG`` or ``gp_dag.SubsplitDAG``;
    ``schedule_options`` go to its ``branch_length_optimization``); returns the number of sweeps."""
    if method is not None:
        engine.set_optimization_method(method)
    engine.reset_optimization_count()
    optimize = dag.branch_length_optimization(**schedule_options)
    populate, marginal = dag.populate_plvs(), dag.marginal_likelihood()
    engine.process_operations(populate)
    engine.process_operations(marginal)
    sweeps = 0
    for _ in range(max_iter):
        engine.process_operations(optimize)
        engine.process_operations(populate)
        engine.process_operations(marginal)
        sweeps += 1
        if float(np.mean(engine.get_branch_length_differences())) < tol:
            break
        engine.increment_optimization_count()
    return sweeps


def single_tree_dag(parent_ids: Sequence[int]) -> SingleTreeDAG:
    """Left ("rotated") clade of a subsplit = the clade holding the smaller taxon id
    (Bitset::SubsplitFromUnorderedClades / CladeCompare, reference src/bitset.cpp:268-272,326-331)."""
    parent_ids = [int(x) for x in parent_ids]
    node_count = len(parent_ids) + 1
    n = (node_count + 1) // 2
    kids: Dict[int, List[int]] = {}
    min_taxon = list(range(n)) + [node_count] * (node_count - n)
    for child, p in enumerate(parent_ids):  # ids are a post-order: children before parents
        kids.setdefault(p, []).append(child)
        min_taxon[p] = min(min_taxon[p], min_taxon[child])
    children = {k: tuple(sorted(v, key=lambda c: min_taxon[c])) for k, v in kids.items()}
    parent = {}
    for k, (left, right) in children.items():
        parent[left] = (k, True)
        parent[right] = (k, False)
    return SingleTreeDAG(n, node_count, node_count, children, parent, node_count - 1)


# ------------------------------------------------------------------------------------
# ctypes mirror of the GPU executor (include/bito_amd_gp.h)

GP_SYMBOLS = [
    "bito_amd_gp_create", "bito_amd_gp_destroy", "bito_amd_gp_last_error", "bito_amd_gp_set_branch_lengths",
    "bito_amd_gp_get_branch_lengths", "bito_amd_gp_set_sbn_parameters", "bito_amd_gp_get_sbn_parameters",
    "bito_amd_gp_process_operations", "bito_amd_gp_log_marginal_likelihood",
    "bito_amd_gp_per_gpcsp_log_likelihoods", "bito_amd_gp_log_likelihood_matrix",
    "bito_amd_gp_log_likelihood_and_first_two_derivatives", "bito_amd_gp_get_branch_length_differences",
    "bito_amd_gp_set_optimization_method", "bito_amd_gp_set_significant_digits_for_optimization",
    "bito_amd_gp_reset_optimization_count", "bito_amd_gp_increment_optimization_count",
    "bito_amd_gp_grow_spare", "bito_amd_gp_copy_gpcsp_data", "bito_amd_gp_process_operation_batches",
    "bito_amd_gp_per_gpcsp_log_likelihoods_range", "bito_amd_gp_branch_lengths_range",
    "bito_amd_gp_grow", "bito_amd_gp_get_plv", "bito_amd_gp_rescaling_counts", "bito_amd_gp_get_plv_as_reference",
    "bito_amd_gp_schedule_operations", "bito_amd_gp_set_optimizer_trace", "bito_amd_gp_get_optimizer_trace",
]


def _lib():
    L = _capi.lib()
    if not getattr(L, "_gp_ready", False):
        dp, vp = C.POINTER(C.c_double), C.c_void_p
        L.bito_amd_gp_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), dp, C.c_int32,
                                         C.c_int32, C.c_double, C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.bito_amd_gp_destroy.restype = None
        L.bito_amd_gp_destroy.argtypes = [vp]
        L.bito_amd_gp_last_error.restype = C.c_char_p
        L.bito_amd_gp_last_error.argtypes = [vp]
        L.bito_amd_gp_set_optimization_method.argtypes = [vp, C.c_int32]
        L.bito_amd_gp_set_significant_digits_for_optimization.argtypes = [vp, C.c_int32]
        L.bito_amd_gp_reset_optimization_count.argtypes = [vp]
        L.bito_amd_gp_increment_optimization_count.argtypes = [vp]
        for name in ("set_branch_lengths", "get_branch_lengths", "set_sbn_parameters", "get_sbn_parameters",
                     "per_gpcsp_log_likelihoods", "log_likelihood_matrix", "log_marginal_likelihood",
                     "get_branch_length_differences"):
            getattr(L, f"bito_amd_gp_{name}").argtypes = [vp, dp]
        L.bito_amd_gp_process_operations.argtypes = [vp, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.bito_amd_gp_log_likelihood_and_first_two_derivatives.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, dp]
        ip = C.POINTER(C.c_int64)
        L.bito_amd_gp_grow_spare.argtypes = [vp, C.c_int64, C.c_int64]
        L.bito_amd_gp_copy_gpcsp_data.argtypes = [vp, ip, ip, C.c_int64]
        L.bito_amd_gp_process_operation_batches.argtypes = [vp, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, ip, C.c_int64]
        L.bito_amd_gp_per_gpcsp_log_likelihoods_range.argtypes = [vp, C.c_int64, C.c_int64, dp]
        L.bito_amd_gp_branch_lengths_range.argtypes = [vp, C.c_int64, C.c_int64, dp]
        L.bito_amd_gp_grow.argtypes = [vp, C.c_int32, C.c_int32, ip, ip]
        L.bito_amd_gp_get_plv.argtypes = [vp, C.c_int64, dp]
        L.bito_amd_gp_rescaling_counts.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(C.c_int32)]
        L.bito_amd_gp_get_plv_as_reference.argtypes = [vp, C.c_int64, dp, C.POINTER(C.c_int32)]
        i32p = C.POINTER(C.c_int32)
        L.bito_amd_gp_schedule_operations.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                                      i32p, i32p, i32p, ip]
        L.bito_amd_gp_set_optimizer_trace.argtypes = [vp, C.c_int64]
        L.bito_amd_gp_get_optimizer_trace.argtypes = [vp, dp, C.c_int64, ip]
        L._gp_ready = True
    return L


def schedule_operations(stream: OpStream, reorder: bool = True):
    """The order in which the executor runs ``stream`` (bito_amd_gp_schedule_operations: host arithmetic only, no GPU):
    returns (scheduled OpStream, launch index per op, level per op, launch kinds) -- kinds 0 per-pattern operations,
    1 concurrent optimisations, 2 UpdateSBNProbabilities."""
    ops, side = stream.arrays()
    n = len(ops)
    out = np.zeros(n, dtype=OP_DTYPE)
    launch = np.zeros(max(n, 1), dtype=np.int32)
    level = np.zeros(max(n, 1), dtype=np.int32)
    kinds = np.zeros(max(n, 1), dtype=np.int32)
    count = C.c_int64(0)
    i32p = C.POINTER(C.c_int32)
    rc = _lib().bito_amd_gp_schedule_operations(ops.ctypes.data, n, side.ctypes.data, len(side), 1 if reorder else 0,
                                                out.ctypes.data, launch.ctypes.data_as(i32p), level.ctypes.data_as(i32p),
                                                kinds.ctypes.data_as(i32p), C.byref(count))
    if rc:
        raise BitoAmdError(rc, "bito_amd_gp_schedule_operations rejected the stream")
    scheduled = OpStream()
    scheduled.ops = [tuple(int(v) for v in row) for row in out.tolist()]
    scheduled.side = list(stream.side)
    return scheduled, launch[:n].copy(), level[:n].copy(), kinds[:count.value].copy()


class GPEngine:
    """Mirror of the reference ``GPEngine`` (src/gp_engine.hpp:24-141), hot-path subset."""

    def __init__(self, patterns: np.ndarray, weights: np.ndarray, node_count: int, gpcsp_count: int,
                 rescaling_threshold: float = 1e-40, device_id: int = 0):
        L = _lib()
        self.patterns = np.ascontiguousarray(patterns, dtype=np.int32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        n, Pn = self.patterns.shape
        self.pattern_count, self.gpcsp_count, self.node_count = Pn, gpcsp_count, node_count
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.bito_amd_gp_create(device_id, n, Pn, self.patterns.ctypes.data_as(C.POINTER(C.c_int32)),
                                  self.weights.ctypes.data_as(C.POINTER(C.c_double)), node_count, gpcsp_count,
                                  rescaling_threshold, C.byref(h), err, 512)
        if rc:
            raise BitoAmdError(rc, err.value.decode())
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().bito_amd_gp_destroy(self._h)
            self._h = None

    def _check(self, rc):
        if rc:
            raise BitoAmdError(rc, _lib().bito_amd_gp_last_error(self._h).decode())

    def _vec(self, fn, count):
        out = np.zeros(count)
        self._check(fn(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def set_branch_lengths(self, bl):
        bl = np.ascontiguousarray(bl, dtype=np.float64)
        assert bl.shape == (self.gpcsp_count,)
        self._check(_lib().bito_amd_gp_set_branch_lengths(self._h, bl.ctypes.data_as(C.POINTER(C.c_double))))

    def get_branch_lengths(self):
        return self._vec(_lib().bito_amd_gp_get_branch_lengths, self.gpcsp_count)

    def get_branch_length_differences(self):
        return self._vec(_lib().bito_amd_gp_get_branch_length_differences, self.gpcsp_count)

    def set_optimization_method(self, method: int):
        self._check(_lib().bito_amd_gp_set_optimization_method(self._h, int(method)))

    def set_significant_digits_for_optimization(self, digits: int):
        self._check(_lib().bito_amd_gp_set_significant_digits_for_optimization(self._h, int(digits)))

    def reset_optimization_count(self):
        self._check(_lib().bito_amd_gp_reset_optimization_count(self._h))

    def increment_optimization_count(self):
        self._check(_lib().bito_amd_gp_increment_optimization_count(self._h))

    def set_sbn_parameters(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        self._check(_lib().bito_amd_gp_set_sbn_parameters(self._h, q.ctypes.data_as(C.POINTER(C.c_double))))

    def get_sbn_parameters(self):
        return self._vec(_lib().bito_amd_gp_get_sbn_parameters, self.gpcsp_count)

    def process_operations(self, stream: OpStream):
        ops, side = stream.arrays()
        self._check(_lib().bito_amd_gp_process_operations(self._h, ops.ctypes.data, len(ops), side.ctypes.data, len(side)))

    # -- diagnostics: the Brent optimisers' function evaluations as rows (gpcsp, x, f, kind) --
    def start_optimizer_trace(self, capacity: int = 1 << 16):
        self._trace_capacity = int(capacity)
        self._check(_lib().bito_amd_gp_set_optimizer_trace(self._h, self._trace_capacity))

    def optimizer_trace(self) -> np.ndarray:
        rows = np.zeros((self._trace_capacity, 4))
        made = C.c_int64(0)
        self._check(_lib().bito_amd_gp_get_optimizer_trace(self._h, rows.ctypes.data_as(C.POINTER(C.c_double)),
                                                           self._trace_capacity, C.byref(made)))
        if made.value > self._trace_capacity:
            raise BitoAmdError(_capi.ERR_STATE, f"optimiser trace overflow: {made.value} rows, capacity {self._trace_capacity}")
        return rows[:made.value].copy()

    def stop_optimizer_trace(self):
        self._check(_lib().bito_amd_gp_set_optimizer_trace(self._h, 0))

    # -- spare slots and side-by-side sub-streams (NNI proposals; src/gp_engine.cpp:196-211,401-409) --
    def grow_spare(self, spare_plv_count: int, spare_gpcsp_count: int):
        self._check(_lib().bito_amd_gp_grow_spare(self._h, int(spare_plv_count), int(spare_gpcsp_count)))

    def copy_gpcsp_data(self, src: Sequence[int], dst: Sequence[int]):
        a = np.ascontiguousarray(src, dtype=np.int64)
        b = np.ascontiguousarray(dst, dtype=np.int64)
        assert a.shape == b.shape
        ip = C.POINTER(C.c_int64)
        self._check(_lib().bito_amd_gp_copy_gpcsp_data(self._h, a.ctypes.data_as(ip), b.ctypes.data_as(ip), len(a)))

    def process_operation_batches(self, streams: Sequence[OpStream]):
        """Independent sub-streams (disjoint destinations) run side by side in one launch."""
        merged, offsets = OpStream(), [0]
        for s in streams:
            merged.extend(s)
            offsets.append(len(merged.ops))
        ops, side = merged.arrays()
        off = np.asarray(offsets, dtype=np.int64)
        self._check(_lib().bito_amd_gp_process_operation_batches(
            self._h, ops.ctypes.data, len(ops), side.ctypes.data, len(side), off.ctypes.data_as(C.POINTER(C.c_int64)),
            len(streams)))

    def _range(self, fn, first, count):
        out = np.zeros(count)
        self._check(fn(self._h, int(first), int(count), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def get_per_gpcsp_log_likelihoods_range(self, first: int, count: int):
        return self._range(_lib().bito_amd_gp_per_gpcsp_log_likelihoods_range, first, count)

    def get_branch_lengths_range(self, first: int, count: int):
        return self._range(_lib().bito_amd_gp_branch_lengths_range, first, count)

    def grow(self, new_node_count: int, new_gpcsp_count: int, node_reindexer=None, gpcsp_reindexer=None):
        """GrowPLVs + GrowGPCSPs with reindexers (old index -> new index), after the DAG has grown."""
        ip = C.POINTER(C.c_int64)
        nr = None if node_reindexer is None else np.ascontiguousarray(node_reindexer, dtype=np.int64)
        gr = None if gpcsp_reindexer is None else np.ascontiguousarray(gpcsp_reindexer, dtype=np.int64)
        if nr is not None and nr.shape != (new_node_count,) or gr is not None and gr.shape != (new_gpcsp_count,):
            raise BitoAmdError(_capi.ERR_BAD_ARG, "a reindexer has one entry per index of the grown engine")
        self._check(_lib().bito_amd_gp_grow(self._h, int(new_node_count), int(new_gpcsp_count),
                                            None if nr is None else nr.ctypes.data_as(ip),
                                            None if gr is None else gr.ctypes.data_as(ip)))
        self.node_count, self.gpcsp_count = int(new_node_count), int(new_gpcsp_count)

    def get_plv(self, plv: int) -> np.ndarray:
        out = np.zeros((4, self.pattern_count))
        self._check(_lib().bito_amd_gp_get_plv(self._h, int(plv), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def get_rescaling_counts(self, first: int = 0, count=None) -> np.ndarray:
        """The reference's ``rescaling_counts_`` (one per PLV) for PLVs ``first .. first + count - 1`` (default: the
        6 * node_count PLVs of the DAG)."""
        count = 6 * self.node_count - first if count is None else count
        out = np.zeros(count, dtype=np.int32)
        self._check(_lib().bito_amd_gp_rescaling_counts(self._h, int(first), int(count), out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def get_plv_as_reference(self, plv: int):
        """-> (values [4][pattern_count] as the reference's GetPLV would hold them, its rescaling count)"""
        out = np.zeros((4, self.pattern_count))
        cnt = C.c_int32()
        self._check(_lib().bito_amd_gp_get_plv_as_reference(self._h, int(plv), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt)))
        return out, int(cnt.value)

    def get_log_marginal_likelihood(self) -> float:
        return float(self._vec(_lib().bito_amd_gp_log_marginal_likelihood, 1)[0])

    def get_per_gpcsp_log_likelihoods(self):
        return self._vec(_lib().bito_amd_gp_per_gpcsp_log_likelihoods, self.gpcsp_count)

    def get_log_likelihood_matrix(self):
        return self._vec(_lib().bito_amd_gp_log_likelihood_matrix, self.gpcsp_count * self.pattern_count).reshape(
            self.gpcsp_count, self.pattern_count)

    def log_likelihood_and_first_two_derivatives(self, gpcsp: int, rootward: int, leafward: int):
        out = np.zeros(3)
        self._check(_lib().bito_amd_gp_log_likelihood_and_first_two_derivatives(
            self._h, gpcsp, rootward, leafward, out.ctypes.data_as(C.POINTER(C.c_double))))
        return tuple(out)
