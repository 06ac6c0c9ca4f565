"""The hot-path subset of bito's Python instance API, backed by the GPU engine.

Method names and argument meaning follow the pybind11 module of the reference
(src/pybito.cpp:289-573: ``rooted_instance`` / ``unrooted_instance``), restricted
to what sits on the likelihood path: reading trees and alignments,
``prepare_for_phylo_likelihood``, the parameter matrix and its block views,
``log_likelihoods``, ``phylo_gradients``, ``set_rescaling``.  SBN training,
sampling and DAG functionality are out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _capi, treeio
from .engine import BitoAmdError, Engine, PhyloGradient, PhyloModelSpecification
from .site_pattern import CodonSitePattern, SitePattern


# PhyloGradientFlagOptions of the reference all default to "on", stick-breaking included
# (src/phylo_flags.hpp:322-343); rooted instances add ratios_root_height and the log-det-Jacobian gradient.
DEFAULT_GRADIENT_FLAGS = (_capi.GRAD_SUBSTITUTION_MODEL | _capi.GRAD_SITE_MODEL | _capi.GRAD_CLOCK_MODEL |
                          _capi.GRAD_STICKBREAKING)
DEFAULT_ROOTED_GRADIENT_FLAGS = (DEFAULT_GRADIENT_FLAGS | _capi.GRAD_RATIOS_ROOT_HEIGHT |
                                 _capi.GRAD_LOG_DET_JACOBIAN_GRADIENT)


class _Tree:
    """pybito ``UnrootedTree`` / ``RootedTree``: ``branch_lengths`` is a writable numpy
    view (the reference exposes the vector through the buffer protocol,
    test/test_bito.py:37-38)."""

    def __init__(self, parsed: treeio.ParsedTree):
        self._parent_ids = parsed.parent_ids
        self.branch_lengths = parsed.branch_lengths
        self.leaf_count = parsed.leaf_count
        self.rates = np.ones(parsed.node_count - 1)
        self.rate_count = 1
        # time-tree state of RootedTree (reference src/rooted_tree.hpp); empty until dates are set
        self.node_bounds = np.zeros(0)
        self.node_heights = np.zeros(0)
        self.height_ratios = np.zeros(0)
        self._engine_ref = None

    def parent_id_vector(self) -> List[int]:
        return [int(x) for x in self._parent_ids]

    def initialize_time_tree_using_height_ratios(self, height_ratios):
        """``RootedTree::InitializeTimeTreeUsingHeightRatios`` (reference src/rooted_tree.cpp:101-121,
        pybito ``initialize_time_tree_using_height_ratios``): node heights and branch lengths from
        the ratios, evaluated on the device."""
        if self.node_bounds.size == 0:
            raise RuntimeError("Have you set dates for your time trees?")
        if self._engine_ref is None or self._engine_ref() is None:
            raise RuntimeError("Engine not available. Call prepare_for_phylo_likelihood to make an engine for "
                               "phylogenetic likelihood computation.")
        ratios = np.asarray(height_ratios, dtype=np.float64)
        heights, bl = self._engine_ref().time_trees_from_height_ratios(self._parent_ids[None, :],
                                                                       self.node_bounds[None, :], ratios[None, :])
        self.height_ratios = ratios.copy()
        self.node_heights[:] = heights[0]
        self.branch_lengths[:] = bl[0]


class _TreeCollection:
    def __init__(self, trees: List[_Tree]):
        self.trees = trees


class _GenericInstance:
    _rooted = False

    def __init__(self, name: str = ""):
        self.name = name
        self.tree_collection = _TreeCollection([])
        self._taxon_names: List[str] = []
        self._alignment: Dict[str, str] = {}
        self._engine: Optional[Engine] = None
        self._params = np.zeros((0, 0))
        self._rescaling = False
        self._device_id = 0

    # -- loading (reference src/generic_sbn_instance.hpp:286-330) ------------
    def _load(self, coll: treeio.TreeCollection):
        for t in coll.trees:
            if t.rooted != self._rooted:
                kind = "bifurcating" if self._rooted else "trifurcating"
                raise RuntimeError(f"Expected a tree with a {kind} root.")
        self.tree_collection = _TreeCollection([_Tree(t) for t in coll.trees])
        self._taxon_names = list(coll.taxon_names)

    def read_newick_file(self, path: str, sort_taxa: bool = True):
        self._load(treeio.read_newick_file(path, sort_taxa))

    def read_nexus_file(self, path: str, sort_taxa: bool = True):
        self._load(treeio.read_nexus_file(path))

    def read_fasta_file(self, path: str):
        self._alignment = treeio.read_fasta(path)

    def taxon_names(self) -> List[str]:
        return list(self._taxon_names)

    def tree_count(self) -> int:
        return len(self.tree_collection.trees)

    def load_duplicates_of_first_tree(self, number_of_times: int):
        first = self.tree_collection.trees[0]
        trees = []
        for _ in range(number_of_times):
            parsed = treeio.ParsedTree(first._parent_ids.copy(), first.branch_lengths.copy(), first.leaf_count)
            trees.append(_Tree(parsed))
        self.tree_collection = _TreeCollection(trees)

    # -- engine life cycle (reference src/generic_sbn_instance.hpp:235-284,380-386)
    def prepare_for_phylo_likelihood(self, model_specification: PhyloModelSpecification, thread_count: int = 1,
                                     beagle_flags: Sequence = (), use_tip_states: bool = True,
                                     tree_count_option: Optional[int] = None, device_id: Optional[int] = None,
                                     devices: Optional[Sequence[int]] = None):
        """``devices``: the GPUs behind this instance's engine (default: one, ``device_id``) -- what the reference's
        ``thread_count`` FatBeagle instances are (src/generic_sbn_instance.cpp MakeEngine, src/engine.cpp:10-31): every
        collection-level call is sharded over them."""
        if thread_count == 0:
            raise RuntimeError("Thread count needs to be strictly positive.")
        if not self._alignment:
            raise RuntimeError("Load an alignment into your instance before preparing for phylogenetic likelihood.")
        # the codon model reads the alignment three nucleotides at a time (states 0..60, 61 = gap)
        pattern_type = CodonSitePattern if model_specification.substitution == "GY94" else SitePattern
        site_pattern = pattern_type(self._alignment, self._taxon_names)
        if self._engine is not None:
            self._engine.close()
        self._engine = Engine(model_specification, site_pattern.patterns, site_pattern.weights,
                              device_id=self._device_id if device_id is None else device_id,
                              use_tip_states=use_tip_states, devices=devices)
        self.resize_phylo_model_params(tree_count_option)

    def resize_phylo_model_params(self, tree_count_option: Optional[int] = None):
        count = tree_count_option if tree_count_option is not None else self.tree_count()
        if count == 0:
            raise RuntimeError("Please add trees to your instance by sampling or loading before preparing for "
                               "phylogenetic likelihood calculation if you aren't going to specify a tree count.")
        self._params = self._get_engine().default_params(count)

    def _get_engine(self) -> Engine:
        if self._engine is None:
            raise RuntimeError("Engine not available. Call prepare_for_phylo_likelihood to make an engine for "
                               "phylogenetic likelihood computation.")
        return self._engine

    def get_phylo_model_params(self) -> np.ndarray:
        return self._params

    def get_phylo_model_param_block_map(self) -> Dict[str, np.ndarray]:
        """Writable views into the parameter matrix, one per block name
        (BlockSpecification::ParameterBlockMapOf, reference src/block_specification.cpp:100-111)."""
        return {k: self._params[:, s:s + ln] for k, (s, ln) in self._get_engine().block_map().items()}

    def set_rescaling(self, use_rescaling: bool):
        self._rescaling = bool(use_rescaling)

    # -- wire format -----------------------------------------------------------
    def _wire(self):
        trees = self.tree_collection.trees
        if not trees:
            raise RuntimeError("No trees loaded.")
        pid = np.stack([t._parent_ids for t in trees]).astype(np.int32)
        bl = np.stack([np.asarray(t.branch_lengths, dtype=np.float64) for t in trees])
        if self._params.shape[0] != len(trees):
            raise RuntimeError("We param_matrix needs as many rows as we have trees.")
        return pid, bl

    def log_likelihoods(self) -> np.ndarray:
        raise NotImplementedError

    def phylo_gradients(self) -> List[PhyloGradient]:
        raise NotImplementedError


class unrooted_instance(_GenericInstance):
    """pybito ``unrooted_instance`` (reference src/pybito.cpp:425-573)."""
    _rooted = False

    def log_likelihoods(self) -> np.ndarray:
        pid, bl = self._wire()
        return self._get_engine().log_likelihoods(pid, bl, self._params, rescaling=self._rescaling)

    def phylo_gradients(self, flags: int = DEFAULT_GRADIENT_FLAGS) -> List[PhyloGradient]:
        pid, bl = self._wire()
        out = self._get_engine().gradients(pid, bl, self._params, rescaling=self._rescaling, flags=flags)
        return _to_gradients(out)


class rooted_instance(_GenericInstance):
    """pybito ``rooted_instance`` (reference src/pybito.cpp:289-421), time trees included: tip
    dates, height ratios, the log-det-Jacobian of the height transform and the
    ``ratios_root_height`` gradient (reference src/rooted_sbn_instance.cpp:43-131)."""
    _rooted = True

    def __init__(self, name: str = ""):
        super().__init__(name)
        self._tip_dates: Optional[np.ndarray] = None
        self._init_from_branch_lengths = False

    # -- tip dates (reference src/rooted_tree_collection.cpp:30-81) -------------
    def set_dates_to_be_constant(self, initialize_time_trees_using_branch_lengths: bool):
        self._process_dates(np.zeros(len(self._taxon_names)), initialize_time_trees_using_branch_lengths)

    def parse_dates_from_taxon_names(self, initialize_time_trees_using_branch_lengths: bool):
        self._process_dates(treeio.parse_dates_from_taxon_names(self._taxon_names),
                            initialize_time_trees_using_branch_lengths)

    def parse_dates_from_csv(self, csv_path: str, initialize_time_trees_using_branch_lengths: bool):
        self._process_dates(treeio.parse_dates_from_csv(csv_path, self._taxon_names),
                            initialize_time_trees_using_branch_lengths)

    def tip_dates(self) -> Dict[str, float]:
        """``RootedTreeCollection::GetTagDateMap`` keyed by taxon name."""
        if self._tip_dates is None:
            return {}
        return {nm: float(d) for nm, d in zip(self._taxon_names, self._tip_dates)}

    def _process_dates(self, dates: np.ndarray, initialize: bool):
        self._tip_dates = np.asarray(dates, dtype=np.float64)
        self._init_from_branch_lengths = bool(initialize)
        for t in self.tree_collection.trees:  # RootedTree::SetTipDates (src/rooted_tree.cpp:36-44)
            t.node_heights = np.zeros(2 * t.leaf_count - 1)
            t.node_heights[:t.leaf_count] = self._tip_dates
            t.rates = np.ones(2 * t.leaf_count - 2)
            t.rate_count = 1
            t.height_ratios = np.zeros(0)
        if self._engine is not None:
            self._initialize_time_trees()

    def _initialize_time_trees(self):
        """SetNodeBoundsUsingDates for every tree and, if asked, InitializeTimeTreeUsingBranchLengths
        -- on the device, so it waits for the engine (the reference runs it on the host at parse
        time; the observable state after prepare_for_phylo_likelihood is the same)."""
        trees = self.tree_collection.trees
        if self._tip_dates is None or not trees:
            return
        eng = self._get_engine()
        import weakref
        ref = weakref.ref(eng)
        pid, bl = self._wire_trees()
        if self._init_from_branch_lengths:
            bounds, heights, ratios = eng.time_trees_from_branch_lengths(pid, bl, self._tip_dates)
        else:
            bounds, heights, ratios = self._bounds_only(pid), None, None
        for i, t in enumerate(trees):
            t.node_bounds = bounds[i].copy()
            t._engine_ref = ref
            if heights is not None:
                t.node_heights = heights[i].copy()
                t.height_ratios = ratios[i].copy()

    def _bounds_only(self, pid: np.ndarray) -> np.ndarray:
        """RootedTree::SetNodeBoundsUsingDates alone (src/rooted_tree.cpp:46-60): the latest tip
        date below every node."""
        n = len(self._tip_dates)
        bounds = np.full((pid.shape[0], 2 * n - 1), -np.inf)
        bounds[:, :n] = self._tip_dates
        rows = np.arange(pid.shape[0])
        for child in range(2 * n - 2):  # ids are post-order: children before parents
            bounds[rows, pid[:, child]] = np.maximum(bounds[rows, pid[:, child]], bounds[:, child])
        return bounds

    def prepare_for_phylo_likelihood(self, *args, **kwargs):
        super().prepare_for_phylo_likelihood(*args, **kwargs)
        self._initialize_time_trees()

    def _wire_trees(self):
        trees = self.tree_collection.trees
        pid = np.stack([t._parent_ids for t in trees]).astype(np.int32)
        bl = np.stack([np.asarray(t.branch_lengths, dtype=np.float64) for t in trees])
        return pid, bl

    def _rates(self) -> np.ndarray:
        return np.stack([np.asarray(t.rates, dtype=np.float64) for t in self.tree_collection.trees])

    def _time_state(self, need_ratios: bool):
        trees = self.tree_collection.trees
        if any(t.node_bounds.size == 0 or t.node_heights.size == 0 for t in trees) or \
                (need_ratios and any(t.height_ratios.size == 0 for t in trees)):
            raise RuntimeError("Time trees have not been initialized: set tip dates (parse_dates_from_taxon_names, "
                               "parse_dates_from_csv, set_dates_to_be_constant) with "
                               "initialize_time_trees_using_branch_lengths=True or set height ratios.")
        heights = np.stack([t.node_heights for t in trees])
        bounds = np.stack([t.node_bounds for t in trees])
        ratios = np.stack([t.height_ratios for t in trees]) if need_ratios else None
        return heights, bounds, ratios

    def log_likelihoods(self, include_log_det_jacobian: bool = True) -> np.ndarray:
        """``RootedSBNInstance::LogLikelihoods``; the log-det-Jacobian of the height transform is
        part of the value by default (include_log_det_jacobian_likelihood, reference
        src/phylo_flags.hpp:347-354), which needs initialised time trees."""
        pid, bl = self._wire()
        eng = self._get_engine()
        if not include_log_det_jacobian:
            return eng.log_likelihoods(pid, bl, self._params, rates=self._rates(), rescaling=self._rescaling)
        heights, bounds, _ = self._time_state(False)
        return eng.time_tree_log_likelihoods(pid, bl, heights, bounds, self._params, rates=self._rates(),
                                             rescaling=self._rescaling, include_log_det_jacobian=True)

    def unrooted_log_likelihoods(self) -> np.ndarray:
        """``RootedSBNInstance::UnrootedLogLikelihoods``: no clock rates, no Jacobian."""
        pid, bl = self._wire()
        return self._get_engine().log_likelihoods(pid, bl, self._params, rescaling=self._rescaling)

    def log_det_jacobian_of_height_transform(self) -> np.ndarray:
        pid, _ = self._wire()
        heights, bounds, _ = self._time_state(False)
        return self._get_engine().log_det_jacobian(pid, heights, bounds)

    def gradient_log_det_jacobian_of_height_transform(self) -> np.ndarray:
        pid, _ = self._wire()
        heights, bounds, ratios = self._time_state(True)
        return self._get_engine().gradient_log_det_jacobian(pid, heights, bounds, ratios)

    def phylo_gradients(self, flags: int = DEFAULT_ROOTED_GRADIENT_FLAGS) -> List[PhyloGradient]:
        pid, bl = self._wire()
        heights, bounds, ratios = self._time_state(True)  # the reference throws on uninitialised time trees
        rate_counts = {t.rate_count for t in self.tree_collection.trees}
        if len(rate_counts) != 1:
            raise RuntimeError("All trees of a collection must share one clock parameterisation.")
        out = self._get_engine().time_tree_gradients(pid, bl, heights, bounds, ratios, self._params,
                                                     rates=self._rates(), rate_count=rate_counts.pop(),
                                                     rescaling=self._rescaling, flags=flags)
        return _to_gradients(out)


def _to_gradients(out: Dict[str, np.ndarray]) -> List[PhyloGradient]:
    res = []
    for i, ll in enumerate(out["log_likelihood"]):
        grad = {k: np.atleast_1d(v[i]) for k, v in out.items() if k != "log_likelihood"}
        res.append(PhyloGradient(float(ll), grad))
    return res
