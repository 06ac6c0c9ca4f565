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
from .site_pattern import SitePattern


# PhyloGradientFlagOptions of the reference all default to "on", stick-breaking included
# (src/phylo_flags.hpp:322-343); ratios_root_height is a time-tree transform outside the GPU path.
DEFAULT_GRADIENT_FLAGS = (_capi.GRAD_SUBSTITUTION_MODEL | _capi.GRAD_SITE_MODEL | _capi.GRAD_CLOCK_MODEL |
                          _capi.GRAD_STICKBREAKING)


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

    def parent_id_vector(self) -> List[int]:
        return [int(x) for x in self._parent_ids]


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
                                     tree_count_option: Optional[int] = None, device_id: Optional[int] = None):
        if thread_count == 0:
            raise RuntimeError("Thread count needs to be strictly positive.")
        if not self._alignment:
            raise RuntimeError("Load an alignment into your instance before preparing for phylogenetic likelihood.")
        site_pattern = SitePattern(self._alignment, self._taxon_names)
        if self._engine is not None:
            self._engine.close()
        self._engine = Engine(model_specification, site_pattern.patterns, site_pattern.weights,
                              device_id=self._device_id if device_id is None else device_id,
                              use_tip_states=use_tip_states)
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
    """pybito ``rooted_instance`` (reference src/pybito.cpp:289-421); time-tree
    parameterisations (height ratios, log-det-Jacobian) are not on the GPU path."""
    _rooted = True

    def _rates(self) -> np.ndarray:
        return np.stack([np.asarray(t.rates, dtype=np.float64) for t in self.tree_collection.trees])

    def log_likelihoods(self) -> np.ndarray:
        pid, bl = self._wire()
        return self._get_engine().log_likelihoods(pid, bl, self._params, rates=self._rates(),
                                                  rescaling=self._rescaling)

    def phylo_gradients(self, flags: int = DEFAULT_GRADIENT_FLAGS) -> List[PhyloGradient]:
        pid, bl = self._wire()
        out = self._get_engine().gradients(pid, bl, self._params, rates=self._rates(), rescaling=self._rescaling,
                                           flags=flags)
        return _to_gradients(out)


def _to_gradients(out: Dict[str, np.ndarray]) -> List[PhyloGradient]:
    res = []
    for i, ll in enumerate(out["log_likelihood"]):
        grad = {k: np.atleast_1d(v[i]) for k, v in out.items() if k != "log_likelihood"}
        res.append(PhyloGradient(float(ll), grad))
    return res
