"""Host-side mirror of the reference ``Engine`` on top of the C ABI.

``Engine.log_likelihoods`` / ``Engine.gradients`` take a whole tree collection in
wire format (parent-id vectors + branch lengths + one parameter row per tree) and
return what ``Engine::LogLikelihoods`` / ``Engine::Gradients`` return in the
reference (src/engine.hpp:33-61).  All arithmetic happens in libbito_amd.so on the
GPU; this module only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _capi


class BitoAmdError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the reference raises
    std::runtime_error -> Python RuntimeError through pybind11)."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


@dataclass
class PhyloModelSpecification:
    """reference src/phylo_model.hpp:13-17 / pybito ``PhyloModelSpecification``."""
    substitution: str
    site: str
    clock: str


@dataclass
class PhyloGradient:
    """reference src/phylo_gradient.hpp:10-35 / pybito ``PhyloGradient``."""
    log_likelihood: float
    gradient: Dict[str, np.ndarray] = field(default_factory=dict)


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Engine:
    def __init__(self, model: PhyloModelSpecification, patterns: np.ndarray, weights: np.ndarray,
                 device_id: int = 0, use_tip_states: bool = True, arena_bytes: int = 0, devices=None,
                 host_threads: int = 0):
        """``devices``: HIP ordinals of the GPUs this engine drives (default: ``[device_id]``); the counterpart of
        the reference's ``thread_count`` (src/engine.cpp:10-31).  Blocking calls shard their trees over them.
        ``host_threads``: host threads a blocking call may use for checking and packing its inputs and copying its
        results out (0: min(8, usable CPUs); 1: the calling thread alone)."""
        self._h = None
        L = _capi.lib()
        self.patterns = np.ascontiguousarray(patterns, dtype=np.int32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        if self.patterns.ndim != 2 or self.patterns.shape[1] != self.weights.shape[0]:
            raise BitoAmdError(_capi.ERR_BAD_ARG, "patterns must be [taxon_count][pattern_count] and match weights")
        n, P = self.patterns.shape
        devs = [int(device_id)] if devices is None else [int(d) for d in devices]
        if not devs:
            raise BitoAmdError(_capi.ERR_BAD_ARG, "Device count needs to be strictly positive.")
        dev_array = (C.c_int32 * len(devs))(*devs)
        spec = _capi.EngineSpec(devs[0], int(use_tip_states), arena_bytes, len(devs), int(host_threads), dev_array)
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.bito_amd_engine_create(C.byref(spec), model.substitution.encode(), model.site.encode(),
                                      model.clock.encode(), n, P, _ip(self.patterns), _dp(self.weights),
                                      C.byref(h), err, 512)
        if rc:
            raise BitoAmdError(rc, err.value.decode())
        self._h = h
        self.model = model
        self.taxon_count = n
        self.pattern_count = P
        self.param_count = L.bito_amd_engine_param_count(h)
        self.category_count = L.bito_amd_engine_category_count(h)
        self.state_count = L.bito_amd_engine_state_count(h)
        self.tree_count = 0
        self._node_count = 2 * n - 1
        self._resident_shape = None
        self.device_count = L.bito_amd_engine_device_count(h)

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib().bito_amd_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- BlockSpecification -------------------------------------------------
    def block_map(self) -> Dict[str, Tuple[int, int]]:
        """BlockSpecification::GetMap; fixed for the engine's lifetime, so asked of the library once."""
        cached = getattr(self, "_block_map", None)
        if cached is not None:
            return dict(cached)
        L = _capi.lib()
        out = {}
        name = C.create_string_buffer(64)
        s, ln = C.c_int32(), C.c_int32()
        for i in range(L.bito_amd_engine_block_count(self._h)):
            L.bito_amd_engine_block(self._h, i, name, 64, C.byref(s), C.byref(ln))
            out[name.value.decode()] = (s.value, ln.value)
        self._block_map = dict(out)
        return out

    def default_params(self, tree_count: int) -> np.ndarray:
        """Defaults of the reference's model classes (GTR rates 1/6 and frequencies 1/4,
        src/substitution_model.hpp:82-89; HKY kappa 1; Weibull shape 1; clock rate 1)."""
        p = np.zeros((tree_count, self.param_count))
        for key, (s, ln) in self.block_map().items():
            if key == "substitution_model_frequencies":
                p[:, s:s + ln] = 0.25
            elif key == "substitution_model_rates":
                p[:, s:s + ln] = 1.0 / 6 if ln == 6 else 1.0  # GTR rates; HKY kappa; GY94 kappa, omega
            elif key in ("Weibull_shape", "clock_rate"):
                p[:, s:s + ln] = 1.0
        return p

    # -- helpers -------------------------------------------------------------
    def _check(self, rc: int):
        if rc:
            raise BitoAmdError(rc, _capi.lib().bito_amd_engine_last_error(self._h).decode())

    def _prep(self, parent_ids, branch_lengths, rates, params):
        parent_ids = np.ascontiguousarray(parent_ids, dtype=np.int32)
        branch_lengths = np.ascontiguousarray(branch_lengths, dtype=np.float64)
        if branch_lengths.ndim != 2 or parent_ids.ndim != 2:
            raise BitoAmdError(_capi.ERR_BAD_ARG, "parent_ids and branch_lengths must be 2-D (one row per tree)")
        T, M = branch_lengths.shape
        if parent_ids.shape != (T, M - 1):
            raise BitoAmdError(_capi.ERR_BAD_ARG, "parent_ids must have one entry fewer per tree than branch_lengths")
        rooted = int(M == 2 * self.taxon_count - 1)
        if rates is not None:
            rates = np.ascontiguousarray(rates, dtype=np.float64)
            if rates.shape != (T, M - 1):
                raise BitoAmdError(_capi.ERR_BAD_ARG, "rates must be [tree_count][node_count-1]")
        if params is None:
            params = self.default_params(T)
        params = np.ascontiguousarray(params, dtype=np.float64)
        # "We param_matrix needs as many rows as we have trees." (reference fat_beagle.hpp:170-171)
        if params.shape != (T, self.param_count):
            raise BitoAmdError(_capi.ERR_BAD_ARG,
                               f"param matrix needs shape ({T}, {self.param_count}), got {params.shape}")
        return parent_ids, branch_lengths, rates, params, T, M, rooted

    # -- Engine::LogLikelihoods / Gradients -----------------------------------
    def log_likelihoods(self, parent_ids, branch_lengths, params=None, rates=None, rescaling=False) -> np.ndarray:
        parent_ids, branch_lengths, rates, params, T, M, rooted = self._prep(parent_ids, branch_lengths, rates, params)
        out = np.zeros(T)
        self._check(_capi.lib().bito_amd_engine_log_likelihoods(
            self._h, T, rooted, M, _ip(parent_ids), _dp(branch_lengths), _dp(rates), _dp(params), int(rescaling),
            _dp(out)))
        self._set_resident(T, M)
        return out

    def log_likelihoods_into(self, parent_ids: np.ndarray, branch_lengths: np.ndarray, params: np.ndarray,
                             out: np.ndarray, rescaling: bool = False):
        """``log_likelihoods`` without per-call allocations or conversions, for callers that evaluate in a loop:
        C-contiguous arrays of the exact dtypes (int32 parent ids, float64 elsewhere), unrooted or rooted by shape,
        results written into ``out`` [T]."""
        T, M = self._lean_check(parent_ids, branch_lengths, params, out)
        self._check(_capi.lib().bito_amd_engine_log_likelihoods(
            self._h, T, int(M == 2 * self.taxon_count - 1), M, _ip(parent_ids), _dp(branch_lengths), None,
            _dp(params) if self.param_count else None, int(rescaling), _dp(out)))
        self._set_resident(T, M)

    def gradients_into(self, parent_ids: np.ndarray, branch_lengths: np.ndarray, params: np.ndarray,
                       out_ll: np.ndarray, out_branch: np.ndarray, rescaling: bool = False):
        """``gradients`` (log-likelihoods + branch-length gradients) into caller-owned arrays ``out_ll`` [T] and
        ``out_branch`` [T][2n-1]; the span of the reference's ``Engine::Gradients`` call -- host trees and
        parameter rows in, host results out (src/fat_beagle.hpp:173-181) -- with nothing else around it."""
        T, M = self._lean_check(parent_ids, branch_lengths, params, out_ll)
        if out_branch.dtype != np.float64 or not out_branch.flags.c_contiguous or \
                out_branch.shape != (T, 2 * self.taxon_count - 1):
            raise BitoAmdError(_capi.ERR_BAD_ARG, "out_branch must be a C-contiguous float64 [T][2n-1] array")
        self._check(_capi.lib().bito_amd_engine_gradients(
            self._h, T, int(M == 2 * self.taxon_count - 1), M, _ip(parent_ids), _dp(branch_lengths), None,
            _dp(params) if self.param_count else None, int(rescaling), 0, 0.0, _dp(out_ll), _dp(out_branch), None,
            None, None))
        self._set_resident(T, M)

    def _lean_check(self, parent_ids, branch_lengths, params, out_ll):
        ok = (parent_ids.dtype == np.int32 and branch_lengths.dtype == np.float64 and out_ll.dtype == np.float64 and
              parent_ids.flags.c_contiguous and branch_lengths.flags.c_contiguous and out_ll.flags.c_contiguous and
              branch_lengths.ndim == 2 and parent_ids.ndim == 2)
        if not ok:
            raise BitoAmdError(_capi.ERR_BAD_ARG, "the *_into calls take C-contiguous int32 / float64 arrays as they are")
        T, M = branch_lengths.shape
        if parent_ids.shape != (T, M - 1) or out_ll.shape != (T,):
            raise BitoAmdError(_capi.ERR_BAD_ARG, "parent_ids must be [T][M-1], branch_lengths [T][M], out [T]")
        if self.param_count and (params is None or params.dtype != np.float64 or not params.flags.c_contiguous or
                                 params.shape != (T, self.param_count)):
            raise BitoAmdError(_capi.ERR_BAD_ARG, f"param matrix needs shape ({T}, {self.param_count}), float64")
        return T, M

    def _set_resident(self, T: int, M: int):
        """Every call that takes a tree collection leaves it resident on the device(s): ``update`` / ``run`` /
        ``download`` then refer to it (the C side keeps the same record)."""
        self.tree_count = T
        self._node_count = 2 * self.taxon_count - 1
        self._resident_shape = (T, M)

    def gradients(self, parent_ids, branch_lengths, params=None, rates=None, rescaling=False, flags=0,
                  fd_delta=1e-6) -> Dict[str, np.ndarray]:
        parent_ids, branch_lengths, rates, params, T, M, rooted = self._prep(parent_ids, branch_lengths, rates, params)
        N = 2 * self.taxon_count - 1
        ll = np.zeros(T)
        branch = np.zeros((T, N))
        bm = self.block_map()
        sub_len = bm["entire_substitution"][1] if "entire_substitution" in bm else 0
        site = np.zeros(T) if flags & _capi.GRAD_SITE_MODEL else None
        subst = np.zeros((T, max(sub_len, 1))) if flags & _capi.GRAD_SUBSTITUTION_MODEL else None
        clock = np.zeros(T) if flags & _capi.GRAD_CLOCK_MODEL else None
        self._check(_capi.lib().bito_amd_engine_gradients(
            self._h, T, rooted, M, _ip(parent_ids), _dp(branch_lengths), _dp(rates), _dp(params), int(rescaling),
            flags, fd_delta, _dp(ll), _dp(branch), _dp(site), _dp(subst), _dp(clock)))
        self._set_resident(T, M)
        return self._package(flags, rooted, ll, branch, site, subst, clock)

    def _package(self, flags, rooted, ll, branch, site, subst, clock) -> Dict[str, np.ndarray]:
        bm = self.block_map()
        sub_len = bm["entire_substitution"][1] if "entire_substitution" in bm else 0
        out = {"log_likelihood": ll, "branch_lengths": branch}
        if site is not None and self.category_count > 1:
            out["site_model"] = site
        if subst is not None and sub_len:
            if flags & _capi.GRAD_STICKBREAKING:
                rl = bm["substitution_model_rates"][1]
                sub_len = (rl - 1 if rl == 6 else rl) + 3
            out["substitution_model"] = subst[:, :sub_len]
            # rates first, then frequencies (reference src/fat_beagle.cpp:528-534)
            n_freq = 3 if flags & _capi.GRAD_STICKBREAKING else 4
            out["substitution_model_rates"] = subst[:, :sub_len - n_freq]
            out["substitution_model_frequencies"] = subst[:, sub_len - n_freq:sub_len]
        if clock is not None and rooted:
            out["clock_model"] = clock
        return out

    # -- time trees (reference src/rooted_tree.cpp, src/rooted_gradient_transforms.cpp) --------
    def _node_array(self, a, shape, what: str) -> np.ndarray:
        """A float64 array of exactly ``shape``: the C ABI takes bare pointers and reads T*(2n-1) (node
        arrays), T*(n-1) (height ratios) or n (tip dates) doubles from it, so the shape is checked here."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != tuple(shape):
            raise BitoAmdError(_capi.ERR_BAD_ARG, f"{what} must have shape {tuple(shape)}, got {a.shape}")
        return a

    def _tt(self, parent_ids, **arrays):
        """parent ids of rooted trees + named arrays; the name's prefix selects the expected shape:
        ``node_*`` / ``branch_*`` -> (T, 2n-1), ``height_ratios`` / ``height_gradient`` -> (T, n-1) (one entry
        per internal node), ``tip_dates`` -> (n,)."""
        pid = np.ascontiguousarray(parent_ids, dtype=np.int32)
        n, N = self.taxon_count, 2 * self.taxon_count - 1
        if pid.ndim != 2 or pid.shape[1] != N - 1:
            raise BitoAmdError(_capi.ERR_BAD_ARG, f"parent_ids must be [tree_count][{N - 1}] (rooted trees)")
        T = pid.shape[0]
        out = []
        for name, a in arrays.items():
            shape = (n,) if name == "tip_dates" else (T, n - 1) if name.startswith("height_") else (T, N)
            out.append(self._node_array(a, shape, name))
        return (pid, T) + tuple(out)

    def time_trees_from_branch_lengths(self, parent_ids, branch_lengths, tip_dates):
        """-> (node_bounds, node_heights, height_ratios); RuntimeError for a tree that is not
        time-calibrated (RootedTree::InitializeTimeTreeUsingBranchLengths)."""
        pid, T, bl, dates = self._tt(parent_ids, branch_lengths=branch_lengths, tip_dates=tip_dates)
        n, N = self.taxon_count, 2 * self.taxon_count - 1
        bounds, heights, ratios = np.zeros((T, N)), np.zeros((T, N)), np.zeros((T, n - 1))
        self._check(_capi.lib().bito_amd_engine_time_trees_from_branch_lengths(
            self._h, T, _ip(pid), _dp(bl), _dp(dates), _dp(bounds), _dp(heights), _dp(ratios)))
        return bounds, heights, ratios

    def time_trees_from_height_ratios(self, parent_ids, node_bounds, height_ratios):
        """-> (node_heights, branch_lengths) (RootedTree::InitializeTimeTreeUsingHeightRatios)."""
        pid, T, bounds, ratios = self._tt(parent_ids, node_bounds=node_bounds, height_ratios=height_ratios)
        N = 2 * self.taxon_count - 1
        heights, bl = np.zeros((T, N)), np.zeros((T, N))
        self._check(_capi.lib().bito_amd_engine_time_trees_from_height_ratios(
            self._h, T, _ip(pid), _dp(bounds), _dp(ratios), _dp(heights), _dp(bl)))
        return heights, bl

    def log_det_jacobian(self, parent_ids, node_heights, node_bounds) -> np.ndarray:
        pid, T, heights, bounds = self._tt(parent_ids, node_heights=node_heights, node_bounds=node_bounds)
        out = np.zeros(T)
        self._check(_capi.lib().bito_amd_engine_log_det_jacobian(self._h, T, _ip(pid), _dp(heights), _dp(bounds),
                                                                 _dp(out)))
        return out

    def gradient_log_det_jacobian(self, parent_ids, node_heights, node_bounds, height_ratios) -> np.ndarray:
        pid, T, heights, bounds, ratios = self._tt(parent_ids, node_heights=node_heights, node_bounds=node_bounds, height_ratios=height_ratios)
        out = np.zeros((T, self.taxon_count - 1))
        self._check(_capi.lib().bito_amd_engine_gradient_log_det_jacobian(
            self._h, T, _ip(pid), _dp(heights), _dp(bounds), _dp(ratios), _dp(out)))
        return out

    def ratio_gradient_of_height_gradient(self, parent_ids, node_heights, node_bounds, height_ratios,
                                          height_gradient) -> np.ndarray:
        pid, T, heights, bounds, ratios, hg = self._tt(
            parent_ids, node_heights=node_heights, node_bounds=node_bounds, height_ratios=height_ratios,
            height_gradient=height_gradient)
        out = np.zeros((T, self.taxon_count - 1))
        self._check(_capi.lib().bito_amd_engine_ratio_gradient_of_height_gradient(
            self._h, T, _ip(pid), _dp(heights), _dp(bounds), _dp(ratios), _dp(hg), _dp(out)))
        return out

    def time_tree_log_likelihoods(self, parent_ids, branch_lengths, node_heights, node_bounds, params=None,
                                  rates=None, rescaling=False, include_log_det_jacobian=True) -> np.ndarray:
        parent_ids, branch_lengths, rates, params, T, M, rooted = self._prep(parent_ids, branch_lengths, rates, params)
        if not rooted:
            raise BitoAmdError(_capi.ERR_BAD_TREE, "time trees are rooted")
        N = 2 * self.taxon_count - 1
        heights = self._node_array(node_heights, (T, N), "node_heights")
        bounds = self._node_array(node_bounds, (T, N), "node_bounds")
        out = np.zeros(T)
        self._check(_capi.lib().bito_amd_engine_time_tree_log_likelihoods(
            self._h, T, _ip(parent_ids), _dp(branch_lengths), _dp(rates), _dp(heights), _dp(bounds), _dp(params),
            int(rescaling), int(include_log_det_jacobian), _dp(out)))
        self._set_resident(T, M)
        return out

    def time_tree_gradients(self, parent_ids, branch_lengths, node_heights, node_bounds, height_ratios, params=None,
                            rates=None, rate_count=1, rescaling=False, flags=0, fd_delta=1e-6):
        """``Engine::Gradients(RootedTreeCollection)`` for time trees: the ``gradients`` outputs plus
        ``ratios_root_height`` and a strict or per-branch ``clock_model``."""
        parent_ids, branch_lengths, rates, params, T, M, rooted = self._prep(parent_ids, branch_lengths, rates, params)
        if not rooted:
            raise BitoAmdError(_capi.ERR_BAD_TREE, "time trees are rooted")
        n, N = self.taxon_count, 2 * self.taxon_count - 1
        heights = self._node_array(node_heights, (T, N), "node_heights")
        bounds = self._node_array(node_bounds, (T, N), "node_bounds")
        ratios = self._node_array(height_ratios, (T, n - 1), "height_ratios")
        ll, branch = np.zeros(T), np.zeros((T, N))
        bm = self.block_map()
        sub_len = bm["entire_substitution"][1] if "entire_substitution" in bm else 0
        site = np.zeros(T) if flags & _capi.GRAD_SITE_MODEL else None
        subst = np.zeros((T, max(sub_len, 1))) if flags & _capi.GRAD_SUBSTITUTION_MODEL else None
        clock = np.zeros((T, 1 if rate_count == 1 else N - 1)) if flags & _capi.GRAD_CLOCK_MODEL else None
        ratio_grad = np.zeros((T, n - 1)) if flags & _capi.GRAD_RATIOS_ROOT_HEIGHT else None
        self._check(_capi.lib().bito_amd_engine_time_tree_gradients(
            self._h, T, _ip(parent_ids), _dp(branch_lengths), _dp(rates), int(rate_count), _dp(heights), _dp(bounds),
            _dp(ratios), _dp(params), int(rescaling), flags, fd_delta, _dp(ll), _dp(branch), _dp(site), _dp(subst),
            _dp(clock), _dp(ratio_grad)))
        self._set_resident(T, M)
        out = self._package(flags, rooted, ll, branch, site, subst, clock)
        if ratio_grad is not None:
            out["ratios_root_height"] = ratio_grad
        return out

    # -- HBM-resident batch ---------------------------------------------------
    def upload(self, parent_ids, branch_lengths, params=None, rates=None):
        parent_ids, branch_lengths, rates, params, T, M, rooted = self._prep(parent_ids, branch_lengths, rates, params)
        self._check(_capi.lib().bito_amd_engine_upload(self._h, T, rooted, M, _ip(parent_ids), _dp(branch_lengths),
                                                       _dp(rates), _dp(params)))
        self._set_resident(T, M)

    def update(self, branch_lengths=None, params=None):
        """New branch lengths and/or parameter rows for the resident batch; the C side copies
        T*in_node_count and T*param_count doubles, so the shapes must be the uploaded ones."""
        shape = getattr(self, "_resident_shape", None)
        if shape is None:
            raise BitoAmdError(_capi.ERR_STATE, "no batch is resident: call upload first")
        bl = None if branch_lengths is None else np.ascontiguousarray(branch_lengths, dtype=np.float64)
        pr = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
        if bl is not None and bl.shape != shape:
            raise BitoAmdError(_capi.ERR_BAD_ARG, f"branch_lengths must have the uploaded shape {shape}, got {bl.shape}")
        if pr is not None and pr.shape != (shape[0], self.param_count):
            raise BitoAmdError(_capi.ERR_BAD_ARG,
                               f"param matrix needs shape ({shape[0]}, {self.param_count}), got {pr.shape}")
        self._check(_capi.lib().bito_amd_engine_update(self._h, _dp(bl), _dp(pr)))

    def run(self, want_gradient: bool, rescaling: bool = False):
        self._check(_capi.lib().bito_amd_engine_run(self._h, int(want_gradient), int(rescaling)))

    def sync(self):
        self._check(_capi.lib().bito_amd_engine_sync(self._h))

    def download(self, want_gradient: bool = True):
        ll = np.zeros(self.tree_count)
        grad = np.zeros((self.tree_count, self._node_count)) if want_gradient else None
        self._check(_capi.lib().bito_amd_engine_download(self._h, ll.ctypes.data,
                                                         None if grad is None else grad.ctypes.data))
        return ll, grad

    def download_to(self, ll_ptr: int, grad_ptr: Optional[int]):
        """Copy results to raw (host or device) addresses, e.g. torch tensor data_ptr()."""
        self._check(_capi.lib().bito_amd_engine_download(self._h, ll_ptr, grad_ptr))

    def download_async(self, ll_ptr: int, grad_ptr: Optional[int]):
        """The same copies enqueued on the engine's stream without waiting (see ``stream_handle``)."""
        self._check(_capi.lib().bito_amd_engine_download_async(self._h, ll_ptr, grad_ptr))

    def results_async(self, consumer_stream: int):
        """Device addresses (log-likelihoods, gradients) of the last pass enqueued, after making the HIP stream
        ``consumer_stream`` wait for it; nothing is copied and nothing enqueued on the engine's stream.  The
        log-likelihood buffer is one of a ring of four (see ``bito_amd_engine_results_async``)."""
        ll, grad = C.c_void_p(), C.c_void_p()
        self._check(_capi.lib().bito_amd_engine_results_async(self._h, C.c_void_p(consumer_stream), C.byref(ll), C.byref(grad)))
        return int(ll.value or 0), int(grad.value or 0)

    def stream_handle(self) -> int:
        """The engine's hipStream_t as an integer, e.g. for ``torch.cuda.ExternalStream``."""
        return int(_capi.lib().bito_amd_engine_stream(self._h) or 0)

    def set_kernel(self, kernel: int):
        self._check(_capi.lib().bito_amd_engine_set_kernel(self._h, kernel))

    def read_general_model(self, tree: int) -> Dict[str, np.ndarray]:
        """Diagnostics (general-state kernels): the model record the set-up kernel built for one tree."""
        buf = np.zeros(3 * 4096 + 3 * 64 + 48)
        self._check(_capi.lib().bito_amd_engine_read_general_model(self._h, tree, _dp(buf), buf.size))
        m = 3 * 4096
        return {"V": buf[:4096].reshape(64, 64), "Vinv": buf[4096:8192].reshape(64, 64),
                "Q": buf[8192:m].reshape(64, 64), "lambda": buf[m:m + 64], "pi": buf[m + 64:m + 128],
                "cat_rate": buf[m + 192:m + 208], "cat_weight": buf[m + 208:m + 224]}

    def kernel_name(self) -> str:
        return _capi.lib().bito_amd_engine_kernel_name(self._h).decode()

    def kernel_form(self) -> str:
        """how the last walk_pipe_kernel pass ran (waves per SIMD, pattern groups per wave, classes); diagnostics"""
        return _capi.lib().bito_amd_engine_kernel_form(self._h).decode()

    def kernel_timing(self, enable: bool):
        self._check(_capi.lib().bito_amd_engine_kernel_timing(self._h, int(enable)))

    def kernel_elapsed(self):
        kern, launches = C.c_double(), C.c_int32()
        self._check(_capi.lib().bito_amd_engine_kernel_elapsed(self._h, C.byref(kern), C.byref(launches)))
        return kern.value, launches.value

    def kernel_span_sum(self) -> float:
        """The launches' spans of the last ``kernel_elapsed`` added up, in ms (what a profiler's per-launch durations
        sum to; ``kernel_elapsed`` itself reports the union of the spans: overlapping chunks counted once)."""
        return float(_capi.lib().bito_amd_engine_kernel_span_sum(self._h))

    def time_runs(self, want_gradient: bool, rescaling: bool, steps: int):
        total, kern, launches = C.c_double(), C.c_double(), C.c_int32()
        self._check(_capi.lib().bito_amd_engine_time_runs(self._h, int(want_gradient), int(rescaling), steps,
                                                          C.byref(total), C.byref(kern), C.byref(launches)))
        return total.value, kern.value, launches.value


def version() -> str:
    return _capi.lib().bito_amd_version().decode()
