"""ctypes wrapper around the CPU oracle (oracle/libbito_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under bito_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbito_oracle.so")

GRAD_SUBSTITUTION_MODEL = 1
GRAD_SITE_MODEL = 2
GRAD_CLOCK_MODEL = 4
GRAD_STICKBREAKING = 8


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("bito_oracle.c", "time_tree_oracle.c", "bito_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(map(os.path.getmtime, srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.oracle_engine_create.restype = C.c_void_p
        L.oracle_engine_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, ip, dp, C.c_char_p, C.c_int]
        L.oracle_engine_destroy.argtypes = [C.c_void_p]
        L.oracle_engine_param_count.argtypes = [C.c_void_p]
        L.oracle_engine_category_count.argtypes = [C.c_void_p]
        L.oracle_engine_block_count.argtypes = [C.c_void_p]
        L.oracle_engine_block.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, ip, ip]
        L.oracle_engine_last_error.restype = C.c_char_p
        L.oracle_engine_last_error.argtypes = [C.c_void_p]
        L.oracle_engine_log_likelihoods.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp,
                                                    C.c_int, dp]
        L.oracle_engine_gradients.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp, C.c_int,
                                              C.c_int, C.c_double, dp, dp, dp, dp, dp]
        L.oracle_substitution_model.argtypes = [C.c_char_p, dp, dp, dp, dp, dp, dp]
        L.oracle_weibull_rates.argtypes = [C.c_int, C.c_double, dp, dp, dp]
        L.oracle_transition_matrix.argtypes = [dp, dp, dp, C.c_double, dp]
        L.oracle_time_tree_bounds.argtypes = [C.c_int, ip, dp, dp]
        L.oracle_time_tree_from_branch_lengths.argtypes = [C.c_int, ip, dp, dp, dp, dp, dp]
        L.oracle_time_tree_from_height_ratios.argtypes = [C.c_int, ip, dp, dp, dp, dp]
        L.oracle_log_det_jacobian.restype = C.c_double
        L.oracle_log_det_jacobian.argtypes = [C.c_int, ip, dp, dp]
        L.oracle_height_gradient.argtypes = [C.c_int, ip, dp, dp, dp]
        L.oracle_ratio_gradient_of_height_gradient.argtypes = [C.c_int, ip, dp, dp, dp, dp, dp]
        L.oracle_gradient_log_det_jacobian.argtypes = [C.c_int, ip, dp, dp, dp, dp]
        L.oracle_ratio_gradient_of_branch_gradient.argtypes = [C.c_int, ip, dp, dp, dp, dp, dp, C.c_int, dp]
        _lib = L
    return _lib


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class OracleError(RuntimeError):
    pass


class OracleEngine:
    """Mirror of the reference ``Engine`` (src/engine.hpp:26-68) on the CPU oracle."""

    def __init__(self, substitution: str, site: str, clock: str, patterns: np.ndarray, weights: np.ndarray,
                 thread_count: int = 1, use_tip_states: bool = True):
        self._h = None
        L = lib()
        self.patterns = np.ascontiguousarray(patterns, dtype=np.int32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        n, P = self.patterns.shape
        err = C.create_string_buffer(256)
        h = L.oracle_engine_create(substitution.encode(), site.encode(), clock.encode(), thread_count,
                                   int(use_tip_states), n, P, _ip(self.patterns), _dp(self.weights), err, 256)
        if not h:
            raise OracleError(err.value.decode())
        self._h = h
        self.taxon_count = n
        self.param_count = L.oracle_engine_param_count(h)
        self.category_count = L.oracle_engine_category_count(h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_engine_destroy(self._h)
            self._h = None

    def block_map(self) -> Dict[str, Tuple[int, int]]:
        L = lib()
        out = {}
        name = C.create_string_buffer(64)
        s, ln = C.c_int(), C.c_int()
        for i in range(L.oracle_engine_block_count(self._h)):
            L.oracle_engine_block(self._h, i, name, 64, C.byref(s), C.byref(ln))
            out[name.value.decode()] = (s.value, ln.value)
        return out

    def default_params(self, tree_count: int) -> np.ndarray:
        """PhyloModel defaults of the reference: GTR/HKY rates & freqs equal, shape 1, clock rate 1."""
        p = np.zeros((tree_count, self.param_count))
        for key, (s, ln) in self.block_map().items():
            if key == "substitution_model_frequencies":
                p[:, s:s + ln] = 0.25
            elif key == "substitution_model_rates":
                p[:, s:s + ln] = 1.0 / 6 if ln == 6 else 1.0
            elif key in ("Weibull_shape", "clock_rate"):
                p[:, s:s + ln] = 1.0
        return p

    def _prep(self, parent_ids, branch_lengths, rates, params):
        parent_ids = np.ascontiguousarray(parent_ids, dtype=np.int32)
        branch_lengths = np.ascontiguousarray(branch_lengths, dtype=np.float64)
        T, M = branch_lengths.shape
        assert parent_ids.shape == (T, M - 1)
        rooted = int(M == 2 * self.taxon_count - 1)
        if rates is not None:
            rates = np.ascontiguousarray(rates, dtype=np.float64)
            assert rates.shape == (T, M - 1)
        if params is None:
            params = self.default_params(T)
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(T, self.param_count)
        return parent_ids, branch_lengths, rates, params, T, M, rooted

    def log_likelihoods(self, parent_ids, branch_lengths, params=None, rates=None, rescaling=False):
        parent_ids, branch_lengths, rates, params, T, M, rooted = self._prep(parent_ids, branch_lengths, rates, params)
        out = np.zeros(T)
        rc = lib().oracle_engine_log_likelihoods(self._h, T, rooted, M, _ip(parent_ids), _dp(branch_lengths),
                                                 _dp(rates), _dp(params), int(rescaling), _dp(out))
        if rc:
            raise OracleError(lib().oracle_engine_last_error(self._h).decode())
        return out

    def gradients(self, parent_ids, branch_lengths, params=None, rates=None, rescaling=False, flags=0,
                  fd_delta=1e-6):
        parent_ids, branch_lengths, rates, params, T, M, rooted = self._prep(parent_ids, branch_lengths, rates, params)
        N = 2 * self.taxon_count - 1
        ll = np.zeros(T)
        branch = np.zeros((T, N))
        bm = self.block_map()
        sub_len = bm["entire_substitution"][1] if "entire_substitution" in bm else 0
        site = np.zeros(T) if flags & GRAD_SITE_MODEL else None
        subst = np.zeros((T, max(sub_len, 1))) if flags & GRAD_SUBSTITUTION_MODEL else None
        clock = np.zeros(T) if flags & GRAD_CLOCK_MODEL else None
        rc = lib().oracle_engine_gradients(self._h, T, rooted, M, _ip(parent_ids), _dp(branch_lengths), _dp(rates),
                                           _dp(params), int(rescaling), flags, fd_delta, _dp(ll), _dp(branch),
                                           _dp(site), _dp(subst), _dp(clock))
        if rc:
            raise OracleError(lib().oracle_engine_last_error(self._h).decode())
        out = {"log_likelihood": ll, "branch_lengths": branch}
        if site is not None:
            out["site_model"] = site
        if subst is not None:
            if flags & GRAD_STICKBREAKING:
                rl = bm["substitution_model_rates"][1]
                sub_len = (rl - 1 if rl == 6 else rl) + 3
            out["substitution_model"] = subst[:, :sub_len]
        if clock is not None:
            out["clock_model"] = clock
        return out


def substitution_model(name: str, params: Optional[np.ndarray] = None):
    Q, V, Vi = np.zeros(16), np.zeros(16), np.zeros(16)
    lam, pi = np.zeros(4), np.zeros(4)
    p = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
    rc = lib().oracle_substitution_model(name.encode(), _dp(p), _dp(Q), _dp(V), _dp(Vi), _dp(lam), _dp(pi))
    if rc:
        raise OracleError(f"substitution model error {rc}")
    return Q.reshape(4, 4), V.reshape(4, 4), Vi.reshape(4, 4), lam, pi


def weibull_rates(category_count: int, shape: float):
    r, w, d = np.zeros(category_count), np.zeros(category_count), np.zeros(category_count)
    lib().oracle_weibull_rates(category_count, shape, _dp(r), _dp(w), _dp(d))
    return r, w, d


def transition_matrix(V, Vinv, lam, t: float) -> np.ndarray:
    P = np.zeros(16)
    V = np.ascontiguousarray(V, dtype=np.float64).reshape(-1)
    Vinv = np.ascontiguousarray(Vinv, dtype=np.float64).reshape(-1)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    lib().oracle_transition_matrix(_dp(V), _dp(Vinv), _dp(lam), float(t), _dp(P))
    return P.reshape(4, 4)


class TimeTree:
    """One RootedTree with its time parameterisation (reference src/rooted_tree.hpp): built from
    branch lengths + tip dates, or re-set from height ratios."""

    def __init__(self, parent_ids, branch_lengths, tip_dates):
        self.parent_ids = np.ascontiguousarray(parent_ids, dtype=np.int32)
        self.n = (self.parent_ids.shape[0] + 2) // 2
        N = 2 * self.n - 1
        self.tip_dates = np.ascontiguousarray(tip_dates, dtype=np.float64)
        self.branch_lengths = np.ascontiguousarray(branch_lengths, dtype=np.float64).copy()
        self.node_bounds = np.zeros(N)
        self.node_heights = np.zeros(N)
        self.height_ratios = np.zeros(self.n - 1)
        rc = lib().oracle_time_tree_from_branch_lengths(self.n, _ip(self.parent_ids), _dp(self.branch_lengths),
                                                        _dp(self.tip_dates), _dp(self.node_bounds),
                                                        _dp(self.node_heights), _dp(self.height_ratios))
        if rc:
            raise RuntimeError("Tree isn't time-calibrated in RootedTree::InitializeTimeTreeUsingBranchLengths.")

    def initialize_time_tree_using_height_ratios(self, ratios):
        self.height_ratios = np.ascontiguousarray(ratios, dtype=np.float64).copy()
        lib().oracle_time_tree_from_height_ratios(self.n, _ip(self.parent_ids), _dp(self.node_bounds),
                                                  _dp(self.height_ratios), _dp(self.node_heights),
                                                  _dp(self.branch_lengths))

    def log_det_jacobian(self) -> float:
        return lib().oracle_log_det_jacobian(self.n, _ip(self.parent_ids), _dp(self.node_heights),
                                             _dp(self.node_bounds))

    def gradient_log_det_jacobian(self) -> np.ndarray:
        out = np.zeros(self.n - 1)
        lib().oracle_gradient_log_det_jacobian(self.n, _ip(self.parent_ids), _dp(self.node_heights),
                                               _dp(self.node_bounds), _dp(self.height_ratios), _dp(out))
        return out

    def ratio_gradient_of_height_gradient(self, height_gradient) -> np.ndarray:
        hg = np.ascontiguousarray(height_gradient, dtype=np.float64)
        out = np.zeros(self.n - 1)
        lib().oracle_ratio_gradient_of_height_gradient(self.n, _ip(self.parent_ids), _dp(self.node_heights),
                                                       _dp(self.node_bounds), _dp(self.height_ratios), _dp(hg),
                                                       _dp(out))
        return out

    def ratio_gradient_of_branch_gradient(self, branch_gradient, rates, include_log_det_jacobian=True):
        bg = np.ascontiguousarray(branch_gradient, dtype=np.float64)
        rt = np.ascontiguousarray(rates, dtype=np.float64)
        out = np.zeros(self.n - 1)
        lib().oracle_ratio_gradient_of_branch_gradient(self.n, _ip(self.parent_ids), _dp(self.node_heights),
                                                       _dp(self.node_bounds), _dp(self.height_ratios), _dp(rt),
                                                       _dp(bg), int(include_log_det_jacobian), _dp(out))
        return out
