"""ctypes wrapper around the general-state-count CPU oracle (oracle/libgs_oracle.so, gs_oracle.c):
the 61-state codon model of BASELINE config 5, and the same code path at S = 4 (GTR) for pinning it
to the reference's goldens.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline legs of
the measurement scripts.  Nothing under bito_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs_oracle.so")
MS = 64  # padded matrix dimension of gs_oracle.c

_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "gs_oracle.c")
        if not os.path.exists(_LIB_PATH) or (os.path.exists(src) and os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libgs_oracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(_LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.gs_engine_create.restype = C.c_void_p
        L.gs_engine_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, ip, dp, C.c_char_p, C.c_int]
        L.gs_engine_destroy.argtypes = [C.c_void_p]
        for f in ("gs_engine_param_count", "gs_engine_state_count", "gs_engine_category_count"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.gs_engine_last_error.restype = C.c_char_p
        L.gs_engine_last_error.argtypes = [C.c_void_p]
        L.gs_engine_evaluate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp, C.c_int, dp, dp, dp]
        L.gs_substitution_model.argtypes = [C.c_char_p, dp, dp, dp, dp, dp, dp]
        L.gs_transition_matrix.argtypes = [dp, dp, dp, C.c_double, dp]
        L.gs_det_exp.restype = C.c_double
        L.gs_det_exp.argtypes = [C.c_double]
        L.gs_codon_table.argtypes = [ip]
        L.gs_codon_state.argtypes = [C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class GsOracleError(RuntimeError):
    pass


def codon_table() -> np.ndarray:
    """[61][3] nucleotides (A,C,G,T = 0..3) of every sense codon, in state order."""
    t = np.zeros((61, 3), dtype=np.int32)
    assert lib().gs_codon_table(_ip(t)) == 61
    return t


def codon_state(a: int, b: int, c: int) -> int:
    return lib().gs_codon_state(int(a), int(b), int(c))


def substitution_model(name: str, params):
    """(Q, V, Vinv, lambda, pi) of one parameter row, trimmed to the model's state count."""
    p = np.ascontiguousarray(params, dtype=np.float64)
    Q, V, Vi = (np.zeros(MS * MS) for _ in range(3))
    lam, pi = np.zeros(MS), np.zeros(MS)
    rc = lib().gs_substitution_model(name.encode(), _dp(p), _dp(Q), _dp(V), _dp(Vi), _dp(lam), _dp(pi))
    if rc:
        raise GsOracleError(f"substitution model error {rc}")
    S = 4 if name == "GTR" else 61
    return tuple(m.reshape(MS, MS)[:S, :S] for m in (Q, V, Vi)) + (lam, pi[:S])


def transition_matrix_padded(Vp, Vip, lam, t: float) -> np.ndarray:
    P = np.zeros(MS * MS)
    lib().gs_transition_matrix(_dp(np.ascontiguousarray(Vp)), _dp(np.ascontiguousarray(Vip)),
                               _dp(np.ascontiguousarray(lam)), float(t), _dp(P))
    return P.reshape(MS, MS)


class GsOracleEngine:
    """Engine mirror (reference src/engine.hpp:26-68) for "GTR" (4 states) or "GY94" (61 codon states)."""

    def __init__(self, substitution: str, site: str, patterns, weights, thread_count: int = 1):
        self._h = None
        L = lib()
        self.patterns = np.ascontiguousarray(patterns, dtype=np.int32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        n, P = self.patterns.shape
        err = C.create_string_buffer(256)
        h = L.gs_engine_create(substitution.encode(), site.encode(), thread_count, n, P, _ip(self.patterns),
                               _dp(self.weights), err, 256)
        if not h:
            raise GsOracleError(err.value.decode())
        self._h = h
        self.taxon_count = n
        self.param_count = L.gs_engine_param_count(h)
        self.state_count = L.gs_engine_state_count(h)
        self.category_count = L.gs_engine_category_count(h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gs_engine_destroy(self._h)
            self._h = None

    def _run(self, parent_ids, branch_lengths, params, rates, rescaling, want_gradient, want_site=False):
        parent_ids = np.ascontiguousarray(parent_ids, dtype=np.int32)
        branch_lengths = np.ascontiguousarray(branch_lengths, dtype=np.float64)
        T, M = branch_lengths.shape
        assert parent_ids.shape == (T, M - 1)
        rooted = int(M == 2 * self.taxon_count - 1)
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(T, self.param_count)
        if rates is not None:
            rates = np.ascontiguousarray(rates, dtype=np.float64)
        ll = np.zeros(T)
        grad = np.zeros((T, 2 * self.taxon_count - 1)) if want_gradient else None
        site = np.zeros(T) if want_site else None
        rc = lib().gs_engine_evaluate(self._h, T, rooted, M, _ip(parent_ids), _dp(branch_lengths), _dp(rates),
                                      _dp(params), int(rescaling), _dp(ll), _dp(grad), _dp(site))
        if rc:
            raise GsOracleError(lib().gs_engine_last_error(self._h).decode())
        return (ll, grad, site) if want_site else (ll, grad)

    def log_likelihoods(self, parent_ids, branch_lengths, params, rates=None, rescaling=False):
        return self._run(parent_ids, branch_lengths, params, rates, rescaling, False)[0]

    def gradients(self, parent_ids, branch_lengths, params, rates=None, rescaling=False, site_model=False):
        res = self._run(parent_ids, branch_lengths, params, rates, rescaling, True, site_model)
        out = {"log_likelihood": res[0], "branch_lengths": res[1]}
        if site_model:
            out["site_model"] = res[2]
        return out
