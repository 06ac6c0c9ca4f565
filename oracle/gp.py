"""ctypes wrapper of the GPEngine CPU oracle (oracle/libgp_oracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libgp_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
        L = C.CDLL(_PATH)
        dp, vp = C.POINTER(C.c_double), C.c_void_p
        L.gp_oracle_create.restype = vp
        L.gp_oracle_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), dp, C.c_int, C.c_int, C.c_double]
        L.gp_oracle_destroy.argtypes = [vp]
        L.gp_oracle_set_branch_lengths.argtypes = [vp, dp]
        L.gp_oracle_set_sbn_parameters.argtypes = [vp, dp]
        L.gp_oracle_get_sbn_parameters.argtypes = [vp, dp]
        L.gp_oracle_process.argtypes = [vp, C.c_void_p, C.c_int, C.c_void_p]
        L.gp_oracle_log_marginal_likelihood.restype = C.c_double
        L.gp_oracle_log_marginal_likelihood.argtypes = [vp]
        L.gp_oracle_per_gpcsp_log_likelihoods.argtypes = [vp, dp]
        L.gp_oracle_derivatives.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, dp]
        L.gp_oracle_transition_matrix.argtypes = [C.c_double, dp]
        L.gp_oracle_get_branch_lengths.argtypes = [vp, dp]
        L.gp_oracle_get_branch_length_differences.argtypes = [vp, dp]
        L.gp_oracle_set_optimization_method.argtypes = [vp, C.c_int]
        L.gp_oracle_set_significant_digits.argtypes = [vp, C.c_int]
        L.gp_oracle_reset_optimization_count.argtypes = [vp]
        L.gp_oracle_increment_optimization_count.argtypes = [vp]
        L.gp_oracle_grow_spare.argtypes = [vp, C.c_int, C.c_int]
        L.gp_oracle_copy_gpcsp_data.argtypes = [vp, C.c_int, C.c_int]
        L.gp_oracle_per_gpcsp_log_likelihoods_range.argtypes = [vp, C.c_int, C.c_int, dp]
        L.gp_oracle_branch_lengths_range.argtypes = [vp, C.c_int, C.c_int, dp]
        L.gp_oracle_grow.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.gp_oracle_get_plv.argtypes = [vp, C.c_int, dp]
        L.gp_oracle_rescaling_counts.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.gp_oracle_set_trace.argtypes = [vp, dp, C.c_int]
        L.gp_oracle_trace_rows.restype = C.c_int
        L.gp_oracle_trace_rows.argtypes = [vp]
        L.gp_oracle_set_eval_noise.argtypes = [vp, C.c_double, C.c_uint64]
        _lib = L
    return _lib


class OracleGPEngine:
    def __init__(self, patterns, weights, node_count, gpcsp_count, rescaling_threshold=1e-40):
        self.patterns = np.ascontiguousarray(patterns, dtype=np.int32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        n, P = self.patterns.shape
        self.gpcsp_count = gpcsp_count
        self._h = lib().gp_oracle_create(n, P, self.patterns.ctypes.data_as(C.POINTER(C.c_int)),
                                         self.weights.ctypes.data_as(C.POINTER(C.c_double)), node_count, gpcsp_count,
                                         rescaling_threshold)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gp_oracle_destroy(self._h)
            self._h = None

    def set_branch_lengths(self, bl):
        bl = np.ascontiguousarray(bl, dtype=np.float64)
        lib().gp_oracle_set_branch_lengths(self._h, bl.ctypes.data_as(C.POINTER(C.c_double)))

    def get_branch_lengths(self):
        out = np.zeros(self.gpcsp_count)
        lib().gp_oracle_get_branch_lengths(self._h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def get_branch_length_differences(self):
        out = np.zeros(self.gpcsp_count)
        lib().gp_oracle_get_branch_length_differences(self._h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def set_optimization_method(self, method: int):
        lib().gp_oracle_set_optimization_method(self._h, int(method))

    def set_significant_digits_for_optimization(self, digits: int):
        lib().gp_oracle_set_significant_digits(self._h, int(digits))

    def reset_optimization_count(self):
        lib().gp_oracle_reset_optimization_count(self._h)

    def increment_optimization_count(self):
        lib().gp_oracle_increment_optimization_count(self._h)

    # test instruments: every function evaluation of the Brent optimiser as rows (edge, x, f, kind, margins), and a relative
    # perturbation of the evaluations' values (what rounding noise of that size does to the iterates)
    def start_optimizer_trace(self, capacity=1 << 16):
        """rows of 6: edge, x, f, kind, the smallest relative margin of the comparisons that chose the point, the
        smallest |difference| of the comparisons of its value with the best points so far (gp_oracle.c)"""
        self._trace = np.zeros((capacity, 6))
        lib().gp_oracle_set_trace(self._h, self._trace.ctypes.data_as(C.POINTER(C.c_double)), capacity)

    def optimizer_trace(self):
        rows = lib().gp_oracle_trace_rows(self._h)
        if rows > len(self._trace):
            raise RuntimeError(f"optimiser trace overflow: {rows} rows, capacity {len(self._trace)}")
        return self._trace[:rows].copy()

    def set_eval_noise(self, relative, seed=1):
        lib().gp_oracle_set_eval_noise(self._h, float(relative), int(seed))

    def set_sbn_parameters(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        lib().gp_oracle_set_sbn_parameters(self._h, q.ctypes.data_as(C.POINTER(C.c_double)))

    def get_sbn_parameters(self):
        q = np.zeros(self.gpcsp_count)
        lib().gp_oracle_get_sbn_parameters(self._h, q.ctypes.data_as(C.POINTER(C.c_double)))
        return q

    def process_operations(self, stream):
        ops, side = stream.arrays()
        rc = lib().gp_oracle_process(self._h, ops.ctypes.data, len(ops), side.ctypes.data)
        if rc:
            raise RuntimeError(f"gp oracle: op stream rejected ({rc})")

    # spare slots (GPEngine::GrowSparePLVs / GrowSpareGPCSPs, CopyGPCSPData; src/gp_engine.cpp:196-211,401-409)
    def grow_spare(self, spare_plv_count, spare_gpcsp_count):
        lib().gp_oracle_grow_spare(self._h, int(spare_plv_count), int(spare_gpcsp_count))

    def copy_gpcsp_data(self, src, dst):
        for a, b in zip(src, dst):
            lib().gp_oracle_copy_gpcsp_data(self._h, int(a), int(b))

    def process_operation_batches(self, streams):
        for s in streams:  # independent sub-streams: any order gives the same result
            self.process_operations(s)

    def get_per_gpcsp_log_likelihoods_range(self, first, count):
        out = np.zeros(count)
        lib().gp_oracle_per_gpcsp_log_likelihoods_range(self._h, int(first), int(count), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def get_branch_lengths_range(self, first, count):
        out = np.zeros(count)
        lib().gp_oracle_branch_lengths_range(self._h, int(first), int(count), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def grow(self, new_node_count, new_gpcsp_count, node_reindexer=None, gpcsp_reindexer=None):
        ip = C.POINTER(C.c_int64)
        nr = None if node_reindexer is None else np.ascontiguousarray(node_reindexer, dtype=np.int64)
        gr = None if gpcsp_reindexer is None else np.ascontiguousarray(gpcsp_reindexer, dtype=np.int64)
        lib().gp_oracle_grow(self._h, int(new_node_count), int(new_gpcsp_count),
                             None if nr is None else nr.ctypes.data_as(ip), None if gr is None else gr.ctypes.data_as(ip))
        self.gpcsp_count = int(new_gpcsp_count)

    def get_plv(self, plv):
        out = np.zeros((4, self.patterns.shape[1]))
        lib().gp_oracle_get_plv(self._h, int(plv), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def get_rescaling_counts(self, first, count):
        """rescaling_counts_ of the reference (src/gp_engine.hpp:300-330): one count per PLV"""
        out = np.zeros(count, dtype=np.int32)
        lib().gp_oracle_rescaling_counts(self._h, int(first), int(count), out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def get_log_marginal_likelihood(self):
        return lib().gp_oracle_log_marginal_likelihood(self._h)

    def get_per_gpcsp_log_likelihoods(self):
        out = np.zeros(self.gpcsp_count)
        lib().gp_oracle_per_gpcsp_log_likelihoods(self._h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def log_likelihood_and_first_two_derivatives(self, gpcsp, rootward, leafward):
        out = np.zeros(3)
        lib().gp_oracle_derivatives(self._h, gpcsp, rootward, leafward, out.ctypes.data_as(C.POINTER(C.c_double)))
        return tuple(out)


def transition_matrix(t):
    P = np.zeros(16)
    lib().gp_oracle_transition_matrix(float(t), P.ctypes.data_as(C.POINTER(C.c_double)))
    return P.reshape(4, 4)
