"""ctypes binding of the C ABI in include/bito_amd.h (libbito_amd.so).

There is no CPU fallback: if the HIP library is missing, or no MI355X is
visible when an engine is created, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BITO_AMD_LIB", os.path.join(_HERE, "libbito_amd.so"))  # env override: dev experiments

OK = 0
ERR_BAD_MODEL, ERR_BAD_PARAMS, ERR_BAD_TREE, ERR_BAD_ARG, ERR_DEVICE, ERR_STATE = -1, -2, -3, -4, -5, -6

GRAD_SUBSTITUTION_MODEL = 1
GRAD_SITE_MODEL = 2
GRAD_CLOCK_MODEL = 4
GRAD_STICKBREAKING = 8
GRAD_RATIOS_ROOT_HEIGHT = 16
GRAD_LOG_DET_JACOBIAN_GRADIENT = 32

KERNEL_AUTO, KERNEL_HBM_ARENA, KERNEL_LDS, KERNEL_LDS_TREE, KERNEL_GENERAL, KERNEL_LDS_PIPE, KERNEL_LDS_PIPE2 = 0, 1, 2, 3, 4, 5, 6

# Every symbol include/bito_amd.h declares (tests check that the library exports them all).
SYMBOLS = [
    "bito_amd_engine_create", "bito_amd_engine_destroy", "bito_amd_engine_last_error",
    "bito_amd_engine_param_count", "bito_amd_engine_device_count", "bito_amd_engine_category_count", "bito_amd_engine_state_count", "bito_amd_engine_block_count",
    "bito_amd_engine_block", "bito_amd_engine_log_likelihoods", "bito_amd_engine_gradients",
    "bito_amd_engine_upload", "bito_amd_engine_update", "bito_amd_engine_run", "bito_amd_engine_sync",
    "bito_amd_engine_download", "bito_amd_engine_download_async", "bito_amd_engine_results_async", "bito_amd_engine_stream", "bito_amd_engine_set_kernel", "bito_amd_plan_pipe_walk", "bito_amd_count_unstored_nodes", "bito_amd_engine_time_runs",
    "bito_amd_engine_kernel_timing", "bito_amd_engine_kernel_elapsed", "bito_amd_engine_kernel_span_sum", "bito_amd_engine_kernel_name", "bito_amd_engine_kernel_form",
    "bito_amd_version", "bito_amd_engine_read_general_model",
    "bito_amd_engine_time_trees_from_branch_lengths", "bito_amd_engine_time_trees_from_height_ratios",
    "bito_amd_engine_log_det_jacobian", "bito_amd_engine_gradient_log_det_jacobian",
    "bito_amd_engine_ratio_gradient_of_height_gradient", "bito_amd_engine_time_tree_log_likelihoods",
    "bito_amd_engine_time_tree_gradients",
]


class EngineSpec(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("use_tip_states", C.c_int32), ("arena_bytes", C.c_uint64),
                ("device_count", C.c_int32), ("host_threads", C.c_int32), ("devices", C.POINTER(C.c_int32))]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C bito_amd/csrc). bito_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.bito_amd_version.restype = C.c_char_p
    L.bito_amd_engine_create.restype = C.c_int
    L.bito_amd_engine_create.argtypes = [C.POINTER(EngineSpec), C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32,
                                         C.c_int32, ip, dp, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.bito_amd_engine_destroy.restype = None
    L.bito_amd_engine_destroy.argtypes = [vp]
    L.bito_amd_engine_last_error.restype = C.c_char_p
    L.bito_amd_engine_last_error.argtypes = [vp]
    for name in ("param_count", "category_count", "state_count", "block_count", "device_count"):
        fn = getattr(L, f"bito_amd_engine_{name}")
        fn.restype = C.c_int32
        fn.argtypes = [vp]
    L.bito_amd_engine_block.argtypes = [vp, C.c_int32, C.c_char_p, C.c_size_t, ip, ip]
    L.bito_amd_engine_log_likelihoods.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, ip, dp, dp, dp, C.c_int32, dp]
    L.bito_amd_engine_gradients.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, ip, dp, dp, dp, C.c_int32,
                                            C.c_int32, C.c_double, dp, dp, dp, dp, dp]
    L.bito_amd_engine_upload.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, ip, dp, dp, dp]
    L.bito_amd_engine_update.argtypes = [vp, dp, dp]
    L.bito_amd_engine_run.argtypes = [vp, C.c_int32, C.c_int32]
    L.bito_amd_engine_sync.argtypes = [vp]
    # destinations may be host or device addresses: pass raw integers
    L.bito_amd_engine_download.argtypes = [vp, vp, vp]
    L.bito_amd_engine_download_async.argtypes = [vp, vp, vp]
    L.bito_amd_plan_pipe_walk.argtypes = [C.c_int32] * 5 + [ip]
    L.bito_amd_count_unstored_nodes.argtypes = [C.c_int32] * 4 + [ip, C.c_int32, ip]
    L.bito_amd_engine_results_async.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp)]
    L.bito_amd_engine_stream.restype = vp
    L.bito_amd_engine_stream.argtypes = [vp]
    L.bito_amd_engine_set_kernel.argtypes = [vp, C.c_int32]
    L.bito_amd_engine_time_runs.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, dp, dp, ip]
    L.bito_amd_engine_kernel_timing.argtypes = [vp, C.c_int32]
    L.bito_amd_engine_kernel_elapsed.argtypes = [vp, dp, ip]
    L.bito_amd_engine_kernel_span_sum.argtypes = [vp]
    L.bito_amd_engine_kernel_span_sum.restype = C.c_double
    L.bito_amd_engine_read_general_model.argtypes = [vp, C.c_int32, dp, C.c_size_t]
    L.bito_amd_engine_kernel_name.restype = C.c_char_p
    L.bito_amd_engine_kernel_name.argtypes = [vp]
    L.bito_amd_engine_kernel_form.restype = C.c_char_p
    L.bito_amd_engine_kernel_form.argtypes = [vp]
    L.bito_amd_engine_time_trees_from_branch_lengths.argtypes = [vp, C.c_int32, ip, dp, dp, dp, dp, dp]
    L.bito_amd_engine_time_trees_from_height_ratios.argtypes = [vp, C.c_int32, ip, dp, dp, dp, dp]
    L.bito_amd_engine_log_det_jacobian.argtypes = [vp, C.c_int32, ip, dp, dp, dp]
    L.bito_amd_engine_gradient_log_det_jacobian.argtypes = [vp, C.c_int32, ip, dp, dp, dp, dp]
    L.bito_amd_engine_ratio_gradient_of_height_gradient.argtypes = [vp, C.c_int32, ip, dp, dp, dp, dp, dp]
    L.bito_amd_engine_time_tree_log_likelihoods.argtypes = [vp, C.c_int32, ip, dp, dp, dp, dp, dp, C.c_int32,
                                                            C.c_int32, dp]
    L.bito_amd_engine_time_tree_gradients.argtypes = [vp, C.c_int32, ip, dp, dp, C.c_int32, dp, dp, dp, dp,
                                                      C.c_int32, C.c_int32, C.c_double, dp, dp, dp, dp, dp, dp]
    _lib = L
    return L
