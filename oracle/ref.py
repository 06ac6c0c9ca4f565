"""ctypes wrapper of oracle/_ref/libbito_ref.so: the parts of the REFERENCE itself that compile from their own sources
without BEAGLE or Eigen (oracle/ref_shim.cpp, oracle/Makefile target `ref`) -- site patterns, topologies and their ids,
the one-dimensional optimisers.  Test infrastructure only.  The library is built in the
container that holds /root/reference; elsewhere `available()` is False unless the built file travelled with the tree."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libbito_ref.so")
REFERENCE = os.environ.get("BITO_REFERENCE", "/root/reference")
_lib = None


def available() -> bool:
    return os.path.exists(_PATH) or os.path.isdir(os.path.join(REFERENCE, "src"))


def lib():
    global _lib
    if _lib is None:
        if os.path.isdir(os.path.join(REFERENCE, "src")):  # (make decides whether anything is out of date)
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref", f"REF={REFERENCE}"], stdout=subprocess.DEVNULL)
        L = C.CDLL(_PATH)
        vp, dp, lp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)
        i32p = C.POINTER(C.c_int32)
        L.ref_polished_parent_ids.argtypes = [C.c_int, C.c_int, i32p, i32p, i32p, lp]
        L.ref_parent_id_round_trip.argtypes = [lp, C.c_int, lp]
        L.ref_detrifurcate.argtypes = [lp, C.c_int, dp, lp, dp]
        L.ref_gp_operation.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
        L.ref_site_pattern.restype = vp
        L.ref_site_pattern.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
        L.ref_free_site_pattern.argtypes = [vp]
        L.ref_pattern_count.argtypes = [vp]
        L.ref_sequence_count.argtypes = [vp]
        L.ref_patterns.argtypes = [vp, C.POINTER(C.c_int32)]
        L.ref_weights.argtypes = [vp, dp]
        L.ref_brent_minimize.argtypes = [VALUE_FN, vp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_double, dp, dp]
        L.ref_brent_minimize_with_gradients.argtypes = [DERIVATIVE_FN, vp, C.c_double, C.c_double, C.c_double, C.c_int,
                                                        C.c_uint64, C.c_double, dp, dp]
        for name in ("ref_gradient_ascent", "ref_logspace_gradient_ascent"):
            fn = getattr(L, name)
            fn.restype = C.c_double
            fn.argtypes = [DERIVATIVE_FN, vp, C.c_double, C.c_int, C.c_double, C.c_double, C.c_uint64]
        L.ref_newton.restype = C.c_double
        L.ref_newton.argtypes = [DERIVATIVE_FN, vp, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint64]
        _lib = L
    return _lib


VALUE_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)
DERIVATIVE_FN = C.CFUNCTYPE(None, C.c_double, C.c_void_p, C.POINTER(C.c_double))


def polished_parent_ids(children, leaf_taxon, root) -> np.ndarray:
    """Node::Polish on a topology given as nested children: children[k] = list of node k's children in order ([] for a
    leaf, whose taxon id is leaf_taxon[k]).  Returns Node::ParentIdVector() of the polished topology (src/node.cpp:383-402)."""
    count = len(children)
    start = np.zeros(count + 1, dtype=np.int32)
    for k, c in enumerate(children):
        start[k + 1] = start[k] + len(c)
    flat = np.array([c for cs in children for c in cs] or [0], dtype=np.int32)
    taxon = np.array([leaf_taxon.get(k, -1) if isinstance(leaf_taxon, dict) else leaf_taxon[k] for k in range(count)], dtype=np.int32)
    out = np.zeros(count - 1, dtype=np.int64)
    i32p = C.POINTER(C.c_int32)
    rc = lib().ref_polished_parent_ids(count, int(root), start.ctypes.data_as(i32p), flat.ctypes.data_as(i32p),
                                       taxon.ctypes.data_as(i32p), out.ctypes.data_as(C.POINTER(C.c_int64)))
    if rc:
        raise RuntimeError("the reference rejected the topology")
    return out


def parent_id_round_trip(parent_ids) -> np.ndarray:
    """Node::OfParentIdVector(ids)->ParentIdVector() (src/node.hpp:207,343)"""
    ids = np.ascontiguousarray(parent_ids, dtype=np.int64)
    out = np.zeros_like(ids)
    if lib().ref_parent_id_round_trip(ids.ctypes.data_as(C.POINTER(C.c_int64)), len(ids), out.ctypes.data_as(C.POINTER(C.c_int64))):
        raise RuntimeError("the reference rejected the parent-id vector")
    return out


def detrifurcate(parent_ids, branch_lengths):
    """UnrootedTree::Detrifurcate (src/unrooted_tree.cpp:27-37) of the unrooted tree with these parent ids and branch lengths
    (by node id, the root's last): (parent ids, branch lengths) of the rooted tree, one node more"""
    ids = np.ascontiguousarray(parent_ids, dtype=np.int64)
    bl = np.ascontiguousarray(branch_lengths, dtype=np.float64)
    assert len(bl) == len(ids) + 1
    out_ids, out_bl = np.zeros(len(ids) + 1, dtype=np.int64), np.zeros(len(ids) + 2)
    if lib().ref_detrifurcate(ids.ctypes.data_as(C.POINTER(C.c_int64)), len(ids), bl.ctypes.data_as(C.POINTER(C.c_double)),
                              out_ids.ctypes.data_as(C.POINTER(C.c_int64)), out_bl.ctypes.data_as(C.POINTER(C.c_double))):
        raise RuntimeError("UnrootedTree::Detrifurcate given a non-trifurcating tree.")
    return out_ids, out_bl


class SitePattern:
    """SitePattern(Alignment::ReadFasta(fasta), {PackInts(id, 1): name}) (src/site_pattern.hpp:17-25)"""

    def __init__(self, fasta: str, taxon_names):
        names = (C.c_char_p * len(taxon_names))(*[n.encode() for n in taxon_names])
        h = lib().ref_site_pattern(fasta.encode(), names, len(taxon_names))
        if not h:
            raise RuntimeError(f"the reference's SitePattern rejected {fasta}")
        n, P = lib().ref_sequence_count(h), lib().ref_pattern_count(h)
        self.patterns = np.zeros((n, P), dtype=np.int32)
        self.weights = np.zeros(P)
        lib().ref_patterns(h, self.patterns.ctypes.data_as(C.POINTER(C.c_int32)))
        lib().ref_weights(h, self.weights.ctypes.data_as(C.POINTER(C.c_double)))
        lib().ref_free_site_pattern(h)


def gp_operation(opcode: int):
    """(alternative index in the reference's std::variant GPOperation, a, b, c, count) of the reference's operation for an
    opcode of include/bito_amd_gp.h, built with field values 11, 22, 33 (src/gp_operation.hpp:24-167)"""
    out = (C.c_uint64 * 5)()
    if lib().ref_gp_operation(int(opcode), out):
        raise ValueError(opcode)
    return tuple(int(v) for v in out)


def _value(fn):
    return VALUE_FN(lambda x, ctx: float(fn(x)))


def _derivatives(fn):
    def call(x, ctx, out):
        values = fn(x)
        for k, v in enumerate(values):
            out[k] = float(v)

    return DERIVATIVE_FN(call)


def brent_minimize(fn, guess, lo, hi, significant_digits, max_iter, step_size, with_gradients=False):
    """Optimization::BrentMinimize / BrentMinimizeWithGradients (src/optimization.hpp:71-331) on a Python function
    (with gradients: x -> (f, f'))"""
    x, fx = C.c_double(0), C.c_double(0)
    if with_gradients:
        cb = _derivatives(fn)
        lib().ref_brent_minimize_with_gradients(cb, None, guess, lo, hi, significant_digits, max_iter, step_size, C.byref(x), C.byref(fx))
    else:
        cb = _value(fn)
        lib().ref_brent_minimize(cb, None, guess, lo, hi, significant_digits, max_iter, step_size, C.byref(x), C.byref(fx))
    return x.value, fx.value


def gradient_ascent(fn, x, significant_digits, step_size, min_x, max_iter, log_space=False):
    cb = _derivatives(fn)
    run = lib().ref_logspace_gradient_ascent if log_space else lib().ref_gradient_ascent
    return run(cb, None, x, significant_digits, step_size, min_x, max_iter)


def newton(fn, x, significant_digits, epsilon, min_x, max_x, max_iter):
    cb = _derivatives(fn)
    return lib().ref_newton(cb, None, x, significant_digits, epsilon, min_x, max_x, max_iter)
