"""The C ABI from a plain C++ program (examples/engine_amd.hpp + tests/cabi_client.cpp, built with
g++ against libbito_amd.so by __graft_entry__.build()): same numbers as the ctypes route, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import bito_amd
from bito_amd import workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, "tests", "cabi_client.bin")


def _ensure_client():
    if not os.path.exists(CLIENT):  # normally built by __graft_entry__.build()
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bito_amd", "csrc"), "../../tests/cabi_client.bin"])


def test_client_is_built_and_links_only_the_c_abi():
    _ensure_client()
    out = subprocess.run(["ldd", CLIENT], stdout=subprocess.PIPE, text=True).stdout
    assert "libbito_amd.so" in out and "torch" not in out and "python" not in out


@pytest.mark.gpu
@pytest.mark.parametrize("device_count", [1, 2])
def test_cpp_client_matches_ctypes_route(tmp_path, device_count):
    """device_count = 1: the single-device engine; 2: the same call over two device slots (both GPU 0 on the one-GPU
    box: the code path of an engine over N devices), results equal bit for bit -- every tree is walked whole by one
    workgroup at this batch size, whichever slot it went to"""
    _ensure_client()
    w = workloads.ds1_gtr_weibull4(1).subset(6)
    case = tmp_path / "case.txt"
    with open(case, "w") as fh:
        fh.write(f"{w.substitution} {w.site} {w.clock}\n{w.patterns.shape[0]} {w.patterns.shape[1]}\n")
        fh.write(" ".join(str(int(x)) for x in w.patterns.reshape(-1)) + "\n")
        fh.write(" ".join(repr(float(x)) for x in w.weights) + "\n")
        fh.write(f"0 {w.tree_count} {w.parent_ids.shape[1] + 1}\n")
        fh.write(" ".join(str(int(x)) for x in w.parent_ids.reshape(-1)) + "\n")
        fh.write(" ".join(repr(float(x)) for x in w.branch_lengths.reshape(-1)) + "\n")
        fh.write(f"{w.params.shape[1]}\n" + " ".join(repr(float(x)) for x in w.params.reshape(-1)) + "\n")
    proc = subprocess.run([CLIENT, str(case), str(device_count)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr
    lines = proc.stdout.strip().splitlines()
    ll = np.array([float(ln.split()[1]) for ln in lines if ln.startswith("ll ")])
    grad = np.array([[float(x) for x in ln.split()[1:]] for ln in lines if ln.startswith("grad")])
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
    ref = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert np.array_equal(ll, ref["log_likelihood"]) and np.array_equal(grad, ref["branch_lengths"])
    assert any(ln.startswith("error-path ok") and "parent id" in ln for ln in lines)
    assert f"devices {device_count}" in lines


def test_cpp_client_rejects_zero_devices(tmp_path):
    """(checked before any device is touched: runs without a GPU)"""
    _ensure_client()
    w = workloads.ds1_gtr_weibull4(1).subset(1)
    case = tmp_path / "case.txt"
    with open(case, "w") as fh:
        fh.write(f"JC69 constant none\n{w.patterns.shape[0]} {w.patterns.shape[1]}\n")
        fh.write(" ".join(str(int(x)) for x in w.patterns.reshape(-1)) + "\n")
        fh.write(" ".join(repr(float(x)) for x in w.weights) + "\n")
        fh.write(f"0 1 {w.parent_ids.shape[1] + 1}\n")
        fh.write(" ".join(str(int(x)) for x in w.parent_ids.reshape(-1)) + "\n")
        fh.write(" ".join(repr(float(x)) for x in w.branch_lengths.reshape(-1)) + "\n0\n")
    proc = subprocess.run([CLIENT, str(case), "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert proc.returncode == 1 and "strictly positive" in proc.stderr
