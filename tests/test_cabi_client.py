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


# ---- seam 1: the BEAGLE subset from a C++ translation unit (tests/beagle_client.cpp) -------------------------------

BEAGLE_CLIENT = os.path.join(ROOT, "tests", "beagle_client.bin")


def _ensure_beagle_client():
    if not os.path.exists(BEAGLE_CLIENT):  # normally built by __graft_entry__.build()
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bito_amd", "csrc"), "../../tests/beagle_client.bin"])


def test_beagle_client_compiles_against_the_header_alone_and_links():
    """SURVEY.md 8b seam 1: include/bito_amd_beagle.h compiles as C++ (g++ -Wall -Wextra, no HIP) and the 17 symbols
    FatBeagle calls (reference src/fat_beagle.cpp:31-373) resolve against libbito_amd.so."""
    _ensure_beagle_client()
    out = subprocess.run(["ldd", BEAGLE_CLIENT], stdout=subprocess.PIPE, text=True).stdout
    assert "libbito_amd.so" in out and "torch" not in out and "python" not in out
    with open(os.path.join(ROOT, "tests", "beagle_client.cpp")) as fh:
        text = fh.read()
    includes = [ln for ln in text.splitlines() if ln.startswith("#include \"")]
    assert includes == ['#include "../include/bito_amd_beagle.h"']
    from beagle_driver import BEAGLE_SYMBOLS

    for sym in BEAGLE_SYMBOLS:
        assert sym + "(" in text, sym
    undefined = subprocess.run(["nm", "-u", BEAGLE_CLIENT], stdout=subprocess.PIPE, text=True).stdout
    assert sum(1 for sym in BEAGLE_SYMBOLS if f" {sym}" in undefined) == len(BEAGLE_SYMBOLS)


def _beagle_case(path, patterns, weights, V, Vinv, lam, pi, Q, rates, props, parent_ids, branch_lengths):
    with open(path, "w") as fh:
        fh.write(f"{patterns.shape[0]} {patterns.shape[1]} {len(rates)}\n")
        for arr in (patterns, weights, V, Vinv, lam, pi, Q, rates, props, parent_ids, branch_lengths):
            a = np.asarray(arr).reshape(-1)
            fh.write(" ".join(str(int(x)) if np.issubdtype(a.dtype, np.integer) else repr(float(x)) for x in a) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("use_tip_states,rescaling", [(1, 0), (0, 1)])
def test_beagle_client_reproduces_the_reference_goldens(tmp_path, use_tip_states, rescaling):
    """FatBeagle's call sequence from C++ against the DS1 JC69 goldens (pybeagle log-likelihoods and the physher gradient,
    reference src/unrooted_sbn_instance.hpp:245-348) and, under GTR + weibull+4, against the batched engine."""
    import json

    from oracle import oracle
    from bito_amd import treeio
    from bito_amd.site_pattern import SitePattern

    _ensure_beagle_client()
    data = os.path.join(ROOT, "tests", "golden", "data")
    with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as fh:
        g = json.load(fh)["ds1_jc69"]
    tc = treeio.read_nexus_file(os.path.join(data, g["trees"])) if g["trees"].endswith(".t") else treeio.read_newick_file(os.path.join(data, g["trees"]))
    sp = SitePattern(treeio.read_fasta(os.path.join(data, g["fasta"])), tc.taxon_names)
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    Q, V, Vi, lam, pi = oracle.substitution_model("JC69")

    def run(case):
        proc = subprocess.run([BEAGLE_CLIENT, str(case), str(use_tip_states), str(rescaling)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, timeout=300)
        assert proc.returncode == 0, proc.stderr
        rows = dict(ln.split(" ", 1) for ln in proc.stdout.strip().splitlines())
        return rows["impl"], float(rows["ll"]), float(rows["ll_from_gradient"]), np.array([float(x) for x in rows["gradient"].split()])

    case = tmp_path / "ds1_jc69.txt"
    _beagle_case(case, sp.patterns, sp.weights, V, Vi, lam, pi, Q, [1.0], [1.0], pid[9], bl[9])
    impl, ll, ll2, grad = run(case)
    assert "bito_amd" in impl
    assert abs(ll - g["log_likelihoods"][9]) < 5e-10 and abs(ll2 - g["log_likelihoods"][9]) < 5e-10
    assert np.abs(np.sort(grad) - g["last_tree_sorted_branch_gradient"]).max() < g["gradient_tol"]
    # GTR + weibull+4: the shim driven from C++ against the batched engine on the same tree
    w = workloads.ds1_gtr_weibull4(1).subset(2)
    Q, V, Vi, lam, pi = oracle.substitution_model("GTR", w.params[0, :10])
    rates, props, _ = oracle.weibull_rates(4, w.params[0, 10])
    case = tmp_path / "ds1_gtr.txt"
    _beagle_case(case, w.patterns, w.weights, V, Vi, lam, pi, Q, rates, props, w.parent_ids[1], w.branch_lengths[1])
    _, ll, ll2, grad = run(case)
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
    ref = eng.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=bool(rescaling))
    assert abs(ll - ref["log_likelihood"][1]) < 1e-10 + 2e-14 * abs(ll) and abs(ll2 - ll) < 1e-10
    assert np.abs(grad - ref["branch_lengths"][1]).max() < 1e-6
