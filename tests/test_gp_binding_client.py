"""Seam 3 bound the way INTEGRATION.md says -- GPEngine::ProcessOperations as a visitor that flattens the reference's
std::variant GPOperation into bito_amd_gp_op records (src/gp_engine.hpp:74, src/gp_operation.hpp:24-167) -- as a compiled
program: tests/gp_binding_client.cpp includes the REFERENCE's own gp_operation.hpp and include/bito_amd_gp.h (plain g++;
built by oracle/Makefile's `ref` target where the reference is present) and replays the three schedules of
GPInstance::EstimateBranchLengths on the multi-tree DAG of ds1-reduced-5 from a case file.  Its numbers must be the ctypes route's,
bit for bit.  On the CPU the program runs against the emulated library (tests/hip_emu); `-m gpu` runs it on the device."""
import os
import subprocess

import numpy as np
import pytest

from bito_amd import gp
from oracle import ref

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLIENT = os.path.join(ROOT, "oracle", "_ref", "gp_binding_client.bin")

pytestmark = pytest.mark.skipif(not (ref.available() or os.path.exists(CLIENT)), reason="the reference's sources are not in this container")


def _case(path, sp, dag, bl, q, streams):
    with open(path, "w") as fh:
        fh.write(f"{sp.patterns.shape[0]} {sp.patterns.shape[1]} {dag.node_count} {dag.gpcsp_count}\n")
        fh.write(" ".join(str(int(x)) for x in sp.patterns.reshape(-1)) + "\n")
        fh.write(" ".join(repr(float(x)) for x in sp.weights) + "\n")
        fh.write(" ".join(repr(float(x)) for x in bl) + "\n" + " ".join(repr(float(x)) for x in q) + "\n")
        fh.write(f"{len(streams)}\n")
        for s in streams:
            fh.write(f"{len(s.ops)}\n")
            for opcode, count, a, b, c in s.ops:
                if opcode == gp.PREP_FOR_MARGINALIZATION:
                    fh.write(f"9 {a} 0 0 {count} " + " ".join(str(x) for x in s.side[b:b + count]) + "\n")
                else:
                    fh.write(f"{opcode} {a} {b} {c}\n")


def _run_client(case, env=None):
    ref.lib()  # (builds oracle/_ref, the client with it, where the reference is present)
    done = subprocess.run([CLIENT, str(case)], capture_output=True, text=True, timeout=600, env=env)
    assert done.returncode == 0, done.stderr
    rows = dict(ln.split(" ", 1) for ln in done.stdout.strip().splitlines())
    return float(rows["marginal"]), np.array([float(x) for x in rows["per_gpcsp"].split()]), np.array([float(x) for x in rows["branch_lengths"].split()])


def _instance(data_dir):
    import test_gp

    sp, dag, bl = test_gp._composite_case(data_dir, "ds1-reduced-5.fasta", "ds1-reduced-5.nwk")
    bl = np.maximum(bl, 0.01)
    q = dag.uniform_on_topological_support_prior()
    streams = [dag.populate_plvs(), dag.compute_likelihoods(), dag.branch_length_optimization(), dag.populate_plvs(),
               dag.compute_likelihoods(), dag.marginal_likelihood()]
    return sp, dag, bl, q, streams


def _through_ctypes(sp, dag, bl, q, streams):
    eng = gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
    eng.set_branch_lengths(bl)
    eng.set_sbn_parameters(q)
    for s in streams:
        eng.process_operations(s)
    return eng.get_log_marginal_likelihood(), eng.get_per_gpcsp_log_likelihoods(), eng.get_branch_lengths()


def test_binding_client_on_the_emulated_library(data_dir, tmp_path):
    import ctypes as C

    from bito_amd import _capi
    from test_engine_emulated import AS_PRODUCT, EMU

    built = subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hip_emu")], capture_output=True, text=True)
    assert built.returncode == 0, built.stdout + built.stderr
    os.makedirs(AS_PRODUCT, exist_ok=True)
    link = os.path.join(AS_PRODUCT, "libbito_amd.so")
    if not os.path.islink(link):
        os.symlink(os.path.join("..", "libbito_amd_emu.so"), link)
    sp, dag, bl, q, streams = _instance(data_dir)
    case = tmp_path / "case.txt"
    _case(case, sp, dag, bl, q, streams)
    env = dict(os.environ, LD_LIBRARY_PATH=AS_PRODUCT + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    marginal, per, after = _run_client(case, env)
    keep = _capi._lib
    _capi._lib = C.CDLL(EMU)
    try:
        want = _through_ctypes(sp, dag, bl, q, streams)
    finally:
        _capi._lib = keep
    assert marginal == want[0] and np.array_equal(per, want[1]) and np.array_equal(after, want[2])
    assert np.abs(after - bl).max() > 1e-3 and np.isfinite(marginal)


@pytest.mark.gpu
def test_binding_client_on_the_device(data_dir, tmp_path):
    sp, dag, bl, q, streams = _instance(data_dir)
    case = tmp_path / "case.txt"
    _case(case, sp, dag, bl, q, streams)
    marginal, per, after = _run_client(case)
    want = _through_ctypes(sp, dag, bl, q, streams)
    assert marginal == want[0] and np.array_equal(per, want[1]) and np.array_equal(after, want[2])
