"""A small blocking call whose parameter rows all equal the row the last such call's tree 0 had copies that tree's model
(rate matrix, eigensystem, category rates) instead of forming it again (bito_amd/csrc/kernels.hpp, DeviceBatch::model_reuse;
worker.cpp, WorkerStageEnd): the reference forms the eigensystem on every call (src/fat_beagle.cpp:186-216,
UpdateSubstitutionModelInBeagle).  Same bits either way, and no stale model when the rows change."""
import os
import subprocess
import sys

import numpy as np
import pytest

import bito_amd
from bito_amd import _capi, workloads
from oracle import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _close(a, b, atol, rtol):
    return bool(np.all(np.abs(np.asarray(a) - np.asarray(b)) <= atol + rtol * np.abs(np.asarray(b))))


def _engines(w):
    spec = bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock)
    return bito_amd.Engine(spec, w.patterns, w.weights), oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)


def _same_bits(a, b):
    return all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in ("log_likelihood", "branch_lengths"))


def test_repeated_and_changing_parameter_rows():
    w = workloads.ds1_gtr_weibull4(1).subset(64)
    gpu, cpu = _engines(w)
    rng = np.random.default_rng(11)
    row_a = w.params[0].copy()
    row_b = row_a.copy()
    row_b[:4] = rng.dirichlet([5, 5, 5, 5])
    row_b[4:10] = rng.dirichlet([3] * 6)
    row_b[10] = 0.7
    A = np.tile(row_a, (64, 1))
    B = np.tile(row_b, (64, 1))
    mixed = A.copy()
    mixed[1::2] = row_b  # tree 0 has row A, every other tree row B: no reuse, and the cache then holds A's model
    tail = B.copy()
    tail[-1] = row_a    # every row but the last equals the cached one

    def check(params):
        out = gpu.gradients(w.parent_ids, w.branch_lengths, params, flags=_capi.GRAD_SITE_MODEL)
        ref = cpu.gradients(w.parent_ids, w.branch_lengths, params, flags=oracle.GRAD_SITE_MODEL)
        assert _close(out["log_likelihood"], ref["log_likelihood"], 1e-10, 2e-14)
        assert _close(out["branch_lengths"], ref["branch_lengths"], 1e-6, 1e-9)
        assert _close(out["site_model"], ref["site_model"], 1e-6, 1e-9)
        return out

    first = check(A)             # forms the models, leaves tree 0's
    again = check(A)             # copies it
    assert _same_bits(first, again)
    other = check(B)             # rows changed: formed again
    assert not np.array_equal(other["log_likelihood"], first["log_likelihood"])
    assert _same_bits(check(B), other)
    check(mixed)                 # tree 0 = A: the cache holds A's model now
    assert _same_bits(check(A), first)
    check(B)
    check(tail)                  # all rows but one equal the cached row: formed again, none copied
    assert _same_bits(check(B), other)
    # log-likelihood-only calls and calls on fewer trees share the cache
    assert np.array_equal(gpu.log_likelihoods(w.parent_ids[:7], w.branch_lengths[:7], B[:7]), other["log_likelihood"][:7]) or \
        _close(gpu.log_likelihoods(w.parent_ids[:7], w.branch_lengths[:7], B[:7]), other["log_likelihood"][:7], 1e-10, 2e-14)


def test_rescaled_calls_and_the_hbm_walk_share_it():
    w = workloads.ds1_gtr_weibull4(1).subset(20)
    gpu, cpu = _engines(w)
    P = np.tile(w.params[3], (20, 1))
    for rescaling in (False, True, True, False):
        out = gpu.gradients(w.parent_ids, w.branch_lengths, P, rescaling=rescaling)
        ref = cpu.gradients(w.parent_ids, w.branch_lengths, P, rescaling=rescaling)
        assert _close(out["log_likelihood"], ref["log_likelihood"], 1e-10, 2e-14)
        assert _close(out["branch_lengths"], ref["branch_lengths"], 1e-6, 1e-9)


def test_same_bits_with_the_cache_off():
    """BITO_AMD_MODEL_CACHE=0 (read once per process): a second process repeats three calls, every value bit for bit."""
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bito_amd\n"
        "from bito_amd import workloads\n"
        "w = workloads.ds1_gtr_weibull4(1).subset(48)\n"
        "gpu = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)\n"
        "P = np.tile(w.params[5], (48, 1))\n"
        "acc = []\n"
        "for _ in range(3):\n"
        "    o = gpu.gradients(w.parent_ids, w.branch_lengths, P)\n"
        "    acc.append(o['log_likelihood'].tobytes().hex() + o['branch_lengths'].tobytes().hex())\n"
        "import hashlib\n"
        "print(hashlib.sha256(''.join(acc).encode()).hexdigest())\n")
    digests = []
    for value in ("1", "0"):
        proc = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                              env={**os.environ, "BITO_AMD_MODEL_CACHE": value})
        assert proc.returncode == 0, proc.stderr[-2000:]
        digests.append(proc.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1]


def test_general_state_path_keeps_models_but_not_topologies():
    """61 states: the rate matrices and 64 x 64 eigensystems of the batch before stand when the parameter rows repeat
    (worker.cpp, UploadModelIndex); the trees' topologies and branch lengths are set up on every pass all the same --
    other trees, other branch lengths, then other rows, blocking calls and passes over a resident batch, each against
    the CPU checker."""
    from oracle import gs
    from test_gpu_parity import _random_rooted_parent_ids

    rng = np.random.default_rng(61)
    n, P, T = 9, 40, 6
    patterns = rng.integers(0, 61, (n, P)).astype(np.int32)
    weights = rng.integers(1, 4, P).astype(np.float64)
    gpu = bito_amd.Engine(bito_amd.PhyloModelSpecification("GY94", "constant", "none"), patterns, weights)
    cpu = gs.GsOracleEngine("GY94", "constant", patterns, weights, 4)
    params = gpu.default_params(T)
    params[:, :4] = rng.dirichlet([5, 5, 5, 5])
    params[:, 4] = 2.0
    params[:, 5] = 0.4

    def check(pid, bl, par):
        out = gpu.gradients(pid, bl, par)
        ref = cpu.gradients(pid, bl, par)
        assert _close(out["log_likelihood"], ref["log_likelihood"], 1e-10, 2e-14)
        assert _close(out["branch_lengths"], ref["branch_lengths"], 1e-6, 1e-9)
        return out

    pid = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)])
    bl = rng.exponential(0.1, (T, 2 * n - 1))
    bl[:, -1] = 0.0
    first = check(pid, bl, params)
    assert _same_bits(check(pid, bl, params), first)           # same rows: the models stand
    pid2 = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)])
    bl2 = rng.exponential(0.05, (T, 2 * n - 1))
    bl2[:, -1] = 0.0
    check(pid2, bl2, params)                                    # other trees, the same rows
    other = params.copy()
    other[3, 5] = 0.9
    check(pid2, bl2, other)                                     # one row changed
    assert _same_bits(check(pid, bl, params), first)
    # passes over the resident batch: new branch lengths, then new rows
    gpu.upload(pid, bl, params)
    gpu.run(True)
    ll, grad = gpu.download()
    assert np.array_equal(ll, first["log_likelihood"]) and np.array_equal(grad, first["branch_lengths"])
    gpu.update(branch_lengths=bl2)
    gpu.run(True)
    ll, grad = gpu.download()
    ref = cpu.gradients(pid, bl2, params)
    assert _close(ll, ref["log_likelihood"], 1e-10, 2e-14) and _close(grad, ref["branch_lengths"], 1e-6, 1e-9)
    gpu.update(params=other)
    gpu.run(True)
    ll, grad = gpu.download()
    ref = cpu.gradients(pid, bl2, other)
    assert _close(ll, ref["log_likelihood"], 1e-10, 2e-14) and _close(grad, ref["branch_lengths"], 1e-6, 1e-9)
