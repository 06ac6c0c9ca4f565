"""The 61-state codon path against known answers that do NOT come from its own algorithm.

tests/golden/codon_fixtures.json is written by scripts/gen_codon_fixtures.py: GY94 built from its definition,
P(t) = exp(Qt) by a Taylor series in 80-bit extended precision (no eigendecomposition), pruning over the
uncompressed codon columns of fluA.fa, analytic gradients spot-checked by central differences.  The reference has no
codon model, so this is the pin for S = 61: oracle/gs_oracle.c and the GPU kernels share one
algorithm (symmetrised eigendecomposition), and agreement between them alone would only show consistency.

Tolerances.  GPU and oracle agree to 1e-10 (tests/test_gpu_general.py); both differ from the extended-precision
values by what FP64 leaves of P = V exp(L t) V^-1 for codons two nucleotide changes apart -- P_ij = O(t^2) ~ 1e-10
formed from terms of order one carries an absolute error of 1e-16, a relative error of 1e-6, on entries that decide
every column needing a double change on one branch.  Measured: 2.0e-6 on log-likelihoods of
-4.7e3 (4e-10 relative), 1.3e-6 relative on gradients of up to 4e4.  The bars below are 1e-5 / 5e-6: wide enough for
that conditioning, narrow enough that a wrong rate matrix, genetic code, frequency or weight would fail by orders
of magnitude (omega or kappa off by 1e-6 moves these log-likelihoods by 1e-3)."""
import json
import os

import numpy as np
import pytest

from bito_amd import workloads

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "codon_fixtures.json")) as fh:
    FIXTURES = json.load(fh)["cases"]

LL_ATOL = 1e-5
GRAD_RTOL = 5e-6


def _inputs(case):
    w = workloads.flua_codon(len(case["log_likelihoods"]), site=case["site"])
    pid = np.array(case["parent_ids"], dtype=np.int32)
    bl = np.array(case["branch_lengths"])
    par = np.array(case["params"])
    # the fixture stores its inputs; the workload builder must still produce the same ones
    assert np.array_equal(pid, w.parent_ids) and np.array_equal(bl, w.branch_lengths) and np.array_equal(par, w.params)
    return w, pid, bl, par


def _check(out, case):
    want_ll = np.array(case["log_likelihoods"])
    want_grad = np.array(case["branch_gradients"])
    assert np.abs(out["log_likelihood"] - want_ll).max() < LL_ATOL
    assert (np.abs(out["branch_lengths"] - want_grad) / np.maximum(1.0, np.abs(want_grad))).max() < GRAD_RTOL
    assert np.all(out["branch_lengths"][:, -1] == 0.0)  # the root has no branch


@pytest.mark.parametrize("case", FIXTURES, ids=[c["name"] for c in FIXTURES])
def test_cpu_checker_against_independent_codon_values(case):
    from oracle import gs

    w, pid, bl, par = _inputs(case)
    eng = gs.GsOracleEngine("GY94", case["site"], w.patterns, w.weights, 2)
    _check(eng.gradients(pid, bl, par), case)
    assert np.abs(eng.log_likelihoods(pid, bl, par) - np.array(case["log_likelihoods"])).max() < LL_ATOL


def test_fixture_is_sensitive_to_the_model():
    """the bar is far below what a wrong parameter does: kappa off by 1e-4 fails it"""
    from oracle import gs

    case = FIXTURES[0]
    w, pid, bl, par = _inputs(case)
    eng = gs.GsOracleEngine("GY94", case["site"], w.patterns, w.weights, 2)
    wrong = par.copy()
    wrong[:, 4] += 1e-4
    assert np.abs(eng.log_likelihoods(pid, bl, wrong) - np.array(case["log_likelihoods"])).max() > 10 * LL_ATOL


@pytest.mark.gpu
@pytest.mark.parametrize("case", FIXTURES, ids=[c["name"] for c in FIXTURES])
def test_gpu_against_independent_codon_values(case):
    import bito_amd

    w, pid, bl, par = _inputs(case)
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification("GY94", case["site"], "none"), w.patterns, w.weights)
    _check(eng.gradients(pid, bl, par), case)
    assert eng.kernel_name() == "gs_walk_kernel"
    assert np.abs(eng.log_likelihoods(pid, bl, par) - np.array(case["log_likelihoods"])).max() < LL_ATOL
