"""walk_pipe_kernel with two waves per SIMD (round 4; bito_amd/csrc/walk_pipe.hip, layout 2): trees of up to 28 taxa in
the one-image-per-branch form, eight waves of 256 registers per workgroup.  AUTO does not take it (it measured slower
than one wave per SIMD with four pattern groups, profiles/r4_pipe_two_waves.md); pinned through
BITO_AMD_KERNEL_LDS_PIPE2 it is held to the same bars as every other kernel -- the reference path it restates is
src/fat_beagle.cpp:49-169 (post-order partials, root likelihood, pre-order partials, edge derivatives)."""
import numpy as np
import pytest

import bito_amd
from bito_amd import _capi, workloads
from oracle import oracle

pytestmark = pytest.mark.gpu

LL_ATOL, LL_RTOL, GRAD_ATOL, GRAD_RTOL = 1e-10, 2e-14, 1e-6, 1e-9


def _close(a, b, atol, rtol):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))


def _pair(w, threads=8):
    gpu = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
    gpu.set_kernel(_capi.KERNEL_LDS_PIPE2)
    return gpu, oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, threads)


def test_config3_trees_match_the_oracle():
    w = workloads.ds1_gtr_weibull4(1)
    gpu, cpu = _pair(w)
    out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=oracle.GRAD_SITE_MODEL)
    assert gpu.kernel_name() == "walk_pipe_kernel" and "two waves per SIMD" in gpu.kernel_form()
    assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
    assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    assert _close(out["site_model"], ref["site_model"], GRAD_ATOL, GRAD_RTOL)
    assert _close(gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params), ref["log_likelihood"], LL_ATOL, LL_RTOL)
    # the one-wave form on the same trees: the same values to a tenth of the tolerances (other image layout, other sums)
    gpu.set_kernel(_capi.KERNEL_LDS_PIPE)
    one = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert "two waves" not in gpu.kernel_form()
    assert _close(out["log_likelihood"], one["log_likelihood"], 0.1 * LL_ATOL, LL_RTOL)
    assert _close(out["branch_lengths"], one["branch_lengths"], 0.1 * GRAD_ATOL, GRAD_RTOL)


@pytest.mark.parametrize("sub,site", [("JC69", "constant"), ("HKY", "weibull+2"), ("GTR", "weibull+4")])
def test_one_two_and_four_rate_categories(sub, site):
    w = workloads.ds1_gtr_weibull4(1).subset(40)
    w.substitution, w.site = sub, site
    gpu, cpu = _pair(w)
    rng = np.random.default_rng(5)
    params = gpu.default_params(w.tree_count)
    if sub != "JC69":
        params[:, :4] = rng.dirichlet([5, 5, 5, 5], w.tree_count)
        params[:, 4:4 + (6 if sub == "GTR" else 1)] = (rng.dirichlet([3] * 6, w.tree_count) if sub == "GTR"
                                                        else rng.uniform(0.5, 4.0, (w.tree_count, 1)))
    if site != "constant":
        params[:, -1] = rng.uniform(0.3, 2.0, w.tree_count)
    out = gpu.gradients(w.parent_ids, w.branch_lengths, params)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, params)
    assert "two waves per SIMD" in gpu.kernel_form()
    assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
    assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)


def test_trees_that_fail_the_guard_run_on_the_one_wave_form_behind_the_others():
    """The two-wave form keeps one image per branch (P^T = Pi P Pi^-1), which holds to the rounding error of P's own
    entries: a tree takes it only when its shortest branch times its smallest off-diagonal rate clears the bound of
    DESIGN.md section 3.  The others -- here every eighth tree, with branches of 1e-9 -- form class B."""
    w = workloads.ds1_gtr_weibull4(1).subset(64)
    bl = w.branch_lengths.copy()
    bl[::8, 3] = 1e-9
    bl[::8, 17] = 0.0
    gpu, cpu = _pair(w)
    out = gpu.gradients(w.parent_ids, bl, w.params)
    ref = cpu.gradients(w.parent_ids, bl, w.params)
    form = gpu.kernel_form()
    assert "two waves per SIMD" in form and "one wave per SIMD" in form, form
    assert int(form.split()[0]) <= 56  # (class A: at most the 56 trees whose branches were left alone)
    assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
    assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    # ... and when most trees fail it, the pinned form says that it cannot run
    bl[:, 3] = 1e-9
    with pytest.raises(bito_amd.BitoAmdError, match="two-wave form"):
        gpu.gradients(w.parent_ids, bl, w.params)
    # (AUTO takes the one-wave kernel with (P, P^T) pairs for them)
    gpu.set_kernel(_capi.KERNEL_AUTO)
    out = gpu.gradients(w.parent_ids, bl, w.params)
    ref = cpu.gradients(w.parent_ids, bl, w.params)
    assert gpu.kernel_name() == "walk_pipe_kernel" and "two waves" not in gpu.kernel_form()
    assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)


def test_resident_passes_follow_new_branch_lengths():
    w = workloads.ds1_gtr_weibull4(1)
    gpu, cpu = _pair(w)
    gpu.upload(w.parent_ids, w.branch_lengths, w.params)
    gpu.run(True)
    ll, grad = gpu.download()
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert "two waves per SIMD" in gpu.kernel_form()
    assert _close(ll, ref["log_likelihood"], LL_ATOL, LL_RTOL) and _close(grad, ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    # new branch lengths, some of them too short for the one-image form: the classes follow
    bl = w.branch_lengths * 0.5
    bl[5::10, 7] = 2e-9
    gpu.update(bl, w.params)
    gpu.run(True)
    ll, grad = gpu.download()
    ref = cpu.gradients(w.parent_ids, bl, w.params)
    assert "one wave per SIMD" in gpu.kernel_form() and "two waves per SIMD" in gpu.kernel_form()
    assert _close(ll, ref["log_likelihood"], LL_ATOL, LL_RTOL) and _close(grad, ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)


def test_other_tree_sizes_rooted_and_unrooted():
    from test_gpu_parity import _random_rooted_parent_ids

    rng = np.random.default_rng(11)
    for n, rooted in ((3, False), (4, True), (9, False), (16, True), (28, False), (28, True)):
        P, T = 150, 33
        patterns = rng.integers(0, 4, (n, P)).astype(np.int32)
        patterns[rng.random((n, P)) < 0.05] = 4
        weights = rng.integers(1, 5, P).astype(np.float64)
        if rooted:
            pid = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)])
        else:
            pid = np.stack([workloads.random_unrooted_tree(n, rng, 0.1).parent_ids for _ in range(T)]).astype(np.int32)
        bl = rng.exponential(0.1, (T, pid.shape[1] + 1)) + 1e-4
        bl[:, -1] = 0.0
        gpu = bito_amd.Engine(bito_amd.PhyloModelSpecification("GTR", "weibull+4", "none"), patterns, weights)
        cpu = oracle.OracleEngine("GTR", "weibull+4", "none", patterns, weights, 8)
        gpu.set_kernel(_capi.KERNEL_LDS_PIPE2)
        params = gpu.default_params(T)
        out = gpu.gradients(pid, bl, params)
        ref = cpu.gradients(pid, bl, params)
        assert "two waves per SIMD" in gpu.kernel_form(), (n, rooted, gpu.kernel_form())
        assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL), (n, rooted)
        assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL), (n, rooted)


def test_more_than_28_taxa_are_declined():
    rng = np.random.default_rng(3)
    n, P, T = 29, 64, 4
    patterns = rng.integers(0, 4, (n, P)).astype(np.int32)
    pid = np.stack([workloads.random_unrooted_tree(n, rng, 0.1).parent_ids for _ in range(T)]).astype(np.int32)
    bl = rng.exponential(0.1, (T, pid.shape[1] + 1)) + 1e-3
    gpu = bito_amd.Engine(bito_amd.PhyloModelSpecification("JC69", "weibull+4", "none"), patterns, np.ones(P))
    gpu.set_kernel(_capi.KERNEL_LDS_PIPE2)
    with pytest.raises(bito_amd.BitoAmdError, match="two-wave form"):
        gpu.gradients(pid, bl, gpu.default_params(T))
