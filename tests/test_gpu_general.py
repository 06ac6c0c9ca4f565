"""Parity of the general-state-count HIP kernels (gs_kernels.hip) through the C ABI: the 61-state
codon model of BASELINE config 5 against oracle/gs_oracle.c, and the same kernels at S = 4 against
the reference's goldens.  Needs a real MI355X: run with ``-m gpu``.

Tolerances (BASELINE.json north_star): 1e-10 on log-likelihoods, 1e-6 on gradients."""
import json
import os

import numpy as np
import pytest

import bito_amd
from bito_amd import _capi, treeio, workloads
from bito_amd.site_pattern import SitePattern
from oracle import gs

from test_gpu_parity import grad_close, ll_close, spec

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_goldens.json")) as fh:
    GOLD = json.load(fh)


def _general(sub, site, patterns, weights):
    eng = bito_amd.Engine(spec(sub, site), patterns, weights)
    eng.set_kernel(_capi.KERNEL_GENERAL)
    return eng


def test_general_kernels_reproduce_ds1_jc69_goldens(data_dir):
    """src/unrooted_sbn_instance.hpp:245-287 through the general kernels at S = 4."""
    g = GOLD["ds1_jc69"]
    tc = treeio.read_nexus_file(os.path.join(data_dir, g["trees"]))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, g["fasta"])), tc.taxon_names)
    eng = _general("JC69", "constant", sp.patterns, sp.weights)
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    out = eng.gradients(pid, bl)
    assert eng.kernel_name() == "gs_walk_kernel"
    assert np.abs(out["log_likelihood"] - g["log_likelihoods"]).max() < 5e-10
    last = np.sort(out["branch_lengths"][-1])
    assert np.abs(last - g["last_tree_sorted_branch_gradient"]).max() < g["gradient_tol"]
    ll = eng.log_likelihoods(pid, bl)
    assert np.array_equal(ll, out["log_likelihood"])


def test_general_kernels_match_oracle_on_headline_model():
    """GTR + weibull+4 on DS1 (config 3's model) through the general kernels vs the general oracle."""
    w = workloads.ds1_gtr_weibull4(1).subset(12)
    eng = _general(w.substitution, w.site, w.patterns, w.weights)
    cpu = gs.GsOracleEngine("GTR", w.site, w.patterns, w.weights, 8)
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])


@pytest.mark.parametrize("distinct", [1, 16, 64])
def test_codon_model_setup_is_bitwise_the_oracles(distinct):
    """Rate matrix, Jacobi eigensystem and F1x4 frequencies of the set-up kernel equal the oracle's bit
    for bit (same operation order, no FMA contraction) -- errors there are coherent across patterns.  With one row for
    all trees, and with 64 trees that carry 64 different (kappa, omega) rows: 64 eigensystems side by side in one launch
    of gs_eigen_kernel (the reference hands every tree its own row, src/fat_beagle.hpp:173-181), every one the
    checker's to the last bit (the alignment's first 16 patterns: the models do not depend on it)."""
    T = 2 if distinct == 1 else distinct
    w = workloads.flua_codon(T)
    if distinct > 1:
        w.params = workloads.codon_rows(T, distinct)
        w.patterns, w.weights = np.ascontiguousarray(w.patterns[:, :16]), np.ascontiguousarray(w.weights[:16])
        assert len(np.unique(w.params, axis=0)) == distinct
    eng = bito_amd.Engine(spec(w.substitution, w.site), w.patterns, w.weights)
    assert eng.state_count == 61 and eng.param_count == 6
    eng.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
    Q, V, Vi = (np.zeros(64 * 64) for _ in range(3))
    lam, pi = np.zeros(64), np.zeros(64)
    for t in range(1 if distinct == 1 else 0, T):
        got = eng.read_general_model(t)
        assert gs.lib().gs_substitution_model(b"GY94", gs._dp(np.ascontiguousarray(w.params[t])), gs._dp(Q), gs._dp(V),
                                              gs._dp(Vi), gs._dp(lam), gs._dp(pi)) == 0
        assert np.array_equal(got["pi"], pi), t
        assert np.array_equal(got["Q"], Q.reshape(64, 64)), t
        assert np.array_equal(got["lambda"], lam), t
        assert np.array_equal(got["V"], V.reshape(64, 64)), t
        assert np.array_equal(got["Vinv"], Vi.reshape(64, 64)), t


@pytest.mark.parametrize("site", ["constant", "weibull+3"])
def test_codon_model_matches_oracle(site):
    """BASELINE config 5: fluA as codons, GY94, log-likelihood + branch gradient vs oracle/gs_oracle.c."""
    w = workloads.flua_codon(5, site)
    eng = bito_amd.Engine(spec(w.substitution, w.site), w.patterns, w.weights)
    cpu = gs.GsOracleEngine("GY94", site, w.patterns, w.weights, 8)
    params = w.params.copy()
    params[1, :4] = [0.1, 0.2, 0.3, 0.4]  # rows differ: the model is per tree
    params[2, 4:6] = [1.0, 1.0]
    out = eng.gradients(w.parent_ids, w.branch_lengths, params)
    assert eng.kernel_name() == "gs_walk_kernel"
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, params)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    assert np.all(out["branch_lengths"][:, -1] == 0.0)
    ll = eng.log_likelihoods(w.parent_ids, w.branch_lengths, params)
    assert ll_close(ll, ref["log_likelihood"])


def test_codon_model_errors():
    w = workloads.flua_codon(1)
    eng = bito_amd.Engine(spec("GY94", "constant"), w.patterns, w.weights)
    bad = w.params.copy()
    bad[0, 4] = -1.0
    with pytest.raises(bito_amd.BitoAmdError, match="kappa and omega"):
        eng.log_likelihoods(w.parent_ids, w.branch_lengths, bad)
    bad = w.params.copy()
    bad[0, 0] = 0.5
    with pytest.raises(bito_amd.BitoAmdError, match="frequencies do not sum to 1"):
        eng.log_likelihoods(w.parent_ids, w.branch_lengths, bad)


def test_codon_model_parameter_gradients():
    """FatBeagle::Gradient's model blocks (src/fat_beagle.cpp:401-508) on the codon model: the site-model
    gradient (second traversal with d r_c / d shape) and the finite-difference substitution-model
    gradient (kappa, omega first, then the nucleotide frequencies), checked against the CPU
    restatement's site-model pass and central differences of its log-likelihood."""
    w = workloads.flua_codon(2, "weibull+3")
    eng = bito_amd.Engine(spec(w.substitution, w.site), w.patterns, w.weights)
    cpu = gs.GsOracleEngine("GY94", w.site, w.patterns, w.weights, 8)
    flags = _capi.GRAD_SITE_MODEL | _capi.GRAD_SUBSTITUTION_MODEL
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, flags=flags, fd_delta=1e-6)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, site_model=True)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    # (a finite difference in the shape is no yardstick here: the O(t^2) entries of P(t) carry 1e-6
    # relative rounding noise that is not smooth in t r_c; the analytic pass is compared instead)
    assert grad_close(out["site_model"], ref["site_model"])

    def fd(column, eps):
        plus, minus = w.params.copy(), w.params.copy()
        plus[:, column] += eps
        minus[:, column] -= eps
        return (cpu.log_likelihoods(w.parent_ids, w.branch_lengths, plus) -
                cpu.log_likelihoods(w.parent_ids, w.branch_lengths, minus)) / (2 * eps)

    want = np.stack([fd(4, 1e-6), fd(5, 1e-6)] + [fd(k, 1e-6) for k in range(4)], axis=1)
    assert np.abs(out["substitution_model"] - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    # the main pass is the resident one afterwards
    ll, grad = eng.download()
    assert np.array_equal(ll, out["log_likelihood"]) and np.array_equal(grad, out["branch_lengths"])


def test_codon_resident_batch_update():
    """upload once, refresh branch lengths and parameters in place (vip's particle loop)."""
    w = workloads.flua_codon(6)
    eng = bito_amd.Engine(spec(w.substitution, w.site), w.patterns, w.weights)
    cpu = gs.GsOracleEngine("GY94", w.site, w.patterns, w.weights, 8)
    eng.upload(w.parent_ids, w.branch_lengths, w.params)
    eng.run(True)
    ll0, g0 = eng.download()
    bl = w.branch_lengths * 1.25
    params = w.params.copy()
    params[3:, 4] = 1.7  # two distinct models in the batch now
    eng.update(bl, params)
    eng.run(True)
    ll1, g1 = eng.download()
    ref = cpu.gradients(w.parent_ids, bl, params)
    assert ll_close(ll1, ref["log_likelihood"]) and grad_close(g1, ref["branch_lengths"])
    assert not np.allclose(ll0, ll1)


def test_codon_model_parity_over_many_trees_and_models():
    """48 trees, each with its own branch-length scale (0.1x .. 10x), nucleotide frequencies, kappa and
    omega: the set-up arithmetic (rate matrix, eigensystem, deterministic exp, fma-chain P(t)) is
    bit-compatible with the CPU restatement, so parity does not depend on how well a tree happens to
    be conditioned."""
    T = 48
    w = workloads.flua_codon(T, "weibull+2")
    rng = np.random.default_rng(99)
    bl = w.branch_lengths * np.exp(rng.uniform(np.log(0.1), np.log(10.0), (T, 1)))
    params = w.params.copy()
    f = rng.dirichlet([8, 8, 8, 8], T)
    params[:, :4] = f
    params[:, 4] = rng.uniform(0.5, 6.0, T)
    params[:, 5] = rng.uniform(0.05, 1.5, T)
    params[:, 6] = rng.uniform(0.3, 2.0, T)
    eng = bito_amd.Engine(spec(w.substitution, w.site), w.patterns, w.weights)
    cpu = gs.GsOracleEngine("GY94", w.site, w.patterns, w.weights, 16)
    out = eng.gradients(w.parent_ids, bl, params)
    ref = cpu.gradients(w.parent_ids, bl, params)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])


@pytest.mark.parametrize("site", ["constant", "weibull+2"])
def test_codon_rescaling(site):
    """Engine's `rescaling` argument (BEAGLE manual scaling, src/fat_beagle.cpp:353-364) on the general-state
    kernels: same results as without on fluA, and finite results where the unscaled partials underflow."""
    w = workloads.flua_codon(3, site)
    eng = bito_amd.Engine(spec(w.substitution, w.site), w.patterns, w.weights)
    cpu = gs.GsOracleEngine("GY94", site, w.patterns, w.weights, 8)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=True)
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=True)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    assert ll_close(eng.log_likelihoods(w.parent_ids, w.branch_lengths, w.params, rescaling=True), ref["log_likelihood"])
    # 300 taxa of random codons on a random tree: every site likelihood is far below 1e-308
    n, P, T = 300, 24, 2
    rng = np.random.default_rng(5)
    patterns = rng.integers(0, 62, (n, P)).astype(np.int32)
    trees = [workloads.random_unrooted_tree(n, np.random.default_rng(50 + i), 0.05) for i in range(T)]
    pid = np.stack([t.parent_ids for t in trees]).astype(np.int32)
    bl = np.stack([t.branch_lengths for t in trees])
    bl[:, -1] = 0.0
    params = np.tile(w.params[:1], (T, 1))
    eng = bito_amd.Engine(spec("GY94", site), patterns, np.ones(P))
    cpu = gs.GsOracleEngine("GY94", site, patterns, np.ones(P), 8)
    ref = cpu.gradients(pid, bl, params, rescaling=True)
    out = eng.gradients(pid, bl, params, rescaling=True)
    assert np.all(np.isfinite(ref["log_likelihood"])) and ref["log_likelihood"].max() < -20000
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    plain = eng.log_likelihoods(pid, bl, params, rescaling=False)
    assert not np.all(np.isfinite(plain))  # which is why the argument exists


def test_general_kernels_rooted_hky_eight_categories(data_dir):
    """Rooted trees with per-branch rates (src/fat_beagle.cpp:86-90), HKY written as GTR inside the general
    set-up kernel, eight rate categories: the general kernels against the pinned 4-state CPU oracle.  Two
    correct FP64 set-ups differ coherently across patterns, hence 2e-9 / 1e-5."""
    from oracle import oracle

    tc = treeio.read_newick_file(os.path.join(data_dir, "fluA.tree"))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, "fluA.fa")), tc.taxon_names)
    T = 3
    pid = np.tile(tc.parent_id_matrix(), (T, 1))
    bl = np.tile(tc.branch_length_matrix(), (T, 1)) * np.array([[1.0], [0.5], [2.0]])
    rates = np.full((T, pid.shape[1]), 0.001) * np.random.default_rng(3).uniform(0.5, 2.0, (T, pid.shape[1]))
    params = np.tile(np.array([0.1, 0.2, 0.3, 0.4, 3.0, 0.6]), (T, 1))  # pi | kappa | shape
    eng = _general("HKY", "weibull+8", sp.patterns, sp.weights)
    cpu = oracle.OracleEngine("HKY", "weibull+8", "none", sp.patterns, sp.weights, 4)
    out = eng.gradients(pid, bl, params, rates=rates)
    ref = cpu.gradients(pid, bl, params, rates=rates)
    assert eng.kernel_name() == "gs_walk_kernel"
    assert np.abs(out["log_likelihood"] - ref["log_likelihood"]).max() < 2e-9
    scale = np.maximum(1.0, np.abs(ref["branch_lengths"]))
    assert (np.abs(out["branch_lengths"] - ref["branch_lengths"]) / scale).max() < 1e-5
    res = eng.gradients(pid, bl, params, rates=rates, rescaling=True)
    assert np.abs(res["log_likelihood"] - out["log_likelihood"]).max() < 1e-9
    assert (np.abs(res["branch_lengths"] - out["branch_lengths"]) / scale).max() < 1e-8


def test_codon_full_size_batch_properties():
    """BASELINE config 5 at the bench's batch size (1024 trees): properties that need no oracle.  A tree's
    result does not depend on what else is in the batch (bitwise), runs are bit-reproducible (no atomics),
    doubling every pattern weight doubles log-likelihoods and gradients exactly, and the first trees agree
    with the CPU restatement."""
    w = workloads.flua_codon(1024)
    eng = bito_amd.Engine(spec(w.substitution, w.site), w.patterns, w.weights)
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    again = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert np.array_equal(out["log_likelihood"], again["log_likelihood"])
    assert np.array_equal(out["branch_lengths"], again["branch_lengths"])
    assert np.all(np.isfinite(out["log_likelihood"])) and np.all(np.isfinite(out["branch_lengths"]))
    k = 37
    pick = np.arange(1024 - k, 1024)
    part = eng.gradients(w.parent_ids[pick], w.branch_lengths[pick], w.params[pick])
    assert np.array_equal(part["log_likelihood"], out["log_likelihood"][pick])
    assert np.array_equal(part["branch_lengths"], out["branch_lengths"][pick])
    twice = bito_amd.Engine(spec(w.substitution, w.site), w.patterns, 2.0 * w.weights)
    dbl = twice.gradients(w.parent_ids[:64], w.branch_lengths[:64], w.params[:64])
    assert np.array_equal(dbl["log_likelihood"], 2.0 * out["log_likelihood"][:64])
    assert np.array_equal(dbl["branch_lengths"], 2.0 * out["branch_lengths"][:64])
    cpu = gs.GsOracleEngine("GY94", w.site, w.patterns, w.weights, 8)
    ref = cpu.gradients(w.parent_ids[:8], w.branch_lengths[:8], w.params[:8])
    assert ll_close(out["log_likelihood"][:8], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"][:8], ref["branch_lengths"])


def test_instance_mirror_with_the_codon_model(data_dir):
    """The pybind-name mirror (bito_amd/instance.py) on the codon model: an unrooted instance reads
    DS1.fasta as codons when the model specification says GY94, same calls as for DNA."""
    inst = bito_amd.unrooted_instance("ds1")
    inst.read_nexus_file(os.path.join(data_dir, "DS1.subsampled_10.t"))
    inst.read_fasta_file(os.path.join(data_dir, "DS1.fasta"))
    inst.prepare_for_phylo_likelihood(bito_amd.PhyloModelSpecification("GY94", "weibull+2", "none"), 1)
    blocks = inst.get_phylo_model_param_block_map()
    blocks["substitution_model_frequencies"][:] = [0.3, 0.2, 0.25, 0.25]
    blocks["substitution_model_rates"][:] = [2.0, 0.4]
    blocks["Weibull_shape"][:] = 0.8
    ll = inst.log_likelihoods()
    grads = inst.phylo_gradients(flags=0)
    from bito_amd.site_pattern import CodonSitePattern

    sp = CodonSitePattern(treeio.read_fasta(os.path.join(data_dir, "DS1.fasta")), inst.taxon_names())
    assert sp.patterns.max() <= 61 and sp.weights.sum() == 1949 // 3
    cpu = gs.GsOracleEngine("GY94", "weibull+2", sp.patterns, sp.weights, 8)
    pid = np.stack([t._parent_ids for t in inst.tree_collection.trees]).astype(np.int32)
    bl = np.stack([np.asarray(t.branch_lengths) for t in inst.tree_collection.trees])
    ref = cpu.gradients(pid, bl, inst.get_phylo_model_params())
    assert ll_close(ll, ref["log_likelihood"])
    assert grad_close(np.stack([g.gradient["branch_lengths"] for g in grads]), ref["branch_lengths"])
