"""Pins the CPU oracle against every known-answer value the reference's own
tests hold for the FatBeagle/Engine path.  CPU only."""
import json
import os

import numpy as np
import pytest

from bito_amd import site_pattern, treeio
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_goldens.json")) as fh:
    GOLD = json.load(fh)

# The oracle's own tightness against the 17-digit pybeagle/physher values
# (tighter than the reference's 1.1e-4: pins ~1e-10 relative agreement).
LL_TIGHT = 5e-10


def _load(data_dir, fasta, trees):
    path = os.path.join(data_dir, trees)
    tc = treeio.read_nexus_file(path) if trees.endswith(".t") else treeio.read_newick_file(path)
    sp = site_pattern.SitePattern(treeio.read_fasta(os.path.join(data_dir, fasta)), tc.taxon_names)
    return tc, sp


def test_hello_jc69(data_dir):
    g = GOLD["hello_jc69_unrooted"]
    tc, sp = _load(data_dir, g["fasta"], g["trees"])
    assert sp.patterns.shape == (3, 15) and sp.weights.sum() == 31
    eng = oracle.OracleEngine("JC69", "constant", "strict", sp.patterns, sp.weights, 2)
    ll = eng.log_likelihoods(tc.parent_id_matrix(), tc.branch_length_matrix())
    assert abs(ll[0] - g["log_likelihood"]) < g["tol"]


def test_hello_vip_branch_lengths(data_dir):
    g = GOLD["hello_jc69_vip_branch_lengths"]
    tc, sp = _load(data_dir, "hello.fasta", "hello.nwk")
    bl = tc.branch_length_matrix()
    for name, value in g["branch_lengths_by_taxon"].items():
        bl[0, tc.taxon_names.index(name)] = value
    eng = oracle.OracleEngine("JC69", "constant", "strict", sp.patterns, sp.weights, 1)
    ll = eng.log_likelihoods(tc.parent_id_matrix(), bl)[0]
    # the reference test compares an ELBO estimate built on this value at 1e-6 relative;
    # the underlying tree log-likelihood agrees to the branch lengths' 6 printed digits
    assert abs(ll - g["log_likelihood"]) < 1e-4


@pytest.mark.parametrize("tip_states", [True, False])
@pytest.mark.parametrize("rescaling", [False, True])
def test_ds1_jc69(data_dir, tip_states, rescaling):
    g = GOLD["ds1_jc69"]
    tc, sp = _load(data_dir, g["fasta"], g["trees"])
    assert sp.patterns.shape == (27, 934)
    eng = oracle.OracleEngine("JC69", "constant", "strict", sp.patterns, sp.weights, 2, tip_states)
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    ll = eng.log_likelihoods(pid, bl, rescaling=rescaling)
    assert np.abs(ll - g["log_likelihoods"]).max() < LL_TIGHT
    grads = eng.gradients(pid, bl, rescaling=rescaling)
    assert np.abs(grads["log_likelihood"] - g["log_likelihoods"]).max() < LL_TIGHT
    last = np.sort(grads["branch_lengths"][-1])
    assert np.abs(last - g["last_tree_sorted_branch_gradient"]).max() < g["gradient_tol"]
    # zeros are the root and the fixed node (fat_beagle.cpp:148,553)
    assert grads["branch_lengths"][-1][-1] == 0.0 and grads["branch_lengths"][-1][-2] == 0.0


@pytest.mark.parametrize("tip_states", [True, False])
def test_ds1_jc69_weibull(data_dir, tip_states):
    g = GOLD["ds1_jc69_weibull4_shape0.1"]
    tc, sp = _load(data_dir, g["fasta"], g["trees"])
    eng = oracle.OracleEngine("JC69", "weibull+4", "strict", sp.patterns, sp.weights, 2, tip_states)
    params = eng.default_params(len(tc.trees))
    params[:, eng.block_map()["Weibull_shape"][0]] = g["shape"]
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    for rescaling in (False, True):
        ll = eng.log_likelihoods(pid, bl, params, rescaling=rescaling)
        assert np.abs(ll - g["log_likelihoods"]).max() < LL_TIGHT
        grads = eng.gradients(pid, bl, params, rescaling=rescaling)
        assert np.abs(grads["branch_lengths"][:, 0] - g["branch_gradient_entry0"]).max() < 2e-6


def _flu(data_dir):
    tc, sp = _load(data_dir, "fluA.fa", "fluA.tree")
    assert tc.trees[0].rooted and sp.taxon_count == 69
    rates = np.full((1, tc.trees[0].node_count - 1), 0.001)
    return tc, sp, rates


def test_flua_rooted_jc69(data_dir):
    g = GOLD["flua_jc69_strict"]
    tc, sp, rates = _flu(data_dir)
    eng = oracle.OracleEngine("JC69", "constant", "strict", sp.patterns, sp.weights, 1)
    ll = eng.log_likelihoods(tc.parent_id_matrix(), tc.branch_length_matrix(), rates=rates)
    assert abs(ll[0] - g["log_likelihood"]) < 2e-6
    # strict-clock gradient agrees with a central difference in the rate
    # (the reference's own check, src/rooted_sbn_instance.hpp:74-95,321-327)
    grads = eng.gradients(tc.parent_id_matrix(), tc.branch_length_matrix(), rates=rates,
                          flags=oracle.GRAD_CLOCK_MODEL)
    eps = 1e-8
    lp = eng.log_likelihoods(tc.parent_id_matrix(), tc.branch_length_matrix(), rates=rates + eps)[0]
    lm = eng.log_likelihoods(tc.parent_id_matrix(), tc.branch_length_matrix(), rates=rates - eps)[0]
    assert abs(grads["clock_model"][0] - (lp - lm) / (2 * eps)) < 1e-3  # the reference's tolerance


def test_flua_weibull_site_gradient(data_dir):
    g = GOLD["flua_jc69_weibull4_shape0.1"]
    tc, sp, rates = _flu(data_dir)
    eng = oracle.OracleEngine("JC69", "weibull+4", "strict", sp.patterns, sp.weights, 1)
    params = eng.default_params(1)
    params[:, eng.block_map()["Weibull_shape"][0]] = g["shape"]
    grads = eng.gradients(tc.parent_id_matrix(), tc.branch_length_matrix(), params, rates=rates,
                          flags=oracle.GRAD_SITE_MODEL)
    assert abs(grads["log_likelihood"][0] - g["log_likelihood"]) < 1e-9
    assert abs(grads["site_model"][0] - g["site_model_gradient"]) < 1e-6


def test_flua_gtr(data_dir):
    g = GOLD["flua_gtr"]
    tc, sp, rates = _flu(data_dir)
    eng = oracle.OracleEngine("GTR", "constant", "strict", sp.patterns, sp.weights, 1)
    bm = eng.block_map()
    assert bm["substitution_model_frequencies"] == (0, 4) and bm["substitution_model_rates"] == (4, 6)
    params = eng.default_params(1)
    params[0, 0:4] = g["frequencies"]
    params[0, 4:10] = g["rates"]
    grads = eng.gradients(tc.parent_id_matrix(), tc.branch_length_matrix(), params, rates=rates,
                          flags=oracle.GRAD_SUBSTITUTION_MODEL | oracle.GRAD_STICKBREAKING)
    assert abs(grads["log_likelihood"][0] - g["log_likelihood"]) < g["tol"]
    assert np.abs(grads["substitution_model"][0] - g["substitution_model_gradient"]).max() < g["gradient_tol"]


def test_flua_hky(data_dir):
    g = GOLD["flua_hky"]
    tc, sp, rates = _flu(data_dir)
    eng = oracle.OracleEngine("HKY", "constant", "strict", sp.patterns, sp.weights, 1)
    params = eng.default_params(1)
    params[0, 0:4] = g["frequencies"]
    params[0, 4] = g["kappa"]
    grads = eng.gradients(tc.parent_id_matrix(), tc.branch_length_matrix(), params, rates=rates,
                          flags=oracle.GRAD_SUBSTITUTION_MODEL | oracle.GRAD_STICKBREAKING)
    assert abs(grads["log_likelihood"][0] - g["log_likelihood"]) < 1e-8
    assert np.abs(grads["substitution_model"][0] - g["substitution_model_gradient"]).max() < g["gradient_tol"]


def test_substitution_model_known_answers():
    g = GOLD["gtr_eigenvalues_r"]
    _, _, _, lam, _ = oracle.substitution_model("GTR", np.array(g["frequencies"] + g["rates"]))
    assert np.abs(np.sort(lam) - np.sort(g["eigenvalues"])).max() < g["tol"]
    # JC69 == GTR(equal) == HKY(kappa 1) eigenvalues (src/substitution_model.hpp:126-145)
    jc = oracle.substitution_model("JC69")
    gtr = oracle.substitution_model("GTR", np.array([0.25] * 4 + [1 / 6] * 6))
    hky = oracle.substitution_model("HKY", np.array([0.25] * 4 + [1.0]))
    assert np.abs(np.sort(jc[3]) - np.sort(gtr[3])).max() < 1e-12
    assert np.abs(np.sort(jc[3]) - np.sort(hky[3])).max() < 1e-12
    # HKY == GTR with kappa-shaped rates, Q included (src/substitution_model.hpp:153-167)
    hky = oracle.substitution_model("HKY", np.array([0.1, 0.2, 0.3, 0.4, 3.0]))
    gtr = oracle.substitution_model("GTR", np.array([0.1, 0.2, 0.3, 0.4, 0.1, 0.3, 0.1, 0.1, 0.3, 0.1]))
    assert np.allclose(hky[0], gtr[0], rtol=1e-12, atol=1e-14)
    assert np.abs(np.sort(hky[3]) - np.sort(gtr[3])).max() < 1e-12
    for Q, V, Vi, lam, pi in (jc, gtr, hky):
        assert np.allclose(V @ Vi, np.eye(4), atol=1e-12)
        assert np.allclose(V @ np.diag(lam) @ Vi, Q, atol=1e-12)


def test_weibull_and_transition_known_answers():
    g = GOLD["weibull_rates_r"]
    r1, w1, _ = oracle.weibull_rates(4, 1.0)
    r2, w2, d2 = oracle.weibull_rates(4, 0.1)
    assert np.abs(r1 - g["shape_1"]).max() < g["tol"] and np.abs(r2 - g["shape_0.1"]).max() < g["tol"]
    assert np.allclose(w1, 0.25) and abs(r1 @ w1 - 1) < 1e-12 and abs(r2 @ w2 - 1) < 1e-12
    eps = 1e-7
    fd = (oracle.weibull_rates(4, 0.1 + eps)[0] - oracle.weibull_rates(4, 0.1 - eps)[0]) / (2 * eps)
    assert np.allclose(d2, fd, rtol=1e-5, atol=1e-9)
    g = GOLD["jc69_p_0.75"]
    _, V, Vi, lam, _ = oracle.substitution_model("JC69")
    P = oracle.transition_matrix(V, Vi, lam, g["t"])
    assert abs(P[0, 0] - g["diag"]) < g["tol"] and abs(P[0, 1] - g["offdiag"]) < g["tol"]


def test_ds1_jc69_equals_gtr_equal(data_dir):
    """test/test_bito.py:97-122 of the reference: JC69 == GTR with equal rates and
    frequencies on the 100 DS1 topologies with every branch length 0.1."""
    tc, sp = _load(data_dir, "DS1.fasta", "DS1.100_topologies.nwk")
    assert len(tc.trees) == 100 and not tc.trees[0].rooted
    pid = tc.parent_id_matrix()
    bl = np.full_like(tc.branch_length_matrix(), 0.1)
    jc = oracle.OracleEngine("JC69", "constant", "none", sp.patterns, sp.weights, 4)
    gtr = oracle.OracleEngine("GTR", "constant", "none", sp.patterns, sp.weights, 4)
    a = jc.log_likelihoods(pid, bl)
    b = gtr.log_likelihoods(pid, bl)
    assert np.abs(a - b).max() < 1e-9
    ga = jc.gradients(pid, bl)["branch_lengths"]
    gb = gtr.gradients(pid, bl)["branch_lengths"]
    assert np.abs(ga - gb).max() < 1e-8


def test_gradient_matches_finite_differences(data_dir):
    """Branch gradient vs central differences of the oracle's own log-likelihood
    for the headline model (GTR + weibull+4), which the reference has no golden for."""
    tc, sp = _load(data_dir, "DS1.fasta", "DS1.subsampled_10.t")
    eng = oracle.OracleEngine("GTR", "weibull+4", "none", sp.patterns, sp.weights, 4)
    params = eng.default_params(1)
    params[0, 0:4] = [0.1, 0.2, 0.3, 0.4]
    params[0, 4:10] = [0.05, 0.1, 0.15, 0.20, 0.25, 0.25]
    params[0, 10] = 0.5
    pid = tc.parent_id_matrix()[3:4]
    bl = tc.branch_length_matrix()[3:4]
    g = eng.gradients(pid, bl, params)["branch_lengths"][0]
    M = bl.shape[1]
    n = 27
    root = M - 1  # trifurcating root id 2n-3: no branch
    for b in list(range(0, M - 1, 7)) + [M - 2]:
        h = 1e-6 * max(bl[0, b], 1e-3)
        up, dn = bl.copy(), bl.copy()
        up[0, b] += h
        dn[0, b] -= h
        fd = (eng.log_likelihoods(pid, up, params)[0] - eng.log_likelihoods(pid, dn, params)[0]) / (2 * h)
        assert abs(fd - g[b]) < 1e-6 * max(1.0, abs(g[b])) * 50, (b, fd, g[b])
    assert g[root] == 0.0 and g[2 * n - 2] == 0.0


def test_error_paths(data_dir):
    tc, sp = _load(data_dir, "hello.fasta", "hello.nwk")
    with pytest.raises(oracle.OracleError, match="Substitution model not known"):
        oracle.OracleEngine("F81", "constant", "none", sp.patterns, sp.weights)
    with pytest.raises(oracle.OracleError, match="Site model not known"):
        oracle.OracleEngine("JC69", "gamma", "none", sp.patterns, sp.weights)
    with pytest.raises(oracle.OracleError, match="Thread count"):
        oracle.OracleEngine("JC69", "constant", "none", sp.patterns, sp.weights, 0)
    eng = oracle.OracleEngine("GTR", "constant", "none", sp.patterns, sp.weights)
    bad = eng.default_params(1)
    bad[0, 0] = 0.5
    with pytest.raises(oracle.OracleError, match="frequencies do not sum"):
        eng.log_likelihoods(tc.parent_id_matrix(), tc.branch_length_matrix(), bad)
    bad = eng.default_params(1)
    bad[0, 5] = 0.5
    with pytest.raises(oracle.OracleError, match="rates do not sum"):
        eng.log_likelihoods(tc.parent_id_matrix(), tc.branch_length_matrix(), bad)


def test_rooted_tree_example_time_parameterisation():
    """RootedTree doctest (reference src/rooted_tree.hpp:133-168): exact equality."""
    g = GOLD["rooted_tree_example"]
    tt = oracle.TimeTree(g["parent_ids"], g["branch_lengths"], g["tip_dates"])
    assert list(tt.height_ratios) == g["height_ratios"]
    assert list(tt.node_heights) == g["node_heights"]
    assert list(tt.node_bounds) == g["node_bounds"]
    tt.initialize_time_tree_using_height_ratios(g["new_height_ratios"])
    assert list(tt.node_heights) == g["new_node_heights"]
    assert list(tt.branch_lengths[:-1]) == g["new_branch_lengths"]
    with pytest.raises(RuntimeError):
        oracle.TimeTree(g["parent_ids"], [2.0, 1.5, 2.0, 1.5, 2.5, 2.5, 0.0], g["tip_dates"])


def test_flua_ratio_gradient_and_log_det_jacobian(data_dir):
    """fluA: log-det-Jacobian and the 68 ratio / root-height gradients
    (reference src/rooted_sbn_instance.hpp:277-307)."""
    g = GOLD["flua_time_tree"]
    tc, sp, rates = _flu(data_dir)
    dates = treeio.parse_dates_from_taxon_names(tc.taxon_names)
    tt = oracle.TimeTree(tc.parent_id_matrix()[0], tc.branch_length_matrix()[0], dates)
    assert abs(tt.log_det_jacobian() - g["log_det_jacobian"]) < 1e-6
    eng = oracle.OracleEngine("JC69", "constant", "strict", sp.patterns, sp.weights, 1)
    grads = eng.gradients(tc.parent_id_matrix(), tc.branch_length_matrix(), rates=rates)
    assert abs(grads["log_likelihood"][0] - g["log_likelihood"]) < g["tol"]
    ratio = tt.ratio_gradient_of_branch_gradient(grads["branch_lengths"][0], rates[0])
    assert np.abs(ratio - g["ratios_root_height_gradient"]).max() < g["gradient_tol"]
    # round trip: ratios -> heights / branch lengths reproduces the tree
    bl = tt.branch_lengths.copy()
    tt.initialize_time_tree_using_height_ratios(tt.height_ratios)
    assert np.abs(tt.branch_lengths[:-1] - bl[:-1]).max() < 1e-4  # the file's branch lengths are rounded
    # gradient of the log-det-Jacobian agrees with central differences in the ratios
    base = tt.height_ratios.copy()
    gj = tt.gradient_log_det_jacobian()
    for i in (0, 17, len(base) - 1):
        eps = 1e-6 * max(1.0, abs(base[i]))
        vals = []
        for sgn in (1, -1):
            r = base.copy()
            r[i] += sgn * eps
            tt.initialize_time_tree_using_height_ratios(r)
            vals.append(tt.log_det_jacobian())
        assert abs((vals[0] - vals[1]) / (2 * eps) - gj[i]) < 1e-5 * max(1.0, abs(gj[i]))
