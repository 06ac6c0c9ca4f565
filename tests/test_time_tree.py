"""Time trees on the GPU: RootedTree's height-ratio parameterisation, the
log-det-Jacobian of the height transform and the ratios_root_height / clock gradients, against
the reference's goldens and the CPU oracle.  The GPU tests read like the reference's own
(src/rooted_tree.hpp:133-168, src/rooted_sbn_instance.hpp:277-345,432-455)."""
import json
import os

import numpy as np
import pytest

import bito_amd
from bito_amd import _capi, treeio
from bito_amd.site_pattern import SitePattern
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_goldens.json")) as fh:
    GOLD = json.load(fh)


def spec(sub, site, clock="strict"):
    return bito_amd.PhyloModelSpecification(sub, site, clock)


def make_flu_instance(data_dir, initialize_time_trees, model=("JC69", "constant", "strict")):
    """MakeFluInstance (reference src/rooted_sbn_instance.hpp:262-275)."""
    inst = bito_amd.rooted_instance("charlie")
    inst.read_newick_file(os.path.join(data_dir, "fluA.tree"), False)  # as the reference: ReadNewickFile(..., false)
    inst.parse_dates_from_taxon_names(initialize_time_trees)
    inst.read_fasta_file(os.path.join(data_dir, "fluA.fa"))
    inst.prepare_for_phylo_likelihood(spec(*model), 1)
    for tree in inst.tree_collection.trees:
        tree.rates[:] = 0.001
    return inst


# ---- CPU: host logic ---------------------------------------------------------------------------

def test_parsing_dates(data_dir):
    """reference src/rooted_sbn_instance.hpp:432-450"""
    inst = bito_amd.rooted_instance("charlie")
    inst.read_nexus_file(os.path.join(data_dir, "test_beast_tree_parsing.nexus"), False)
    inst.parse_dates_from_taxon_names(True)
    dates = sorted(inst.tip_dates().values())
    assert dates[0] == 0 and dates[-1] == 80.0
    alt = bito_amd.rooted_instance("betty")
    alt.read_nexus_file(os.path.join(data_dir, "test_beast_tree_parsing.nexus"), False)
    alt.parse_dates_from_csv(os.path.join(data_dir, "test_beast_tree_parsing.csv"), True)
    assert inst.tip_dates() == alt.tip_dates()
    with pytest.raises(RuntimeError, match="Couldn't parse a date"):
        treeio.parse_dates_from_taxon_names(["mars", "saturn_1"])


def test_time_tree_symbols_are_declared():
    with open(os.path.join(HERE, "..", "include", "bito_amd.h")) as fh:
        header = fh.read()
    for sym in _capi.SYMBOLS:
        assert sym + "(" in header, sym


# ---- GPU ---------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_rooted_tree_example_on_device():
    """RootedTree doctest (reference src/rooted_tree.hpp:133-168), exact equality."""
    g = GOLD["rooted_tree_example"]
    patterns = np.zeros((4, 1), dtype=np.int32)
    eng = bito_amd.Engine(spec("JC69", "constant"), patterns, np.ones(1))
    pid = np.array([g["parent_ids"]], dtype=np.int32)
    bounds, heights, ratios = eng.time_trees_from_branch_lengths(pid, [g["branch_lengths"]], g["tip_dates"])
    assert list(ratios[0]) == g["height_ratios"]
    assert list(heights[0]) == g["node_heights"]
    assert list(bounds[0]) == g["node_bounds"]
    heights2, bl2 = eng.time_trees_from_height_ratios(pid, bounds, [g["new_height_ratios"]])
    assert list(heights2[0]) == g["new_node_heights"]
    assert list(bl2[0][:-1]) == g["new_branch_lengths"]
    with pytest.raises(bito_amd.BitoAmdError, match="time-calibrated"):
        eng.time_trees_from_branch_lengths(pid, [[2.0, 1.5, 2.0, 1.5, 2.5, 2.5, 0.0]], g["tip_dates"])


@pytest.mark.gpu
def test_rooted_instance_gradients(data_dir):
    """reference src/rooted_sbn_instance.hpp:277-307 (physher goldens)"""
    g = GOLD["flua_time_tree"]
    inst = make_flu_instance(data_dir, True)
    likelihood = inst.log_likelihoods()
    assert abs(likelihood[0] - (g["log_likelihood"] + g["log_det_jacobian"])) < 1e-4
    assert abs(inst.log_det_jacobian_of_height_transform()[0] - g["log_det_jacobian"]) < 1e-6
    assert abs(inst.log_likelihoods(include_log_det_jacobian=False)[0] - g["log_likelihood"]) < 1e-4
    gradients = inst.phylo_gradients()
    assert np.abs(gradients[0].gradient["ratios_root_height"] - g["ratios_root_height_gradient"]).max() < 1e-4
    assert abs(gradients[0].log_likelihood - g["log_likelihood"]) < 1e-4
    assert set(gradients[0].gradient) == {"branch_lengths", "clock_model", "ratios_root_height"}


@pytest.mark.gpu
def test_uninitialized_time_trees_raise(data_dir):
    """reference src/rooted_sbn_instance.hpp:452-455"""
    inst = make_flu_instance(data_dir, False)
    with pytest.raises(RuntimeError):
        inst.phylo_gradients()


@pytest.mark.gpu
def test_rooted_instance_goldens_on_an_engine_over_two_device_slots(data_dir):
    """Engine::Gradients(RootedTreeCollection) runs over every FatBeagle of the engine (reference src/engine.cpp:94-119):
    the physher goldens of src/rooted_sbn_instance.hpp:277-307 on a collection of three copies of the fluA tree, sharded
    over two device slots (GPU 0 named twice: 1 + 2 trees, each slot on its own host thread)."""
    import copy

    g = GOLD["flua_time_tree"]
    inst = bito_amd.rooted_instance("charlie")
    inst.read_newick_file(os.path.join(data_dir, "fluA.tree"), False)
    inst.tree_collection.trees.extend(copy.deepcopy(inst.tree_collection.trees[0]) for _ in range(2))
    inst.parse_dates_from_taxon_names(True)
    inst.read_fasta_file(os.path.join(data_dir, "fluA.fa"))
    inst.prepare_for_phylo_likelihood(spec("JC69", "constant", "strict"), 2, devices=[0, 0])
    assert inst._get_engine().device_count == 2
    for tree in inst.tree_collection.trees:
        tree.rates[:] = 0.001
    likelihood = inst.log_likelihoods()
    assert likelihood.shape == (3,) and np.abs(likelihood - (g["log_likelihood"] + g["log_det_jacobian"])).max() < 1e-4
    assert np.abs(inst.log_det_jacobian_of_height_transform() - g["log_det_jacobian"]).max() < 1e-6
    gradients = inst.phylo_gradients()
    assert len(gradients) == 3
    for grad in gradients:
        assert np.abs(grad.gradient["ratios_root_height"] - g["ratios_root_height_gradient"]).max() < 1e-4
        assert abs(grad.log_likelihood - g["log_likelihood"]) < 1e-4
    # the batch stays resident, one block per slot: passes over it and downloads address it through the shard list
    eng = inst._get_engine()
    ll, grad = eng.download()
    assert ll.shape == (3,) and np.abs(ll - g["log_likelihood"]).max() < 1e-4 and grad.shape[0] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0]], ids=["one-slot", "two-slots", "three-slots"])
def test_time_tree_transforms_match_oracle(data_dir, devices):
    tc = treeio.read_newick_file(os.path.join(data_dir, "fluA.tree"))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, "fluA.fa")), tc.taxon_names)
    dates = treeio.parse_dates_from_taxon_names(tc.taxon_names)
    pid1, bl1 = tc.parent_id_matrix(), tc.branch_length_matrix()
    n = sp.taxon_count
    # a batch of distinct time trees on the same topology: perturbed height ratios
    ref0 = oracle.TimeTree(pid1[0], bl1[0], dates)
    rng = np.random.default_rng(5)
    T = 7
    refs, ratios = [], []
    for t in range(T):
        r = ref0.height_ratios.copy()
        if t:
            r[:-1] = np.clip(r[:-1] * rng.uniform(0.8, 1.2, n - 2), 1e-3, 0.999)
            r[-1] *= rng.uniform(1.0, 1.3)
        tt = oracle.TimeTree(pid1[0], bl1[0], dates)
        tt.initialize_time_tree_using_height_ratios(r)
        refs.append(tt)
        ratios.append(r)
    pid = np.repeat(pid1, T, axis=0)
    ratios = np.array(ratios)
    # (seven trees over two or three device slots -- GPU 0 named more than once: blocks of 3 + 4, of 2 + 2 + 3)
    eng = bito_amd.Engine(spec("JC69", "weibull+4"), sp.patterns, sp.weights, devices=devices)
    bounds = np.stack([tt.node_bounds for tt in refs])
    heights, bl = eng.time_trees_from_height_ratios(pid, bounds, ratios)
    assert np.abs(heights - np.stack([tt.node_heights for tt in refs])).max() < 1e-12
    assert np.abs(bl[:, :-1] - np.stack([tt.branch_lengths[:-1] for tt in refs])).max() < 1e-12
    # from branch lengths: bounds, heights, ratios round-trip
    b2, h2, r2 = eng.time_trees_from_branch_lengths(pid, bl, dates)
    assert np.array_equal(b2, bounds) and np.abs(h2 - heights).max() < 1e-10 and np.abs(r2 - ratios).max() < 1e-10
    ldj = eng.log_det_jacobian(pid, heights, bounds)
    assert np.abs(ldj - [tt.log_det_jacobian() for tt in refs]).max() < 1e-11
    gldj = eng.gradient_log_det_jacobian(pid, heights, bounds, ratios)
    ref_gldj = np.stack([tt.gradient_log_det_jacobian() for tt in refs])
    assert np.abs(gldj - ref_gldj).max() < 1e-9 * max(1.0, np.abs(ref_gldj).max())
    hg = rng.normal(size=(T, n - 1))
    rg = eng.ratio_gradient_of_height_gradient(pid, heights, bounds, ratios, hg)
    ref_rg = np.stack([tt.ratio_gradient_of_height_gradient(hg[t]) for t, tt in enumerate(refs)])
    assert np.abs(rg - ref_rg).max() < 1e-9 * max(1.0, np.abs(ref_rg).max())
    # the composed gradient call: strict and per-branch clock, with and without the Jacobian term
    cpu = oracle.OracleEngine("JC69", "weibull+4", "strict", sp.patterns, sp.weights, 2)
    params = eng.default_params(T)
    params[:, eng.block_map()["Weibull_shape"][0]] = 0.3
    rates = np.full((T, 2 * n - 2), 0.001) * (1.0 + (np.arange(2 * n - 2) % 3))
    ref = cpu.gradients(pid, bl, params, rates=rates, flags=oracle.GRAD_SITE_MODEL)
    for rate_count in (1, 2 * n - 2):
        for jac in (0, _capi.GRAD_LOG_DET_JACOBIAN_GRADIENT):
            flags = _capi.GRAD_SITE_MODEL | _capi.GRAD_CLOCK_MODEL | _capi.GRAD_RATIOS_ROOT_HEIGHT | jac
            out = eng.time_tree_gradients(pid, bl, heights, bounds, ratios, params, rates=rates,
                                          rate_count=rate_count, flags=flags)
            assert np.abs(out["log_likelihood"] - ref["log_likelihood"]).max() < 1e-9
            assert np.abs(out["branch_lengths"] - ref["branch_lengths"]).max() < 1e-6 * np.abs(ref["branch_lengths"]).max()
            assert np.abs(out["site_model"] - ref["site_model"]).max() < 1e-6 * max(1.0, np.abs(ref["site_model"]).max())
            want_ratio = np.stack([tt.ratio_gradient_of_branch_gradient(ref["branch_lengths"][t], rates[t], bool(jac))
                                   for t, tt in enumerate(refs)])
            assert np.abs(out["ratios_root_height"] - want_ratio).max() < 1e-6 * max(1.0, np.abs(want_ratio).max())
            per_branch = ref["branch_lengths"][:, :-1] * bl[:, :-1]  # ClockGradient, fat_beagle.cpp:379-399
            want_clock = per_branch.sum(axis=1, keepdims=True) if rate_count == 1 else per_branch
            assert out["clock_model"].shape == want_clock.shape
            assert np.abs(out["clock_model"] - want_clock).max() < 1e-6 * max(1.0, np.abs(want_clock).max())
    with pytest.raises(bito_amd.BitoAmdError, match="number of rates"):
        eng.time_tree_gradients(pid, bl, heights, bounds, ratios, params, rates=rates, rate_count=3,
                                flags=_capi.GRAD_CLOCK_MODEL)
    # an error in a later slot's block names the caller's tree
    if devices is not None:
        off = bl.copy()
        off[T - 1, 3] += 0.5
        with pytest.raises(bito_amd.BitoAmdError, match=rf"time-calibrated.*\(tree {T - 1}\)"):
            eng.time_trees_from_branch_lengths(pid, off, dates)
    # log-likelihood with the Jacobian of the height transform
    ll = eng.time_tree_log_likelihoods(pid, bl, heights, bounds, params, rates=rates)
    assert np.abs(ll - (ref["log_likelihood"] + ldj)).max() < 1e-9


@pytest.mark.gpu
def test_height_ratio_setter_on_tree(data_dir):
    """pybito ``RootedTree.initialize_time_tree_using_height_ratios`` through the instance."""
    inst = make_flu_instance(data_dir, True)
    tree = inst.tree_collection.trees[0]
    before = inst.log_likelihoods()[0]
    ratios = tree.height_ratios.copy()
    ratios[-1] *= 1.1  # a taller root
    tree.initialize_time_tree_using_height_ratios(ratios)
    assert tree.node_heights[-1] == ratios[-1]
    after = inst.log_likelihoods()[0]
    assert after != before
    # the ratio gradient is the derivative of LL + log|J| in the ratios: central difference in the root height
    g = inst.phylo_gradients()[0].gradient["ratios_root_height"]
    eps = 1e-4 * ratios[-1]
    vals = []
    for sgn in (1, -1):
        r = ratios.copy()
        r[-1] += sgn * eps
        tree.initialize_time_tree_using_height_ratios(r)
        vals.append(inst.log_likelihoods()[0])
    assert abs((vals[0] - vals[1]) / (2 * eps) - g[-1]) < 1e-4 * max(1.0, abs(g[-1]))
