"""Parity of the HIP engine (through the C ABI) against the CPU oracle and the
reference's golden values.  Needs a real MI355X: run with ``-m gpu``.

Tolerances (BASELINE.json north_star): 1e-10 on log-likelihoods, 1e-6 on gradients;
for log-likelihoods whose magnitude makes 1e-10 smaller than a few ulps the bound is
relative (LL_RTOL).
"""
import json
import os

import numpy as np
import pytest

import bito_amd
from bito_amd import _capi, treeio, workloads
from bito_amd.site_pattern import SitePattern
from oracle import oracle

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_goldens.json")) as fh:
    GOLD = json.load(fh)

LL_ATOL = 1e-10
LL_RTOL = 2e-14
GRAD_ATOL = 1e-6
GRAD_RTOL = 1e-9


class _Close:
    """Truthy when |a-b| <= atol + rtol*|b| everywhere; repr shows the worst offender."""

    def __init__(self, a, b, atol, rtol):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        self.err = np.abs(a - b)
        self.bound = atol + rtol * np.abs(b)
        self.ok = bool(np.all(self.err <= self.bound)) and a.shape == b.shape
        k = int(np.argmax(self.err - self.bound)) if self.err.size else 0
        self.msg = (f"max|a-b|={self.err.max() if self.err.size else 0:.3e} at flat index {k}: "
                    f"a={a.reshape(-1)[k] if a.size else None!r} b={b.reshape(-1)[k] if b.size else None!r} "
                    f"bound={self.bound.reshape(-1)[k] if self.err.size else 0:.3e}")

    def __bool__(self):
        return self.ok

    def __repr__(self):
        return self.msg


def ll_close(a, b):
    return _Close(a, b, LL_ATOL, LL_RTOL)


def grad_close(a, b):
    return _Close(a, b, GRAD_ATOL, GRAD_RTOL)


def spec(sub, site, clock="none"):
    return bito_amd.PhyloModelSpecification(sub, site, clock)


def engines(sub, site, clock, patterns, weights, threads=8):
    gpu = bito_amd.Engine(spec(sub, site, clock), patterns, weights)
    cpu = oracle.OracleEngine(sub, site, clock, patterns, weights, threads)
    return gpu, cpu


def load(data_dir, fasta, trees):
    path = os.path.join(data_dir, trees)
    tc = treeio.read_nexus_file(path) if trees.endswith(".t") else treeio.read_newick_file(path)
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, fasta)), tc.taxon_names)
    return tc, sp


def test_hello_jc69_instance_api(data_dir):
    """Reads like the reference's own test (src/unrooted_sbn_instance.hpp:236-244)."""
    inst = bito_amd.unrooted_instance("charlie")
    inst.read_newick_file(os.path.join(data_dir, "hello.nwk"), False)
    inst.read_fasta_file(os.path.join(data_dir, "hello.fasta"))
    inst.prepare_for_phylo_likelihood(spec("JC69", "constant", "strict"), 2)
    for ll in inst.log_likelihoods():
        assert abs(ll - -84.852358) < 0.000001
    grads = inst.phylo_gradients()
    assert abs(grads[0].log_likelihood - -84.852358) < 0.000001
    assert grads[0].gradient["branch_lengths"].shape == (5,)


@pytest.mark.parametrize("rescaling", [False, True])
def test_ds1_jc69_goldens(data_dir, rescaling):
    g = GOLD["ds1_jc69"]
    tc, sp = load(data_dir, g["fasta"], g["trees"])
    gpu, cpu = engines("JC69", "constant", "strict", sp.patterns, sp.weights)
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    ll = gpu.log_likelihoods(pid, bl, rescaling=rescaling)
    assert np.abs(ll - g["log_likelihoods"]).max() < 5e-10  # 17-digit pybeagle values
    assert ll_close(ll, cpu.log_likelihoods(pid, bl, rescaling=rescaling))
    out = gpu.gradients(pid, bl, rescaling=rescaling)
    ref = cpu.gradients(pid, bl, rescaling=rescaling)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    last = np.sort(out["branch_lengths"][-1])
    assert np.abs(last - g["last_tree_sorted_branch_gradient"]).max() < g["gradient_tol"]
    assert out["branch_lengths"][-1][-1] == 0.0 and out["branch_lengths"][-1][-2] == 0.0


def test_ds1_jc69_weibull_goldens(data_dir):
    g = GOLD["ds1_jc69_weibull4_shape0.1"]
    tc, sp = load(data_dir, g["fasta"], g["trees"])
    gpu, cpu = engines("JC69", "weibull+4", "strict", sp.patterns, sp.weights)
    params = gpu.default_params(len(tc.trees))
    params[:, gpu.block_map()["Weibull_shape"][0]] = g["shape"]
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    for rescaling in (False, True):
        ll = gpu.log_likelihoods(pid, bl, params, rescaling=rescaling)
        assert np.abs(ll - g["log_likelihoods"]).max() < 5e-10
        out = gpu.gradients(pid, bl, params, rescaling=rescaling)
        assert np.abs(out["branch_lengths"][:, 0] - g["branch_gradient_entry0"]).max() < 2e-6
        ref = cpu.gradients(pid, bl, params, rescaling=rescaling)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"])
        assert grad_close(out["branch_lengths"], ref["branch_lengths"])


def test_config2_ds1_jc69_100_topologies():
    w = workloads.ds1_jc69(1)
    gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights)
    ll = gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
    assert ll_close(ll, cpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params))
    # JC69 == GTR with equal rates and frequencies (reference test/test_bito.py:97-122)
    gtr = bito_amd.Engine(spec("GTR", "constant"), w.patterns, w.weights)
    ll_gtr = gtr.log_likelihoods(w.parent_ids, w.branch_lengths)
    assert np.abs(ll - ll_gtr).max() < 1e-9


def test_config3_ds1_gtr_weibull4_vs_oracle():
    w = workloads.ds1_gtr_weibull4(1)
    gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights)
    out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    assert ll_close(gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params), ref["log_likelihood"])
    # rescaled == unrescaled (reference src/unrooted_sbn_instance.hpp:289-311)
    res = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=True)
    assert np.abs(res["log_likelihood"] - out["log_likelihood"]).max() < 1e-9
    assert grad_close(res["branch_lengths"], out["branch_lengths"])


@pytest.mark.parametrize("kernel", [_capi.KERNEL_HBM_ARENA, _capi.KERNEL_LDS, _capi.KERNEL_LDS_TREE, _capi.KERNEL_LDS_PIPE])
@pytest.mark.parametrize("model", [("JC69", "constant"), ("HKY", "weibull+2"), ("GTR", "weibull+4")])
def test_every_traversal_kernel_matches_oracle(kernel, model):
    """The four traversal kernels (HBM arena, LDS one wave/SIMD, LDS tree-resident images, LDS with
    hand-scheduled loops) are interchangeable: same inputs, same answers, for 1, 2 and 4 rate categories."""
    sub, site = model
    w = workloads.ds1_gtr_weibull4(1).subset(12)
    gpu, cpu = engines(sub, site, "none", w.patterns, w.weights, 4)
    gpu.set_kernel(kernel)
    params = gpu.default_params(12)
    bm = gpu.block_map()
    if "substitution_model_frequencies" in bm:
        params[:, bm["substitution_model_frequencies"][0]:][:, :4] = [0.15, 0.35, 0.3, 0.2]
    if site != "constant":
        params[:, bm["Weibull_shape"][0]] = np.linspace(0.4, 1.6, 12)
    out = gpu.gradients(w.parent_ids, w.branch_lengths, params)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, params)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    assert ll_close(gpu.log_likelihoods(w.parent_ids, w.branch_lengths, params), ref["log_likelihood"])
    # (the HBM-arena walk for up to 4 rate categories is walk_hbm_cat_kernel: one wave per category)
    expect = {_capi.KERNEL_HBM_ARENA: "walk_hbm_cat_kernel", _capi.KERNEL_LDS: "walk_lds_kernel",
              _capi.KERNEL_LDS_TREE: "walk_tree_kernel", _capi.KERNEL_LDS_PIPE: "walk_pipe_kernel"}[kernel]
    assert gpu.kernel_name() == expect


def test_flua_rooted_with_rates(data_dir):
    g = GOLD["flua_jc69_strict"]
    tc, sp = load(data_dir, "fluA.fa", "fluA.tree")
    rates = np.full((1, tc.trees[0].node_count - 1), g["clock_rate"])
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    gpu, cpu = engines("JC69", "constant", "strict", sp.patterns, sp.weights, 1)
    ll = gpu.log_likelihoods(pid, bl, rates=rates)
    assert abs(ll[0] - g["log_likelihood"]) < 2e-6
    out = gpu.gradients(pid, bl, rates=rates, flags=_capi.GRAD_CLOCK_MODEL)
    ref = cpu.gradients(pid, bl, rates=rates, flags=oracle.GRAD_CLOCK_MODEL)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert np.abs(out["branch_lengths"] - ref["branch_lengths"]).max() < 1e-6 * max(1.0, np.abs(ref["branch_lengths"]).max())
    assert abs(out["clock_model"][0] - ref["clock_model"][0]) < 1e-6 * abs(ref["clock_model"][0])
    # GTR / HKY log-likelihood goldens of the reference (pinned to 1e-3 / 1e-4 there)
    for key, sub, row in (("flua_gtr", "GTR", GOLD["flua_gtr"]["frequencies"] + GOLD["flua_gtr"]["rates"]),
                          ("flua_hky", "HKY", GOLD["flua_hky"]["frequencies"] + [GOLD["flua_hky"]["kappa"]])):
        gpu2, cpu2 = engines(sub, "constant", "strict", sp.patterns, sp.weights, 1)
        params = gpu2.default_params(1)
        params[0, :len(row)] = row
        ll2 = gpu2.log_likelihoods(pid, bl, params, rates=rates)
        assert abs(ll2[0] - GOLD[key]["log_likelihood"]) < GOLD[key]["tol"]
        assert ll_close(ll2, cpu2.log_likelihoods(pid, bl, params, rates=rates))


@pytest.mark.parametrize("categories", [1, 2, 3, 5, 8])
def test_category_counts(categories):
    w = workloads.ds1_gtr_weibull4(1).subset(6)
    site = f"weibull+{categories}"
    gpu, cpu = engines("HKY", site, "none", w.patterns, w.weights, 4)
    params = gpu.default_params(6)
    bm = gpu.block_map()
    params[:, bm["substitution_model_frequencies"][0]:][:, :4] = [0.3, 0.2, 0.1, 0.4]
    params[:, bm["substitution_model_rates"][0]] = np.linspace(0.5, 4.0, 6)
    params[:, bm["Weibull_shape"][0]] = np.linspace(0.3, 2.0, 6)
    for rescaling in (False, True):
        out = gpu.gradients(w.parent_ids, w.branch_lengths, params, rescaling=rescaling)
        ref = cpu.gradients(w.parent_ids, w.branch_lengths, params, rescaling=rescaling)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"])
        assert grad_close(out["branch_lengths"], ref["branch_lengths"])


def test_edge_cases(data_dir):
    tc, sp = load(data_dir, "hello.fasta", "hello.nwk")
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    # one pattern; an all-gap column; heavy weights
    for patterns, weights in (
        (sp.patterns[:, :1], sp.weights[:1]),
        (np.full((3, 1), 4, dtype=np.int32), np.array([7.0])),
        (np.concatenate([sp.patterns, np.full((3, 2), 4, dtype=np.int32)], axis=1),
         np.concatenate([sp.weights, [3.0, 1e6]])),
    ):
        gpu, cpu = engines("JC69", "weibull+4", "none", patterns, weights, 1)
        out = gpu.gradients(pid, bl)
        ref = cpu.gradients(pid, bl)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"])
        assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    # zero and tiny branch lengths
    gpu, cpu = engines("GTR", "constant", "none", sp.patterns, sp.weights, 1)
    bl0 = bl.copy()
    bl0[0, 0] = 0.0
    bl0[0, 1] = 1e-12
    out = gpu.gradients(pid, bl0)
    ref = cpu.gradients(pid, bl0)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    # an empty collection gives empty results, not an error (FatBeagleParallelize over no trees)
    assert gpu.log_likelihoods(pid[:0], bl[:0]).shape == (0,)
    empty = gpu.gradients(pid[:0], bl[:0])
    assert empty["log_likelihood"].shape == (0,) and empty["branch_lengths"].shape[0] == 0
    assert ll_close(gpu.log_likelihoods(pid, bl0), ref["log_likelihood"])  # and the engine carries on
    # two-taxon rooted tree: the smallest bifurcating tree
    pats = sp.patterns[:2]
    gpu, cpu = engines("JC69", "constant", "none", pats, sp.weights, 1)
    pid2 = np.array([[2, 2]], dtype=np.int32)
    bl2 = np.array([[0.1, 0.25, 0.0]])
    out = gpu.gradients(pid2, bl2)
    ref = cpu.gradients(pid2, bl2)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])


def test_pattern_counts_around_tile_edges():
    w = workloads.ds1_gtr_weibull4(1).subset(3)
    for P in (63, 64, 65, 255, 256, 257, 934):
        gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns[:, :P], w.weights[:P], 3)
        out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
        ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"]), P
        assert grad_close(out["branch_lengths"], ref["branch_lengths"]), P


def test_model_parameter_gradients(data_dir):
    """site_model / substitution_model / clock_model entries of PhyloGradient: the reference's
    fluA goldens (src/rooted_sbn_instance.hpp:347-430) and the oracle on DS1."""
    tc, sp = load(data_dir, "fluA.fa", "fluA.tree")
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    rates = np.full((1, tc.trees[0].node_count - 1), 0.001)
    allflags = (_capi.GRAD_SUBSTITUTION_MODEL | _capi.GRAD_SITE_MODEL | _capi.GRAD_CLOCK_MODEL |
                _capi.GRAD_STICKBREAKING)
    # Weibull shape gradient
    g = GOLD["flua_jc69_weibull4_shape0.1"]
    gpu, cpu = engines("JC69", "weibull+4", "strict", sp.patterns, sp.weights, 1)
    params = gpu.default_params(1)
    params[:, gpu.block_map()["Weibull_shape"][0]] = g["shape"]
    out = gpu.gradients(pid, bl, params, rates=rates, flags=allflags)
    assert abs(out["log_likelihood"][0] - g["log_likelihood"]) < 1e-8
    assert abs(out["site_model"][0] - g["site_model_gradient"]) < 1e-6
    assert "substitution_model" not in out  # JC69 has no free parameters (fat_beagle.cpp:525)
    # GTR and HKY finite-difference gradients in stick-breaking coordinates
    for key, sub, row in (("flua_gtr", "GTR", GOLD["flua_gtr"]["frequencies"] + GOLD["flua_gtr"]["rates"]),
                          ("flua_hky", "HKY", GOLD["flua_hky"]["frequencies"] + [GOLD["flua_hky"]["kappa"]])):
        gpu2, cpu2 = engines(sub, "constant", "strict", sp.patterns, sp.weights, 1)
        p2 = gpu2.default_params(1)
        p2[0, :len(row)] = row
        out2 = gpu2.gradients(pid, bl, p2, rates=rates, flags=allflags)
        ref2 = cpu2.gradients(pid, bl, p2, rates=rates, flags=oracle.GRAD_SUBSTITUTION_MODEL | oracle.GRAD_STICKBREAKING |
                              oracle.GRAD_CLOCK_MODEL)
        assert np.abs(out2["substitution_model"][0] - GOLD[key]["substitution_model_gradient"]).max() < GOLD[key]["gradient_tol"]
        assert np.abs(out2["substitution_model"] - ref2["substitution_model"]).max() < 2e-3  # FD of two FP64 codes, delta 1e-6
        assert abs(out2["clock_model"][0] - ref2["clock_model"][0]) < 1e-6 * abs(ref2["clock_model"][0])
        assert out2["substitution_model_rates"].shape[1] + out2["substitution_model_frequencies"].shape[1] == \
            out2["substitution_model"].shape[1]
        # the main batch stays resident and consistent after the composed call
        ll_again, grad_again = gpu2.download()
        assert np.array_equal(ll_again, out2["log_likelihood"]) and np.array_equal(grad_again, out2["branch_lengths"])
    # unrooted, headline model, identity transform, every kernel path that supports it
    w = workloads.ds1_gtr_weibull4(1).subset(5)
    gpu3, cpu3 = engines(w.substitution, w.site, w.clock, w.patterns, w.weights, 5)
    out3 = gpu3.gradients(w.parent_ids, w.branch_lengths, w.params,
                          flags=_capi.GRAD_SITE_MODEL | _capi.GRAD_SUBSTITUTION_MODEL)
    ref3 = cpu3.gradients(w.parent_ids, w.branch_lengths, w.params,
                          flags=oracle.GRAD_SITE_MODEL | oracle.GRAD_SUBSTITUTION_MODEL)
    assert np.abs(out3["site_model"] - ref3["site_model"]).max() < 1e-6 * max(1.0, np.abs(ref3["site_model"]).max())
    assert out3["substitution_model"].shape == (5, 10)
    assert np.abs(out3["substitution_model"] - ref3["substitution_model"]).max() < 1e-3 * max(1.0, np.abs(ref3["substitution_model"]).max())
    assert grad_close(out3["branch_lengths"], ref3["branch_lengths"])
    # the instance API returns every key by default, like the reference's default flags
    inst = bito_amd.unrooted_instance("x")
    inst.read_nexus_file(os.path.join(data_dir, "DS1.subsampled_10.t"))
    inst.read_fasta_file(os.path.join(data_dir, "DS1.fasta"))
    inst.prepare_for_phylo_likelihood(spec("GTR", "weibull+4", "strict"), 1)
    grads = inst.phylo_gradients()
    assert set(grads[0].gradient) == {"branch_lengths", "site_model", "substitution_model", "substitution_model_rates",
                                      "substitution_model_frequencies"}
    assert grads[0].gradient["substitution_model"].shape == (8,)


def test_resident_batch_interface():
    w = workloads.ds1_gtr_weibull4(1).subset(20)
    gpu = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns, w.weights)
    direct = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    gpu.upload(w.parent_ids, w.branch_lengths, w.params)
    gpu.run(True)
    gpu.sync()
    ll, grad = gpu.download()
    assert np.array_equal(ll, direct["log_likelihood"]) and np.array_equal(grad, direct["branch_lengths"])
    # new branch lengths and parameters without re-uploading the topologies
    bl2 = w.branch_lengths * 1.5
    p2 = w.params.copy()
    p2[:, 10] = 0.8
    gpu.update(bl2, p2)
    gpu.run(True)
    ll2, grad2 = gpu.download()
    fresh = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns, w.weights)
    again = fresh.gradients(w.parent_ids, bl2, p2)
    assert np.array_equal(ll2, again["log_likelihood"]) and np.array_equal(grad2, again["branch_lengths"])
    total, kern, launches = gpu.time_runs(True, False, 3)
    assert total > 0 and kern > 0 and launches >= 3 and kern <= total * 1.05


@pytest.mark.parametrize("kernel", [_capi.KERNEL_AUTO, _capi.KERNEL_HBM_ARENA, _capi.KERNEL_LDS])
def test_passes_in_flight_with_updates_between_them(kernel):
    """Passes enqueued back to back with new inputs in between: the set-up of a pass runs on its own stream into
    one of three buffer sets, beside the traversals of earlier passes, and `update` rewrites the inputs it reads
    -- every pass must see the inputs of its own moment.  Eight passes without a wait (results copied out by
    copies enqueued behind each pass), log-likelihood-only passes mixed in, each against a fresh engine."""
    w = workloads.ds1_gtr_weibull4(3).subset(300)
    gpu = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns, w.weights)
    gpu.set_kernel(kernel)
    gpu.upload(w.parent_ids, w.branch_lengths, w.params)
    outs, inputs = [], []
    for k in range(8):
        bl = w.branch_lengths * (1.0 + 0.07 * k)
        par = w.params.copy()
        par[:, 10] = 0.4 + 0.1 * k
        gpu.update(bl, par)
        grad = k % 3 != 2
        gpu.run(grad, False)
        ll_out, grad_out = np.zeros(300), np.zeros((300, 53))
        gpu.download_async(ll_out.ctypes.data, grad_out.ctypes.data if grad else None)
        outs.append((ll_out, grad_out, grad))
        inputs.append((bl, par))
    gpu.sync()
    fresh = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns, w.weights)
    fresh.set_kernel(kernel)
    for (ll_out, grad_out, grad), (bl, par) in zip(outs, inputs):
        if grad:
            ref = fresh.gradients(w.parent_ids, bl, par)
            assert np.array_equal(ll_out, ref["log_likelihood"])
            assert np.array_equal(grad_out, ref["branch_lengths"])
        else:
            assert np.array_equal(ll_out, fresh.log_likelihoods(w.parent_ids, bl, par))


def test_full_size_batch_properties():
    """BASELINE config 3 at bench size: results are bit-reproducible from run to run, a tree's results depend
    on the rest of the batch only through the order of the pattern-tile sums (the launcher splits a batch into
    whole-tree and run-of-tiles units by its size: rounding level, held to a tenth of the parity tolerances
    here), and agree with the oracle on a sample."""
    w = workloads.ds1_gtr_weibull4(8)  # 800 trees
    gpu = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns, w.weights)
    full = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    again = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert np.array_equal(full["log_likelihood"], again["log_likelihood"])
    assert np.array_equal(full["branch_lengths"], again["branch_lengths"])
    sl = slice(300, 400)
    part = gpu.gradients(w.parent_ids[sl], w.branch_lengths[sl], w.params[sl])
    assert np.abs(part["log_likelihood"] - full["log_likelihood"][sl]).max() <= 0.1 * LL_ATOL
    assert np.abs(part["branch_lengths"] - full["branch_lengths"][sl]).max() <= 0.1 * GRAD_ATOL
    part_again = gpu.gradients(w.parent_ids[sl], w.branch_lengths[sl], w.params[sl])
    assert np.array_equal(part["log_likelihood"], part_again["log_likelihood"])
    assert np.array_equal(part["branch_lengths"], part_again["branch_lengths"])
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    idx = np.arange(0, 800, 37)
    ref = cpu.gradients(w.parent_ids[idx], w.branch_lengths[idx], w.params[idx])
    assert ll_close(full["log_likelihood"][idx], ref["log_likelihood"])
    assert grad_close(full["branch_lengths"][idx], ref["branch_lengths"])
    # permuting the site patterns changes nothing beyond summation order
    perm = np.random.default_rng(0).permutation(w.patterns.shape[1])
    gpu_p = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns[:, perm], w.weights[perm])
    permuted = gpu_p.gradients(w.parent_ids[:50], w.branch_lengths[:50], w.params[:50])
    assert np.abs(permuted["log_likelihood"] - full["log_likelihood"][:50]).max() < 1e-9
    assert grad_close(permuted["branch_lengths"], full["branch_lengths"][:50])


def test_large_tree_with_rescaling_vs_oracle():
    """Config-4-shaped input (many taxa, rescaling on) at a size the oracle finishes in seconds."""
    w = workloads.synthetic_gtr_weibull4(n=300, P=1500, tree_count=3)
    gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights, 3)
    out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=True)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=True)
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    assert np.all(np.isfinite(out["branch_lengths"]))


def test_config4_full_size_sixteen_trees():
    """BASELINE config 4 shape (1000 taxa x 10000 patterns, rescaling on): sixteen of its 1000 trees against the oracle,
    two from the block of every rank of the 8-GPU sharding (trees 125 r and 125 r + 124: the first and the last tree a
    rank walks)."""
    picks = [125 * r + k for r in range(8) for k in (0, 124)]
    parts = [workloads.synthetic_gtr_weibull4(n=1000, P=10000, tree_count=1, first_tree=t) for t in picks]
    w = parts[0]
    for other in parts[1:]:
        assert np.array_equal(w.patterns, other.patterns)
    w.parent_ids = np.concatenate([x.parent_ids for x in parts])
    w.branch_lengths = np.concatenate([x.branch_lengths for x in parts])
    w.params = np.concatenate([x.params for x in parts])
    gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=True)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=True)
    assert gpu.kernel_name() == "walk_hbm_cat_kernel"
    assert np.all(np.isfinite(out["log_likelihood"]))
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert np.abs(out["branch_lengths"] - ref["branch_lengths"]).max() <= GRAD_ATOL + 1e-9 * np.abs(ref["branch_lengths"]).max()


def test_error_behaviour(data_dir):
    tc, sp = load(data_dir, "hello.fasta", "hello.nwk")
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    with pytest.raises(bito_amd.BitoAmdError, match="Substitution model not known"):
        bito_amd.Engine(spec("F81", "constant"), sp.patterns, sp.weights)
    with pytest.raises(bito_amd.BitoAmdError, match="Site model not known"):
        bito_amd.Engine(spec("JC69", "gamma"), sp.patterns, sp.weights)
    with pytest.raises(bito_amd.BitoAmdError, match="Clock model not known"):
        bito_amd.Engine(spec("JC69", "constant", "relaxed"), sp.patterns, sp.weights)
    eng = bito_amd.Engine(spec("GTR", "constant"), sp.patterns, sp.weights)
    bad = eng.default_params(1)
    bad[0, 0] = 0.5
    with pytest.raises(bito_amd.BitoAmdError, match="frequencies do not sum to 1"):
        eng.log_likelihoods(pid, bl, bad)
    bad = eng.default_params(1)
    bad[0, 5] = 0.9
    with pytest.raises(bito_amd.BitoAmdError, match="rates do not sum to 1"):
        eng.log_likelihoods(pid, bl, bad)
    with pytest.raises(bito_amd.BitoAmdError, match="param matrix"):
        eng.log_likelihoods(pid, bl, eng.default_params(2))
    with pytest.raises(bito_amd.BitoAmdError, match="not a valid internal id"):
        eng.log_likelihoods(np.array([[3, 3, 1]], dtype=np.int32), bl)
    with pytest.raises(bito_amd.BitoAmdError, match="does not match"):
        eng.log_likelihoods(np.array([[4, 4, 5, 5, 6, 6]], dtype=np.int32), np.ones((1, 7)))
    fresh = bito_amd.Engine(spec("JC69", "constant"), sp.patterns, sp.weights)
    with pytest.raises(bito_amd.BitoAmdError, match="no batch is resident"):
        fresh.run(False)
    # the engine is still usable after an error
    assert np.isfinite(eng.log_likelihoods(pid, bl)[0])


@pytest.mark.parametrize("use_tip_states", [True, False])
@pytest.mark.parametrize("rescaling", [False, True])
def test_beagle_shim_with_fatbeagle_call_sequence(data_dir, use_tip_states, rescaling):
    """SURVEY 8b seam 1: the 17 BEAGLE symbols FatBeagle calls, driven in FatBeagle's own
    order (tests/beagle_driver.py), reproduce the reference's DS1 goldens and the oracle."""
    from beagle_driver import FatBeagleDriver

    g = GOLD["ds1_jc69"]
    tc, sp = load(data_dir, g["fasta"], g["trees"])
    Q, V, Vi, lam, pi = oracle.substitution_model("JC69")
    drv = FatBeagleDriver(sp.patterns, sp.weights, V, Vi, lam, pi, Q, [1.0], [1.0], use_tip_states)
    assert "bito_amd" in drv.impl
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    for t in (0, 9):
        ll = drv.log_likelihood(pid[t], bl[t], rescaling)
        assert abs(ll - g["log_likelihoods"][t]) < 5e-10
    ll, grad = drv.gradient(pid[9], bl[9], rescaling)
    assert abs(ll - g["log_likelihoods"][9]) < 5e-10
    assert np.abs(np.sort(grad) - g["last_tree_sorted_branch_gradient"]).max() < g["gradient_tol"]
    drv.close()
    # GTR + weibull+4 against the oracle
    w = workloads.ds1_gtr_weibull4(1).subset(3)
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 1)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
    Q, V, Vi, lam, pi = oracle.substitution_model("GTR", w.params[0, :10])
    rates, props, _ = oracle.weibull_rates(4, w.params[0, 10])
    drv = FatBeagleDriver(w.patterns, w.weights, V, Vi, lam, pi, Q, rates, props, use_tip_states)
    for t in range(3):
        ll, grad = drv.gradient(w.parent_ids[t], w.branch_lengths[t], rescaling)
        assert ll_close(ll, ref["log_likelihood"][t])
        assert grad_close(grad, ref["branch_lengths"][t])
    drv.close()


def _random_rooted_parent_ids(n, rng):
    """Random rooted bifurcating topology with bito ids (leaves 0..n-1, internal ids in post-order)."""
    import itertools

    counter = itertools.count()
    nodes = [("leaf", i) for i in range(n)]
    rng.shuffle(nodes)
    while len(nodes) > 1:
        i, j = sorted(rng.choice(len(nodes), 2, replace=False))
        b, a = nodes.pop(j), nodes.pop(i)
        nodes.append(("node", a, b, next(counter)))
    parents, next_id = {}, [n]

    def walk(t):
        if t[0] == "leaf":
            return t[1]
        kids = [walk(t[1]), walk(t[2])]
        me = next_id[0]
        next_id[0] += 1
        for k in kids:
            parents[k] = me
        return me

    root = walk(nodes[0])
    assert root == 2 * n - 2
    return np.array([parents[v] for v in range(2 * n - 2)], dtype=np.int32)


@pytest.mark.parametrize("kernel", [_capi.KERNEL_LDS, _capi.KERNEL_HBM_ARENA, _capi.KERNEL_LDS_PIPE])
@pytest.mark.parametrize("n", [3, 4, 5, 7, 12, 29, 33, 38, 41, 48, 49, 52, 56, 57, 64, 65])
def test_random_shapes_rooted_and_unrooted(kernel, n):
    """Random topologies of many shapes -- caterpillars to balanced trees, cherries as first or second
    child, roots over a tip -- with gaps in the alignment, rooted and unrooted, 1, 2 and 4 categories:
    the kernels that rebuild cherries and forward vectors in registers against the oracle."""
    rng = np.random.default_rng(1000 + n)
    P, T = 37, 9
    patterns = rng.integers(0, 5, (n, P)).astype(np.int32)  # 4 = gap
    weights = rng.integers(1, 5, P).astype(np.float64)
    for site, C in (("constant", 1), ("weibull+2", 2), ("weibull+4", 4)):
        for rooted in (False, True):
            if rooted:
                pid = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)])
                M = 2 * n - 1
            else:
                pid = np.stack([workloads.random_unrooted_tree(n, rng, 0.1).parent_ids for _ in range(T)]).astype(np.int32)
                M = 2 * n - 2
            bl = rng.exponential(0.1, (T, M))
            bl[:, -1] = 0.0
            gpu, cpu = engines("GTR", site, "none", patterns, weights, 4)
            gpu.set_kernel(kernel)
            params = gpu.default_params(T)
            params[:, :4] = rng.dirichlet([5, 5, 5, 5], T)
            params[:, 4:10] = rng.dirichlet([3] * 6, T)
            if C > 1:
                params[:, 10] = rng.uniform(0.3, 2.0, T)
            if kernel == _capi.KERNEL_LDS_PIPE and n > 64:  # 4n - 4 image registers, a mask register per tip: up to 64 taxa
                with pytest.raises(bito_amd.BitoAmdError, match="pipelined LDS kernel was forced"):
                    gpu.gradients(pid, bl, params)
                continue
            try:
                out = gpu.gradients(pid, bl, params)
            except bito_amd.BitoAmdError:
                # (beyond some 45 taxa a random tree's stored vectors may not fit a wave's share of LDS)
                assert kernel == _capi.KERNEL_LDS_PIPE and n > 41
                continue
            ref = cpu.gradients(pid, bl, params)
            assert ll_close(out["log_likelihood"], ref["log_likelihood"]), (site, rooted)
            assert grad_close(out["branch_lengths"], ref["branch_lengths"]), (site, rooted)
            assert ll_close(gpu.log_likelihoods(pid, bl, params), ref["log_likelihood"]), (site, rooted)


@pytest.mark.parametrize("site", ["constant", "weibull+2", "weibull+3", "weibull+4"])
@pytest.mark.parametrize("rescaling", [False, True])
def test_hbm_arena_walk_one_wave_per_category(site, rescaling):
    """walk_hbm_cat_kernel (one wave per rate category, per-category power-of-two rescaling, one pending vector per
    thread in LDS) on trees with many nodes over two internal children, pattern counts that leave a tile one
    pattern full, gaps, rooted and unrooted; log-likelihood-only passes and the site-model pass included."""
    rng = np.random.default_rng(4242)
    n, T = 60, 6
    for P, rooted in ((65, False), (130, True)):
        patterns = rng.integers(0, 5, (n, P)).astype(np.int32)
        weights = rng.integers(1, 5, P).astype(np.float64)
        if rooted:
            pid = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)])
            M = 2 * n - 1
        else:
            pid = np.stack([workloads.random_unrooted_tree(n, rng, 0.1).parent_ids for _ in range(T)]).astype(np.int32)
            M = 2 * n - 2
        bl = rng.exponential(0.1, (T, M))
        bl[:, -1] = 0.0
        gpu, cpu = engines("HKY", site, "none", patterns, weights, 4)
        gpu.set_kernel(_capi.KERNEL_HBM_ARENA)
        params = gpu.default_params(T)
        params[:, :4] = rng.dirichlet([5, 5, 5, 5], T)
        if site != "constant":
            params[:, -1] = rng.uniform(0.3, 2.0, T)
        flags = _capi.GRAD_SITE_MODEL if site != "constant" else 0
        out = gpu.gradients(pid, bl, params, rescaling=rescaling, flags=flags)
        assert gpu.kernel_name() == "walk_hbm_cat_kernel"
        ref = cpu.gradients(pid, bl, params, rescaling=rescaling, flags=oracle.GRAD_SITE_MODEL if flags else 0)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"]), (P, rooted)
        assert grad_close(out["branch_lengths"], ref["branch_lengths"]), (P, rooted)
        if flags:
            assert grad_close(out["site_model"], ref["site_model"]), (P, rooted)
        assert ll_close(gpu.log_likelihoods(pid, bl, params, rescaling=rescaling), ref["log_likelihood"]), (P, rooted)


def test_hbm_arena_walk_in_chunks():
    """A PLV arena too small for the batch: the HBM-arena walk runs it in several launches (chunks of trees), with
    gradients, rescaling and the site-model gradient, against the oracle."""
    rng = np.random.default_rng(77)
    n, P, T = 50, 150, 11
    patterns = rng.integers(0, 5, (n, P)).astype(np.int32)
    weights = rng.integers(1, 4, P).astype(np.float64)
    pid = np.stack([workloads.random_unrooted_tree(n, rng, 0.1).parent_ids for _ in range(T)]).astype(np.int32)
    bl = rng.exponential(0.1, (T, 2 * n - 2))
    bl[:, -1] = 0.0
    per_tree = (n - 1) * 4 * 4 * 256 * 8  # (n - 1) vectors of 4 categories x 4 states x the padded patterns, doubles
    gpu = bito_amd.Engine(spec("HKY", "weibull+4"), patterns, weights, arena_bytes=3 * per_tree)  # three trees at a time
    gpu.set_kernel(_capi.KERNEL_HBM_ARENA)  # (50 taxa: AUTO takes walk_pipe_kernel's wide layout when rescaling is off)
    cpu = oracle.OracleEngine("HKY", "weibull+4", "none", patterns, weights, 4)
    params = gpu.default_params(T)
    params[:, :4] = rng.dirichlet([5, 5, 5, 5], T)
    params[:, -1] = rng.uniform(0.3, 2.0, T)
    for rescaling in (False, True):
        out = gpu.gradients(pid, bl, params, rescaling=rescaling, flags=_capi.GRAD_SITE_MODEL)
        assert gpu.kernel_name() == "walk_hbm_cat_kernel"
        ref = cpu.gradients(pid, bl, params, rescaling=rescaling, flags=oracle.GRAD_SITE_MODEL)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"])
        assert grad_close(out["branch_lengths"], ref["branch_lengths"])
        assert grad_close(out["site_model"], ref["site_model"])
        assert ll_close(gpu.log_likelihoods(pid, bl, params, rescaling=rescaling), ref["log_likelihood"])


def test_one_rate_category_underflows_without_rescaling():
    """90 taxa, short branches, no rescaling: the slowest rate category's site likelihoods underflow for some
    patterns while the patterns' likelihoods (the sum over categories) stay normal numbers.  The derivatives must
    stay finite and right (a per-category ratio num_c / den_c would be 0/0 there: found by scripts/gpu_fuzz.py)."""
    rng = np.random.default_rng(90)
    n, P, T = 90, 65, 8
    patterns = rng.integers(0, 4, (n, P)).astype(np.int32)
    weights = rng.integers(1, 9, P).astype(np.float64)
    pid = np.stack([workloads.random_unrooted_tree(n, rng, 0.1).parent_ids for _ in range(T)]).astype(np.int32)
    bl = rng.exponential(0.01, (T, 2 * n - 2))
    bl[:, -1] = 0.0
    gpu, cpu = engines("HKY", "weibull+4", "none", patterns, weights, 4)
    params = gpu.default_params(T)
    params[:, :4] = rng.dirichlet([5, 5, 5, 5], T)
    params[:, -1] = 0.3  # a heavy-tailed rate distribution: the slowest category is very slow
    out = gpu.gradients(pid, bl, params)
    assert gpu.kernel_name() == "walk_hbm_cat_kernel"
    ref = cpu.gradients(pid, bl, params)
    assert np.all(np.isfinite(ref["branch_lengths"])) and np.all(np.isfinite(out["branch_lengths"]))
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert np.allclose(out["branch_lengths"], ref["branch_lengths"], rtol=1e-9, atol=GRAD_ATOL)


def test_site_model_gradient_fused_equals_second_pass():
    """The LDS traversal yields the site-model gradient in the same pass (per-category edge sums weighted by
    (d r_c / d shape) / r_c); the other kernels run FatBeagle's second traversal with the rate derivatives
    (src/fat_beagle.cpp:538-550).  Same numbers, and the oracle's."""
    w = workloads.ds1_gtr_weibull4(1).subset(9)
    gpu, cpu = engines(w.substitution, w.site, "none", w.patterns, w.weights, 4)
    params = w.params.copy()
    params[:, 10] = np.linspace(0.3, 1.9, 9)
    flags = _capi.GRAD_SITE_MODEL
    fused = gpu.gradients(w.parent_ids, w.branch_lengths, params, flags=flags)
    assert gpu.kernel_name() == "walk_pipe_kernel"
    gpu.set_kernel(_capi.KERNEL_LDS)
    fused_lds = gpu.gradients(w.parent_ids, w.branch_lengths, params, flags=flags)
    assert gpu.kernel_name() == "walk_lds_kernel"
    gpu.set_kernel(_capi.KERNEL_HBM_ARENA)
    twice = gpu.gradients(w.parent_ids, w.branch_lengths, params, flags=flags)
    assert gpu.kernel_name() == "walk_hbm_cat_kernel"
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, params, flags=oracle.GRAD_SITE_MODEL)
    assert grad_close(fused["site_model"], ref["site_model"])
    assert grad_close(fused_lds["site_model"], ref["site_model"])
    assert grad_close(twice["site_model"], ref["site_model"])
    assert grad_close(fused["branch_lengths"], ref["branch_lengths"])


@pytest.mark.parametrize("kernel", [_capi.KERNEL_LDS, _capi.KERNEL_LDS_PIPE])
@pytest.mark.parametrize("run", [1, 3, 5, 15])
def test_tile_runs_of_the_lds_walk(run, kernel, monkeypatch):
    """A workgroup of the LDS walk takes a run of consecutive pattern tiles of its tree (the launcher picks the
    run length from the batch size; here it is forced): same results whatever the run length, against the
    oracle, with and without the fused site-model gradient."""
    monkeypatch.setenv("BITO_AMD_LDS_TILE_RUN", str(run))
    w = workloads.ds1_gtr_weibull4(1).subset(12)
    gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    gpu.set_kernel(kernel)
    out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
    assert gpu.kernel_name() == ("walk_lds_kernel" if kernel == _capi.KERNEL_LDS else "walk_pipe_kernel")
    assert ll_close(out["log_likelihood"], ref["log_likelihood"])
    assert grad_close(out["branch_lengths"], ref["branch_lengths"])
    assert grad_close(out["site_model"], ref["site_model"])
    assert ll_close(gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params), ref["log_likelihood"])


@pytest.mark.parametrize("whole", [0, 5, 12])
@pytest.mark.parametrize("run", [1, 5])
def test_whole_tree_units_of_the_pipe_walk(run, whole, monkeypatch):
    """walk_pipe_kernel gives the first trees of a batch a workgroup each (all tiles, one partial row) and walks
    the rest in runs of tiles (the launcher picks the split from the batch size; here it is forced): same
    results whatever the split, the unused partial rows of a whole-tree unit being zero."""
    monkeypatch.setenv("BITO_AMD_LDS_TILE_RUN", str(run))
    monkeypatch.setenv("BITO_AMD_PIPE_WHOLE_TREES", str(whole))
    w = workloads.ds1_gtr_weibull4(1).subset(12)
    gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    gpu.set_kernel(_capi.KERNEL_LDS_PIPE)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
    for _ in range(2):  # (the second pass finds the partial rows of the first)
        out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
        assert gpu.kernel_name() == "walk_pipe_kernel"
        assert ll_close(out["log_likelihood"], ref["log_likelihood"])
        assert grad_close(out["branch_lengths"], ref["branch_lengths"])
        assert grad_close(out["site_model"], ref["site_model"])
    assert ll_close(gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params), ref["log_likelihood"])


_RESULTS_RING_SCRIPT = r"""
import sys
import numpy as np
import torch
torch.cuda.init()  # (torch's HIP runtime first: the engine's library brings its own)
sys.path.insert(0, sys.argv[1])
import bito_amd
from bito_amd import workloads


class DeviceVector:
    def __init__(self, address, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (address, False), "version": 2}


w = workloads.ds1_gtr_weibull4(1).subset(12)
gpu = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
gpu.upload(w.parent_ids, w.branch_lengths, w.params)
stream = torch.cuda.Stream()
views, expected = [], []
for k in range(4):
    gpu.update(w.branch_lengths * (1.0 + 0.1 * k), None)
    gpu.run(True, False)
    ll_address, grad_address = gpu.results_async(stream.cuda_stream)
    with torch.cuda.stream(stream):
        views.append(torch.as_tensor(DeviceVector(ll_address, 12), device="cuda"))
        grad_now = torch.as_tensor(DeviceVector(grad_address, 12 * 53), device="cuda").clone()
    ll, grad = gpu.download(True)
    expected.append(ll.copy())
    stream.synchronize()
    assert np.array_equal(grad_now.cpu().numpy().reshape(12, 53), grad)
assert len({v.data_ptr() for v in views}) == 4
for k in range(4):  # passes 1..3 have not touched pass 0's buffer
    assert np.array_equal(views[k].cpu().numpy(), expected[k])
assert not np.array_equal(expected[0], expected[1])
print("ring ok")
"""


def test_results_in_place_ring():
    """bito_amd_engine_results_async hands out device addresses instead of copying: per-tree log-likelihoods of
    pass k stay where they are until three more passes have been enqueued (a ring of four buffers), and the
    consumer's stream is ordered behind the pass.  (In a process of its own: the consumer here is torch, whose
    bundled HIP runtime has to be the first one the process initialises.)"""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, "-c", _RESULTS_RING_SCRIPT, os.path.dirname(HERE)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "ring ok" in out.stdout, out.stderr[-2000:]


def _shaped_rooted_parent_ids(n, shape):
    """Rooted bifurcating topologies of extreme shape with bito ids: 'caterpillar' (one cherry, every other
    internal node has a tip child) or 'balanced' (as many cherries as a tree can have)."""
    parents, next_id = {}, [n]

    def join(a, b):
        me = next_id[0]
        next_id[0] += 1
        parents[a] = parents[b] = me
        return me

    def balanced(leaves):
        if len(leaves) == 1:
            return leaves[0]
        half = len(leaves) // 2
        return join(balanced(leaves[:half]), balanced(leaves[half:]))

    if shape == "caterpillar":
        top = join(0, 1)
        for leaf in range(2, n):
            top = join(top, leaf)
    else:
        top = balanced(list(range(n)))
    assert top == 2 * n - 2
    return np.array([parents[v] for v in range(2 * n - 2)], dtype=np.int32)


@pytest.mark.parametrize("site", ["constant", "weibull+2", "weibull+4"])
@pytest.mark.parametrize("n", [5, 16, 27, 29, 32, 33, 38, 45, 48, 49, 53, 56, 60, 64])
def test_pipe_walk_on_extreme_tree_shapes(n, site):
    """walk_pipe_kernel keeps one LDS cell per internal node that is not a cherry, and sizes its cells by the
    tree of the batch with the FEWEST cherries: a caterpillar (one cherry: the most cells, so fewer pattern
    groups per wave from 27 taxa on; beyond 32 taxa the tip masks need the wider register file of the two-group
    loops) and a balanced tree (the most cherries, the longest step bodies) in one
    batch with a random tree, every category count, against the oracle."""
    rng = np.random.default_rng(n)
    P = 131
    patterns = rng.integers(0, 4, (n, P)).astype(np.int32)
    patterns[rng.random((n, P)) < 0.05] = 4
    weights = rng.integers(1, 5, P).astype(np.float64)
    pid = np.stack([_shaped_rooted_parent_ids(n, "caterpillar"), _shaped_rooted_parent_ids(n, "balanced"),
                    _random_rooted_parent_ids(n, rng)])
    bl = rng.exponential(0.1, (3, 2 * n - 1))
    if n > 38:
        bl = np.maximum(bl, 1e-4)  # (39 taxa and more: the one-image form is used from 9e-7 on)
    bl[:, -1] = 0.0
    gpu, cpu = engines("GTR", site, "none", patterns, weights, 3)
    gpu.set_kernel(_capi.KERNEL_LDS_PIPE)
    params = gpu.default_params(3)
    params[:, :4] = rng.dirichlet([5, 5, 5, 5], 3)
    params[:, 4:10] = rng.dirichlet([3] * 6, 3)
    for trees in (slice(0, 3), slice(0, 1), slice(1, 2)):  # mixed batch, caterpillars only, balanced only
        try:
            out = gpu.gradients(pid[trees], bl[trees], params[trees])
        except bito_amd.BitoAmdError:
            # a 38-taxon caterpillar keeps 35 vectors per wave: with one rate category (16 patterns per group)
            # they do not fit beside the tip masks, and the forced kernel says so
            assert n >= 33 and trees.start == 0
            continue
        ref = cpu.gradients(pid[trees], bl[trees], params[trees])
        assert gpu.kernel_name() == "walk_pipe_kernel"
        assert ll_close(out["log_likelihood"], ref["log_likelihood"])
        assert grad_close(out["branch_lengths"], ref["branch_lengths"])
        assert ll_close(gpu.log_likelihoods(pid[trees], bl[trees], params[trees]), ref["log_likelihood"])


def test_auto_kernel_choice_at_the_pipe_walk_limits():
    """AUTO: walk_pipe_kernel up to 64 taxa, the size it is built for (from 49: the wide register layout; trees that keep
    too many vectors for two pattern groups per wave run with one and half-size cells -- few of them since round 4 folds
    pitchforks into their parents' steps) -- from 39 taxa on (one matrix image per branch, the reversible
    form of the pre-order recursion) only when no branch is shorter than 1e-6; everything else, and rescaling at any
    size, goes to the HBM-arena walk (walk_hbm_cat_kernel); each against the oracle."""
    rng = np.random.default_rng(29)
    for n, rescaling, shortest, expect in ((38, False, 0.0, "walk_pipe_kernel"), (41, False, 9.2e-7, "walk_pipe_kernel"),
                                           (41, False, 1e-7, "walk_hbm_cat_kernel"), (41, False, 0.0, "walk_hbm_cat_kernel"),
                                           (58, False, 1e-3, "walk_pipe_kernel"), (29, True, 0.0, "walk_hbm_cat_kernel"),
                                           # (round 3's sizes behind the earlier cases, whose random draws stay as they were)
                                           (49, False, 1e-3, "walk_pipe_kernel"), (52, False, 1e-3, "walk_pipe_kernel"),
                                           (56, False, 1e-3, "walk_pipe_kernel"), (64, False, 1e-3, "walk_pipe_kernel"),
                                           (52, False, 1e-8, "walk_hbm_cat_kernel"), (66, False, 1e-3, "walk_hbm_cat_kernel"),
                                           (60, False, 1e-3, "walk_pipe_kernel"), (64, False, 1e-8, "walk_hbm_cat_kernel"),
                                           (65, False, 1e-3, "walk_hbm_cat_kernel")):
        patterns = rng.integers(0, 4, (n, 70)).astype(np.int32)
        weights = np.ones(70)
        pid = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(4)])
        bl = np.maximum(rng.exponential(0.1, (4, 2 * n - 1)), 1e-3)
        bl[rng.random(bl.shape) < 0.1] = shortest  # a tenth of the branches are as short as the case says
        bl[:, -1] = 0.0
        gpu, cpu = engines("HKY", "weibull+4", "none", patterns, weights, 4)
        out = gpu.gradients(pid, bl, rescaling=rescaling)
        ref = cpu.gradients(pid, bl, rescaling=rescaling)
        assert gpu.kernel_name() == expect, (n, shortest)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"]), (n, shortest)
        fin = np.isfinite(ref["branch_lengths"])
        assert np.array_equal(fin, np.isfinite(out["branch_lengths"]))
        assert np.allclose(out["branch_lengths"][fin], ref["branch_lengths"][fin], rtol=1e-9, atol=GRAD_ATOL), (n, shortest)
        if n == 41 and shortest == 0.0:  # the kernel itself, when asked for, says why it does not take the batch
            gpu.set_kernel(_capi.KERNEL_LDS_PIPE)
            with pytest.raises(bito_amd.BitoAmdError, match="branch lengths of 9e-7 and more"):
                gpu.gradients(pid, bl)


def test_reversible_form_guard_counts_the_rate_matrix_too():
    """39 taxa and more: the one-image-per-branch form needs every off-diagonal entry of P(t) ~ t Q_ij well above its
    own rounding error.  With a large kappa and rare nucleotides some Q_ij are a hundred times smaller than in the
    matrices the branch-length bound was measured on, so the same branch lengths that go to walk_pipe_kernel under
    mild parameters must go to the HBM-arena walk under extreme ones -- and parity must hold either way."""
    rng = np.random.default_rng(31)
    n, T = 41, 6
    patterns = rng.integers(0, 4, (n, 90)).astype(np.int32)
    weights = np.ones(90)
    pid = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)])
    bl = np.maximum(rng.exponential(0.1, (T, 2 * n - 1)), 1e-3)
    bl[rng.random(bl.shape) < 0.2] = 2e-6  # a fifth of the branches just above the bound
    bl[:, -1] = 0.0
    gpu, cpu = engines("HKY", "weibull+4", "none", patterns, weights, 4)
    for freqs, kappa, expect in (([0.25, 0.25, 0.25, 0.25], 2.0, "walk_pipe_kernel"),
                                 ([0.03, 0.47, 0.47, 0.03], 60.0, "walk_hbm_cat_kernel")):
        params = gpu.default_params(T)
        params[:, :4] = freqs
        params[:, 4] = kappa
        params[:, 5] = 0.5
        out = gpu.gradients(pid, bl, params)
        ref = cpu.gradients(pid, bl, params)
        assert gpu.kernel_name() == expect, (freqs, kappa)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"])
        assert np.allclose(out["branch_lengths"], ref["branch_lengths"], rtol=1e-9, atol=GRAD_ATOL), (freqs, kappa)
    # new parameter rows for a resident batch move the choice as well
    mild = gpu.default_params(T)
    mild[:, 4] = 2.0
    gpu.upload(pid, bl, mild)
    gpu.run(True)
    assert gpu.kernel_name() == "walk_pipe_kernel"
    gpu.update(None, params)
    gpu.run(True)
    assert gpu.kernel_name() == "walk_hbm_cat_kernel"
    ll, grad = gpu.download()
    assert ll_close(ll, ref["log_likelihood"]) and np.allclose(grad, ref["branch_lengths"], rtol=1e-9, atol=GRAD_ATOL)


def test_large_batches_of_larger_trees_every_tree_every_pass():
    """1600 trees of 64 taxa, three passes per kernel: EVERY tree of every pass must agree between the HBM-arena walk
    (AUTO's choice at this size) and walk_lds_kernel, and a sample with the oracle.  (Round 2 found walk_lds_kernel
    returning wrong gradients for a few trees per pass, different ones each time, at 55 taxa and more when a
    workgroup walked a run of tiles: step descriptors whose scalar loads were still in flight were visible to the
    compiler, which reused their registers)"""
    n, T = 64, 1600
    w = workloads.synthetic_gtr_weibull4(n=n, P=400, tree_count=T)
    gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    gpu.set_kernel(_capi.KERNEL_HBM_ARENA)
    good = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert gpu.kernel_name() == "walk_hbm_cat_kernel"
    sel = np.r_[0:3, T - 3:T]
    ref = cpu.gradients(w.parent_ids[sel], w.branch_lengths[sel], w.params[sel])
    assert ll_close(good["log_likelihood"][sel], ref["log_likelihood"])
    assert grad_close(good["branch_lengths"][sel], ref["branch_lengths"])
    # (walk_pipe_kernel takes 39 taxa and more when t_min x Q_min / 0.2 >= 9e-7: this model's smallest rate is 0.026)
    bl_floor = np.maximum(w.branch_lengths, 1e-4)
    bl_floor[:, -1] = 0.0
    for kernel in (_capi.KERNEL_AUTO, _capi.KERNEL_LDS, _capi.KERNEL_LDS_PIPE):
        gpu.set_kernel(kernel)
        if kernel == _capi.KERNEL_LDS_PIPE:  # the wide layout, one group per wave: every tree of every pass as well
            gpu.set_kernel(_capi.KERNEL_HBM_ARENA)
            good = gpu.gradients(w.parent_ids, bl_floor, w.params)
            gpu.set_kernel(kernel)
            w.branch_lengths = bl_floor
        for _ in range(3):
            out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
            assert ll_close(out["log_likelihood"], good["log_likelihood"])
            assert grad_close(out["branch_lengths"], good["branch_lengths"])


@pytest.mark.parametrize("trees", [1, 255, 257, 1030])
def test_pipe_walk_unit_queue_sizes(trees):
    """The resident workgroups of walk_pipe_kernel take units of work from a queue: fewer units than
    workgroups, one more unit than workgroups, and a batch large enough for whole-tree units (first seven
    eighths) followed by runs of tiles -- every tree against the oracle on a sample, and twice in a row (the
    queue resets itself)."""
    w = workloads.ds1_gtr_weibull4(11).subset(trees)
    gpu, cpu = engines(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    first = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    again = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert gpu.kernel_name() == "walk_pipe_kernel"
    assert np.array_equal(first["log_likelihood"], again["log_likelihood"])
    assert np.array_equal(first["branch_lengths"], again["branch_lengths"])
    idx = np.unique(np.concatenate([np.arange(0, trees, 97), [trees - 1], np.arange(max(0, trees - 140), trees, 13)]))
    ref = cpu.gradients(w.parent_ids[idx], w.branch_lengths[idx], w.params[idx])
    assert ll_close(first["log_likelihood"][idx], ref["log_likelihood"])
    assert grad_close(first["branch_lengths"][idx], ref["branch_lengths"])


def test_pipe_walk_in_two_classes():
    """One tree with few cherries would halve the pattern groups per wave of a whole walk_pipe_kernel batch (the
    LDS region of a wave is sized by the tree that keeps the most vectors): the engine walks such a batch in
    two launches, the trees that fit four groups per wave and the others.  40 trees of 27 taxa -- caterpillars,
    balanced and random ones, interleaved -- against the oracle, with the gradient and without, and again after
    new branch lengths (the classes are a property of the topologies)."""
    n, P = 27, 300
    rng = np.random.default_rng(272)
    patterns = rng.integers(0, 4, (n, P)).astype(np.int32)
    patterns[rng.random((n, P)) < 0.03] = 4
    weights = rng.integers(1, 4, P).astype(np.float64)
    shapes = [_shaped_rooted_parent_ids(n, "caterpillar"), _shaped_rooted_parent_ids(n, "balanced")]
    pid = np.stack([shapes[t % 2] if t % 3 else _random_rooted_parent_ids(n, rng) for t in range(40)])
    bl = rng.exponential(0.1, (40, 2 * n - 1))
    bl[:, -1] = 0.0
    gpu, cpu = engines("GTR", "weibull+4", "none", patterns, weights, 8)
    params = gpu.default_params(40)
    params[:, :4] = rng.dirichlet([5, 5, 5, 5], 40)
    params[:, 4:10] = rng.dirichlet([3] * 6, 40)
    params[:, 10] = rng.uniform(0.3, 2.0, 40)
    for scale in (1.0, 0.5):
        out = gpu.gradients(pid, bl * scale, params, flags=_capi.GRAD_SITE_MODEL)
        ref = cpu.gradients(pid, bl * scale, params, flags=_capi.GRAD_SITE_MODEL)
        assert gpu.kernel_name() == "walk_pipe_kernel"
        assert ll_close(out["log_likelihood"], ref["log_likelihood"])
        assert grad_close(out["branch_lengths"], ref["branch_lengths"])
        assert grad_close(out["site_model"], ref["site_model"])
        assert ll_close(gpu.log_likelihoods(pid, bl * scale, params), ref["log_likelihood"])


@pytest.mark.parametrize("n", [33, 35, 36, 38])
def test_pipe_walk_four_groups_per_wave_beyond_32_taxa(n):
    """33 to 38 taxa with four pattern groups per wave (layout 3 of walk_pipe.hip: 40 mask registers instead of 32;
    two groups before round 4): random trees (those that keep few enough vectors for four groups -- at 38 taxa the
    engine walks the batch in two classes, most trees with two groups), every category count, the site-model pass,
    log-likelihood only; against the oracle."""
    rng = np.random.default_rng(3300 + n)
    P, T = 300, 48  # (32 trees and more: the engine may walk a batch in two classes)
    patterns = rng.integers(0, 4, (n, P)).astype(np.int32)
    patterns[rng.random((n, P)) < 0.03] = 4
    weights = rng.integers(1, 4, P).astype(np.float64)
    pid = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)])
    bl = rng.exponential(0.1, (T, 2 * n - 1))
    bl[:, -1] = 0.0
    for site in ("weibull+4", "weibull+2", "constant"):
        gpu, cpu = engines("GTR", site, "none", patterns, weights, 8)
        params = gpu.default_params(T)
        params[:, :4] = rng.dirichlet([5, 5, 5, 5], T)
        params[:, 4:10] = rng.dirichlet([3] * 6, T)
        if site != "constant":
            params[:, 10] = rng.uniform(0.3, 2.0, T)
        flags = _capi.GRAD_SITE_MODEL if site != "constant" else 0
        out = gpu.gradients(pid, bl, params, flags=flags)
        assert gpu.kernel_name() == "walk_pipe_kernel"
        if n <= 36 and site == "weibull+4":
            assert "x 4 pattern groups" in gpu.kernel_form(), gpu.kernel_form()
        ref = cpu.gradients(pid, bl, params, flags=oracle.GRAD_SITE_MODEL if flags else 0)
        assert ll_close(out["log_likelihood"], ref["log_likelihood"]), site
        assert grad_close(out["branch_lengths"], ref["branch_lengths"]), site
        if flags:
            assert grad_close(out["site_model"], ref["site_model"]), site
        assert ll_close(gpu.log_likelihoods(pid, bl, params), ref["log_likelihood"]), site


def test_general_kernel_model_index_follows_the_resident_batch():
    """Selecting the general-state kernels AFTER a batch was uploaded under another kernel choice must not
    reuse the model index of an earlier batch: set_kernel(GENERAL), upload A, set_kernel(AUTO), upload B (same
    tree count, other parameter rows), set_kernel(GENERAL), run."""
    w = workloads.ds1_gtr_weibull4(1).subset(6)
    gpu = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns, w.weights)
    rng = np.random.default_rng(5)
    pa = w.params.copy()  # batch A: all rows equal -> every tree shares record 0
    pb = w.params.copy()  # batch B: every tree its own Weibull shape
    pb[:, 10] = 0.3 + rng.random(6)
    gpu.set_kernel(_capi.KERNEL_GENERAL)
    gpu.upload(w.parent_ids, w.branch_lengths, pa)
    gpu.run(True)
    gpu.set_kernel(_capi.KERNEL_AUTO)
    gpu.upload(w.parent_ids, w.branch_lengths, pb)
    gpu.set_kernel(_capi.KERNEL_GENERAL)
    gpu.run(True)
    ll, grad = gpu.download()
    assert gpu.kernel_name().startswith("gs_walk")

    def fresh_general(params, want_gradient):
        eng = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns, w.weights)
        eng.set_kernel(_capi.KERNEL_GENERAL)
        eng.upload(w.parent_ids, w.branch_lengths, params)
        eng.run(want_gradient)
        return eng.download(want_gradient)

    ll_ref, grad_ref = fresh_general(pb, True)  # the same kernels with the index built at upload: bit-equal
    assert np.array_equal(ll, ll_ref) and np.array_equal(grad, grad_ref)
    # (the general-state set-up is a different eigensolver from the 4-state one: 1e-9 against the 4-state oracle)
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, pb)
    assert np.abs(ll - ref["log_likelihood"]).max() < 1e-9
    assert grad_close(grad, ref["branch_lengths"]), grad_close(grad, ref["branch_lengths"])
    # and update() of the rows alone, again under another selection
    gpu.set_kernel(_capi.KERNEL_AUTO)
    gpu.update(None, pa)
    gpu.set_kernel(_capi.KERNEL_GENERAL)
    gpu.run(False)
    ll, _ = gpu.download(False)
    assert np.array_equal(ll, fresh_general(pa, False)[0])


def test_resident_update_and_time_tree_shapes_are_checked():
    """The C ABI reads T*M / T*param_count / T*(2n-1) doubles from bare pointers; the host mirror rejects
    arrays of any other shape before they cross the boundary."""
    w = workloads.ds1_gtr_weibull4(1).subset(4)
    gpu = bito_amd.Engine(spec(w.substitution, w.site, w.clock), w.patterns, w.weights)
    with pytest.raises(bito_amd.BitoAmdError, match="no batch is resident"):
        gpu.update(w.branch_lengths)
    gpu.upload(w.parent_ids, w.branch_lengths, w.params)
    with pytest.raises(bito_amd.BitoAmdError, match="uploaded shape"):
        gpu.update(w.branch_lengths[:2])
    with pytest.raises(bito_amd.BitoAmdError, match="uploaded shape"):
        gpu.update(w.branch_lengths[:, :-1])
    with pytest.raises(bito_amd.BitoAmdError, match="param matrix"):
        gpu.update(None, w.params[:, :-1])
    gpu.update(w.branch_lengths, w.params)  # the right shapes still pass
    n = gpu.taxon_count
    pid = np.zeros((2, 2 * n - 2), dtype=np.int32)
    with pytest.raises(bito_amd.BitoAmdError, match="node_heights must have shape"):
        gpu.log_det_jacobian(pid, np.zeros((2, n)), np.zeros((2, 2 * n - 1)))
    with pytest.raises(bito_amd.BitoAmdError, match="height_ratios must have shape"):
        gpu.time_trees_from_height_ratios(pid, np.zeros((2, 2 * n - 1)), np.zeros((2, n)))
    with pytest.raises(bito_amd.BitoAmdError, match="tip_dates must have shape"):
        gpu.time_trees_from_branch_lengths(pid, np.zeros((2, 2 * n - 1)), np.zeros(n + 1))
    with pytest.raises(bito_amd.BitoAmdError, match="rooted trees"):
        gpu.log_det_jacobian(pid[:, :-1], np.zeros((2, 2 * n - 1)), np.zeros((2, 2 * n - 1)))


def test_beagle_shim_rejects_out_of_range_indices(data_dir):
    """Every buffer, matrix and scale index of an operation list is range-checked before anything is
    launched (BEAGLE_ERROR_OUT_OF_RANGE = -5), negative child indices included; a resource list selects the
    device."""
    import ctypes as C

    from beagle_driver import BEAGLE_OP_NONE, FLAG_SCALING_MANUAL, FatBeagleDriver, InstanceDetails, Operation

    tc, sp = load(data_dir, "hello.fasta", "hello.nwk")
    V, Vinv = np.eye(4), np.eye(4)
    drv = FatBeagleDriver(sp.patterns, sp.weights, V, Vinv, np.zeros(4), np.full(4, 0.25), np.zeros((4, 4)), [1.0], [1.0])
    n, N = drv.n, drv.N
    good = (n, BEAGLE_OP_NONE, BEAGLE_OP_NONE, 0, 0, 1, 1)
    out_of_range = -5
    for field, value in [(0, -1), (0, 10**6), (3, -1), (3, 10**6), (5, -7), (5, 10**6), (4, -1), (4, 2 * N),
                         (6, 2 * N), (1, 10**6), (1, -2), (2, -3)]:
        row = list(good)
        row[field] = value
        ops = (Operation * 1)(Operation(*row))
        assert drv.lib.beagleUpdatePartials(drv.inst, ops, 1, BEAGLE_OP_NONE) == out_of_range, (field, value)
        assert drv.lib.beagleUpdatePrePartials(drv.inst, ops, 1, BEAGLE_OP_NONE) == out_of_range, (field, value)
    ops = (Operation * 1)(Operation(*good))
    assert drv.lib.beagleUpdatePartials(drv.inst, ops, 1, 10**6) == out_of_range  # cumulative scale index
    assert drv.lib.beagleUpdatePartials(drv.inst, ops, 1, BEAGLE_OP_NONE) == 0
    bad = np.array([10**6], dtype=np.int32)
    ok = np.array([0], dtype=np.int32)
    grad = np.zeros(1)
    gp = grad.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    for post, pre, dm in [(bad, ok, ok), (ok, bad, ok), (ok, ok, bad), (-bad, ok, ok)]:
        assert drv.lib.beagleCalculateEdgeDerivatives(drv.inst, ip(post), ip(pre), ip(dm), ip(ok), 1, None, gp,
                                                      None) == out_of_range
    ll = C.c_double()
    assert drv.lib.beagleCalculateRootLogLikelihoods(drv.inst, ip(ok), ip(ok), ip(ok), ip(bad), 1,
                                                     C.byref(ll)) == out_of_range
    drv.close()
    # resource list: device 0 is accepted and reported; a list without any existing device is refused
    info = InstanceDetails()
    lib = drv.lib
    res = (C.c_int * 2)(99, 0)
    inst = lib.beagleCreateInstance(3, 7, 3, 4, 15, 1, 10, 1, 8, res, 2, 0, FLAG_SCALING_MANUAL, C.byref(info))
    assert inst >= 0 and info.resourceNumber == 0
    assert lib.beagleFinalizeInstance(inst) == 0
    res = (C.c_int * 1)(99)
    assert lib.beagleCreateInstance(3, 7, 3, 4, 15, 1, 10, 1, 8, res, 1, 0, FLAG_SCALING_MANUAL, C.byref(info)) == -6  # BEAGLE_ERROR_NO_RESOURCE
