"""Pins the general-state-count CPU oracle (oracle/gs_oracle.c, the 61-state codon path of BASELINE
config 5).  The reference has no model with more than four states, so: (1) the state-count-generic
code path is run at S = 4 with the reference's GTR matrix and checked against the reference's own
goldens and against the pinned 4-state oracle; (2) the GY94 matrix builder, which has no reference
counterpart, is checked by its defining properties; (3) codon gradients are checked by central finite
differences.  CPU only."""
import json
import os

import numpy as np
import pytest

from bito_amd import site_pattern, treeio, workloads
from oracle import gs, oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_goldens.json")) as fh:
    GOLD = json.load(fh)


def _load(data_dir, fasta, trees):
    path = os.path.join(data_dir, trees)
    tc = treeio.read_nexus_file(path) if trees.endswith(".t") else treeio.read_newick_file(path)
    sp = site_pattern.SitePattern(treeio.read_fasta(os.path.join(data_dir, fasta)), tc.taxon_names)
    return tc, sp


def _equal_gtr(T, shape=None):
    row = [0.25] * 4 + [1.0 / 6] * 6 + ([] if shape is None else [shape])
    return np.tile(np.array(row), (T, 1))


def test_generic_path_reproduces_ds1_jc69_goldens(data_dir):
    """src/unrooted_sbn_instance.hpp:245-287 (pybeagle log-likelihoods, physher gradient) through the
    S-generic code with GTR(equal) = JC69 (the identity test/test_bito.py:97-122 relies on)."""
    g = GOLD["ds1_jc69"]
    tc, sp = _load(data_dir, g["fasta"], g["trees"])
    eng = gs.GsOracleEngine("GTR", "constant", sp.patterns, sp.weights, 4)
    pid, bl = tc.parent_id_matrix(), tc.branch_length_matrix()
    for rescaling in (False, True):
        out = eng.gradients(pid, bl, _equal_gtr(len(tc.trees)), rescaling=rescaling)
        assert np.abs(out["log_likelihood"] - g["log_likelihoods"]).max() < 5e-10
        last = np.sort(out["branch_lengths"][-1])
        assert np.abs(last - g["last_tree_sorted_branch_gradient"]).max() < g["gradient_tol"]


def test_generic_path_reproduces_ds1_weibull_goldens(data_dir):
    """src/unrooted_sbn_instance.hpp:314-348: JC69 + weibull+4, shape 0.1 (physher)."""
    g = GOLD["ds1_jc69_weibull4_shape0.1"]
    tc, sp = _load(data_dir, g["fasta"], g["trees"])
    eng = gs.GsOracleEngine("GTR", "weibull+4", sp.patterns, sp.weights, 4)
    out = eng.gradients(tc.parent_id_matrix(), tc.branch_length_matrix(), _equal_gtr(len(tc.trees), g["shape"]))
    assert np.abs(out["log_likelihood"] - g["log_likelihoods"]).max() < 5e-10
    assert np.abs(out["branch_lengths"][:, 0] - g["branch_gradient_entry0"]).max() < 2e-6


def test_generic_path_reproduces_flua_gtr_golden(data_dir):
    """src/rooted_sbn_instance.hpp:347-376: fluA, rooted, GTR, strict clock rate 0.001 (tolerance 1e-3
    in the reference; the golden came from another tool)."""
    g = GOLD["flua_gtr"]
    tc, sp = _load(data_dir, "fluA.fa", "fluA.tree")
    rates = np.full((1, tc.trees[0].node_count - 1), 0.001)
    eng = gs.GsOracleEngine("GTR", "constant", sp.patterns, sp.weights)
    params = np.array([workloads.GTR_FREQS + workloads.GTR_RATES])
    ll = eng.log_likelihoods(tc.parent_id_matrix(), tc.branch_length_matrix(), params, rates=rates)
    assert abs(ll[0] - g["log_likelihood"]) < 1e-3


def test_generic_path_equals_pinned_four_state_oracle():
    """Headline model (GTR + weibull+4) on DS1: same numbers as bito_oracle.c, which is pinned to the
    reference's goldens (tests/test_oracle_golden.py)."""
    w = workloads.ds1_gtr_weibull4(1).subset(6)
    ref = oracle.OracleEngine(w.substitution, w.site, "none", w.patterns, w.weights, 4)
    eng = gs.GsOracleEngine("GTR", "weibull+4", w.patterns, w.weights, 4)
    a = ref.gradients(w.parent_ids, w.branch_lengths, w.params)
    b = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    # two correct FP64 set-ups (different Jacobi ordering, fma chain in P(t)) differ by a few 1e-10 on
    # |LL| ~ 8000: P(t) rounding is coherent across all site patterns
    assert np.abs(a["log_likelihood"] - b["log_likelihood"]).max() < 1e-9
    assert np.abs(a["branch_lengths"] - b["branch_lengths"]).max() < 1e-6


def test_gtr_eigenvalues_match_r():
    """src/substitution_model.hpp:146-167 (eigenvalues from R), through the padded Jacobi solver."""
    g = GOLD["gtr_eigenvalues_r"]
    _, _, _, lam, _ = gs.substitution_model("GTR", np.array(g["frequencies"] + g["rates"]))
    top4 = np.sort(np.sort(lam)[:3].tolist() + [np.abs(lam).min()])
    assert np.abs(np.sort(g["eigenvalues"]) - top4).max() < 1e-4


CODON_PARAMS = np.array([0.3, 0.2, 0.25, 0.25, 2.5, 0.3])  # freqs A,C,G,T | kappa, omega


def test_gy94_rate_matrix_properties():
    Q, V, Vinv, lam, pi = gs.substitution_model("GY94", CODON_PARAMS)
    tab = gs.codon_table()
    assert Q.shape == (61, 61) and abs(pi.sum() - 1) < 1e-15
    assert np.abs(Q.sum(axis=1)).max() < 1e-14  # rows sum to zero
    assert abs(-(pi * np.diag(Q)).sum() - 1.0) < 1e-14  # one expected substitution per unit time
    flux = pi[:, None] * Q
    assert np.abs(flux - flux.T).max() < 1e-16  # detailed balance
    f = CODON_PARAMS[:4]
    raw = f[tab[:, 0]] * f[tab[:, 1]] * f[tab[:, 2]]
    assert np.abs(pi - raw / raw.sum()).max() < 1e-16  # F1x4
    # structure: zero for multi-nucleotide changes; kappa on transitions; omega on amino-acid changes
    aa = [site_pattern.amino_acid(*c) for c in tab]
    scale = None
    for i in range(61):
        for j in range(61):
            if i == j:
                continue
            diff = [k for k in range(3) if tab[i, k] != tab[j, k]]
            if len(diff) != 1:
                assert Q[i, j] == 0.0
                continue
            x, y = tab[i, diff[0]], tab[j, diff[0]]
            expect = pi[j] * (CODON_PARAMS[4] if (x ^ y) == 2 else 1.0) * (CODON_PARAMS[5] if aa[i] != aa[j] else 1.0)
            scale = Q[i, j] / expect if scale is None else scale
            assert abs(Q[i, j] / expect - scale) < 1e-12
    # the eigendecomposition reconstructs Q, and P(t) equals scipy's matrix exponential
    from scipy.linalg import expm
    full = gs.lib()
    Qf, Vf, Vif = (np.zeros(64 * 64) for _ in range(3))
    lamf, pif = np.zeros(64), np.zeros(64)
    assert full.gs_substitution_model(b"GY94", gs._dp(CODON_PARAMS), gs._dp(Qf), gs._dp(Vf), gs._dp(Vif), gs._dp(lamf),
                                      gs._dp(pif)) == 0
    Vf, Vif = Vf.reshape(64, 64), Vif.reshape(64, 64)
    assert np.abs((Vf * lamf) @ Vif - Qf.reshape(64, 64)).max() < 1e-13
    for t in (0.01, 0.3, 2.0):
        P = gs.transition_matrix_padded(Vf, Vif, lamf, t)[:61, :61]
        assert np.abs(P - expm(Q * t)).max() < 1e-13
        assert np.abs(P.sum(axis=1) - 1).max() < 1e-13


def _flu_codon(data_dir, T=1, seed=5):
    tc = treeio.read_newick_file(os.path.join(data_dir, "fluA.tree"))
    sp = site_pattern.CodonSitePattern(treeio.read_fasta(os.path.join(data_dir, "fluA.fa")), tc.taxon_names)
    pid = np.tile(tc.parent_id_matrix(), (T, 1))
    rng = np.random.default_rng(seed)
    bl = np.tile(tc.branch_length_matrix(), (T, 1)) * 0.002 * rng.uniform(0.5, 1.5, (T, pid.shape[1] + 1))
    bl[:, -1] = 0
    return sp, pid, bl


def test_codon_site_patterns(data_dir):
    sp, _, _ = _flu_codon(data_dir)
    assert sp.patterns.shape == (69, 242) and sp.weights.sum() == 329  # 987 nt = 329 codon columns
    assert sp.patterns.max() == 61 and sp.patterns.min() >= 0
    assert site_pattern.codon_state_vector("AAAAACTAAATGNNNTGAAT").tolist() == [0, 1, 61, gs.codon_state(0, 3, 2), 61, 61]


def test_codon_gradient_is_the_derivative_of_the_log_likelihood(data_dir):
    sp, pid, bl = _flu_codon(data_dir)
    keep = slice(0, 40)  # a slice of the patterns keeps this quick; every pattern is an independent term
    eng = gs.GsOracleEngine("GY94", "weibull+2", sp.patterns[:, keep], sp.weights[keep], 8)
    params = np.array([list(CODON_PARAMS) + [0.7]])
    out = eng.gradients(pid, bl, params)
    branches = [0, 7, 68, 69, 100, 135]
    eps = 1e-6
    plus = np.repeat(bl, len(branches), axis=0)
    minus = plus.copy()
    for k, b in enumerate(branches):
        plus[k, b] += eps
        minus[k, b] -= eps
    P = np.repeat(params, len(branches), axis=0)
    Pid = np.repeat(pid, len(branches), axis=0)
    fd = (eng.log_likelihoods(Pid, plus, P) - eng.log_likelihoods(Pid, minus, P)) / (2 * eps)
    assert np.abs(fd - out["branch_lengths"][0, branches]).max() < 2e-5 * max(1.0, np.abs(fd).max())
    assert out["branch_lengths"][0, -1] == 0.0
    # rescaling leaves the results unchanged
    res = eng.gradients(pid, bl, params, rescaling=True)
    assert abs(res["log_likelihood"][0] - out["log_likelihood"][0]) < 1e-9
    assert np.abs(res["branch_lengths"] - out["branch_lengths"]).max() < 1e-7


def test_site_model_gradient_generic_path(data_dir):
    """src/rooted_sbn_instance.hpp:409-430: fluA, JC69 + weibull+4 shape 0.1, d/d shape = -5.231329
    (the S-generic site-model pass at S = 4), and the codon model's against a central difference."""
    g = GOLD["flua_jc69_weibull4_shape0.1"]
    tc, sp = _load(data_dir, "fluA.fa", "fluA.tree")
    rates = np.full((1, tc.trees[0].node_count - 1), 0.001)
    eng = gs.GsOracleEngine("GTR", "weibull+4", sp.patterns, sp.weights)
    out = eng.gradients(tc.parent_id_matrix(), tc.branch_length_matrix(), _equal_gtr(1, g["shape"]), rates=rates,
                        site_model=True)
    assert abs(out["log_likelihood"][0] - g["log_likelihood"]) < 1e-9
    assert abs(out["site_model"][0] - g["site_model_gradient"]) < 1e-6
    sp, pid, bl = _flu_codon(data_dir)
    keep = slice(0, 40)
    eng = gs.GsOracleEngine("GY94", "weibull+3", sp.patterns[:, keep], sp.weights[keep], 8)
    params = np.array([list(CODON_PARAMS) + [0.7]])
    got = eng.gradients(pid, bl, params, site_model=True)["site_model"][0]
    eps = 1e-6
    plus, minus = params.copy(), params.copy()
    plus[0, 6] += eps
    minus[0, 6] -= eps
    fd = (eng.log_likelihoods(pid, bl, plus)[0] - eng.log_likelihoods(pid, bl, minus)[0]) / (2 * eps)
    assert abs(got - fd) < 1e-5 * max(1.0, abs(fd))
