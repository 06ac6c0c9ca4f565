"""Round 6's changes that a device must confirm (they were built without GPU access, their parity held under tests/hip_emu:
tests/test_engine_emulated.py runs these very functions on the emulated library).
* small calls: tree set-up, step tables and matrix images as ONE launch (walk_pipe.hip, pipe_small_prepare_kernel) -- the
  same functions in the same order as the three-launch route: the same bits in every result;
* walk_hbm_cat_kernel with four-tip subtrees rebuilt where they are used (BITO_AMD_HBM_FOLD=2): against the CPU checker on
  trees that hold every neighbour case is tests/test_engine_emulated.py's; here on mid-size trees against the checker, and
  the three levels against one another;
* other wave counts of the GP optimiser's workgroups (BITO_AMD_GP_OPT_WAVES);
* the final sums of a tree by the last of its runs of tiles (BITO_AMD_PIPE_LAST_UNIT=1): no final-sums launch.
All of them are switches that are OFF by default until a device has run them (scripts/gpu_round6.sh times them)."""
import numpy as np
import pytest

import bito_amd
from bito_amd import workloads
from test_engine_chunks import GRAD_ATOL, GRAD_RTOL, LL_ATOL, LL_RTOL, _close, _Env, _spec


def small_call_results():
    """results of small blocking calls (1 to 100 trees of 6 to 45 taxa; the same rows twice, then other rows: the model
    of the call before copied, then set up afresh) by the fused launch and by the three-launch route"""
    out = {}
    shapes = [(6, 24, 1), (9, 70, 7), (12, 40, 8), (27, 60, 9), (33, 50, 17), (45, 40, 5)]
    if "cpu-emulation" not in bito_amd.version():  # (a hundred trees as fibers take a minute: the device's case)
        shapes.append((27, 130, 100))
    for n, P, T in shapes:
        w = workloads.synthetic_gtr_weibull4(n, P, tree_count=T)
        w.rescaling = False
        for fused in (1, 0):
            with _Env(BITO_AMD_SMALL_PREPARE=fused):
                eng = bito_amd.Engine(_spec(w), w.patterns, w.weights)
            for rep in range(3):
                params = w.params if rep < 2 else workloads.other_bits(w.params, 4)
                r = eng.gradients(w.parent_ids, w.branch_lengths, params)
                assert eng.kernel_name() == "walk_pipe_kernel", eng.kernel_name()
                out[(n, T, rep, fused)] = np.concatenate([r["log_likelihood"].ravel(), r["branch_lengths"].ravel()])
            out[(n, T, "ll", fused)] = eng.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
    return out


@pytest.mark.gpu
def test_small_calls_in_one_set_up_launch_give_the_three_launch_routes_bits():
    out = small_call_results()
    for key, value in out.items():
        if key[3] == 1:
            other = out[key[:3] + (0,)]
            assert np.all(np.isfinite(value)) and np.array_equal(value, other), (key, float(np.abs(value - other).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [41, 65, 100])
def test_hbm_walk_with_four_tip_subtrees_folded_against_the_checker(n):
    """the HBM-arena walk at fold level 2 (four-tip subtrees rebuilt in their parents' steps), 20 trees x 300 patterns,
    with and without rescaling, against the CPU checker"""
    from oracle import oracle

    w = workloads.synthetic_gtr_weibull4(n, 300, tree_count=20)
    eng = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    eng.set_kernel(1)
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    with _Env(BITO_AMD_HBM_FOLD=2):  # (read at every launch)
        for rescaling in (True, False):
            out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
            ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
            assert eng.kernel_name().startswith("walk_hbm_cat")
            assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
            assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)


@pytest.mark.gpu
def test_hbm_walk_fold_levels_agree():
    """BITO_AMD_HBM_FOLD = 2 (four-tip subtrees rebuilt in the step), 1 (pitchforks only, round 4) and 0 on the same 24
    trees of 64 taxa: the same arithmetic in another grouping of the steps -- without rescaling the same bits, with it
    (a folded node's power-of-two rescaling is skipped) a tenth of the bars"""
    w = workloads.synthetic_gtr_weibull4(64, 200, tree_count=24 if "cpu-emulation" not in bito_amd.version() else 8)
    res = {}
    for fold in (2, 1, 0):
        with _Env(BITO_AMD_HBM_FOLD=fold):
            eng = bito_amd.Engine(_spec(w), w.patterns, w.weights)
            eng.set_kernel(1)
            for rescaling in (False, True):
                out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
                assert eng.kernel_name().startswith("walk_hbm_cat")
                res[(fold, rescaling)] = (out["log_likelihood"].copy(), out["branch_lengths"].copy())
    for fold in (1, 0):
        assert np.array_equal(res[(2, False)][0], res[(fold, False)][0]) and np.array_equal(res[(2, False)][1], res[(fold, False)][1])
        assert _close(res[(2, True)][0], res[(fold, True)][0], 0.1 * LL_ATOL, 0.1 * LL_RTOL)
        assert _close(res[(2, True)][1], res[(fold, True)][1], 0.1 * GRAD_ATOL, 0.1 * GRAD_RTOL)
    # (the levels really differ in what they fold: the rescaled gradients carry another rounding)
    assert not np.array_equal(res[(2, True)][1], res[(1, True)][1]) and not np.array_equal(res[(1, True)][1], res[(0, True)][1])


@pytest.mark.gpu
@pytest.mark.parametrize("waves", [16, 1])
def test_other_wave_counts_of_the_optimiser_workgroup(data_dir, waves):
    """BITO_AMD_GP_OPT_WAVES = 16 (gp_optimize_kernel with a pattern per thread at DS1's size) and 1 (no cross-wave sum at
    all); four waves is the default -- another summation order, the same bars: Brent visits the checker's points on fluA and on the DS1 ten-tree DAG
    (tests/gp_trace.py) and ends at its lengths to 1e-8, Newton too; and a scheduled sweep is still bit for bit the sequential
    one (both run the same kernel)."""
    import gp_trace
    import test_gp
    from bito_amd import gp

    sp, tree, flu = test_gp._flu(data_dir)
    cases = [("fluA", sp, flu, flu.branch_lengths(np.full(tree.node_count, 0.01)))]
    dag, sp2 = workloads.ds1_subsplit_dag(10)
    cases.append(("DS1 DAG", sp2, dag, np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count)))
    with _Env(BITO_AMD_GP_OPT_WAVES=waves):
        for name, sp_, dag_, bl0 in cases:
            cpu, bl_cpu = test_gp._traced_sweep(test_gp._oracle_factory, sp_, dag_, bl0, gp.BRENT)
            gpu, bl_gpu = test_gp._traced_sweep(test_gp._gpu_factory, sp_, dag_, bl0, gp.BRENT)
            problems, stats = gp_trace.compare(cpu, gpu)
            assert not problems, (name, problems[:3], stats)
            assert stats["ties"] == 0 and stats["compared"] == len(cpu) == len(gpu), (name, stats)
            assert np.abs(bl_gpu - bl_cpu).max() < 1e-8, name
            results = []
            for factory in (test_gp._gpu_factory, test_gp._oracle_factory):
                eng = factory(sp_, dag_)
                eng.set_branch_lengths(bl0)
                eng.set_optimization_method(gp.NEWTON)
                eng.reset_optimization_count()
                eng.process_operations(dag_.populate_plvs())
                eng.process_operations(dag_.branch_length_optimization())
                results.append(eng.get_branch_lengths())
            assert np.abs(results[0] - results[1]).max() < 1e-8 * max(1.0, np.abs(results[1]).max()), name
        test_gp._scheduled_and_sequential_sweeps_agree([(workloads.ds1_subsplit_dag(10), 1e-40)], (gp.BRENT, gp.NEWTON))


@pytest.mark.gpu
def test_last_unit_of_a_tree_forms_its_final_sums():
    """BITO_AMD_PIPE_LAST_UNIT=1: walk_pipe_kernel's run-of-tiles units count themselves per tree, and the one that counts last
    forms the tree's final sums as reduce_tiles_kernel would (the same rows in the same order: the same bits) and, for a
    blocking call, the workgroup that finishes the launch's last tree stores the completion flag -- no final-sums launch
    behind the traversal.  Blocking calls and passes over a resident batch, batches that are all whole-tree units, all runs of
    tiles, and both (BITO_AMD_PIPE_WHOLE_TREES), against the two-launch route bit for bit; a 2800-tree call in chunks."""
    import os

    res = {}
    for last in (1, 0):
        for n, P, T, whole in ((9, 70, 7, None), (12, 200, 8, 3), (27, 130, 9, 0), (33, 300, 5, 2), (45, 64, 6, None)):
            w = workloads.synthetic_gtr_weibull4(n, P, tree_count=T)
            w.rescaling = False
            env = {"BITO_AMD_PIPE_LAST_UNIT": last}
            if whole is not None:
                env["BITO_AMD_PIPE_WHOLE_TREES"] = whole
            with _Env(**env):
                eng = bito_amd.Engine(_spec(w), w.patterns, w.weights)
                for rep in range(2):
                    r = eng.gradients(w.parent_ids, w.branch_lengths * (1 + 0.01 * rep), w.params)
                    assert eng.kernel_name() == "walk_pipe_kernel"
                    res[(n, rep, last)] = np.concatenate([r["log_likelihood"].ravel(), r["branch_lengths"].ravel()])
                res[(n, "ll", last)] = eng.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
                eng.upload(w.parent_ids, w.branch_lengths, w.params)
                for _ in range(2):
                    eng.run(True, False)
                res[(n, "resident", last)] = np.concatenate([np.ravel(x) for x in eng.download() if x is not None])
        if "cpu-emulation" not in bito_amd.version():  # (chunks on two workers, each with its own counters and flag)
            big = workloads.ds1_gtr_weibull4(28)
            with _Env(BITO_AMD_PIPE_LAST_UNIT=last):
                eng = bito_amd.Engine(_spec(big), big.patterns, big.weights)
                r = eng.gradients(big.parent_ids, big.branch_lengths, big.params)
            res[("chunks", 0, last)] = np.concatenate([r["log_likelihood"].ravel(), r["branch_lengths"].ravel()])
    for key, value in res.items():
        if key[2] == 1:
            other = res[key[:2] + (0,)]
            assert np.all(np.isfinite(value)) and np.array_equal(value, other), (key, float(np.abs(value - other).max()))


@pytest.mark.gpu
def test_seeded_sweeps_with_round_6s_switches_on():
    """scripts/gpu_fuzz.py (random shapes, models, rooted and unrooted, rescaling on and off, against the CPU checker) with
    the forms round 6 built behind switches: the HBM-arena walk pinned with four-tip subtrees folded and trees of up to 333 taxa;
    walk_pipe_kernel pinned with small calls set up in one launch and a tree's final sums formed by its last unit.  (The same
    sweeps ran on the emulated library: profiles/r6_cpu/fuzz_*.log.)"""
    from test_gpu_fuzz import _sweep

    done, declined = _sweep("gpu_fuzz.py", 100, 6106, 1, env={"FUZZ_LARGE_TREES": "1", "BITO_AMD_HBM_FOLD": "2"})
    assert done == 100 and declined == 0
    done, declined = _sweep("gpu_fuzz.py", 100, 6107, 5, env={"BITO_AMD_SMALL_PREPARE": "1", "BITO_AMD_PIPE_LAST_UNIT": "1"})
    assert done == 100 and declined < 100
