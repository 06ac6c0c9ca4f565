"""Path B's executor -- bito_amd/csrc/gp_engine.hip itself, kernels and host code -- executed on the CPU through the
stand-in HIP runtime of tests/hip_emu (every workgroup's threads as fibers, barriers and wave shuffles with their real
semantics), against the CPU checker.  Round 5 changed this file's launch structure (gp_schedule.hpp: the optimisations
of equal optimiser depth as concurrent workgroups), its optimiser kernels (the log-likelihood alone at Brent's trial
points, one workgroup per optimisation of a launch, the evaluation trace) in a round without GPU access: these tests are
what stands in for the `-m gpu` tests of tests/test_gp.py until those run on an MI355X again -- they call the very same
test functions with the library swapped.  What emulation cannot show is listed in tests/hip_emu/hip/hip_runtime.h.
Test infrastructure: the product has no CPU path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import test_gp
from bito_amd import _capi, gp, workloads

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hip_emu", "_build", "libgp_emu.so")


@pytest.fixture(scope="module")
def emulated():
    built = subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hip_emu")], capture_output=True, text=True)
    assert built.returncode == 0, built.stdout + built.stderr
    keep = _capi._lib
    _capi._lib = C.CDLL(EMU)  # (gp._lib() sets the GP entry points' signatures on whatever _capi.lib() returns)
    try:
        yield
    finally:
        _capi._lib = keep


def test_emulated_executor_matches_the_checker(emulated, data_dir):
    """tests/test_gp.py::test_gp_executor_matches_oracle under emulation: hello and fluA schedules, derivatives, PLVs"""
    test_gp.test_gp_executor_matches_oracle(data_dir)


def test_emulated_brent_follows_the_checkers_iterates(emulated, data_dir):
    """tests/test_gp.py::test_device_brent_follows_the_checkers_iterates under emulation"""
    test_gp.test_device_brent_follows_the_checkers_iterates(data_dir)


def test_emulated_scheduled_sweep_is_bitwise_the_sequential_one(emulated):
    """tests/test_gp.py::test_scheduled_sweep_is_bitwise_the_sequential_one under emulation, on the DS1 ten-tree DAG with
    rescaling in play (threshold 1e-4), Brent and Newton: concurrent workgroups of a launch against one optimisation per
    launch in stream order, bit for bit (branch lengths, differences, per-GPCSP log-likelihoods)."""
    test_gp._scheduled_and_sequential_sweeps_agree(((workloads.ds1_subsplit_dag(10), 1e-4),), (gp.BRENT, gp.NEWTON))


def test_emulated_branch_length_optimisation(emulated, data_dir):
    """The heart of tests/test_gp.py::test_gp_branch_length_optimization_on_device under emulation (the whole of it takes
    four minutes as fibers): hello's venus edge to the reference's optimum (src/gp_doctest.cpp:326-346), Newton beating
    Brent; one sweep over fluA's 136 edges with Brent and Newton against the checker -- branch lengths, differences and
    the marginal to 1e-8 --; the second sweep's convergence skip (differences below 1e-15 are left alone only once the
    optimisation count is incremented, src/dag_branch_handler.cpp:127-131)."""
    true_length = 0.0694244266
    brent, _ = test_gp._optimized_venus_length(test_gp._gpu_factory, data_dir, gp.BRENT)
    newton, gpu = test_gp._optimized_venus_length(test_gp._gpu_factory, data_dir, gp.NEWTON)
    newton_cpu, cpu = test_gp._optimized_venus_length(test_gp._oracle_factory, data_dir, gp.NEWTON)
    assert abs(newton - true_length) < 1e-6 and abs(newton - true_length) < abs(brent - true_length)
    assert np.abs(gpu.get_branch_lengths() - cpu.get_branch_lengths()).max() < 1e-12
    sp, tree, dag = test_gp._flu(data_dir)
    bl0 = dag.branch_lengths(np.full(tree.node_count, 0.01))
    for method in (gp.NEWTON, gp.BRENT):
        results = []
        for factory in (test_gp._gpu_factory, test_gp._oracle_factory):
            eng = factory(sp, dag)
            eng.set_branch_lengths(bl0)
            eng.set_optimization_method(method)
            eng.reset_optimization_count()
            eng.process_operations(dag.populate_plvs())
            eng.process_operations(dag.branch_length_optimization())
            first = eng.get_branch_lengths()
            eng.increment_optimization_count()
            eng.process_operations(dag.branch_length_optimization())
            eng.process_operations(dag.populate_plvs())
            eng.process_operations(dag.marginal_likelihood())
            results.append((first, eng.get_branch_lengths(), eng.get_branch_length_differences(), eng.get_log_marginal_likelihood()))
        for x, y in zip(*results):
            assert np.abs(np.asarray(x) - np.asarray(y)).max() < 1e-8, method
        assert results[0][3] > -5000  # the sweeps improved on the starting tree


@pytest.mark.parametrize("fasta,newick", test_gp.COMPOSITE_CASES[:2])
def test_emulated_multi_tree_dags(emulated, data_dir, fasta, newick):
    """composite marginals and branch-length estimation on multi-tree DAGs under emulation"""
    test_gp.test_gp_executor_composite_marginal(data_dir, fasta, newick)
    test_gp.test_estimate_branch_lengths_on_multi_tree_dags_gpu(data_dir, fasta, newick)


def test_emulated_rescaling_counts(emulated, data_dir):
    test_gp.test_gp_rescaling_counts_and_plvs_as_the_reference_holds_them(data_dir, 0.1, 1e-2)


def _gpu_marked(module):
    """(name, function, parameter sets) of every `-m gpu` test of a module, as pytest would expand them"""
    import itertools

    out = []
    for name, fn in sorted(vars(module).items()):
        marks = getattr(fn, "pytestmark", [])
        if not callable(fn) or not name.startswith("test_") or not any(m.name == "gpu" for m in marks):
            continue
        axes = []
        for m in marks:
            if m.name == "parametrize":
                names = [a.strip() for a in m.args[0].split(",")]
                axes.append([dict(zip(names, v if len(names) > 1 else (v,))) for v in m.args[1]])
        for combo in itertools.product(*axes) if axes else [()]:
            kwargs = {}
            for part in combo:
                kwargs.update(part)
            out.append((name, fn, kwargs))
    return out


def _call(fn, kwargs, data_dir):
    import inspect

    if "data_dir" in inspect.signature(fn).parameters:
        kwargs = dict(kwargs, data_dir=data_dir)
    fn(**kwargs)


def test_emulated_nni_and_top_pruning(emulated, data_dir):
    """Every `-m gpu` test of tests/test_nni.py and tests/test_tp.py (f4: batched NNI proposals through
    bito_amd_gp_process_operation_batches, with and without optimiser operations -- gp_block_stream_kernel shares the
    optimiser code round 5 touched --, the engine growing with the DAG, top-pruning scores) under emulation, except the
    one that needs more workgroups than is pleasant as fibers."""
    import test_nni
    import test_tp

    ran, left_out = 0, []
    for module in (test_nni, test_tp):
        for name, fn, kwargs in _gpu_marked(module):
            if name == "test_engine_grows_beyond_one_grid_dimension":
                continue
            try:
                _call(fn, kwargs, data_dir)
                ran += 1
            except AttributeError as err:  # (a test that also drives the per-tree engine: not in the emulated library)
                if "bito_amd_engine" not in str(err):
                    raise
                left_out.append(name)
    assert ran >= 11 and set(left_out) <= {"test_proposed_nni_scores_equal_tree_likelihoods", "test_top_tree_likelihoods_equal_tree_likelihoods"}, (ran, left_out)


def test_emulated_optimizer_trace_entry_points(emulated, data_dir):
    """bito_amd_gp_set_optimizer_trace / _get_optimizer_trace: nothing to read before a trace is started (STATE), rows of
    one optimisation in order with the kinds 0, 1, 2 ..., an overflowing trace reports how many evaluations were made."""
    from bito_amd.engine import BitoAmdError

    sp, tree, dag = test_gp.hello_instance(data_dir)
    eng = test_gp._gpu_factory(sp, dag)
    eng._trace_capacity = 8
    with pytest.raises(BitoAmdError):
        eng.optimizer_trace()
    eng.set_branch_lengths(dag.branch_lengths(tree.branch_lengths))
    eng.process_operations(dag.populate_plvs())
    eng.start_optimizer_trace(4096)
    eng.process_operations(dag.branch_length_optimization())
    rows = eng.optimizer_trace()
    edges = sorted(set(int(e) for e in rows[:, 0]))
    assert len(edges) == 4 and len(rows) > 30
    for e in edges:
        mine = rows[rows[:, 0] == e]
        assert list(mine[:2, 3]) == [0.0, 1.0] and np.all(mine[2:, 3] == 2.0) and mine[0, 1] == mine[1, 1]
    eng.start_optimizer_trace(5)  # (restarts: the buffer is cleared)
    eng.process_operations(dag.branch_length_optimization())
    with pytest.raises(BitoAmdError, match="overflow"):
        eng.optimizer_trace()
    eng.stop_optimizer_trace()
    eng.process_operations(dag.branch_length_optimization())
