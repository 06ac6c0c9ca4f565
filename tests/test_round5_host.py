"""Round 5's changes to the host side of a multi-slot engine (bito_amd/csrc/engine.cpp), written in a round without GPU
access and therefore run LAST by tests/conftest.py: (a) a slot's issuing thread packs its large chunks in ranges that the
engine's shared helper threads claim beside it (host_pool.hpp, SharedPool; before, only a one-slot engine used helpers),
and large result blocks are copied out the same way; (b) the site-model gradient's second traversal (kernels that do not
produce it in the main pass) runs from every slot's own thread at once instead of slot after slot from the calling
thread.  Reference: N FatBeagle instances each on a thread of its own, src/engine.cpp:10-31, src/task_processor.hpp:43-140."""
import numpy as np
import pytest

import bito_amd
from bito_amd import _capi, workloads
from test_engine_chunks import GRAD_ATOL, GRAD_RTOL, LL_ATOL, LL_RTOL, _close, _Env, _spec


@pytest.mark.gpu
@pytest.mark.parametrize("slots,host_threads", [(3, 0), (2, 5), (3, 1)])
def test_large_chunks_of_several_slots_are_packed_by_shared_helpers(slots, host_threads):
    """6000 DS1 trees over two or three device slots (GPU 0 named several times): every slot's block is a chunk of 1024
    trees and one of about 1000-2000, the latter packed in ranges of 512 trees by the slot's thread and the shared
    helpers; results are those of a one-slot engine (a tenth of the tolerances: a tree's chunk only changes the order
    of its pattern-tile sums) and a sample is held to the CPU checker.  host_threads 1 = no helpers: the old path."""
    from oracle import oracle

    w = workloads.ds1_gtr_weibull4(60)
    single = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    one = single.gradients(w.parent_ids, w.branch_lengths, w.params)
    kwargs = dict(devices=[0] * slots)
    if host_threads:
        kwargs["host_threads"] = host_threads
    eng = bito_amd.Engine(_spec(w), w.patterns, w.weights, **kwargs)
    for _ in range(3):  # (threads: more than one try)
        out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
        assert _close(out["log_likelihood"], one["log_likelihood"], 0.1 * LL_ATOL, 0.1 * LL_RTOL)
        assert _close(out["branch_lengths"], one["branch_lengths"], 0.1 * GRAD_ATOL, 0.1 * GRAD_RTOL)
    sel = np.r_[0:4, 1022:1026, 1998:2002, 2999:3003, 5996:6000]
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    ref = cpu.gradients(w.parent_ids[sel], w.branch_lengths[sel], w.params[sel])
    assert _close(out["log_likelihood"][sel], ref["log_likelihood"], LL_ATOL, LL_RTOL)
    assert _close(out["branch_lengths"][sel], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    # errors still name the caller's tree, whichever thread met them
    pid = w.parent_ids.copy()
    pid[4321, 0] = 0
    with pytest.raises(bito_amd.BitoAmdError, match="tree 4321: parent id 0"):
        eng.gradients(pid, w.branch_lengths, w.params)
    again = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert np.array_equal(again["log_likelihood"], out["log_likelihood"])


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [1, 3])
def test_site_gradient_second_pass_from_every_slots_own_thread(slots):
    """weibull+6 (walk_hbm_kernel: no site-model gradient in the main pass) with GRAD_SITE_MODEL on an engine over three
    device slots: every slot's chunks get their second traversal from the slot's own thread.  Against the CPU checker,
    several tries, small chunks (so that a slot has several blocks and some were lent another worker's streams)."""
    from oracle import oracle

    w = workloads.ds1_gtr_weibull4(3).subset(271)
    w.site = "weibull+6"
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=oracle.GRAD_SITE_MODEL)
    with _Env(BITO_AMD_CHUNK_FIRST=16, BITO_AMD_CHUNK_GROWTH=2, BITO_AMD_CHUNK_CAP=64, BITO_AMD_CHUNK_LANES=4):
        eng = bito_amd.Engine(_spec(w), w.patterns, w.weights, devices=[0] * slots)
    for _ in range(3):
        out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
        assert eng.kernel_name().startswith("walk_hbm_kernel")
        assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
        assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
        assert _close(out["site_model"], ref["site_model"], GRAD_ATOL, GRAD_RTOL)
