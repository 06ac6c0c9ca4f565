"""The engine level of the C ABI (bito_amd/csrc/engine.cpp): blocking calls cut into chunks that travel through
the device one behind the other, and engines over several device slots -- the counterpart of the reference's
Engine over thread_count FatBeagle instances (src/engine.cpp:10-31, src/fat_beagle.hpp:151-184).  On the one-GPU box
the device list names GPU 0 twice: every slot is served like a device of its own, so the code path is the N > 1 one.
Chunking of small batches is forced through the BITO_AMD_CHUNK_* variables, which an engine reads when it is created."""
import os

import numpy as np
import pytest

import bito_amd
from bito_amd import _capi, workloads

LL_ATOL, LL_RTOL, GRAD_ATOL, GRAD_RTOL = 1e-10, 2e-14, 1e-6, 1e-9


def _close(a, b, atol, rtol):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))


def _spec(w):
    return bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock)


class _Env:
    def __init__(self, **values):
        self.values = {k: str(v) for k, v in values.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.values}
        os.environ.update(self.values)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


SMALL_CHUNKS = dict(BITO_AMD_CHUNK_FIRST=16, BITO_AMD_CHUNK_GROWTH=2, BITO_AMD_CHUNK_CAP=64, BITO_AMD_CHUNK_LANES=6)


def test_device_count_must_be_positive():
    """"Thread count needs to be strictly positive." (reference src/engine.cpp:14-16); checked before any device is
    touched, so this runs on a machine without a GPU as well."""
    w = workloads.ds1_gtr_weibull4(1).subset(2)
    with pytest.raises(bito_amd.BitoAmdError, match="strictly positive"):
        bito_amd.Engine(_spec(w), w.patterns, w.weights, devices=[])
    import ctypes as C

    L = _capi.lib()
    spec = _capi.EngineSpec(0, 1, 0, -1, 0, None)  # (0 is the C ABI's default, one device)
    handle, err = C.c_void_p(), C.create_string_buffer(256)
    pat = np.ascontiguousarray(w.patterns, dtype=np.int32)
    wts = np.ascontiguousarray(w.weights)
    rc = L.bito_amd_engine_create(C.byref(spec), b"JC69", b"constant", b"none", pat.shape[0], pat.shape[1],
                                  pat.ctypes.data_as(C.POINTER(C.c_int32)), wts.ctypes.data_as(C.POINTER(C.c_double)),
                                  C.byref(handle), err, 256)
    assert rc == _capi.ERR_BAD_ARG and b"strictly positive" in err.value and not handle.value


@pytest.mark.gpu
def test_chunked_call_matches_oracle_and_single_chunk():
    from oracle import oracle

    w = workloads.ds1_gtr_weibull4(3)  # 300 trees
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=oracle.GRAD_SITE_MODEL)
    whole = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    one = whole.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
    with _Env(**SMALL_CHUNKS):
        eng = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
    for got in (one, out):
        assert _close(got["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
        assert _close(got["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
        assert _close(got["site_model"], ref["site_model"], GRAD_ATOL, GRAD_RTOL)
    # a tree's results depend on its chunk only through the order of the pattern-tile sums
    assert _close(out["log_likelihood"], one["log_likelihood"], 0.1 * LL_ATOL, 0.1 * LL_RTOL)
    assert _close(out["branch_lengths"], one["branch_lengths"], 0.1 * GRAD_ATOL, 0.1 * GRAD_RTOL)
    # the call's batch stays resident, spread over the chunks' workers: download returns the same bits, update +
    # run work on it, and the calls that need ONE block say so
    ll, grad = eng.download()
    assert np.array_equal(ll, out["log_likelihood"]) and np.array_equal(grad, out["branch_lengths"])
    eng.update(w.branch_lengths * 1.5)
    eng.run(True)
    ll2, grad2 = eng.download()
    ref2 = cpu.gradients(w.parent_ids, w.branch_lengths * 1.5, w.params)
    assert _close(ll2, ref2["log_likelihood"], LL_ATOL, LL_RTOL) and _close(grad2, ref2["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    with pytest.raises(bito_amd.BitoAmdError, match="several devices or chunks"):
        eng.time_runs(True, False, 1)
    # log-likelihoods only, and a second call on the same workers
    assert _close(eng.log_likelihoods(w.parent_ids, w.branch_lengths, w.params), ref["log_likelihood"], LL_ATOL, LL_RTOL)
    # the lean entry points take the same route
    ll3, grad3 = np.zeros(w.tree_count), np.zeros((w.tree_count, 2 * w.taxon_count - 1))
    eng.gradients_into(np.ascontiguousarray(w.parent_ids, dtype=np.int32), np.ascontiguousarray(w.branch_lengths),
                       np.ascontiguousarray(w.params), ll3, grad3)
    assert np.array_equal(ll3, out["log_likelihood"]) and np.array_equal(grad3, out["branch_lengths"])


@pytest.mark.gpu
def test_site_gradient_by_second_pass_on_a_chunked_call():
    """weibull+6 runs on walk_hbm_kernel, which does not produce the site-model gradient in the main pass: Evaluate
    then makes a second traversal per chunk.  With one host thread a call is cut into three chunks and more, and
    chunks from the third on were lent ANOTHER worker's stream for the call -- the second pass must run on the
    worker's own streams (round 3's advisor finding: the download was ordered behind the wrong stream)."""
    from oracle import oracle

    w = workloads.ds1_gtr_weibull4(2)  # 200 trees
    w.site = "weibull+6"
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=oracle.GRAD_SITE_MODEL)
    with _Env(BITO_AMD_HOST_THREADS=1, **SMALL_CHUNKS):
        eng = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    for _ in range(3):  # (a race: more than one try)
        out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
        assert eng.kernel_name().startswith("walk_hbm_kernel")
        assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
        assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
        assert _close(out["site_model"], ref["site_model"], GRAD_ATOL, GRAD_RTOL)
    # the batch is resident again afterwards, with the main pass's results
    ll, grad = eng.download()
    assert np.array_equal(ll, out["log_likelihood"]) and np.array_equal(grad, out["branch_lengths"])


@pytest.mark.gpu
def test_errors_in_a_later_chunk_name_the_callers_tree():
    w = workloads.ds1_gtr_weibull4(3)
    with _Env(**SMALL_CHUNKS):
        eng = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    pid = w.parent_ids.copy()
    pid[250, 0] = 0
    with pytest.raises(bito_amd.BitoAmdError, match="tree 250: parent id 0"):
        eng.gradients(pid, w.branch_lengths, w.params)
    par = w.params.copy()
    par[123, 0] += 0.5
    with pytest.raises(bito_amd.BitoAmdError, match=r"frequencies do not sum to 1.*\[tree 123\]"):
        eng.log_likelihoods(w.parent_ids, w.branch_lengths, par)
    with pytest.raises(bito_amd.BitoAmdError, match="no batch is resident"):
        eng.run(True)
    # the engine is still usable after an error
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert np.all(np.isfinite(out["log_likelihood"]))


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [2, 3])
def test_engine_over_several_device_slots(slots):
    """one process, one engine, several devices: trees sharded contiguously, every slot driven from the calling thread,
    results gathered into the caller's arrays"""
    from oracle import oracle

    w = workloads.ds1_gtr_weibull4(2).subset(157)  # (a count the slots do not divide)
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    single = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    one = single.gradients(w.parent_ids, w.branch_lengths, w.params)
    with _Env(BITO_AMD_CHUNK_FIRST=16, BITO_AMD_CHUNK_CAP=32, BITO_AMD_CHUNK_LANES=4):
        eng = bito_amd.Engine(_spec(w), w.patterns, w.weights, devices=[0] * slots)
    assert eng.device_count == slots
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
    assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    assert _close(out["log_likelihood"], one["log_likelihood"], 0.1 * LL_ATOL, 0.1 * LL_RTOL)
    # the summed log-likelihood of the collection (the caller's objective) is a host sum over the gathered values
    assert abs(out["log_likelihood"].sum() - ref["log_likelihood"].sum()) < 1e-8
    # resident batch interface: one block per slot
    eng.upload(w.parent_ids, w.branch_lengths, w.params)
    for scale in (1.0, 0.7):
        eng.update(w.branch_lengths * scale)
        eng.run(True)
        ll, grad = eng.download()
        want = cpu.gradients(w.parent_ids, w.branch_lengths * scale, w.params)
        assert _close(ll, want["log_likelihood"], LL_ATOL, LL_RTOL) and _close(grad, want["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    with pytest.raises(bito_amd.BitoAmdError, match="several devices or chunks"):
        eng.results_async(0)
    # fewer trees than slots: the empty slots are skipped
    few = eng.log_likelihoods(w.parent_ids[:1], w.branch_lengths[:1], w.params[:1])
    assert _close(few, ref["log_likelihood"][:1], LL_ATOL, LL_RTOL)


@pytest.mark.gpu
def test_rooted_trees_with_rates_and_rescaling_through_chunks(data_dir):
    """the other input layouts through the chunked route: rooted trees with per-branch rates (input block with a
    rates section), rescaling on (HBM-arena walk), the clock-model gradient"""
    from bito_amd import treeio
    from bito_amd.site_pattern import SitePattern
    from oracle import oracle

    tc = treeio.read_newick_file(os.path.join(data_dir, "fluA.tree"))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, "fluA.fa")), tc.taxon_names)
    T = 70
    rng = np.random.default_rng(11)
    pid = np.tile(tc.parent_id_matrix(), (T, 1))
    bl = np.tile(tc.branch_length_matrix(), (T, 1)) * rng.uniform(0.5, 1.5, (T, pid.shape[1] + 1))
    rates = rng.uniform(0.0005, 0.002, (T, pid.shape[1]))
    with _Env(**SMALL_CHUNKS):
        gpu = bito_amd.Engine(bito_amd.PhyloModelSpecification("HKY", "weibull+4", "strict"), sp.patterns, sp.weights)
    cpu = oracle.OracleEngine("HKY", "weibull+4", "strict", sp.patterns, sp.weights, 8)
    par = gpu.default_params(T)
    par[:, :4] = [0.1, 0.2, 0.3, 0.4]
    par[:, 4] = 3.0
    par[:, 5] = rng.uniform(0.3, 1.2, T)
    for rescaling in (False, True):
        out = gpu.gradients(pid, bl, par, rates=rates, rescaling=rescaling, flags=_capi.GRAD_CLOCK_MODEL | _capi.GRAD_SITE_MODEL)
        ref = cpu.gradients(pid, bl, par, rates=rates, rescaling=rescaling, flags=oracle.GRAD_CLOCK_MODEL | oracle.GRAD_SITE_MODEL)
        assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
        assert _close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
        assert _close(out["clock_model"], ref["clock_model"], GRAD_ATOL, 1e-8)
        assert _close(out["site_model"], ref["site_model"], GRAD_ATOL, 1e-8)


@pytest.mark.gpu
def test_collections_that_mix_short_and_ordinary_branches_are_evaluated_by_kind():
    """39 to 64 taxa: walk_pipe_kernel's one-image-per-branch form needs every branch of a block well above the
    rounding error of its transition matrix, and a worker decides for its whole block -- so the engine sorts a mixed
    collection into the trees that fit (walk_pipe_kernel) and the others (HBM-arena walk) and scatters the results
    back.  Against the oracle, with the positions of the odd trees scattered through the collection."""
    from oracle import oracle

    rng = np.random.default_rng(5)
    n, T = 50, 40
    w = workloads.synthetic_gtr_weibull4(n=n, P=200, tree_count=T)
    bl = np.maximum(w.branch_lengths, 1e-3)
    bl[:, -1] = 0.0
    odd = [3, 4, 17, 39]
    for t in odd:
        bl[t, rng.integers(0, 2 * n - 3)] = 1e-9
    gpu = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    out = gpu.gradients(w.parent_ids, bl, w.params, flags=_capi.GRAD_SITE_MODEL)
    ref = cpu.gradients(w.parent_ids, bl, w.params, flags=oracle.GRAD_SITE_MODEL)
    assert _close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
    assert np.allclose(out["branch_lengths"], ref["branch_lengths"], rtol=GRAD_RTOL, atol=GRAD_ATOL)
    assert _close(out["site_model"], ref["site_model"], GRAD_ATOL, 1e-8)
    assert gpu.kernel_name() in ("walk_pipe_kernel", "walk_hbm_cat_kernel")
    with pytest.raises(bito_amd.BitoAmdError, match="no batch is resident"):
        gpu.run(True)  # (a call evaluated as two collections leaves none of them resident)
    # the same trees without the short branches: one collection, walk_pipe_kernel
    bl2 = np.maximum(bl, 1e-3)
    bl2[:, -1] = 0.0
    out2 = gpu.gradients(w.parent_ids, bl2, w.params)
    assert gpu.kernel_name() == "walk_pipe_kernel"
    assert _close(out2["log_likelihood"], cpu.log_likelihoods(w.parent_ids, bl2, w.params), LL_ATOL, LL_RTOL)
    # errors name the caller's tree
    pid = w.parent_ids.copy()
    pid[30, 0] = 0
    with pytest.raises(bito_amd.BitoAmdError, match="tree 30: parent id 0"):
        gpu.gradients(pid, bl, w.params)


@pytest.mark.gpu
def test_host_threads_check_and_pack_ranges_of_a_large_chunk():
    """A blocking call's host share -- checks, packing, copy-out -- runs in ranges over helper threads for chunks of
    1024 trees and more (engine.cpp, host_pool.hpp).  Same results as the calling thread alone (to the order of the
    pattern-tile sums, which follows the chunk plan), against the oracle on a sample; and the error a serial pass over
    the trees would have met first, whichever thread finds it."""
    from oracle import oracle

    w = workloads.ds1_gtr_weibull4(31)  # 3100 trees: chunks of 1024 + 2076 with helper threads (the second chunk's inputs by a copy command)
    alone = bito_amd.Engine(_spec(w), w.patterns, w.weights, host_threads=1)
    many = bito_amd.Engine(_spec(w), w.patterns, w.weights, host_threads=6)
    a = alone.gradients(w.parent_ids, w.branch_lengths, w.params)
    b = many.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert _close(a["log_likelihood"], b["log_likelihood"], 0.1 * LL_ATOL, 0.1 * LL_RTOL)
    assert _close(a["branch_lengths"], b["branch_lengths"], 0.1 * GRAD_ATOL, 0.1 * GRAD_RTOL)
    sel = np.r_[0:6, 1020:1030, 1366:1374, 3094:3100]  # (chunk and range boundaries among them)
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
    ref = cpu.gradients(w.parent_ids[sel], w.branch_lengths[sel], w.params[sel])
    assert _close(b["log_likelihood"][sel], ref["log_likelihood"], LL_ATOL, LL_RTOL)
    assert _close(b["branch_lengths"][sel], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    # a second call on the same engines (the helpers went back to sleep in between), fresh branch lengths
    bl2 = w.branch_lengths * 1.03125
    a2 = alone.gradients(w.parent_ids, bl2, w.params)
    b2 = many.gradients(w.parent_ids, bl2, w.params)
    assert _close(a2["log_likelihood"], b2["log_likelihood"], 0.1 * LL_ATOL, 0.1 * LL_RTOL)
    assert not np.array_equal(b2["log_likelihood"], b["log_likelihood"])
    # bad trees in two different ranges of the second chunk, and a bad parameter row in between: the first one is named
    pid = w.parent_ids.copy()
    par = w.params.copy()
    pid[2900, 0] = 0
    par[2000, 0] += 0.5
    pid[1500, 3] = 0
    for eng in (alone, many):
        with pytest.raises(bito_amd.BitoAmdError, match="tree 1500: parent id 0 of node 3"):
            eng.gradients(pid, w.branch_lengths, par)
    pid[1500] = w.parent_ids[1500]
    for eng in (alone, many):
        with pytest.raises(bito_amd.BitoAmdError, match=r"frequencies do not sum to 1.*\[tree 2000\]"):
            eng.gradients(pid, w.branch_lengths, par)
    # and the engines still work after a failed call
    c = many.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert np.array_equal(c["log_likelihood"], b["log_likelihood"])


@pytest.mark.gpu
def test_whole_tree_units_write_the_final_sums_the_final_sums_kernel_would():
    """walk_pipe_kernel's whole-tree units store their tree's results themselves (a blocking call's results cross
    PCIe during the traversal); the final-sums kernel would have added zeros to the same numbers: bit for bit the
    results of an engine that sends everything through that kernel (BITO_AMD_PIPE_DIRECT=0), blocking calls and passes
    over a resident batch alike, rooted trees (one zeroed entry) and unrooted ones (two)."""
    w = workloads.ds1_gtr_weibull4(30)  # 3000 trees: whole-tree units for most, runs of tiles for the rest
    direct = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    with _Env(BITO_AMD_PIPE_DIRECT=0):
        plain = bito_amd.Engine(_spec(w), w.patterns, w.weights)
    a = direct.gradients(w.parent_ids, w.branch_lengths, w.params)
    b = plain.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert direct.kernel_name() == plain.kernel_name() == "walk_pipe_kernel"
    assert np.array_equal(a["log_likelihood"], b["log_likelihood"])
    assert np.array_equal(a["branch_lengths"], b["branch_lengths"])
    assert np.all(a["branch_lengths"][:, -2:] == 0.0)  # (unrooted: the root's and the fixed node's entries)
    for eng in (direct, plain):
        eng.upload(w.parent_ids, w.branch_lengths * 1.0625, w.params)
        eng.run(True)
    la, ga = direct.download()
    lb, gb = plain.download()
    assert np.array_equal(la, lb) and np.array_equal(ga, gb)
    assert not np.array_equal(la, a["log_likelihood"])
    ll_only = direct.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
    assert np.array_equal(ll_only, plain.log_likelihoods(w.parent_ids, w.branch_lengths, w.params))
