"""The per-tree engine -- engine.cpp, worker.cpp and the kernels (kernels.hip: set-up, transition matrices,
walk_hbm_kernel, final sums; walk_hbm_cat.hip; time_tree.hip as plain HIP C++; walk_pipe.hip with its gfx950 assembly
interpreted instruction by instruction, tests/hip_emu/gfx950_asm.hpp) -- executed on the CPU through the stand-in HIP
runtime of tests/hip_emu, against the CPU checker.  AUTO routes as in the product: walk_pipe_kernel for up to 64 taxa
and four rate categories without rescaling (its own tests: tests/test_pipe_emulated.py), the HBM-arena walks for
rescaling, five to eight rate categories and larger trees; the pinned-only LDS walks (walk_lds.hip, walk_tree.hip) are in
the build as well -- every kernel of the product is.
What this holds in a round without GPU access: the host side of a blocking call (chunks, device slots each on its own
thread, the shared helper threads that pack large chunks, the site-model gradient's second pass from every slot's thread
-- round 5's changes), and the logic of the kernels.  Small synthetic shapes: a tree costs a third of a second as fibers.
Each test runs in a process of its own (the library under test is chosen when bito_amd is first imported).  Test
infrastructure: the product has no CPU path."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU = os.path.join(HERE, "hip_emu", "_build", "libbito_amd_emu.so")

PRELUDE = '''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import bito_amd
from bito_amd import _capi, workloads
from oracle import oracle
LL_ATOL, LL_RTOL, GRAD_ATOL, GRAD_RTOL = 1e-10, 2e-14, 1e-6, 1e-9
def close(a, b, atol, rtol):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))
def spec(w):
    return bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock)
def small(n, P, T, site="weibull+4"):
    w = workloads.synthetic_gtr_weibull4(n, P, tree_count=T)
    w.site = site
    w.rescaling = False
    if site == "constant":
        w.params = np.ascontiguousarray(w.params[:, :10])
    return w
assert "cpu-emulation" in bito_amd.version()
'''


AS_PRODUCT = os.path.join(HERE, "hip_emu", "_build", "as_product")  # holds libbito_amd.so -> the emulated library


@pytest.fixture(scope="module")
def emulated():
    built = subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hip_emu")], capture_output=True, text=True)
    assert built.returncode == 0, built.stdout + built.stderr
    os.makedirs(AS_PRODUCT, exist_ok=True)
    link = os.path.join(AS_PRODUCT, "libbito_amd.so")
    if not os.path.islink(link):
        os.symlink(os.path.join("..", "libbito_amd_emu.so"), link)


def run_gpu_tests_emulated(args, timeout=900):
    """`pytest -m gpu <args>` as it is, in a process whose bito_amd loads the emulated library (and whose C++ client
    programs, which look libbito_amd.so up through their run path, find it first on LD_LIBRARY_PATH)"""
    env = dict(os.environ, BITO_AMD_LIB=EMU, LD_LIBRARY_PATH=AS_PRODUCT + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    done = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", *args],
                          capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-2000:]
    return done.stdout


def run(body, timeout=600, **env):
    code = PRELUDE.format(root=ROOT, here=HERE) + body
    full = dict(os.environ, BITO_AMD_LIB=EMU, **{k: str(v) for k, v in env.items()})
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=full)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    return done.stdout


def test_emulated_kernels_against_the_checker(emulated):
    """walk_hbm_cat_kernel (one wave per rate category: with and without rescaling, log-likelihood only and with the
    gradient, one and four categories) and walk_hbm_kernel (six categories) on 9- and 23-taxon trees, every result against
    the CPU checker at the bars of tests/test_gpu_parity.py; unrooted and rooted trees."""
    run('''
for n, P, T, site in ((9, 70, 5, "weibull+4"), (23, 40, 3, "weibull+4"), (9, 70, 4, "weibull+6"), (12, 50, 4, "constant")):
    w = small(n, P, T, site)
    gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
    for rescaling in (False, True):
        out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
        ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
        hbm = rescaling or site == "weibull+6"  # (walk_pipe_kernel takes the rest: interpreted, tests/test_pipe_emulated.py)
        assert gpu.kernel_name().startswith("walk_hbm" if hbm else "walk_pipe"), gpu.kernel_name()
        assert close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL), (n, site, rescaling)
        assert close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL), (n, site, rescaling)
        ll = gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
        assert close(ll, ref["log_likelihood"], LL_ATOL, LL_RTOL)
    print(n, site, gpu.kernel_name())
''')


def test_emulated_chunks_slots_and_shared_helpers(emulated):
    """A blocking call cut into chunks, on one slot and on three (GPU 0 named three times: every slot on an issuing thread
    of its own), the slots' larger chunks packed in ranges by the shared helper threads (round 5; BITO_AMD_HOST_MIN_TREES
    brings the threshold down to this test's size): the same results as an engine that never chunks, held to a tenth of
    the tolerances, and to the CPU checker; errors name the caller's tree from whichever thread met them; the batch is
    resident afterwards."""
    run('''
w = small(6, 24, 150)
plain = bito_amd.Engine(spec(w), w.patterns, w.weights, host_threads=1)
want = plain.gradients(w.parent_ids, w.branch_lengths, w.params)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
assert close(want["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
assert close(want["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
for slots, threads in ((1, 4), (3, 4), (3, 1), (2, 6)):
    eng = bito_amd.Engine(spec(w), w.patterns, w.weights, devices=[0] * slots, host_threads=threads)
    assert eng.device_count == slots
    for _ in range(3):
        out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
        assert close(out["log_likelihood"], want["log_likelihood"], 0.1 * LL_ATOL, 0.1 * LL_RTOL), (slots, threads)
        assert close(out["branch_lengths"], want["branch_lengths"], 0.1 * GRAD_ATOL, 0.1 * GRAD_RTOL), (slots, threads)
    pid = w.parent_ids.copy()
    pid[131, 0] = 0
    try:
        eng.gradients(pid, w.branch_lengths, w.params)
        raise SystemExit("a bad tree went through")
    except bito_amd.BitoAmdError as err:
        assert "tree 131: parent id 0" in str(err), str(err)
    again = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert np.array_equal(again["log_likelihood"], out["log_likelihood"])
    ll, grad = eng.download()
    assert np.array_equal(ll, again["log_likelihood"]) and np.array_equal(grad, again["branch_lengths"])
    print(slots, threads, "ok")
''', BITO_AMD_CHUNK_FIRST=8, BITO_AMD_CHUNK_GROWTH=2, BITO_AMD_CHUNK_CAP=40, BITO_AMD_CHUNK_LANES=5, BITO_AMD_HOST_MIN_TREES=8)


def test_emulated_site_gradient_second_pass_per_slot(emulated):
    """weibull+6 with the site-model gradient (walk_hbm_kernel: a second traversal per block) on one slot and on three,
    several chunks per slot: the second pass runs from every slot's own thread (round 5) and the three gradients are the
    checker's."""
    run('''
w = small(7, 30, 41, "weibull+6")
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=oracle.GRAD_SITE_MODEL)
for slots in (1, 3):
    eng = bito_amd.Engine(spec(w), w.patterns, w.weights, devices=[0] * slots)
    for _ in range(2):
        out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
        assert eng.kernel_name().startswith("walk_hbm_kernel")
        assert close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
        assert close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
        assert close(out["site_model"], ref["site_model"], GRAD_ATOL, GRAD_RTOL)
    print(slots, "ok")
''', BITO_AMD_CHUNK_FIRST=4, BITO_AMD_CHUNK_GROWTH=2, BITO_AMD_CHUNK_CAP=16, BITO_AMD_CHUNK_LANES=4)


def test_emulated_codon_model(emulated):
    """BASELINE config 5's path under emulation: two fluA trees under GY94 (61 states), each with its own (kappa, omega)
    row -- model set-up, the 64 x 64 Jacobi eigensolver, transition-matrix images and the walk on
    v_mfma_f64_16x16x4 (emulated with the lane layout and the fma-chain rounding measured on the device,
    profiles/r1_mfma16_probe.json), images moved into LDS by global_load_lds -- against the general-state CPU checker at
    the bars of tests/test_gpu_general.py; then the same rows again (the models of the call before stand) and other rows."""
    run('''
from oracle import gs
w = workloads.flua_codon(2)
w.params = workloads.codon_rows(2, 2)
gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
cpu = gs.GsOracleEngine(w.substitution, w.site, w.patterns, w.weights, 2)
for params in (w.params, w.params, workloads.other_bits(w.params, 4)):
    out = gpu.gradients(w.parent_ids, w.branch_lengths, params)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, params)
    assert gpu.kernel_name() == "gs_walk_kernel"
    assert np.abs(out["log_likelihood"] - ref["log_likelihood"]).max() < 1e-10
    assert np.abs(out["branch_lengths"] - ref["branch_lengths"]).max() < 1e-6 + 1e-9 * np.abs(ref["branch_lengths"]).max()
assert not np.array_equal(w.params[0], w.params[1])
print("codon ok", out["log_likelihood"])
''', timeout=900)


def test_emulated_time_tree_gpu_tests_as_they_are(emulated):
    """tests/test_time_tree.py -m gpu, unchanged, under emulation: the height-ratio transforms, log-det-Jacobian and
    rooted gradients of time_tree.hip on one, two and three device slots, fluA's rooted goldens of the reference among
    them (src/rooted_sbn_instance.hpp:60-330)."""
    out = run_gpu_tests_emulated(["tests/test_time_tree.py"])
    assert "8 passed" in out, out[-500:]


def test_emulated_cpp_clients_as_they_are(emulated):
    """tests/test_cabi_client.py -m gpu, unchanged, under emulation: the plain C++ programs of seam 2 (examples/engine_amd.hpp,
    one and two device slots, bit for bit the ctypes route) and seam 1 (FatBeagle's call sequence over the 17 BEAGLE symbols:
    the DS1 JC69 pybeagle log-likelihood and physher gradient goldens, src/unrooted_sbn_instance.hpp:245-348)."""
    out = run_gpu_tests_emulated(["tests/test_cabi_client.py"])
    assert "4 passed" in out, out[-500:]


def test_emulated_reference_goldens_and_seam_1_as_they_are(emulated):
    """Tests of tests/test_gpu_parity.py -m gpu, unchanged, under emulation (AUTO routes to the HBM-arena walks there): the
    reference's DS1 JC69 goldens -- 17-digit pybeagle log-likelihoods, the physher gradient -- with and without rescaling,
    the DS1 JC69 + weibull+4 goldens, JC69 == GTR(equal) on the 100 DS1 topologies (src/unrooted_sbn_instance.hpp:245-348,
    test/test_bito.py:97-122), fluA rooted with rates, the hello instance API, one to eight rate categories, pattern counts
    around tile edges, the 17-symbol BEAGLE shim in FatBeagle's call sequence, the edge cases."""
    out = run_gpu_tests_emulated(["tests/test_gpu_parity.py", "-n", "4", "-k",
                                  "ds1_jc69_goldens or ds1_jc69_weibull_goldens or config2_ds1 or flua_rooted_with_rates or "
                                  "beagle_shim or category_counts or edge_cases or hello_jc69 or one_rate_category_underflows or "
                                  "pattern_counts_around or resident_update_and_time_tree or hbm_arena_walk_in_chunks"])
    assert "21 passed" in out, out[-600:]


def test_emulated_codon_model_setup_is_bitwise_the_checkers(emulated):
    """tests/test_gpu_general.py::test_codon_model_setup_is_bitwise_the_oracles, unchanged, under emulation: rate matrix,
    F1x4 frequencies and the 64 x 64 round-robin Jacobi eigensystem of gs_model_kernel / gs_eigen_kernel equal the CPU
    checker's bit for bit (round 5 folded the eigensolver's zeroing step into its row update: three barriers per round)."""
    out = run_gpu_tests_emulated(["tests/test_gpu_general.py", "-k", "bitwise and not 64"], timeout=1800)
    # (one row for all trees; 16 trees with 16 different rows; the 64-row case takes three minutes as fibers:
    # profiles/r6_cpu/codon_setup_bitwise_64_rows_emulated.log)
    assert "2 passed" in out, out[-500:]


def test_emulated_hbm_walks_do_not_depend_on_the_batch(emulated):
    """A tree's results from the HBM-arena walks -- config 4's path: rescaling, five to eight rate categories, more than 64
    taxa -- are the same bits alone, inside a batch, and in a call cut into chunks over two device slots: their pattern
    tiles are 64 wide whatever the batch and reduce_tiles_kernel adds them in tile order.  (The dependence DESIGN.md
    section 3 documents is walk_pipe_kernel's: its tiles follow the launch plan of the batch.)  As the reference, whose
    trees do not know of one another (src/fat_beagle.hpp:173-181)."""
    run('''
for site in ("weibull+4", "weibull+6"):
    w = small(14, 300, 9, site)
    for rescaling in (False, True):
        eng = bito_amd.Engine(spec(w), w.patterns, w.weights)
        whole = eng.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
        for t in (0, 4, 8):
            one = eng.gradients(w.parent_ids[t:t + 1], w.branch_lengths[t:t + 1], w.params[t:t + 1], rescaling=rescaling)
            assert one["log_likelihood"][0] == whole["log_likelihood"][t]
            assert np.array_equal(one["branch_lengths"][0], whole["branch_lengths"][t])
        os.environ.update(BITO_AMD_CHUNK_FIRST="2", BITO_AMD_CHUNK_CAP="4", BITO_AMD_CHUNK_GROWTH="2")
        chunked = bito_amd.Engine(spec(w), w.patterns, w.weights, devices=[0, 0])
        for k in ("BITO_AMD_CHUNK_FIRST", "BITO_AMD_CHUNK_CAP", "BITO_AMD_CHUNK_GROWTH"):
            os.environ.pop(k)
        out = chunked.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
        assert np.array_equal(out["log_likelihood"], whole["log_likelihood"])
        assert np.array_equal(out["branch_lengths"], whole["branch_lengths"])
        print(site, rescaling, eng.kernel_name(), "ok")
''')


def test_emulated_pinned_lds_walks_against_the_checker(emulated):
    """walk_lds_kernel (KERNEL_LDS: the first LDS-resident walk, compiled C++ with scalar descriptor prefetches) and
    walk_tree_kernel (KERNEL_LDS_TREE: whole-tree workgroups, cross-lane sums by DPP / permlane swaps / ds_bpermute) --
    kernels AUTO never picks, kept as pinned alternatives -- on the CPU: one, two and four rate categories, rooted and
    unrooted, log-likelihood only and with the gradient; walk_lds_kernel also with the site-model gradient fused into its
    pre-order pass.  With these every kernel of the product runs in the emulated build."""
    run('''
from test_gpu_parity import _random_rooted_parent_ids
for kernel, name in ((_capi.KERNEL_LDS, "walk_lds_kernel"), (_capi.KERNEL_LDS_TREE, "walk_tree_kernel")):
    for n, P, T, site, rooted in ((5, 16, 2, "weibull+4", False), (9, 70, 3, "weibull+4", True), (12, 40, 2, "weibull+2", False),
                                  (23, 65, 2, "constant", True), (27, 130, 1, "weibull+4", False)):
        w = small(n, P, T, site)
        if rooted:
            rng = np.random.default_rng(n)
            w.parent_ids = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)]).astype(np.int32)
            w.branch_lengths = rng.uniform(0.01, 0.4, (T, 2 * n - 1))
            w.branch_lengths[:, -1] = 0.0
        gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
        gpu.set_kernel(kernel)
        cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
        ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
        ll = gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
        assert gpu.kernel_name() == name, gpu.kernel_name()
        assert close(ll, ref["log_likelihood"], LL_ATOL, LL_RTOL), (name, n, site)
        out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
        assert close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL), (name, n, site)
        assert close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL), (name, n, site)
        print(name, n, P, site)
w = small(9, 70, 3)
gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
gpu.set_kernel(_capi.KERNEL_LDS)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
assert gpu.kernel_name() == "walk_lds_kernel"
assert close(out["site_model"], ref["site_model"], 1e-6, 1e-8) and close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
''')


def test_emulated_pitchfork_counts_do_not_depend_on_the_ranges_of_a_call(emulated):
    """A blocking call checks its trees in ranges (one per helper thread); a range counts the pitchforks walk_pipe_kernel
    folds only when its own trees, by their cherries, would not fit beside four pattern groups.  27 taxa, where seven
    unstored nodes fit: 47 trees with seven cherries and two foldable pitchforks, one tree with six cherries and three
    (nine unstored nodes each way).  Cut into ranges, only that tree's range counts pitchforks on its own;
    WorkerStageEnd counts the rest, so that the plan -- four groups for the whole batch and LDS vectors for nine unstored
    nodes, not for seven -- and the results are those of the call checked in one range."""
    run('''
from bito_amd import treeio
rng = np.random.default_rng(11)
n, P, T = 27, 40, 48
w = small(n, P, T)
cherries = lambda pid: sum(1 for v in range(n, 2 * n - 2) if sum(1 for c in range(n) if pid[c] == v) == 2)
def ladder_tree(pitchforks, plain, names):
    # a node over two cherries (stored), then `pitchforks` groups ((a,b),c) and `plain` groups (a,b) joined to the growing
    # ladder one at a time (a pitchfork folds when its sibling is a tip or a stored node), then the remaining tips one by
    # one up to a trifurcating root
    names = list(names)
    text = "((%s,%s),(%s,%s))" % (names.pop(), names.pop(), names.pop(), names.pop())
    for _ in range(pitchforks):
        text = "(%s,((%s,%s),%s))" % (text, names.pop(), names.pop(), names.pop())
    for _ in range(plain):
        text = "(%s,(%s,%s))" % (text, names.pop(), names.pop())
    last = names.pop()
    for name in names:
        text = "(%s,%s)" % (text, name)
    return "(%s,%s);" % (text[1:-1], last)
names = ["t%02d" % i for i in range(n)]
newicks = [ladder_tree(2, 3, rng.permutation(names)) for _ in range(T)]
newicks[31] = ladder_tree(3, 1, rng.permutation(names))
tc = treeio.parse_newick_strings(newicks)
w.parent_ids = np.ascontiguousarray(tc.parent_id_matrix(), dtype=np.int32)
assert w.parent_ids.shape == (T, 2 * n - 3)
import ctypes as C
counts = {}
for fold in (0, 1):
    counts[fold] = np.zeros(T, dtype=np.int32)
    _capi.lib().bito_amd_count_unstored_nodes(n, T, 0, 2 * n - 2, w.parent_ids.ctypes.data_as(C.POINTER(C.c_int32)), fold,
                                              counts[fold].ctypes.data_as(C.POINTER(C.c_int32)))
assert counts[0].tolist() == [7] * 31 + [6] + [7] * 16 and counts[1].tolist() == [9] * T, (counts[0], counts[1])
order = [tc.taxon_names.index(name) for name in names]  # (rows of the alignment in the collection's taxon order)
w.patterns = np.ascontiguousarray(w.patterns[np.argsort(order)])
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
forms, outs = [], []
for threads in (1, 4, 6):
    eng = bito_amd.Engine(spec(w), w.patterns, w.weights, host_threads=threads)
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert eng.kernel_name() == "walk_pipe_kernel"
    assert close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
    assert close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
    forms.append(eng.kernel_form())
    outs.append(out)
assert forms[0] == forms[1] == forms[2] and "4 pattern groups, 16 vectors per wave" in forms[0], forms
for o in outs[1:]:
    assert np.array_equal(o["log_likelihood"], outs[0]["log_likelihood"]) and np.array_equal(o["branch_lengths"], outs[0]["branch_lengths"])
print(forms[0])
''', BITO_AMD_HOST_MIN_TREES=8)


def test_emulated_codon_walk_issues_the_matrix_instructions_the_device_counted(emulated):
    """Config 5's executed-instruction count, held the way walk_pipe_kernel's is (tests/test_pipe_emulated.py): an MI355X
    counted SQ_INSTS_MFMA = 843 055 104 v_mfma_f64_16x16x4 per launch of gs_walk_kernel over 4096 fluA codon trees
    (profiles/r3_v9_codon_issue_pmc.json; every tree the same topology) = 205 824 per tree, the number
    profiles/executed.json gives bench.py.  The emulated kernel, one tree, issues exactly that many through the matrix
    builtin (counted per kernel by the stand-in runtime: the image kernel's 34 816 = 136 branches x 256 are not the
    walk's)."""
    import json

    device = json.load(open(os.path.join(ROOT, "profiles", "r3_v9_codon_issue_pmc.json")))
    assert device["SQ_INSTS_MFMA"] == 843055104.0
    body = PRELUDE.format(root=ROOT, here=HERE) + '''
w = workloads.flua_codon(1)
eng = bito_amd.Engine(spec(w), w.patterns, w.weights)
out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
assert eng.kernel_name() == "gs_walk_kernel" and np.isfinite(out["log_likelihood"]).all()
'''
    done = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, timeout=600,
                          env=dict(os.environ, BITO_AMD_LIB=EMU, HIP_EMU_ASM_COUNT="1"))
    assert done.returncode == 0, done.stderr[-2000:]
    counts = {}
    for line in done.stderr.splitlines():
        if line.startswith("{"):
            counts.update(json.loads(line))
    per_kernel = counts["builtin_mfma_by_kernel"]
    walk = sum(v for k, v in per_kernel.items() if k != "gs_matrices_kernel")  # (the walk is launched through a pointer named `kern`)
    assert per_kernel["gs_matrices_kernel"] == 136 * 256
    assert walk * 4096 == int(device["SQ_INSTS_MFMA"]), (per_kernel, device["SQ_INSTS_MFMA"])
    row = [r for r in json.load(open(os.path.join(ROOT, "profiles", "executed.json"))) if r["kernel"] == "gs_walk_kernel"][0]
    assert row["matrix_instructions_per_tree"] == walk


def test_emulated_codon_image_kernel_forms_give_the_same_bits(emulated, tmp_path):
    """gs_matrices_kernel's sparse dP terms (round 6: four threads per column of dP^T with the column's list of Q in
    registers, lists padded with zero terms; the exponentials of a workgroup's eight jobs taken at once) against round 3's
    loop (lists in LDS, read per term; -DGS_DP_COLUMN=0): the same terms in the same order, so the SAME BITS in every
    log-likelihood, branch gradient and site-model gradient -- a fluA codon tree, constant rates and weibull+3 with the
    site-model gradient (whose second pass multiplies the lists by a negative d rate / d shape: the padding's -0)."""
    emu_dir = os.path.join(HERE, "hip_emu")
    flags = [f for f in subprocess.run(["make", "-s", "-C", emu_dir, "print-flags"], capture_output=True, text=True).stdout.split()]
    assert flags, "tests/hip_emu/Makefile: print-flags"
    obj, lib = str(tmp_path / "gs_kernels_round3.o"), str(tmp_path / "libbito_amd_emu_round3.so")
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-DGS_DP_COLUMN=0", *flags, "-Wno-psabi", "-c", "_build/src/gs_kernels.hip", "-o", obj],
                   check=True, cwd=emu_dir, capture_output=True)
    others = [os.path.join(emu_dir, "_build", f) for f in os.listdir(os.path.join(emu_dir, "_build"))
              if f.endswith(".o") and not f.startswith("gs_kernels")]
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-shared", "-pthread", "-o", lib, obj, *others], check=True)
    body = '''
site = {site!r}
w = workloads.flua_codon(1, site)
gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL if site != "constant" else 0)
assert gpu.kernel_name() == "gs_walk_kernel"
np.save({path!r}, np.concatenate([np.ravel(out[k]) for k in ("log_likelihood", "branch_lengths", "site_model") if k in out and out[k] is not None]))
'''
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    jobs = []
    for site in ("constant", "weibull+3"):
        for tag, library in (("now", EMU), ("round3", lib)):
            path = str(tmp_path / f"{site}_{tag}.npy")
            code = PRELUDE.format(root=ROOT, here=HERE) + body.format(site=site, path=path)
            jobs.append((site, tag, path, [sys.executable, "-c", code], dict(os.environ, BITO_AMD_LIB=library)))
    with ThreadPoolExecutor(4) as pool:
        done = list(pool.map(lambda j: subprocess.run(j[3], env=j[4], capture_output=True, text=True, timeout=1500), jobs))
    for j, d in zip(jobs, done):
        assert d.returncode == 0, d.stdout[-2000:] + d.stderr[-2000:]
    for site in ("constant", "weibull+3"):
        now, old = (np.load(str(tmp_path / f"{site}_{tag}.npy")) for tag in ("now", "round3"))
        assert now.shape == old.shape and now.size >= 135 and np.all(np.isfinite(now))
        assert np.array_equal(now, old), (site, float(np.abs(now - old).max()))


@pytest.mark.parametrize("fold", [2, 1, 0])
def test_emulated_four_tip_subtrees_rebuilt_where_they_are_used(emulated, fold):
    """walk_hbm_cat_kernel with BITO_AMD_HBM_FOLD = 2 (round 6: caterpillars -- a tip and a pitchfork under one node -- and
    twin cherries have no step and no cell: their partials are rebuilt from their four tips' matrix rows in the parent's
    step, their six edges take their derivatives there), 1 (round 4: pitchforks only) and 0, on seeded random trees of 9
    to 41 taxa that hold every case -- a four-tip child beside a tip, a cherry, a pitchfork, a stored node, another
    four-tip child, and right under the root -- against the CPU checker: log-likelihood and gradient, with and without
    rescaling, one and four rate categories; and the three levels agree with one another."""
    out = run('''
import sys
sys.path.insert(0, os.path.join({root!r}, "scripts"))
import sim_hbm_traffic as sim
seen = {{"cat": 0, "twin": 0, "both four-tip": 0, "four-tip under the root": 0, "beside": set()}}
for n, P, T, site in ((9, 70, 6, "weibull+4"), (17, 64, 6, "weibull+4"), (41, 70, 4, "weibull+4"), (24, 40, 5, "constant")):
    w = small(n, P, T, site)
    for t in range(T):
        ch = sim.children_of(np.asarray(w.parent_ids[t]), n)
        cherry, fork, cat, twin = sim.shapes(ch, n)
        seen["cat"] += len(cat)
        seen["twin"] += len(twin)
        four = cat | twin
        for v in range(n, n + len(ch)):
            a, b = ch[v - n]
            if a in four and b in four:
                seen["both four-tip"] += 1
            for x, y in ((a, b), (b, a)):
                if x in four:
                    seen["beside"].add("tip" if y < n else "cherry" if y in cherry else "fork" if y in fork else "four" if y in four else "stored")
                    if v == n + len(ch) - 1:
                        seen["four-tip under the root"] += 1
    gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
    gpu.set_kernel(1)  # the HBM-arena walk, whatever AUTO would take
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
    for rescaling in (False, True):
        res = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
        ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
        assert gpu.kernel_name().startswith("walk_hbm_cat"), gpu.kernel_name()
        assert close(res["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL), (n, site, rescaling)
        assert close(res["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL), (n, site, rescaling)
        ll = gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params, rescaling=rescaling)
        assert close(ll, ref["log_likelihood"], LL_ATOL, LL_RTOL)
        np.save(os.path.join({tmp!r}, "fold%d_%d_%s_%d.npy" % ({fold}, n, site, rescaling)),
                np.concatenate([res["log_likelihood"].ravel(), res["branch_lengths"].ravel()]))
assert seen["cat"] >= 10 and seen["twin"] >= 5 and seen["both four-tip"] >= 1 and seen["four-tip under the root"] >= 1, seen
assert {{"tip", "cherry", "fork", "stored", "four"}} <= seen["beside"], seen
print("shapes", seen)
'''.format(root=ROOT, tmp=FOLD_DIR, fold=fold), timeout=1500, BITO_AMD_HBM_FOLD=fold)
    assert "shapes" in out
    # the levels agree with one another (a tenth of the bars: the same arithmetic in another grouping of the steps)
    import glob

    import numpy as np

    mine = sorted(glob.glob(os.path.join(FOLD_DIR, f"fold{fold}_*.npy")))
    assert len(mine) == 8
    for path in mine:
        other = path.replace(f"fold{fold}_", "fold2_")
        if fold != 2 and os.path.exists(other):
            a, b = np.load(path), np.load(other)
            assert np.all(np.abs(a - b) <= 1e-7 + 1e-10 * np.abs(b)), (path, float(np.abs(a - b).max()))


FOLD_DIR = os.path.join(HERE, "hip_emu", "_build", "fold_levels")
os.makedirs(FOLD_DIR, exist_ok=True)


def test_emulated_round6_gpu_tests_as_they_are(emulated):
    """tests/test_round6.py -m gpu, unchanged, under emulation: small calls whose set-up, step tables and images are one
    launch give the three-launch route's bits, and a tree's last run of tiles forms its final sums as the final-sums kernel
    would; the HBM-arena walk with four-tip subtrees folded against the checker (41
    taxa here) and its three fold levels against one another; sixteen waves per optimiser workgroup on the checker's
    iterates."""
    out = run_gpu_tests_emulated(["tests/test_round6.py", "-k", "small_calls or last_unit or fold_levels or (against_the_checker and 41) or (wave_counts and 16)"],
                                 timeout=2400)
    assert "5 passed" in out, out[-600:]  # (the 65- and 100-taxon cases and one wave per optimiser: profiles/r6_cpu/)
