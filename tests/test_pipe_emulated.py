"""walk_pipe_kernel -- the headline kernel: C++ around three statements of generated gfx950 assembly (the image loader and
the post-order / pre-order loops of bito_amd/csrc/walk_pipe_gen.inc) -- executed on the CPU: the C++ as fibers of the
stand-in runtime (tests/hip_emu/hip/hip_runtime.h), the assembly INTERPRETED instruction by instruction
(tests/hip_emu/gfx950_asm.hpp: VGPR/AGPR/SGPR files that persist across statements, EXEC, the VGPR index mode, s_movrels,
computed jumps, LDS, global_load_lds, v_mfma_f64_4x4x4 in the lane layout the kernel was written to), against the CPU
checker at the bars of tests/test_gpu_parity.py.  Every layout the launch planner can pick is walked: four, two and one
pattern groups per wave, the "many" form (33-36 taxa beside four groups), exact and reversible images (38 / 39 taxa),
the wide form (49-64 taxa), the two-wave form and the two-class launch, tile runs and whole-tree units, the site-model
gradient's second traversal.  What this holds in a round without GPU access: the text of the generated assembly and the
C++ around it compute the reference's numbers, and its s_waitcnt discipline holds by the ISA's guarantees alone (every
register read is checked against the loads in flight: 0 hazards over every layout); what it cannot show: timing, and
the wait states between vector and matrix instructions that the hardware interlocks or the author counted by hand.  Each test runs in a process of its own.  Test
infrastructure: the product has no CPU path."""
import os

import pytest

from test_engine_emulated import AS_PRODUCT, EMU, HERE, ROOT, emulated, run_gpu_tests_emulated  # noqa: F401
from test_engine_emulated import run as run_emulated


def run(body, timeout=600, **env):
    """... with the interpreter's hazard check fatal: a register read before the s_waitcnt that delivers it, an LDS range
    read while a global_load_lds may still be filling it, M0 used without its wait state (gfx950_asm.hpp: HazardLog) --
    what one-instruction-at-a-time execution would otherwise hide -- aborts the run."""
    env.setdefault("HIP_EMU_ASM_HAZARDS", "abort")
    return run_emulated(body, timeout, **env)

CASE = '''
def check(n, P, T, site="weibull+4", kernel=_capi.KERNEL_LDS_PIPE, rooted=False, want="walk_pipe_kernel", form=None, seed=5):
    rng = np.random.default_rng(seed + n)
    w = small(n, P, T, site)
    if rooted:
        from test_gpu_parity import _random_rooted_parent_ids
        w.parent_ids = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)]).astype(np.int32)
        w.branch_lengths = rng.uniform(0.01, 0.4, (T, 2 * n - 1))
        w.branch_lengths[:, -1] = 0.0
    gaps = rng.random(w.patterns.shape) < 0.05
    w.patterns = np.where(gaps, 4, w.patterns).astype(np.int32)
    gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
    gpu.set_kernel(kernel)
    cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
    ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    ll = gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
    assert gpu.kernel_name() == want, gpu.kernel_name()
    if form is not None:
        assert form in gpu.kernel_form(), gpu.kernel_form()
    assert close(ll, ref["log_likelihood"], LL_ATOL, LL_RTOL), (n, P, site, "log-likelihood only")
    out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
    assert close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL), (n, P, site)
    assert close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL), (n, P, site)
    print(n, P, T, site, gpu.kernel_name(), gpu.kernel_form())
    return gpu, w, out
'''


def test_interpreted_pipe_kernel_every_group_count_and_category_count(emulated):
    """Four pattern groups per wave (the planner's choice up to 32 taxa), two and one (forced: BITO_AMD_PIPE_GROUPS), one,
    two and four rate categories (three go to the HBM-arena walk), rooted and unrooted trees, pattern counts either side
    of a tile, gaps in the alignment.  (KERNEL_LDS_PIPE pinned: AUTO keeps a log-likelihood-only pass with one rate
    category on the HBM-arena walk.)"""
    for groups in ("4", "2", "1"):
        run(CASE + '''
for n, P, T, site, rooted in ((3, 40, 2, "weibull+4", False), (9, 70, 3, "weibull+4", False), (9, 70, 2, "weibull+2", True),
                              (12, 130, 2, "constant", False), (27, 65, 2, "weibull+4", True), (5, 200, 7, "constant", True)):
    check(n, P, T, site, rooted=rooted, form="x %s pattern groups" % os.environ["BITO_AMD_PIPE_GROUPS"])
''', BITO_AMD_PIPE_GROUPS=groups)


def test_interpreted_pipe_kernel_every_register_layout(emulated):
    """The layouts that depend on the taxon count: 33-36 taxa beside four groups ("many" masks), two groups with exact
    images up to 38 taxa and reversible-form images (one per branch) from 39, the wide form for 49-64 taxa (64 mask
    registers, images of 126 KB staged through LDS); 65 taxa is the HBM-arena walk's."""
    run(CASE + '''
for n, P in ((33, 40), (36, 130), (38, 70), (39, 70), (44, 33), (48, 100), (49, 40), (52, 100), (56, 70), (57, 33), (64, 70)):
    check(n, P, 2 if n < 49 else 1, rooted=bool(n & 1))
check(65, 40, 1, kernel=_capi.KERNEL_AUTO, want="walk_hbm_cat_kernel")
''', timeout=1500)


def test_interpreted_pipe_kernel_two_wave_form_and_two_classes(emulated):
    """KERNEL_LDS_PIPE2: workgroups of two waves (small trees), and the launch in two classes (trees of a batch sorted into
    whole-tree units and tile runs): the same numbers."""
    run(CASE + '''
for n, P, T, site in ((6, 24, 5, "weibull+4"), (9, 70, 3, "weibull+4"), (16, 16, 4, "constant"), (23, 40, 2, "weibull+2")):
    check(n, P, T, site, kernel=_capi.KERNEL_LDS_PIPE2)
''')


def test_interpreted_pipe_kernel_tile_runs_and_whole_tree_units(emulated):
    """A tree's tiles walked by one workgroup (whole-tree unit: it writes the tree's final sums itself) or cut into runs
    whose partial sums the final-sums kernel adds up; both kinds in one launch."""
    for env in ({"BITO_AMD_LDS_TILE_RUN": "1"}, {"BITO_AMD_LDS_TILE_RUN": "2"}, {"BITO_AMD_PIPE_WHOLE_TREES": "1"},
                {"BITO_AMD_PIPE_DIRECT": "0"}):
        run(CASE + '''
check(9, 300, 3)
check(23, 130, 2, "weibull+2")
''', **env)


def test_interpreted_pipe_kernel_site_model_gradient(emulated):
    """The site-model (Weibull shape) gradient: a second pre-order traversal with the category rates differentiated
    (deriv_mode 1), against the checker's."""
    run('''
w = small(9, 70, 3)
gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
assert gpu.kernel_name() == "walk_pipe_kernel", gpu.kernel_name()
assert close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
assert close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
assert close(out["site_model"], ref["site_model"], 1e-6, 1e-8), (out["site_model"], ref["site_model"])
''')


def test_interpreted_pipe_kernel_ds1_trees_and_golden_values(emulated):
    """DS1's own trees (27 taxa, 1937 -> 410 site patterns, GTR + weibull+4: BASELINE.json's config 3, the workload `value`
    is quoted on) through the interpreted kernel: log-likelihood and gradient of five trees against the checker at the GPU
    tests' bars, and the hello tree's value the reference's doctest holds (src/doctest: -84.852358)."""
    run('''
w = workloads.ds1_gtr_weibull4(1).subset(5)
gpu = bito_amd.Engine(spec(w), w.patterns, w.weights)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 4)
out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
assert gpu.kernel_name() == "walk_pipe_kernel" and "4 pattern groups" in gpu.kernel_form(), gpu.kernel_form()
assert close(out["log_likelihood"], ref["log_likelihood"], LL_ATOL, LL_RTOL)
assert close(out["branch_lengths"], ref["branch_lengths"], GRAD_ATOL, GRAD_RTOL)
from bito_amd import treeio
from bito_amd.site_pattern import SitePattern
data = os.path.join({here!r}, "golden", "data")
tc = treeio.read_newick_file(os.path.join(data, "hello.nwk"))
sp = SitePattern(treeio.read_fasta(os.path.join(data, "hello.fasta")), tc.taxon_names)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification("JC69", "constant", "none"), sp.patterns, sp.weights)
eng.set_kernel(_capi.KERNEL_LDS_PIPE)  # (AUTO: a log-likelihood-only pass with one rate category is the HBM-arena walk's)
ll = eng.log_likelihoods(tc.parent_id_matrix(), tc.branch_length_matrix(), eng.default_params(len(tc.trees)))
assert eng.kernel_name() == "walk_pipe_kernel"
assert abs(ll[0] - -84.852358) < 1e-6, ll
'''.replace("{here!r}", repr(HERE)))


def test_interpreter_issues_the_matrix_instructions_the_device_counted(emulated):
    """The interpreter against the hardware's own counter: an MI355X counted SQ_INSTS_MFMA = 281 502 720 for one launch of
    walk_pipe_kernel<4,4,true,0> over 6400 DS1 trees (the 100 topologies x 64; round 4's last build, per dispatch in
    profiles/r4_v2_pipe_one_wave_pmc_per_dispatch.json).  The interpreter, walking the 100 distinct trees, must issue
    exactly a sixty-fourth of that -- the asm statements' v_mfma_f64_4x4x4 plus the root's builtin -- or it does not walk
    the step tables and loop bodies the device does.  (scripts/emu_pipe_instruction_mix.py prints every class and the
    blocking call's 1024- and 5376-tree chunks, which agree as well.)"""
    import json
    import subprocess
    import sys

    device = json.load(open(os.path.join(ROOT, "profiles", "r4_v2_pipe_one_wave_pmc_per_dispatch.json")))
    full = device["dispatches"]["19"]
    assert full["grid_size"] == 65536 and device["_trees"]["19"] == 6400
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import emu_pipe_instruction_mix as mix

    counts = mix.emulated_counts(100)
    issued = counts["mfma"] + counts["builtin_mfma_f64_4x4x4"]
    assert 64 * issued == int(full["SQ_INSTS_MFMA"]) == 281502720, (issued, full["SQ_INSTS_MFMA"])
    # the classes the compiled C++ around the statements adds to on the device: the interpreter's share is most of each
    for mine, theirs in ((issued + counts["valu_other"], "SQ_INSTS_VALU"), (counts["lds"], "SQ_INSTS_LDS"),
                         (counts["salu"] + counts["branch"], "SQ_INSTS_SALU"), (counts["smem"], "SQ_INSTS_SMEM")):
        assert 0.9 * full[theirs] < 64 * mine <= full[theirs], (theirs, 64 * mine, full[theirs])
    # profiles/executed.json (bench.py's roofline.executed) quotes the same count
    row = [r for r in json.load(open(os.path.join(ROOT, "profiles", "executed.json"))) if r["kernel"] == "walk_pipe_kernel"][0]
    assert abs(row["matrix_instructions_per_tree"] - issued / 100) < 1e-9


def test_gpu_parity_tests_of_the_pipe_walk_as_they_are(emulated):
    """`-m gpu` tests of tests/test_gpu_parity.py, unchanged, in a process whose bito_amd loads the emulated library (the
    interpreter's hazard checks fatal): the reference's golden values (hello JC69, DS1 JC69 + weibull+4: src/doctest
    values), the edge cases the reference tests, pattern counts around tile edges,
    caterpillar / balanced / random trees of 45 to 64 taxa, AUTO's choice at the pipe walk's limits, the reversible
    form's guard, the launch in two classes.  (The whole suite this way: scripts/emu_gpu_suite.sh,
    profiles/r5_emulated/gpu_suite_emulated.log.)"""
    os.environ["HIP_EMU_ASM_HAZARDS"] = "abort"
    try:
        out = run_gpu_tests_emulated(["tests/test_gpu_parity.py", "-k",
                                      "hello_jc69 or ds1_jc69_weibull or edge_cases or tile_edges or extreme_tree_shapes "
                                      "or kernel_choice_at_the_pipe or reversible_form_guard or pipe_walk_in_two_classes"], timeout=900)
    finally:
        os.environ.pop("HIP_EMU_ASM_HAZARDS", None)
    assert " passed" in out and "failed" not in out, out[-500:]
