"""CPU-side tests: host conventions (trees, alignments, site patterns, workloads), the
C ABI library (loads, exports every declared symbol, fails loudly without a GPU).
No compute call reaches a GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest

import bito_amd
from bito_amd import _capi, treeio, workloads
from bito_amd.site_pattern import SitePattern, symbol_vector

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def test_header_symbols_are_exported():
    header = open(os.path.join(ROOT, "include", "bito_amd.h")).read()
    declared = set(re.findall(r"\b(bito_amd_[a-z_]+)\s*\(", header))
    assert declared == set(_capi.SYMBOLS)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libbito_amd.so does not export {name}"
    assert bito_amd.version().startswith("bito_amd")


def test_beagle_shim_symbols_are_exported():
    from beagle_driver import BEAGLE_SYMBOLS

    header = open(os.path.join(ROOT, "include", "bito_amd_beagle.h")).read()
    declared = set(re.findall(r"\b(beagle[A-Z][A-Za-z]+)\s*\(", header))
    assert declared == set(BEAGLE_SYMBOLS) and len(declared) == 17
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libbito_amd.so does not export {name}"


def test_gp_symbols_are_exported():
    from bito_amd import gp

    header = open(os.path.join(ROOT, "include", "bito_amd_gp.h")).read()
    declared = set(re.findall(r"\b(bito_amd_gp_[a-z_]+)\s*\(", header))
    assert declared == set(gp.GP_SYMBOLS)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libbito_amd.so does not export {name}"
    # the op record is the POD the header declares
    assert gp.OP_DTYPE.itemsize == 32


def test_single_tree_gp_schedule_shape():
    from bito_amd import gp

    dag = gp.single_tree_dag([4, 3, 3, 4])  # (0,(1,2)3)4
    assert dag.node_count == 5 and dag.gpcsp_count == 5 and dag.root == 4
    assert dag.children == {3: (1, 2), 4: (0, 3)} and dag.parent[3] == (4, False)
    assert dag.pv(gp.R_LEFT, 4) == 5 * 5 + 4 and dag.edge(4) == 0 and dag.edge(0) == 1
    pop, lik = dag.populate_plvs(), dag.compute_likelihoods()
    ops, side = pop.arrays()
    # 2 internal nodes x 3 + 5 nodes x 3 zeroings, 1 stationary, rootward 2 x 5, leafward 4 x 4 + 2
    assert len(ops) == 6 + 15 + 1 + 10 + 18 and len(side) == 8
    assert [o[0] for o in lik.ops] == [gp.LIKELIHOOD] * 4 + [gp.RESET_MARGINAL_LIKELIHOOD, gp.INCREMENT_MARGINAL_LIKELIHOOD]


def test_no_signature_leaks_torch_or_cxx_types():
    header = open(os.path.join(ROOT, "include", "bito_amd.h")).read()
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # declarations only, comments stripped
    assert "torch" not in code and "std::" not in code and "hip" not in code.lower() and "&" not in code


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_engine_creation_fails_loudly_without_gpu():
    spec = bito_amd.PhyloModelSpecification("JC69", "constant", "none")
    with pytest.raises(bito_amd.BitoAmdError) as exc:
        bito_amd.Engine(spec, np.zeros((3, 5), dtype=np.int32), np.ones(5))
    assert exc.value.code == _capi.ERR_DEVICE and "no CPU fallback" in str(exc.value)


def test_model_errors_precede_device_errors():
    with pytest.raises(bito_amd.BitoAmdError, match="Substitution model not known"):
        bito_amd.Engine(bito_amd.PhyloModelSpecification("K80", "constant", "none"),
                        np.zeros((3, 5), dtype=np.int32), np.ones(5))
    with pytest.raises(bito_amd.BitoAmdError, match="Site model not known"):
        bito_amd.Engine(bito_amd.PhyloModelSpecification("JC69", "invgamma", "none"),
                        np.zeros((3, 5), dtype=np.int32), np.ones(5))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bito_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("CPU oracle and GPU engine", ""), f"{f} mentions the oracle"


def test_product_knows_nothing_of_the_emulation_or_the_reference_build():
    """tests/hip_emu (the stand-in HIP runtime the CPU suite runs the .hip sources under) and oracle/_ref (the reference's
    own code compiled in place) are test infrastructure: the package, the headers, bench.py's measured path and the driver
    entry points' product side do not mention them, and the only way a process loads the emulated library is the
    BITO_AMD_LIB variable that tests set."""
    for base in ("bito_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".inc")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f)).read()
                    for word in ("hip_emu", "HIP_EMULATION", "_ref/", "libbito_ref", "libbito_amd_emu"):
                        assert word not in text, (f, word)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "hip_emu" not in bench and "_ref" not in bench
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "hip_emu" not in entry  # (build() does build oracle/_ref where the reference is present: building the checker is not using it)


def test_newick_ids_follow_polish(data_dir):
    tc = treeio.read_newick_file(os.path.join(data_dir, "hello.nwk"))
    assert tc.taxon_names == ["mars", "saturn", "jupiter"]
    t = tc.trees[0]
    assert not t.rooted and t.node_count == 4
    assert list(t.parent_ids) == [3, 3, 3]
    assert np.allclose(t.branch_lengths, [0.1, 0.1, 0.3, 0.0])
    tc = treeio.read_newick_file(os.path.join(data_dir, "hello.nwk"), sort_taxa=True)
    assert tc.taxon_names == ["jupiter", "mars", "saturn"]
    tc = treeio.read_newick_file(os.path.join(data_dir, "hello_rooted.nwk"))
    t = tc.trees[0]
    # (jupiter,(mars,saturn)): leaves 0,1,2 by first appearance; inner node 3; root 4
    assert t.rooted and list(t.parent_ids) == [4, 3, 3, 4]
    assert np.allclose(t.branch_lengths, [0.0594247559, 0.2072560544, 0.0694244266, 0.01, 0.0])


def test_parse_inline_newick_features():
    coll = treeio.parse_newick_strings(["tree t1 = [&U] ('a b':1e-1[&x=1],(c:2,d:3)lab:0.5,e);"])
    assert coll.taxon_names == ["a b", "c", "d", "e"]
    t = coll.trees[0]
    assert list(t.parent_ids) == [5, 4, 4, 5, 5]
    assert np.allclose(t.branch_lengths, [0.1, 2, 3, 0, 0.5, 0])
    with pytest.raises(RuntimeError, match="not known in our taxon set"):
        treeio.parse_newick_strings(["(a,b,c);", "(a,b,z);"])
    with pytest.raises(RuntimeError, match="Float conversion failed"):
        treeio.parse_newick_strings(["(a:x,b,c);"])


def test_internal_ids_do_not_depend_on_sibling_order():
    """The reference's Node constructor sorts children by the largest leaf id beneath them before
    Polish numbers the internal nodes in post-order (src/node.cpp:33-46,383-402), so one topology has
    one parent-id vector however its Newick string orders siblings.  Expected vectors: the reference's
    own examples (src/node.hpp:313-314,336-339,358)."""
    def ids(newick, **kw):
        return list(treeio.parse_newick_strings([newick], **kw).trees[0].parent_ids)

    taxa = {str(i): i for i in range(7)}
    assert ids("((((0,1),2),(3,4)),5,6);", taxa=taxa) == [7, 7, 8, 9, 9, 11, 11, 8, 10, 10, 11]
    assert ids("(6,((4,3),(2,(1,0))),5);", taxa=taxa) == [7, 7, 8, 9, 9, 11, 11, 8, 10, 10, 11]
    taxa4 = {str(i): i for i in range(4)}
    assert ids("(0,1,(2,3));", taxa=taxa4) == [5, 5, 4, 4, 5]
    assert ids("((3,2),1,0);", taxa=taxa4) == [5, 5, 4, 4, 5]
    assert ids("(0,(1,(2,3)));", taxa=taxa4) == [6, 5, 4, 4, 5, 6]
    assert ids("(((3,2),1),0);", taxa=taxa4) == [6, 5, 4, 4, 5, 6]
    assert ids("(3,(2,(1,0)));", taxa=taxa4) == [4, 4, 5, 6, 5, 6]  # Node::Ladder(4)
    # out-of-order siblings with alphabetical leaf ids: bito gives [5,5,6,6,7,7,7]
    assert ids("((c,d),(a,b),e);", sort_taxa=True) == [5, 5, 6, 6, 7, 7, 7]
    a = treeio.parse_newick_strings(["((a:1,b:2):3,(c:4,d:5):6,e:7);", "(e:7,(d:5,c:4):6,(b:2,a:1):3);"], sort_taxa=True)
    assert np.array_equal(a.trees[0].parent_ids, a.trees[1].parent_ids)
    assert np.array_equal(a.trees[0].branch_lengths, a.trees[1].branch_lengths)  # lengths travel with their child
    with pytest.raises(RuntimeError, match="appears twice"):
        treeio.parse_newick_strings(["((a,b),(a,c));"], taxa={"a": 0, "b": 1, "c": 2})


def test_nexus_translate_order(data_dir):
    tc = treeio.read_nexus_file(os.path.join(data_dir, "DS1.subsampled_10.t"))
    assert len(tc.trees) == 10 and len(tc.taxon_names) == 27
    assert tc.taxon_names[0] == "Alligator_mississippiensis" and tc.taxon_names[26] == "Xenopus_laevis"
    for t in tc.trees:
        assert t.node_count == 52 and not t.rooted
        # post-order ids: every parent id exceeds its children
        assert np.all(t.parent_ids > np.arange(51))


def test_parent_id_vector_round_trip():
    t = treeio.tree_from_parent_ids([5, 5, 6, 7, 7, 6, 8, 8])  # reference src/unrooted_sbn_instance.hpp:225-226
    assert t.leaf_count == 5 and t.rooted and t.node_count == 9


def test_site_pattern_compression(data_dir):
    tc = treeio.read_newick_file(os.path.join(data_dir, "hello.nwk"))
    sp = SitePattern(treeio.read_fasta(os.path.join(data_dir, "hello.fasta")), tc.taxon_names)
    assert sp.patterns.shape == (3, 15) and sp.weights.sum() == 31 and sp.patterns.max() == 4
    # symbol table of the reference (src/site_pattern.hpp:64-69): degenerate codes are gaps
    assert list(symbol_vector("-tgcaTGCA?")) == [4, 3, 2, 1, 0, 3, 2, 1, 0, 4]
    assert list(symbol_vector("RYKMSWBDHVN")) == [4] * 11
    with pytest.raises(RuntimeError, match="Symbol 'Z' not known"):
        symbol_vector("ACZ")
    part = sp.partials(0).reshape(15, 4)
    gap = sp.patterns[0] == 4
    assert np.all(part[gap] == 1.0) and np.all(part[~gap].sum(axis=1) == 1.0)
    # expansion by weights reproduces the column multiset
    tc2, sp2 = workloads.load_ds1("DS1.subsampled_10.t")
    assert sp2.patterns.shape == (27, 934) and sp2.weights.sum() == 1949
    with pytest.raises(RuntimeError, match="not found in alignment"):
        SitePattern({"mars": "A"}, ["mars", "venus"])


def test_workloads_are_deterministic():
    a = workloads.ds1_gtr_weibull4(1)
    b = workloads.ds1_gtr_weibull4(2)
    assert a.parent_ids.shape == (100, 51) and b.tree_count == 200
    assert np.array_equal(a.branch_lengths, b.branch_lengths[:100])
    assert not np.array_equal(b.branch_lengths[:100], b.branch_lengths[100:])
    assert a.branch_lengths[:, :51].min() >= 1e-6 and a.branch_lengths.max() <= 1.0
    assert np.all(a.branch_lengths[:, 51] == 0.0)
    rng = workloads.Xoshiro256ss(20240601)
    assert rng.next_u64() != rng.next_u64()
    s = workloads.synthetic_gtr_weibull4(12, 40, 3)
    t = workloads.synthetic_gtr_weibull4(12, 40, 3)
    assert np.array_equal(s.patterns, t.patterns) and np.array_equal(s.parent_ids, t.parent_ids)
    assert s.parent_ids.shape == (3, 21) and np.all(s.parent_ids > np.arange(21))
    sh = [a.shard(r, 8) for r in range(8)]
    assert sum(x.tree_count for x in sh) == 100
    assert np.array_equal(np.concatenate([x.branch_lengths for x in sh]), a.branch_lengths)


def test_generated_walk_loops_are_current_and_checked(tmp_path):
    """bito_amd/csrc/walk_pipe_gen.inc is generated: the committed file must be what scripts/gen_walk_pipe.py
    emits today, and the compiled kernel must leave the AGPRs that hold a tree's matrix images alone outside its
    own asm statements (the Makefile runs the same check at build time)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = os.path.join(root, "bito_amd", "csrc", "walk_pipe_gen.inc")
    fresh = tmp_path / "walk_pipe_gen.inc"  # (generated beside the tracked file, which keeps its content and time stamp)
    subprocess.run([sys.executable, os.path.join(root, "scripts", "gen_walk_pipe.py"), "--out", str(fresh)], check=True,
                   stdout=subprocess.DEVNULL)
    assert open(committed).read() == fresh.read_text(), "walk_pipe_gen.inc is stale: run scripts/gen_walk_pipe.py"
    listing = os.path.join(root, "bito_amd", "csrc", "walk_pipe.gfx950.s")
    if os.path.exists(listing):  # (written by the build)
        done = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_walk_pipe_asm.py"), listing],
                              capture_output=True, text=True)
        assert done.returncode == 0, done.stderr


def _pipe_plan(n, patterns, categories, trees, min_cherries):
    import ctypes as C

    from bito_amd import _capi

    out = (C.c_int32 * 7)()
    assert _capi.lib().bito_amd_plan_pipe_walk(n, patterns, categories, trees, min_cherries, out) == 0
    keys = ("groups", "patterns_per_workgroup", "tiles", "lds_bytes", "tile_run", "whole_trees", "slots")
    return dict(zip(keys, out))


def test_pipe_walk_planner():
    """Host arithmetic of walk_pipe_kernel's launch plan (no device involved): pattern groups per wave from what
    a wave must keep in LDS, the taxon limits of the register files, and the units of work by batch size."""
    ds1 = _pipe_plan(27, 934, 4, 1600, 7)  # BASELINE config 3: DS1's topologies have seven cherries or more
    assert ds1["groups"] == 4 and ds1["patterns_per_workgroup"] == 64 and ds1["tiles"] == 15 and ds1["slots"] == 18
    assert ds1["lds_bytes"] <= 160 * 1024
    # a thousand trees and more: whole-tree units for seven eighths, runs of a third of a tree behind them
    assert ds1["tile_run"] == 5 and ds1["whole_trees"] == 1400
    big = _pipe_plan(27, 934, 4, 6400, 7)  # the short units stay a third of a tree, about 400 trees of them
    assert big["tile_run"] == 5 and big["whole_trees"] == 6000
    small = _pipe_plan(27, 934, 4, 400, 7)  # fewer: runs only, about four units per workgroup
    assert small["whole_trees"] == 0 and small["tile_run"] == 5
    # a hundred trees: the units go through the 256 resident workgroups in rounds -- 500 units of three tiles (two
    # rounds) beat 1500 of one (six rounds, each paying the tree's images again); ten trees: one tile per unit
    assert _pipe_plan(27, 934, 4, 100, 7)["tile_run"] == 3
    assert _pipe_plan(27, 934, 4, 10, 7)["tile_run"] == 1
    # one more vector per wave than LDS holds beside four groups: two groups
    assert _pipe_plan(27, 934, 4, 1600, 6)["groups"] == 2
    # tip masks: 32 registers beside four groups, 48 beside two (one image per branch: the AGPR file would hold 57
    # taxa); beyond that, or when the stored vectors of the tree with the fewest cherries do not fit LDS: not taken
    assert _pipe_plan(32, 934, 4, 1600, 14)["groups"] == 4
    # 33 to 38 taxa: four groups beside 40 mask registers (round 4) when the stored vectors leave room, else two
    assert _pipe_plan(33, 934, 4, 1600, 14)["groups"] == 4
    assert _pipe_plan(38, 934, 4, 1600, 18)["groups"] == 4
    assert _pipe_plan(38, 934, 4, 1600, 10)["groups"] == 2
    assert _pipe_plan(39, 934, 4, 1600, 19)["groups"] == 2
    assert _pipe_plan(41, 934, 4, 1600, 9)["groups"] == 2
    assert _pipe_plan(48, 934, 4, 1600, 14)["groups"] == 2
    # 38 vectors per wave: 160 KB do not hold them beside two groups (1 KB cells); one group per wave keeps 512-byte
    # cells, twice as many
    assert _pipe_plan(48, 934, 4, 1600, 8)["groups"] == 1
    # 49 to 64 taxa: the wide layout (64 mask registers, 4n - 4 <= 252 image registers), two groups at most
    assert _pipe_plan(49, 934, 4, 1600, 20)["groups"] == 2
    assert _pipe_plan(56, 934, 4, 1600, 22)["groups"] == 2
    assert _pipe_plan(56, 934, 4, 1600, 10)["groups"] == 1  # 44 vectors per wave: one group, half-size cells
    assert _pipe_plan(56, 934, 4, 1600, 1)["groups"] == 1  # a caterpillar: 53 vectors
    assert _pipe_plan(57, 934, 4, 1600, 25)["groups"] == 2  # (the wide kernels own all but four AGPRs: up to 64 taxa)
    assert _pipe_plan(64, 934, 4, 1600, 21)["groups"] == 1
    assert _pipe_plan(64, 934, 4, 1600, 1)["groups"] == 1   # a caterpillar: 61 vectors of 512 bytes per wave
    assert _pipe_plan(65, 934, 4, 1600, 30)["groups"] == 0
    # one rate category: sixteen patterns per group, 256 per workgroup
    jc = _pipe_plan(27, 934, 1, 1600, 7)
    assert jc["groups"] == 4 and jc["patterns_per_workgroup"] == 256 and jc["tiles"] == 4 and jc["lds_bytes"] <= 160 * 1024
    assert _pipe_plan(27, 934, 3, 1600, 7)["groups"] == 0  # 1, 2 or 4 categories
    # every plan fits LDS and keeps at least the vectors the batch's worst tree needs
    for n in range(3, 66):
        for cherries in (1, n // 3, n // 2):
            for categories in (1, 2, 4):
                p = _pipe_plan(n, 500, categories, 1000, cherries)
                if p["groups"]:
                    assert p["lds_bytes"] <= 160 * 1024 and p["slots"] >= max(1, n - 2 - cherries)
                    assert p["tiles"] * p["patterns_per_workgroup"] >= 500 and p["tiles"] % p["tile_run"] == 0


def test_host_pool_runs_every_part_once_and_is_race_free(tmp_path):
    """bito_amd/csrc/host_pool.hpp (the helper threads of a blocking call) on its own, no GPU: tests/host_pool_test.cpp
    built with g++ -- under ThreadSanitizer when its runtime is installed -- runs thousands of jobs on pools of 0 to 7
    helpers that are armed, lingering or asleep when a job arrives."""
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_pool_test.cpp")
    exe = str(tmp_path / "host_pool_test")
    base = [gxx, "-O1", "-g", "-std=c++17", "-pthread", src, "-o", exe]
    built = subprocess.run(base[:5] + ["-fsanitize=thread"] + base[5:], capture_output=True, text=True)
    if built.returncode != 0:
        built = subprocess.run(base, capture_output=True, text=True)
    assert built.returncode == 0, built.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "0 bad" in run.stdout, run.stdout + run.stderr
    assert "ThreadSanitizer" not in run.stderr, run.stderr


def test_unstored_node_count_matches_a_restatement():
    """walk_pipe_kernel keeps no LDS vector for cherries and for the pitchforks it folds into their parents' steps (a tip
    and a cherry under one node whose sibling is a tip or a stored node; of two sibling pitchforks the lower id).  The host
    sizes a tree's LDS slots by this count, so it is held to an independent restatement here, rooted and unrooted
    (the unrooted tree as it is walked: UnrootedTree::Detrifurcate, reference src/unrooted_tree.cpp:27-37)."""
    import ctypes as C

    from test_gpu_parity import _random_rooted_parent_ids

    def restated(par, n, rooted, fold):
        M = len(par) + 1
        kids = {}
        for child, p in enumerate(par):
            kids.setdefault(int(p), []).append(child)
        if not rooted:
            r = M - 1
            a, b, c = kids[r]
            kids[r] = [b, c]
            kids[r + 1] = [a, r]
        root = 2 * n - 2
        up = {c: p for p, cs in kids.items() for c in cs}
        cherry = lambda v: v >= n and v != root and all(x < n for x in kids[v])  # noqa: E731
        fork = lambda v: v >= n and v != root and sorted((x < n, cherry(x)) for x in kids[v]) == [(False, True), (True, False)]  # noqa: E731
        count = 0
        for v in range(n, root):
            if cherry(v):
                count += 1
            elif fold and fork(v):
                sib = [x for x in kids[up[v]] if x != v][0]
                if sib < n or (not cherry(sib) and (not fork(sib) or v < sib)):
                    count += 1
        return count

    rng = np.random.default_rng(17)
    L = _capi.lib()
    for n in (3, 4, 5, 9, 27, 64):
        for rooted in (0, 1):
            T = 40
            if rooted:
                pid = np.stack([_random_rooted_parent_ids(n, rng) for _ in range(T)]).astype(np.int32)
            else:
                pid = np.stack([workloads.random_unrooted_tree(n, rng, 0.1).parent_ids for _ in range(T)]).astype(np.int32)
            pid = np.ascontiguousarray(pid)
            for fold in (0, 1):
                out = np.zeros(T, dtype=np.int32)
                rc = L.bito_amd_count_unstored_nodes(n, T, rooted, pid.shape[1] + 1, pid.ctypes.data_as(C.POINTER(C.c_int32)), fold,
                                                     out.ctypes.data_as(C.POINTER(C.c_int32)))
                assert rc == 0
                assert list(out) == [restated(row, n, rooted, fold) for row in pid], (n, rooted, fold)
    # the DS1 topologies: 17.5 stored vectors per tree without folding, 14.2 with
    w = workloads.ds1_gtr_weibull4(1)
    pid = np.ascontiguousarray(w.parent_ids, dtype=np.int32)
    counts = []
    for fold in (0, 1):
        out = np.zeros(100, dtype=np.int32)
        assert L.bito_amd_count_unstored_nodes(27, 100, 0, 52, pid.ctypes.data_as(C.POINTER(C.c_int32)), fold,
                                               out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        counts.append(26 - out.mean())  # internal nodes - unstored = stored vectors + the root
    assert 17.0 < counts[0] < 18.5 and 13.5 < counts[1] < 15.0, counts


def test_unstored_node_count_rejects_rows_that_are_not_trees():
    """bito_amd_count_unstored_nodes is a public entry point: rows that are not bito topologies (a parent id outside
    [n, M), a parent below its child, an internal node without exactly two children) come back as BAD_TREE instead of
    being used as indices."""
    import ctypes as C

    from test_gpu_parity import _random_rooted_parent_ids

    L = _capi.lib()
    ip = C.POINTER(C.c_int32)
    n = 6
    for rooted in (0, 1):
        rng = np.random.default_rng(5 + rooted)
        good = (_random_rooted_parent_ids(n, rng) if rooted else workloads.random_unrooted_tree(n, rng, 0.1).parent_ids)
        good = np.ascontiguousarray(good, dtype=np.int32)
        M = good.shape[0] + 1
        out = np.zeros(1, dtype=np.int32)
        assert L.bito_amd_count_unstored_nodes(n, 1, rooted, M, good.ctypes.data_as(ip), 1, out.ctypes.data_as(ip)) == 0
        for child, value in ((0, n - 1), (0, -3), (1, M), (2, 10 ** 6), (n, n), (0, int(good[1]))):
            bad = good.copy()
            bad[child] = value
            if np.array_equal(bad, good):
                continue
            rc = L.bito_amd_count_unstored_nodes(n, 1, rooted, M, bad.ctypes.data_as(ip), 1, out.ctypes.data_as(ip))
            assert rc == _capi.ERR_BAD_TREE, (rooted, child, value, rc)
        assert L.bito_amd_count_unstored_nodes(n, 1, rooted, M + 1, good.ctypes.data_as(ip), 1, out.ctypes.data_as(ip)) != 0


def test_bench_parameter_rows_for_cache_misses_and_distinct_models():
    """bench.py's timed loops alternate two sets of parameter rows that differ in the last bit of one rate (no step finds
    the models of the step before), and the codon workload can carry K different (kappa, omega) rows among its trees
    (the reference hands every tree its own row, src/fat_beagle.hpp:173-181)."""
    w = workloads.ds1_gtr_weibull4(1)
    other = workloads.other_bits(w.params, 4)
    assert other.shape == w.params.shape and np.all(other[:, 4] > w.params[:, 4])
    assert np.array_equal(np.delete(other, 4, axis=1), np.delete(w.params, 4, axis=1))
    assert np.abs(other - w.params).max() < 1e-16 and abs(other[0, 4:10].sum() - 1.0) < 1e-3
    for K in (1, 7, 64):
        rows = workloads.codon_rows(64, K)
        assert rows.shape == (64, 6) and len(np.unique(rows, axis=0)) == K
        assert np.array_equal(rows[0], np.array(workloads.CODON_PARAMS)) and np.array_equal(rows[:K], rows[K:2 * K] if 2 * K <= 64 else rows[:K])
        assert np.all(rows[:, :4] == np.array(workloads.CODON_PARAMS[:4]))
        assert np.all((rows[:, 4] >= 1.5) & (rows[:, 4] <= 4.0) & (rows[:, 5] >= 0.1) & (rows[:, 5] <= 0.9))
    assert np.array_equal(workloads.codon_rows(8, 3), workloads.codon_rows(8, 3))
