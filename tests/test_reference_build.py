"""The restatements held to the REFERENCE ITSELF where the reference compiles here: oracle/_ref/libbito_ref.so is built by
oracle/Makefile (`make ref`) from the reference's own source files where they lie under /root/reference/src --
alignment.cpp, site_pattern.cpp, node.cpp, tree.cpp, unrooted_tree.cpp, bitset.cpp, optimization.hpp -- behind the C ABI
of oracle/ref_shim.cpp (no BEAGLE, no Eigen: the likelihood arithmetic and the Newick parser are not buildable in this
container and stay pinned by golden values).  What runs below is the reference's code, not a restatement of it:
SitePattern, Node's sibling order + Polish + parent-id vectors and UnrootedTree::Detrifurcate (A6), and
the five one-dimensional optimisers (8f f1) -- Brent's iterates bit for bit, the tie of hello's edge 3 included.
Skipped where neither /root/reference nor the built library exists."""
import os

import numpy as np
import pytest

from bito_amd import gp, treeio, workloads
from bito_amd.site_pattern import SitePattern
from oracle import gp as ogp
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="the reference's sources are not in this container")

TREE_FILES = ["hello.nwk", "hello_rooted.nwk", "hello_rooted_two_trees.nwk", "five_taxon_rooted.nwk", "ds1-reduced-5.nwk",
              "six_taxon_rooted_simple.nwk", "simplest-hybrid-marginal-all-trees.nwk", "fluA.tree", "DS1.100_topologies.nwk"]


def _raw_structure(line, taxa):
    """the Newick string's own nesting, siblings in the order they are written: (children lists, leaf taxon ids, root)"""
    tok = treeio._Tok(line[line.find("("):])
    root = treeio._parse_node(tok)
    nodes, children, leaf = [], [], {}
    stack = [root]
    index = {}
    while stack:
        nd = stack.pop()
        index[id(nd)] = len(nodes)
        nodes.append(nd)
        stack.extend(reversed(nd.children))
    for k, nd in enumerate(nodes):
        children.append([index[id(ch)] for ch in nd.children])
        if not nd.children:
            leaf[k] = taxa[nd.name]
    return children, leaf, 0


@pytest.mark.parametrize("name", TREE_FILES)
def test_node_ids_are_the_references(data_dir, name):
    """treeio's ids (siblings ordered by the largest leaf id below them, internal ids in post-order) against the
    reference's Node constructor + Node::Polish on the same nesting (src/node.cpp:33-46,383-402); and
    Node::OfParentIdVector / ParentIdVector take treeio's vectors back and forth unchanged (src/node.cpp:511-551)."""
    path = os.path.join(data_dir, name)
    lines = [ln for ln in open(path).read().splitlines() if ln.strip() and "(" in ln]
    tc = treeio.read_newick_file(path)
    taxa = {n: i for i, n in enumerate(tc.taxon_names)}
    assert len(lines) == len(tc.trees)
    for line, tree in list(zip(lines, tc.trees))[:40]:
        children, leaf, root = _raw_structure(line, taxa)
        assert np.array_equal(ref.polished_parent_ids(children, leaf, root), tree.parent_ids)
        assert np.array_equal(ref.parent_id_round_trip(tree.parent_ids), tree.parent_ids)


def test_detrifurcation_is_the_references(data_dir):
    """UnrootedTree::Detrifurcate (src/unrooted_tree.cpp:27-37) on the 100 DS1 topologies: the rooted tree the kernels and
    the checker walk -- the old root keeps children 1 and 2, a new root joins child 0 with it, both new branches zero --
    (a) as parent ids and branch lengths, and (b) through the CPU checker: the unrooted tree and the reference's
    detrifurcated tree, passed as a rooted one, have the same log-likelihood and gradient to the last bit."""
    from oracle import oracle

    w = workloads.ds1_gtr_weibull4(1)
    n, M = w.patterns.shape[0], w.parent_ids.shape[1] + 1
    rooted_ids, rooted_bl = [], []
    for t in range(w.tree_count):
        ids, bl = ref.detrifurcate(w.parent_ids[t], w.branch_lengths[t])
        kids = [c for c in range(M - 1) if w.parent_ids[t][c] == M - 1]
        want = w.parent_ids[t].astype(np.int64).copy()
        want[kids[0]] = M  # child 0 hangs from the new root
        want = np.append(want, M)  # ... and so does the old root
        assert np.array_equal(ids, want)
        assert np.array_equal(bl[:M - 1], w.branch_lengths[t][:M - 1]) and bl[M - 1] == 0.0 and bl[M] == 0.0
        rooted_ids.append(ids)
        rooted_bl.append(bl)
    cpu = oracle.OracleEngine(w.substitution, w.site, "none", w.patterns, w.weights, 8)
    sel = slice(0, 12)
    unrooted = cpu.gradients(w.parent_ids[sel], w.branch_lengths[sel], w.params[sel])
    rooted = cpu.gradients(np.array(rooted_ids, dtype=np.int32)[sel], np.array(rooted_bl)[sel], w.params[sel])
    assert np.array_equal(unrooted["log_likelihood"], rooted["log_likelihood"])
    assert np.array_equal(unrooted["branch_lengths"][:, :M - 1], rooted["branch_lengths"][:, :M - 1])


@pytest.mark.parametrize("fasta,trees", [("hello.fasta", "hello.nwk"), ("DS1.fasta", "DS1.100_topologies.nwk"),
                                         ("fluA.fa", "fluA.tree"), ("five_taxon.fasta", "five_taxon_rooted.nwk"),
                                         ("six_taxon.fasta", "six_taxon_rooted_simple.nwk"),
                                         ("7-taxon-slice-of-ds1.fasta", "simplest-hybrid-marginal-all-trees.nwk")])
def test_site_patterns_are_the_references(data_dir, fasta, trees):
    """SitePattern::Compress (src/site_pattern.cpp:77-115) run by the reference against bito_amd/site_pattern.py: the same
    set of distinct columns with the same weights (the reference's pattern ORDER is the iteration order of an
    unordered_map: only order-independent quantities are compared), rows in taxon-id order, symbols by GetSymbolTable."""
    tc = treeio.read_newick_file(os.path.join(data_dir, trees))
    theirs = ref.SitePattern(os.path.join(data_dir, fasta), tc.taxon_names)
    mine = SitePattern(treeio.read_fasta(os.path.join(data_dir, fasta)), tc.taxon_names)
    assert theirs.patterns.shape == mine.patterns.shape and theirs.weights.sum() == mine.weights.sum()
    as_map = lambda sp: {tuple(int(x) for x in sp.patterns[:, p]): float(sp.weights[p]) for p in range(sp.patterns.shape[1])}  # noqa: E731
    assert as_map(theirs) == as_map(mine)


def _edge_function(data_dir, edge_index):
    """f(x) = (log-likelihood, d/dt, d2/dt2 at branch length t = x) of one edge of hello's sweep, at the state the sweep
    finds it in; evaluated by the CPU checker.  Returns (engine, evaluate, the edge's OptimizeBranchLength operation,
    the operations before it)"""
    import test_gp

    sp, tree, dag = test_gp.hello_instance(data_dir)
    ops = dag.branch_length_optimization().ops
    at = [k for k, op in enumerate(ops) if op[0] == gp.OPTIMIZE_BRANCH_LENGTH][edge_index]
    eng = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
    eng.set_branch_lengths(dag.branch_lengths(tree.branch_lengths))
    eng.process_operations(dag.populate_plvs())
    before = gp.OpStream()
    before.ops = list(ops[:at])
    before.side = list(dag.branch_length_optimization().side)
    eng.process_operations(before)
    _, _, leafward, rootward, edge = ops[at]

    def evaluate(t):
        bl = eng.get_branch_lengths()
        keep = bl[edge]
        bl[edge] = t
        eng.set_branch_lengths(bl)
        out = eng.log_likelihood_and_first_two_derivatives(edge, rootward, leafward)
        bl[edge] = keep
        eng.set_branch_lengths(bl)
        return out

    return eng, evaluate, ops[at], edge


@pytest.mark.parametrize("edge_index", [0, 1, 2, 3])
def test_brent_iterates_are_the_references(data_dir, edge_index):
    """Optimization::BrentMinimize -- the reference's own template, compiled from src/optimization.hpp -- on the negative
    log-likelihood of each of hello's four optimised edges (brent_nongrad_func, src/gp_engine.cpp:605-612; constants of
    src/dag_branch_handler.hpp:266-295), against the checker's restatement (oracle/gp_oracle.c) optimising the same edge:
    the same trial points with the same values in the same order, bit for bit, and the same optimum.  The third of these
    edges is the one whose fifth decision is the tie Brent has by construction (tests/gp_trace.py): the restatement falls
    the way the reference's code falls."""
    eng, evaluate, op, edge = _edge_function(data_dir, edge_index)
    calls = []

    def neg_ll(x):
        f = -evaluate(np.exp(x))[0]
        calls.append((x, f))
        return f

    current = eng.get_branch_lengths()[edge]
    x_ref, fx_ref = ref.brent_minimize(neg_ll, np.log(current), -13.9, 1.1, 10, 1000, 1.0005)
    # the checker on the same edge from the same state
    eng.set_optimization_method(gp.BRENT)
    eng.reset_optimization_count()
    eng.start_optimizer_trace()
    one = gp.OpStream()
    one.ops = [op]
    eng.process_operations(one)
    trace = eng.optimizer_trace()
    mine = [(row[1], row[2]) for row in trace if row[3] >= 1]  # (kind 0 is the handler's own evaluation, before Brent)
    assert len(mine) == len(calls) and len(calls) >= 8
    assert all(a == b for a, b in zip(mine, calls))  # bit for bit
    handler_nll = trace[0][2]
    expect = current if fx_ref > handler_nll else np.exp(x_ref)  # DAGBranchHandler::BrentOptimization keeps the better one
    assert eng.get_branch_lengths()[edge] == expect


def test_gradient_brent_newton_and_the_ascents_are_the_references(data_dir):
    """The other four optimisers of src/optimization.hpp:191-405 -- BrentMinimizeWithGradients, GradientAscent,
    LogSpaceGradientAscent, NewtonRaphsonOptimization -- run by the reference on hello's venus edge with the handler's
    arguments (src/dag_branch_handler.cpp:176-300), against the checker's restatements: the same branch length, bit for bit."""
    for method in (gp.BRENT_WITH_GRADIENTS, gp.GRADIENT_ASCENT, gp.LOGSPACE_GRADIENT_ASCENT, gp.NEWTON):
        eng, evaluate, op, edge = _edge_function(data_dir, 2)
        current = eng.get_branch_lengths()[edge]
        if method == gp.BRENT_WITH_GRADIENTS:
            def f(x):  # brent_grad_func: (-LL, -t dLL/dt) in the log length
                t = np.exp(x)
                v = evaluate(t)
                return (-v[0], -t * v[1], 0.0)
            handler_nll = f(np.log(current))[0]
            x, fx = ref.brent_minimize(f, np.log(current), -13.9, 1.1, 10, 1000, 1.0005, with_gradients=True)
            expect = current if fx > handler_nll else np.exp(x)
        elif method == gp.GRADIENT_ASCENT:
            expect = ref.gradient_ascent(lambda t: evaluate(t), current, 10, 5e-4, -13.9, 1000)
        elif method == gp.LOGSPACE_GRADIENT_ASCENT:
            expect = ref.gradient_ascent(lambda t: evaluate(t), current, 10, 1.0005, np.exp(-13.9), 1000, log_space=True)
        else:
            def g(x):  # the log-length form of src/gp_engine.cpp:643-655
                t = np.exp(x)
                v = evaluate(t)
                f1 = t * v[1]
                return (v[0], f1, f1 + t * t * v[2])
            expect = np.exp(ref.newton(g, np.log(current), 10, 1e-10, -13.9, 1.1, 1000))
        eng.set_optimization_method(method)
        eng.reset_optimization_count()
        one = gp.OpStream()
        one.ops = [op]
        eng.process_operations(one)
        assert eng.get_branch_lengths()[edge] == expect, method


def test_gp_opcodes_are_the_variants_alternative_indices():
    """Seam 3's opcode of an operation is the alternative index of the reference's std::variant GPOperation, and a, b, c are
    the reference's fields in declaration order (src/gp_operation.hpp:24-167; INTEGRATION.md's Flatten visitor): read off
    the reference's own types."""
    names = ["ZERO_PLV", "SET_TO_STATIONARY", "INCREMENT_WITH_WEIGHTED_EVOLVED_PLV", "MULTIPLY", "LIKELIHOOD",
             "OPTIMIZE_BRANCH_LENGTH", "UPDATE_SBN_PROBABILITIES", "RESET_MARGINAL_LIKELIHOOD", "INCREMENT_MARGINAL_LIKELIHOOD",
             "PREP_FOR_MARGINALIZATION"]
    fields = {0: (11, 0, 0), 1: (11, 22, 0), 2: (11, 22, 33), 3: (11, 22, 33), 4: (11, 22, 33), 5: (11, 22, 33), 6: (11, 22, 0),
              7: (0, 0, 0), 8: (11, 22, 33), 9: (11, 0, 0)}
    for opcode, name in enumerate(names):
        assert getattr(gp, name) == opcode
        index, a, b, c, count = ref.gp_operation(opcode)
        assert index == opcode and (a, b, c) == fields[opcode] and count == (3 if opcode == 9 else 0)
